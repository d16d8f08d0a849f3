"""Look-up decoders downstream of the hot path, on the GPU through qinco_lut_* (HBM-bound gather-add).

`reconstruct_from_fixed_codebooks` mirrors qinco/search/search_utils.py:105-115; `PairwiseDecoder` mirrors the
inference part of PairwiseDecoderIVF (qinco/search/pairwise_decoder.py: forward :88-93, map_codes :126-130) with its
state-dict tensors `codebook_MKD` (M_target, K^2, D), `combine_mvals_m` (2, M_target), `ivf_code_map` (ivf_K, IVF_M).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .engine import _CODE_DT, _is_torch

F32 = np.float32


class LutDecoder:
    """out[n] = sum_j tables[j][codes[n, a[j]] * mul + (codes[n, b[j]] if b[j] >= 0 else 0)], fp32, summed in j order."""

    def __init__(self, tables: np.ndarray, a, b=None, mul: int = 1):
        self.lib = _lib.load()
        tables = np.ascontiguousarray(np.asarray(tables, dtype=F32))
        if tables.ndim != 3:
            raise ValueError("tables must be (J, Kt, D)")
        self.J, self.Kt, self.D = tables.shape
        a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
        b = np.full(self.J, -1, np.int32) if b is None else np.ascontiguousarray(np.asarray(b, dtype=np.int32))
        if a.shape != (self.J,) or b.shape != (self.J,):
            raise ValueError("a / b must have one entry per table")
        self._h = C.c_void_p()
        _lib.check(self.lib.qinco_lut_create(tables.ctypes.data_as(_lib.FP), self.J, self.Kt, self.D,
                                             a.ctypes.data_as(C.POINTER(C.c_int32)), b.ctypes.data_as(C.POINTER(C.c_int32)),
                                             int(mul), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.qinco_lut_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, codes):
        """codes (n, Mc) int64/int32/uint8, numpy (host path) or torch CUDA tensor (device path) -> (n, D) float32."""
        if _is_torch(codes) and codes.is_cuda:
            import torch
            if codes.dtype not in (torch.int64, torch.int32, torch.uint8):
                codes = codes.to(torch.int64)
            codes = codes.contiguous()
            n, Mc = codes.shape
            out = torch.empty((n, self.D), dtype=torch.float32, device=codes.device)
            cdt = {torch.int64: _lib.CODE_I64, torch.int32: _lib.CODE_I32, torch.uint8: _lib.CODE_U8}[codes.dtype]
            st = torch.cuda.current_stream(codes.device).cuda_stream
            _lib.check(self.lib.qinco_lut_decode(self._h, codes.data_ptr(), cdt, Mc, n, out.data_ptr(), st))
            return out
        if _is_torch(codes):
            codes = codes.detach().cpu().numpy()
        codes = np.asarray(codes)
        if codes.dtype not in _CODE_DT:
            codes = codes.astype(np.int64)
        codes = np.ascontiguousarray(codes)
        n, Mc = codes.shape
        out = np.empty((n, self.D), dtype=F32)
        _lib.check(self.lib.qinco_lut_decode_host(self._h, codes.ctypes.data, _CODE_DT[codes.dtype], Mc, n, out.ctypes.data))
        return out


def reconstruct_from_fixed_codebooks(codes, codebooks):
    """search_utils.py:105-115: codes (N, M), codebooks (M, K, D) -> sum_m codebooks[m, codes[:, m]]."""
    codebooks = np.asarray(codebooks, dtype=F32)
    M = codebooks.shape[0]
    if np.asarray(codes).shape[1] != M:
        raise AssertionError("codebooks.shape[0] == M")
    dec = LutDecoder(codebooks, a=np.arange(M))
    try:
        return dec(codes)
    finally:
        dec.close()


class PairwiseDecoder:
    """Inference half of PairwiseDecoderIVF: forward(codes_MB, ivf_codes) (pairwise_decoder.py:88-93)."""

    def __init__(self, codebook_MKD, combine_mvals_m, K_base: int, ivf_code_map: Optional[np.ndarray] = None):
        self.codebook_MKD = np.asarray(codebook_MKD, dtype=F32)
        self.combine = np.asarray(combine_mvals_m, dtype=np.int64)
        self.K_base = int(K_base)
        self.ivf_code_map = None if ivf_code_map is None else np.asarray(ivf_code_map, dtype=np.int64)
        self._dec = LutDecoder(self.codebook_MKD, a=self.combine[0], b=self.combine[1], mul=self.K_base)

    def gather_codes(self, codes_MB, ivf_codes=None) -> np.ndarray:
        """The column set map_codes builds before combining (:126-128): [codes ; ivf_code_map[ivf_codes].T] as (n, Mc)."""
        codes = np.asarray(codes_MB).T
        if self.ivf_code_map is not None:
            ivf_codes = np.asarray(ivf_codes)
            assert ivf_codes.ndim == 1
            codes = np.concatenate([codes, self.ivf_code_map[ivf_codes]], axis=1)
        return np.ascontiguousarray(codes.astype(np.int64))

    def forward(self, codes_MB, ivf_codes=None):
        return self._dec(self.gather_codes(codes_MB, ivf_codes))

    __call__ = forward

    def close(self):
        self._dec.close()
