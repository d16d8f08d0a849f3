"""QincoEngine: owns one libqinco_hip handle (= one model on one GPU) and moves buffers across the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .config import QincoConfig

F32 = np.float32
_CODE_DT = {np.dtype(np.int64): _lib.CODE_I64, np.dtype(np.int32): _lib.CODE_I32, np.dtype(np.uint8): _lib.CODE_U8}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class QincoEngine:
    """encode / decode of one QINCo model through the HIP library.

    state_dict: {reference parameter name: array-like fp32} (qinco_base.py:229-260, 432-445); torch tensors
    are accepted (PyTorch is only the weight loader here).
    """

    def __init__(self, cfg: QincoConfig, state_dict: dict, max_batch: int = 8192, device: Optional[int] = None,
                 split_f16: bool = False, diagnostics: Optional[dict] = None):
        """split_f16: opt-in split-fp16 evaluation of the FFN blocks (include/qinco_hip.h, QINCO_CREATE_SPLIT_F16): several
        times the fp32-MFMA throughput, fp32-class accuracy but not the fp32 path's bits.
        diagnostics: qinco_options knobs for A/B runs and the race-detector tests -- ivf_fp32, table_valu, decode_folded,
        table_no_coop, split_no_calibration, no_presel_fusion, no_small_launch, epilogue_select, no_epilogue_select (bools), mlp_variant=(P, VAR) (a non-production kernel instance of
        csrc/shapes.def), table_coop_max."""
        self.lib = _lib.load()
        self.cfg = cfg
        self.split_f16 = bool(split_f16)
        diag = dict(diagnostics or {})
        flags = _lib.CREATE_SPLIT_F16 if self.split_f16 else 0
        for key, bit in (("ivf_fp32", _lib.CREATE_IVF_FP32), ("table_valu", _lib.CREATE_TABLE_VALU),
                         ("decode_folded", _lib.CREATE_DECODE_FOLDED), ("table_no_coop", _lib.CREATE_TABLE_NO_COOP),
                         ("split_no_calibration", _lib.CREATE_SPLIT_NO_CALIBRATION),
                         ("no_presel_fusion", _lib.CREATE_NO_PRESEL_FUSION), ("no_small_launch", _lib.CREATE_NO_SMALL_LAUNCH),
                         ("epilogue_select", _lib.CREATE_EPILOGUE_SELECT), ("no_epilogue_select", _lib.CREATE_NO_EPILOGUE_SELECT)):
            if diag.pop(key, False):
                flags |= bit
        P, var = diag.pop("mlp_variant", None) or (-1, -1)
        opts = _lib.QincoOptions(struct_bytes=C.sizeof(_lib.QincoOptions), create_flags=flags, mlp_P=int(P), mlp_var=int(var),
                                 table_coop_max=int(diag.pop("table_coop_max", -1)))
        if diag:
            raise ValueError(f"unknown diagnostics keys: {sorted(diag)}")
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        if device is not None:
            import torch
            torch.cuda.set_device(device)
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("module."):            # load_model strips the DDP prefix (utils.py:195-199)
                k = k[len("module."):]
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            sd[k] = np.ascontiguousarray(np.asarray(v, dtype=F32))
        self._keep = []  # host arrays must outlive qinco_create only, but keep them for introspection

        def ptr(name, shape):
            if name not in sd:
                raise KeyError(f"state_dict is missing {name}")
            a = sd[name]
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
            self._keep.append(a)
            return a.ctypes.data_as(_lib.FP)

        def parr(n):
            return (_lib.FP * n)()

        M, L, K, D, De, Dh = cfg.M_total, cfg.L, cfg.K, cfg.D, cfg.De, cfg.dh
        if M > 1 and not self.split_f16 and not self.lib.qinco_shape_supported(D, De, Dh):
            from .build import ensure_instance      # a geometry shapes.def does not list: one kernel instance built on demand
            ensure_instance(D, De, Dh)
        w = _lib.QincoWeights()
        w.data_mean = ptr("data_mean", (D,))
        std = float(np.asarray(sd["data_std"]).reshape(-1)[0]) if "data_std" in sd else 0.0
        w.data_std = std
        cb, sub, inp, outp, cw, cbias = parr(M), parr(M), parr(M), parr(M), parr(M), parr(M)
        up, down = parr(max(M * L, 1)), parr(max(M * L, 1))
        for m in range(M):
            p = f"steps.{m}."
            if m == 0 and cfg.ivf:      # IVFBook (qinco_base.py:128-196)
                cb[0] = ptr(p + "ivf_centroids.weight", (cfg.ivf_K, D))
                continue
            cb[m] = ptr(p + "codebook.weight", (K, D))
            if m == 0:
                continue
            if cfg.A > 0:
                sub[m] = ptr(p + "substep.codebook.weight", (K, D))
            cw[m] = ptr(p + "concat.mlp.weight", (De, De + D))
            cbias[m] = ptr(p + "concat.mlp.bias", (De,))
            if De != D:
                inp[m] = ptr(p + "in_proj.weight", (De, D))
                outp[m] = ptr(p + "out_proj.weight", (D, De))
            for l in range(L):
                up[m * L + l] = ptr(p + f"residual_blocks.{l}.up_proj.weight", (Dh, De))
                down[m * L + l] = ptr(p + f"residual_blocks.{l}.down_proj.weight", (De, Dh))
        w.codebook, w.sub_codebook, w.in_proj, w.out_proj = cb, sub, inp, outp
        w.cat_w, w.cat_b, w.up, w.down = cw, cbias, up, down
        desc = _lib.QincoDesc(D=D, De=De, Dh=Dh, L=L, M=M, K=K, A=cfg.A, B=cfg.B,
                              qinco1_mode=int(cfg.qinco1_mode), ivf_K=int(cfg.ivf_K or 0), max_batch=self.max_batch)
        _lib.check(self.lib.qinco_create_opt(C.byref(desc), C.byref(w), C.byref(opts), C.byref(self._h)))
        self._keep = []  # weights now live on the device
        self.device = None
        try:   # the HIP device the handle is bound to (= the current one at create), for stream look-ups on multi-GPU processes
            import sys
            if "torch" in sys.modules:
                self.device = sys.modules["torch"].cuda.current_device()
        except Exception:
            pass
        self.data_mean = sd["data_mean"]
        self.data_std = F32(std)
        self.A, self.B = cfg.A, cfg.B

    # ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.qinco_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_beam(self, A: Optional[int] = None, B: Optional[int] = None):
        A = self.A if A is None else int(A)
        B = self.B if B is None else int(B)
        _lib.check(self.lib.qinco_set_beam(self._h, A, B))
        self.A, self.B = A, B

    # ------------------------------------------------------------------------------------------
    def encode(self, x, code_dtype=np.int64, return_xhat: bool = False, normalised: bool = False, check: Optional[bool] = None):
        """x: (n, D) float32 / uint8, numpy (host path) or torch CUDA tensor (device path, async on the current
        stream).  Returns codes (n, M) [and the normalised reconstruction (n, D)].  normalised=True: x is already
        (x - mean) / std, i.e. QINCoInferenceWrapper.encode instead of forward.
        check (device path): wait for the stream and raise if the split-fp16 form left the fp16 range (qinco_check); default:
        only when the engine was created with split_f16 -- the fp32 path cannot fail on the device and stays asynchronous."""
        M, D = self.cfg.M_total, self.cfg.D
        flags = _lib.FLAG_NORMALISED if normalised else 0
        if _is_torch(x) and x.is_cuda:
            import torch
            if x.dim() != 2 or x.shape[1] != D:
                raise ValueError(f"x must be (n, {D}), got {tuple(x.shape)}")
            if x.dtype not in (torch.float32, torch.uint8):
                x = x.to(torch.float32)
            if x.stride(1) != 1:
                x = x.contiguous()
            n = x.shape[0]
            tdt = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8}
            codes = torch.empty((n, M), dtype=tdt[np.dtype(code_dtype)], device=x.device)
            xhat = torch.empty((n, D), dtype=torch.float32, device=x.device) if return_xhat else None
            st = torch.cuda.current_stream(x.device).cuda_stream
            _lib.check(self.lib.qinco_encode(
                self._h, x.data_ptr(), _lib.X_F32 if x.dtype == torch.float32 else _lib.X_U8,
                x.stride(0) * x.element_size(), n, codes.data_ptr(), _CODE_DT[np.dtype(code_dtype)],
                xhat.data_ptr() if xhat is not None else None, flags, st))
            if self.split_f16 if check is None else check:
                _lib.check(self.lib.qinco_check(self._h, st))
            return (codes, xhat) if return_xhat else codes
        if _is_torch(x):
            x = x.detach().cpu().numpy()
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != D:
            raise ValueError(f"x must be (n, {D}), got {x.shape}")
        if x.dtype != np.uint8:
            x = x.astype(F32, copy=False)
        if x.strides[1] != x.itemsize:
            x = np.ascontiguousarray(x)
        n = x.shape[0]
        codes = np.empty((n, M), dtype=code_dtype)
        xhat = np.empty((n, D), dtype=F32) if return_xhat else None
        _lib.check(self.lib.qinco_encode_host(
            self._h, x.ctypes.data, _lib.X_U8 if x.dtype == np.uint8 else _lib.X_F32, x.strides[0], n,
            codes.ctypes.data, _CODE_DT[np.dtype(code_dtype)], xhat.ctypes.data if xhat is not None else None,
            flags))
        return (codes, xhat) if return_xhat else codes

    def decode(self, codes, normalised: bool = False, check: bool = True):
        """codes: (n, M) int64 / int32 / uint8, numpy or torch CUDA tensor.  Returns (n, D) float32, denormalised
        (forward(step="decode")) unless normalised=True (QINCoInferenceWrapper.decode).  Codes outside [0, K) raise
        IndexError like the reference's indexing does; on the device path that costs a stream synchronisation
        (qinco_check) -- check=False keeps the call asynchronous and leaves the check to a later `check_codes()`."""
        M, D = self.cfg.M_total, self.cfg.D
        flags = _lib.FLAG_NORMALISED if normalised else 0
        if _is_torch(codes) and codes.is_cuda:
            import torch
            if codes.dim() != 2 or codes.shape[1] != M:
                raise ValueError(f"codes must be (n, {M}), got {tuple(codes.shape)}")
            if codes.dtype not in (torch.int64, torch.int32, torch.uint8):
                codes = codes.to(torch.int64)
            codes = codes.contiguous()
            n = codes.shape[0]
            out = torch.empty((n, D), dtype=torch.float32, device=codes.device)
            cdt = {torch.int64: _lib.CODE_I64, torch.int32: _lib.CODE_I32, torch.uint8: _lib.CODE_U8}[codes.dtype]
            st = torch.cuda.current_stream(codes.device).cuda_stream
            _lib.check(self.lib.qinco_decode(self._h, codes.data_ptr(), cdt, n, out.data_ptr(), flags, st))
            if check:
                _lib.check(self.lib.qinco_check(self._h, st))
            return out
        if _is_torch(codes):
            codes = codes.detach().cpu().numpy()
        codes = np.asarray(codes)
        if codes.ndim != 2 or codes.shape[1] != M:
            raise ValueError(f"codes must be (n, {M}), got {codes.shape}")
        if codes.dtype not in _CODE_DT:
            codes = codes.astype(np.int64)
        codes = np.ascontiguousarray(codes)
        n = codes.shape[0]
        out = np.empty((n, D), dtype=F32)
        _lib.check(self.lib.qinco_decode_host(self._h, codes.ctypes.data, _CODE_DT[codes.dtype], n, out.ctypes.data,
                                              flags))
        return out

    def check_codes(self, stream=None):
        """Wait for `stream` (default: the current torch stream OF THE ENGINE'S DEVICE, else the null stream) and raise
        IndexError if a device-path call since the last check saw a code outside [0, K) or a split-fp16 overflow
        (include/qinco_hip.h: qinco_check)."""
        if stream is None:
            import sys
            torch = sys.modules.get("torch")
            stream = (torch.cuda.current_stream(self.device).cuda_stream
                      if torch is not None and torch.cuda.is_available() else None)
        _lib.check(self.lib.qinco_check(self._h, stream))

    # ------------------------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        _lib.check(self.lib.qinco_profile_enable(self._h, int(on)))

    def profile_read(self):
        """Totals of the fused-MLP launches since the last read: event time, launches, algorithmic FLOPs (rows x R_mlp) and the
        FLOPs the matrix pipe executed for them (qinco_profile_read2)."""
        ms, cnt, fl, fx = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib.qinco_profile_read2(self._h, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(fx)))
        return {"mlp_ms": ms.value, "mlp_launches": cnt.value, "mlp_flops": fl.value, "mlp_flops_executed": fx.value}

    def ivf_last_stats(self) -> dict:
        """IVF models: exact-pass candidate pairs and fall-back flag of the last IVF assignment (qinco_ivf_last_stats)."""
        c, f = C.c_int64(), C.c_int32()
        _lib.check(self.lib.qinco_ivf_last_stats(self._h, C.byref(c), C.byref(f)))
        return {"candidates": c.value, "fell_back": bool(f.value)}

    def split_stats(self) -> dict:
        """qinco_split_stats: the split-fp16 form's create-time calibration against the fp32 instance and its run-time
        underflow / overflow statistics (include/qinco_hip.h)."""
        r = _lib.QincoSplitReport()
        _lib.check(self.lib.qinco_split_stats(self._h, C.byref(r)))
        d = {k: getattr(r, k) for k, _ in r._fields_}
        d["lo_subnormal_frac"] = (r.lo_subnormal / r.lo_sampled) if r.lo_sampled else None
        return d

    def describe(self) -> str:
        """qinco_describe: which kernel instances / arithmetic form serve this handle."""
        buf = C.create_string_buffer(512)
        n = self.lib.qinco_describe(self._h, buf, 512)
        if n < 0:
            _lib.check(n)
        return buf.value.decode()

    def flops_per_vector(self, what: str = "encode") -> float:
        fn = self.lib.qinco_flops_per_vector_encode if what == "encode" else self.lib.qinco_flops_per_vector_decode
        return float(fn(self._h))
