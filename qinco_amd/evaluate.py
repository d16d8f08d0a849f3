"""MSE + per-vector timing of the hot path: compute_MSE / AnyVectMSE / Timer of the reference
(qinco/qinco_tasks.py:87-148, qinco/metrics.py:29-58, 182-253), i.e. `task=eval` and `task=eval_time`.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, Iterable, Optional

import numpy as np

from . import _lib
from .engine import _is_torch


class Timer:
    """Accumulating wall-clock timer used as a context manager (metrics.py:182-253)."""

    def __init__(self):
        self.elapsed = 0.0
        self._t0 = None

    def __enter__(self):
        assert self._t0 is None, "Timer is already in use"
        self._t0 = time.time()
        return self

    def __exit__(self, *exc):
        self.elapsed += time.time() - self._t0
        self._t0 = None

    def get(self) -> float:
        return self.elapsed + (time.time() - self._t0 if self._t0 is not None else 0.0)


def sqerr_sum(batch, xhat) -> float:
    """sum((batch - xhat)**2) (AnyVectMSE.update, metrics.py:43-50).  CUDA tensors: one reduction kernel through
    qinco_sqerr_sum (fp64 accumulation); host arrays: numpy in fp64."""
    if _is_torch(batch) and batch.is_cuda and _is_torch(xhat) and xhat.is_cuda:
        import torch
        a = batch.to(torch.float32).contiguous()
        b = xhat.to(torch.float32).contiguous()
        if a.shape != b.shape:
            raise AssertionError(f"xhat.shape={tuple(b.shape)} != batch.shape={tuple(a.shape)}")
        out = C.c_double()
        st = torch.cuda.current_stream(a.device).cuda_stream
        _lib.check(_lib.load().qinco_sqerr_sum(a.data_ptr(), b.data_ptr(), a.numel(), C.byref(out), st))
        return out.value
    a = np.asarray(batch.cpu() if _is_torch(batch) else batch, dtype=np.float64)
    b = np.asarray(xhat.cpu() if _is_torch(xhat) else xhat, dtype=np.float64)
    if a.shape != b.shape:
        raise AssertionError(f"xhat.shape={b.shape} != batch.shape={a.shape}")
    return float(((a - b) ** 2).sum())


def _force(t) -> None:
    """Forces completion before leaving a timer (qinco_tasks.py:112-122 reads the last element back)."""
    if _is_torch(t):
        float(t.reshape(-1)[-1].cpu())
    else:
        float(np.asarray(t).reshape(-1)[-1])


def compute_MSE(model: Callable, val_batches: Iterable, mse_scale: float = 1.0, warm_start: bool = True, dist=None,
                log: Optional[Callable[[str], None]] = None) -> dict:
    """qinco_tasks.py:87-148.  `val_batches` is a re-iterable of (n_b, D) batches (numpy, or torch on the GPU).
    Up to 11 warm-up batches when `warm_start` ("eval" tasks), then every batch is encoded and decoded under separate
    timers that force completion, and sum((x - xhat)^2) is accumulated.  MSE = loss_sum * mse_scale / n
    (metrics.py:51-58); with `dist` (an initialised torch.distributed) the sums of all ranks are combined like
    AnyVectMSE.compute_sync.  Returns MSE, n_vecs, encode/decode seconds and microseconds per vector."""
    if warm_start:
        decoded = None
        for i_batch, batch in enumerate(val_batches):
            decoded = model(model(batch, step="encode"), step="decode")
            if i_batch >= 10:
                break
        if decoded is not None:
            _force(decoded)
    t_encode, t_decode = Timer(), Timer()
    n_vecs, loss_sum = 0, 0.0
    for batch in val_batches:
        n_vecs += len(batch)
        with t_encode:
            codes = model(batch, step="encode")
            _force(codes)
        with t_decode:
            xhat = model(codes, step="decode")
            _force(xhat)
        assert tuple(xhat.shape) == tuple(batch.shape), f"{tuple(xhat.shape)=} != {tuple(batch.shape)=}"
        loss_sum += sqerr_sum(batch, xhat)
    tot_loss, tot_n = loss_sum * mse_scale, n_vecs
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([tot_loss, float(tot_n)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        tot_loss, tot_n = float(t[0]), int(t[1])
    res = {"MSE": tot_loss / max(tot_n, 1), "n_vecs": n_vecs, "encode_s": t_encode.get(), "decode_s": t_decode.get(),
           "encode_us_per_vec": t_encode.get() / max(n_vecs, 1) * 1e6,
           "decode_us_per_vec": t_decode.get() / max(n_vecs, 1) * 1e6}
    if log:
        log(f"MSE: {res['MSE']:.6g}")
        log(f"Encoding time / vector: {res['encode_us_per_vec']:.1f}us")
        log(f"Decoding time / vector: {res['decode_us_per_vec']:.1f}us")
    return res
