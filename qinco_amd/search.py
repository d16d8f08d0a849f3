"""Small-database search after the hot path: encode -> decode -> brute-force L2 top-100 -> recall.

`search_small_db` mirrors run_search_full_direct_small_db (reference qinco/search/search_tasks.py:551-603) and
`compute_recalls` mirrors :275-282; the top-k itself runs on the GPU through qinco_knn_* (fp32-MFMA distance table +
radix select, csrc/knn_kernel.hpp).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _lib
from .engine import _is_torch

F32 = np.float32


class KnnSearcher:
    """ids[q] = argsort_n(|queries[q]|^2 + |db[n]|^2 - 2 queries[q].db[n])[:k]  (stable: ties -> lower n)."""

    def __init__(self, D: int):
        self.lib = _lib.load()
        self.D = int(D)
        self._h = C.c_void_p()
        _lib.check(self.lib.qinco_knn_create(self.D, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.qinco_knn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, db, queries, k: int = 100, return_dist: bool = False):
        """db (n, D), queries (nq, D) float32: both torch CUDA tensors (device path, asynchronous on the current
        stream) or both host arrays (host path).  Returns ids (nq, k) int64 [and distances (nq, k) float32]."""
        D = self.D
        if _is_torch(db) and db.is_cuda:
            import torch
            if not (_is_torch(queries) and queries.is_cuda):
                queries = torch.as_tensor(np.asarray(queries, dtype=F32)).to(db.device)
            db = db.to(torch.float32).contiguous()
            queries = queries.to(torch.float32).contiguous()
            if db.dim() != 2 or queries.dim() != 2 or db.shape[1] != D or queries.shape[1] != D:
                raise ValueError(f"db / queries must be (n, {D})")
            nq = queries.shape[0]
            ids = torch.empty((nq, k), dtype=torch.int64, device=db.device)
            dist = torch.empty((nq, k), dtype=torch.float32, device=db.device) if return_dist else None
            st = torch.cuda.current_stream(db.device).cuda_stream
            _lib.check(self.lib.qinco_knn_search(self._h, db.data_ptr(), db.shape[0], queries.data_ptr(), nq, int(k),
                                                 ids.data_ptr(), dist.data_ptr() if dist is not None else None, st))
            return (ids, dist) if return_dist else ids
        if _is_torch(db):
            db = db.detach().cpu().numpy()
        if _is_torch(queries):
            queries = queries.detach().cpu().numpy()
        db = np.ascontiguousarray(np.asarray(db, dtype=F32))
        queries = np.ascontiguousarray(np.asarray(queries, dtype=F32))
        if db.ndim != 2 or queries.ndim != 2 or db.shape[1] != D or queries.shape[1] != D:
            raise ValueError(f"db / queries must be (n, {D})")
        nq = queries.shape[0]
        ids = np.empty((nq, k), dtype=np.int64)
        dist = np.empty((nq, k), dtype=F32) if return_dist else None
        _lib.check(self.lib.qinco_knn_search_host(self._h, db.ctypes.data, db.shape[0], queries.ctypes.data, nq, int(k),
                                                  ids.ctypes.data, dist.ctypes.data if dist is not None else None))
        return (ids, dist) if return_dist else ids


def compute_recalls(I, gt) -> dict:
    """search_tasks.py:275-282: fraction of queries whose first ground-truth id is among the first `rank` results."""
    I = np.asarray(I.cpu() if _is_torch(I) else I)
    gt = np.asarray(gt.cpu() if _is_torch(gt) else gt)
    assert I.ndim == 2 and gt.ndim == 2
    return {rank: float((I[:, :rank] == gt[:, :1]).sum() / gt.shape[0]) for rank in (1, 10, 100)}


def search_small_db(model: Callable, db, queries, gt, batch: int = 65536, nshort: int = 100,
                    log: Optional[Callable[[str], None]] = None) -> dict:
    """run_search_full_direct_small_db (search_tasks.py:551-603): encode and decode the whole database with
    `model(x, step="encode")` / `model(codes, step="decode")`, keep the reconstructions in HBM, take each query's
    `nshort` nearest reconstructions and score them against `gt` (nq, >=1).  Returns {"recalls": {1,10,100},
    "shortlists": (nq, nshort) int64 numpy, "xhat": the decoded database}.  With torch + a GPU present everything
    stays on the device; otherwise the host entry points are used."""
    D = int(np.asarray(queries).shape[1]) if not _is_torch(queries) else int(queries.shape[1])
    N = db.shape[0]
    try:
        import torch
        on_device = torch.cuda.is_available()
    except ImportError:  # pragma: no cover
        torch, on_device = None, False
    parts = []
    for i0 in range(0, N, batch):
        xb = db[i0:i0 + batch]
        if on_device:
            xb = xb if _is_torch(xb) else torch.from_numpy(np.ascontiguousarray(xb))
            xb = xb.to("cuda")
            if xb.dtype != torch.uint8:
                xb = xb.to(torch.float32)
        codes_MB = model(xb, step="encode")
        parts.append(model(codes_MB, step="decode"))
        if log and (i0 // batch) % 10 == 0:
            log(f"Encoding database, batch {i0 // batch + 1}/{(N + batch - 1) // batch}")
    if on_device:
        xhat = torch.cat(parts, dim=0)
    else:
        xhat = np.concatenate([np.asarray(p) for p in parts], axis=0)
    knn = KnnSearcher(D)
    try:
        shortlists = knn.search(xhat, queries, k=min(nshort, N))
        shortlists = shortlists.cpu().numpy() if _is_torch(shortlists) else shortlists
    finally:
        knn.close()
    recalls = compute_recalls(shortlists, gt)
    if log:
        log("R@1={:.2f}    R@10={:.2f}    R@100={:.2f}".format(*(recalls[r] * 100 for r in (1, 10, 100))))
    return {"recalls": recalls, "shortlists": shortlists, "xhat": xhat}
