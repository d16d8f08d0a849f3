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

    def __init__(self, D: int, filtered: bool = True, filter_min_n: int | None = None, roles: bool | None = None):
        """filtered: large databases take the form that never writes the (queries x n) distance table (csrc/knn_kernel.hpp;
        the same ids and distances bit for bit); filter_min_n: the smallest n that takes it (library default 65536);
        roles: None = library default (off): every wave computes and filters; True (D <= 128): the filtered table runs as one MFMA
        wave + one filter wave per SIMD (csrc/knn_roles_kernel.hpp: the round-6 experiment, measured slower).  Same bits either way."""
        self.lib = _lib.load()
        self.D = int(D)
        self._h = C.c_void_p()
        _lib.check(self.lib.qinco_knn_create(self.D, C.byref(self._h)))
        _lib.check(self.lib.qinco_knn_set_option(self._h, 0, int(bool(filtered))))
        if filter_min_n is not None:
            _lib.check(self.lib.qinco_knn_set_option(self._h, 1, int(filter_min_n)))
        if roles is not None:
            _lib.check(self.lib.qinco_knn_set_option(self._h, 3, int(bool(roles))))

    def roles_stats(self) -> dict:
        """Two-role kernel launches since the last call (synchronises): workgroups with one MFMA wave on every SIMD / that fell
        back to roles by wave index."""
        out = (C.c_int64 * 2)()
        _lib.check(self.lib.qinco_knn_roles_stats(self._h, out))
        return {"spread": int(out[0]), "by_index": int(out[1])}

    def last_stats(self) -> dict:
        """Of the last search (synchronises the device): chunks of queries, those that took the filtered form, those redone
        unfiltered because a candidate list overflowed."""
        out = (C.c_int64 * 3)()
        _lib.check(self.lib.qinco_knn_last_stats(self._h, out))
        return {"chunks": int(out[0]), "filtered": int(out[1]), "redone_unfiltered": int(out[2])}

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


def rerank(xq, cand, k: int, ids=None, codes=None) -> dict:
    """One re-rank stage of run_search_ivf (search_tasks.py:447-472 / 497-507) on the GPU (qinco_rerank, csrc/rerank_kernel.hpp):
    xq (nq, d), cand (nq, ns, d) float32 CUDA tensors; for every query the distances |x|^2 + |c|^2 - 2 x.c to its own ns candidates
    (compute_batch_distances(..., approx=True)), sorted ascending (ties -> the earlier shortlist position), the first k kept.
    Returns {"pos": (nq, k) positions in the shortlist, "dist": (nq, k), "ids": ids (nq, ns) gathered to (nq, k) if given,
    "codes": codes (nq, ns, Mc) int32 gathered to (nq, k, Mc) if given}; asynchronous on the current stream."""
    import torch
    lib = _lib.load()
    xq = xq.to(torch.float32).contiguous()
    cand = cand.to(torch.float32).contiguous()
    nq, ns, d = cand.shape
    if xq.shape != (nq, d):
        raise ValueError(f"xq must be ({nq}, {d}), got {tuple(xq.shape)}")
    dev = cand.device
    pos = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out = {"pos": pos, "dist": dist}
    ids_in = ids_out = codes_in = codes_out = None
    Mc = 0
    if ids is not None:
        ids_in = ids.to(device=dev, dtype=torch.int64).contiguous()
        ids_out = out["ids"] = torch.empty((nq, k), dtype=torch.int64, device=dev)
    if codes is not None:
        codes_in = codes.to(device=dev, dtype=torch.int32).contiguous().reshape(nq, ns, -1)
        Mc = codes_in.shape[2]
        codes_out = out["codes"] = torch.empty((nq, k, Mc), dtype=torch.int32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
    _lib.check(lib.qinco_rerank(xq.data_ptr(), cand.data_ptr(), nq, ns, d, int(k), ptr(ids_in), ptr(codes_in), Mc, pos.data_ptr(),
                                dist.data_ptr(), ptr(ids_out), ptr(codes_out), torch.cuda.current_stream(dev).cuda_stream))
    return out


def rerank_ivf(model: Callable, xq, I, codes_int32, *, nshort: int, mid_reranker: Optional[Callable] = None, ivf_book=None,
               batch_size: int = 12288, topk: int = 100) -> dict:
    """The re-rank stages of run_search_ivf (search_tasks.py:447-507) behind the IVF shortlist that faiss hands over:
      xq (nq, d) queries, I (nq, n_short_ivf) database ids and codes_int32 (nq * n_short_ivf, M + 1) their code rows (IVF id in
      column 0), as search_tasks.py:418-445 assembles them;
      Part 3 (only if nshort < n_short_ivf): approximate reconstructions from the look-up decoder -- mid_reranker(codes_MB,
        ivf_codes) -> (n, d), e.g. qinco_amd.lut.PairwiseDecoder -- plus the IVF centroid ivf_book[codes[:, 0]]
        (model.qinco_model.steps[0].ivf_centroids.weight in the reference), distances to the query, the nshort best kept with
        their ids and code rows (rerank);
      Part 4: QINCo decode of the kept code rows in batches of cfg.search.batch_size = 12 288 through model(codes.T, step="decode")
        (the small-launch form of the fused MLP: one launch per batch);
      Part 5: distances of the decoded vectors to the query, the topk = 100 best ids.
    Everything stays on the GPU.  Returns {"I": (nq, topk) final ids, "I_mid", "codes_mid": the stage-3 survivors (None without
    that stage), "decoded": (nq, nshort, d), "dist": (nq, topk)}."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    xq = torch.as_tensor(np.asarray(xq, dtype=F32) if not _is_torch(xq) else xq).to(dev, torch.float32)
    I = torch.as_tensor(np.asarray(I) if not _is_torch(I) else I).to(dev, torch.int64)
    codes = torch.as_tensor(np.asarray(codes_int32) if not _is_torch(codes_int32) else codes_int32).to(dev, torch.int32)
    nq, n_short_ivf = I.shape
    d, Mc = xq.shape[1], codes.shape[1]
    I_mid = codes_mid = None
    if nshort < n_short_ivf:
        if mid_reranker is None or ivf_book is None:
            raise ValueError("nshort < n_short_ivf needs the mid re-ranker and the IVF centroids (search_tasks.py:447-451)")
        ct = codes.T
        approx = mid_reranker(ct[1:], ct[0])
        approx = torch.as_tensor(approx).to(dev, torch.float32) if not (_is_torch(approx) and approx.is_cuda) else approx
        book = torch.as_tensor(np.asarray(ivf_book, dtype=F32) if not _is_torch(ivf_book) else ivf_book).to(dev, torch.float32)
        approx = approx + book[codes[:, 0].long()]
        r = rerank(xq, approx.reshape(nq, n_short_ivf, d), nshort, ids=I, codes=codes.reshape(nq, n_short_ivf, Mc))
        I_mid, codes_mid = r["ids"], r["codes"]
        I, codes = I_mid, codes_mid.reshape(nq * nshort, Mc)
    else:
        nshort = n_short_ivf
    parts = [model(codes[i:i + batch_size].T, step="decode") for i in range(0, len(codes), batch_size)]
    decoded = torch.cat([p if _is_torch(p) else torch.from_numpy(np.asarray(p)).to(dev) for p in parts]).reshape(nq, nshort, d)
    r = rerank(xq, decoded, min(topk, nshort), ids=I)
    return {"I": r["ids"], "dist": r["dist"], "I_mid": I_mid, "codes_mid": codes_mid, "decoded": decoded}


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
