#!/usr/bin/env python
"""Randomised cases for the small-db search's FILTERED form (csrc/knn_kernel.hpp: thresholds from a strided sample, survivors of the
whole table voted into per-wave lists, per-query candidate sort) against the table form (distance table in HBM + radix select):

    python tests/sweeps/gpu_fuzz_knn.py --seed 1 --count 40 --out gpurun_out/fuzz_knn.jsonl         # GPU box

Per case a database drawn to stress one thing -- gaussian, clustered and SORTED by cluster (the strided sample must stay
representative), exact duplicates (thousands of keys AT the threshold: lists overflow, the chunk is redone on the device), a database
whose first rows are far from every query (a prefix sample would be useless), tiny / huge magnitudes, NaN and inf rows, queries that
are database rows (distance ~0 with cancellation noise, negative zeros) -- and D, n, the number of queries (several chunks, ragged
last chunk) and k at random.  Both forms must return the same ids and the same distance BITS; on a few rows the ids are checked
against a float64 ranking too (the returned distance must be within rounding of the float64 one and no closer row may be missing by
more than the fp32 table's own rounding)."""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

KINDS = ["gauss", "clustered_sorted", "duplicates", "far_prefix", "tiny", "huge", "nan_inf", "queries_are_rows", "heavy_tail"]


def make_case(rs, kind, D, n, nq):
    db = rs.randn(n, D).astype(np.float32)
    q = rs.randn(nq, D).astype(np.float32)
    if kind == "clustered_sorted":
        nc = int(rs.randint(8, 200))
        cen = (4.0 * rs.randn(nc, D)).astype(np.float32)
        lab = np.sort(rs.randint(0, nc, n))
        db = (cen[lab] + 0.3 * rs.randn(n, D)).astype(np.float32)
        q = (cen[rs.randint(0, nc, nq)] + 0.3 * rs.randn(nq, D)).astype(np.float32)
    elif kind == "duplicates":
        base = rs.randn(int(rs.randint(2, 50)), D).astype(np.float32)
        db = base[rs.randint(0, len(base), n)]
        q = (base[rs.randint(0, len(base), nq)] + 0.01).astype(np.float32)
    elif kind == "far_prefix":
        db[: n // 3] += 50.0
    elif kind == "tiny":
        db *= 1e-18
        q *= 1e-18
    elif kind == "huge":
        db *= 1e15
        q *= 1e15
    elif kind == "nan_inf":
        db[rs.choice(n, 50, replace=False), rs.randint(0, D, 50)] = np.nan
        db[rs.choice(n, 20, replace=False), rs.randint(0, D, 20)] = np.inf
        if nq > 3:
            q[1, 0] = np.nan
    elif kind == "queries_are_rows":
        q = db[rs.choice(n, nq)].copy()
        q[::7] = -q[::7] * 0.0      # rows of negative zeros among them
    elif kind == "heavy_tail":
        db = (rs.standard_cauchy((n, D)) * 0.1).astype(np.float32)
        q = (rs.standard_cauchy((nq, D)) * 0.1).astype(np.float32)
    return np.ascontiguousarray(db), np.ascontiguousarray(q)


def main():
    import torch
    from qinco_amd.search import KnnSearcher
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    out = open(a.out, "w") if a.out else None
    bad = 0
    for ci in range(a.count):
        kind = KINDS[ci % len(KINDS)]
        D = int(rs.choice([32, 64, 96, 128, 256, 768]))
        n = int(rs.randint(2_000, 60_000 if D == 768 else 400_000))
        nq = int(rs.choice([1, 7, 33, 500, 2100, 4500])) if D <= 128 else int(rs.choice([1, 33, 700]))
        k = int(rs.choice([1, 10, 100, 100, 300, 511, 513, 2048]))
        k = min(k, n)
        min_n = int(rs.choice([1024, 65536]))
        db, q = make_case(rs, kind, D, n, nq)
        dbt, qt = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
        res = {}
        t0 = time.time()
        for filtered in (True, False):
            knn = KnnSearcher(D, filtered=filtered, filter_min_n=min_n)
            ids, dist = knn.search(dbt, qt, k=k, return_dist=True)
            res[filtered] = (ids.cpu().numpy(), dist.cpu().numpy(), knn.last_stats())
            knn.close()
        (ids_f, dist_f, st_f), (ids_t, dist_t, _) = res[True], res[False]
        same = bool(np.array_equal(ids_f, ids_t) and np.array_equal(dist_f.view(np.uint32), dist_t.view(np.uint32)))
        # float64 check on a few finite rows: the k-th returned distance bounds every row left out, up to the fp32 table's rounding
        f64_ok = True
        if kind not in ("nan_inf",):
            for r in rs.choice(nq, min(nq, 3), replace=False):
                if not np.isfinite(dist_f[r]).all() or not np.isfinite(q[r]).all():
                    continue
                d64 = ((db.astype(np.float64) - q[r].astype(np.float64)) ** 2).sum(1)
                tol = 1e-5 * max(float((q[r].astype(np.float64) ** 2).sum() + (db.astype(np.float64) ** 2).sum(1).max()), 1e-300)
                left_out = np.ones(n, bool)
                left_out[ids_f[r]] = False
                if left_out.any() and d64[left_out].min() < d64[ids_f[r]].max() - 2 * tol:
                    f64_ok = False
                if len(set(ids_f[r].tolist())) != k:
                    f64_ok = False
        rec = {"case": ci, "kind": kind, "D": D, "n": n, "nq": nq, "k": k, "filter_min_n": min_n, **st_f, "forms_equal": same,
               "float64_check": f64_ok, "seconds": round(time.time() - t0, 2)}
        bad += not (same and f64_ok)
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
        del dbt, qt
    print(f"{a.count - bad} of {a.count} cases: both forms equal bit for bit and consistent with float64")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
