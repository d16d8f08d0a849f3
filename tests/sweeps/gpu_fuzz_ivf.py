#!/usr/bin/env python
"""Adversarial inputs for the IVF coarse assignment (csrc/ivf_f16_kernel.hpp: fp16 matrix-core filter, exact fp32 decision):
the filter's error bound is claimed rigorous, so the arg-min must survive whatever the data looks like.

    python tests/sweeps/gpu_fuzz_ivf.py --out gpurun_out/fuzz_ivf.jsonl         # GPU box

Per case: a coarse codebook and vectors drawn to stress one thing -- near-duplicate centroids (distance gaps at rounding
level), exact duplicates (ties: the lower id must win, like argmin), magnitudes from 1e-4 (fp16 subnormals) to 3e4 (the top
of the fp16 range) and beyond it (the exact kernel must take over), vectors ON centroids, heavy tails, one giant outlier
centroid (inflates the bound: many candidates), byte rows, codebooks whose size is not a multiple of 32.  The step-0 code of
every vector is compared with IVFBook.quantize's arg-min of |x|^2 + |c|^2 - 2 x.c in fp32 (qinco_base.py:146-163,
utils.py:336-346, restated in numpy here): it must be that arg-min or sit within a relative 2e-5 of it in the oracle's table
(the bar of tests/test_hip_parity.py), and on exact duplicates it must be the lower id.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

KINDS = ["gauss", "near_dup", "exact_dup", "on_centroid", "heavy_tail", "outlier", "tiny", "huge", "beyond_fp16", "x_beyond_fp16", "bytes"]


def make_case(rs: np.random.RandomState, kind: str, D: int, K: int, n: int):
    scale = 1.0
    c = rs.randn(K, D).astype(np.float32)
    if kind == "near_dup":          # pairs of centroids a few ulps apart
        half = K // 2
        c[half:2 * half] = c[:half] * (1.0 + 1e-6 * rs.randn(half, 1).astype(np.float32))
    elif kind == "exact_dup":
        half = K // 2
        c[half:2 * half] = c[:half]
    elif kind == "heavy_tail":
        c = (rs.standard_cauchy((K, D)).clip(-200, 200)).astype(np.float32)
    elif kind == "outlier":
        c[rs.randint(K)] *= 3000.0
    elif kind == "tiny":
        scale = 1e-4
    elif kind == "huge":
        scale = 3e4 / float(np.abs(c).max())          # |c| up to 3e4: inside the fp16 range, barely
    elif kind == "beyond_fp16":
        scale = 2e5 / float(np.abs(c).max())          # the filter cannot represent these: the exact kernel must do the batch
    c = (c * scale).astype(np.float32)
    pick = rs.randint(0, K, n)
    if kind == "on_centroid":
        x = c[pick].copy()
    elif kind == "x_beyond_fp16":   # inputs the filter cannot represent: its overflow flag must hand the batch to the exact kernel
        x = (1e5 * rs.randn(n, D)).astype(np.float32)
    elif kind == "heavy_tail":
        x = (c[pick] + rs.standard_cauchy((n, D)).clip(-50, 50)).astype(np.float32)
    else:
        x = (c[pick] + 0.3 * scale * rs.randn(n, D)).astype(np.float32)
    mean, std = np.zeros(D, np.float32), np.float32(1.0)
    if kind == "bytes":             # SIFT-like rows: centroids in normalised space, inputs uint8
        mean, std = (20 + 40 * rs.rand(D)).astype(np.float32), np.float32(36.5888)
        x = np.clip(np.rint(x * std + mean), 0, 255).astype(np.uint8)
    return c, x, mean, std


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from qinco_amd import QincoConfig, QincoEngine, synth_state_dict
    rs = np.random.RandomState(a.seed)
    log = open(a.out, "w") if a.out else None
    failures = 0
    cases = []
    for kind in KINDS:
        for D, K in ((128, 4096), (32, 1000), (96, 65536), (256, 2048), (768, 1024), (128, 100)):
            cases.append((kind, D, K, min(int(rs.choice([33, 500, 2000])), 40_000_000 // K)))     # (the oracle's table: n x K floats)
    for i, (kind, D, K, n) in enumerate(cases):
        rec = dict(i=i, kind=kind, D=D, ivf_K=K, n=n)
        t0 = time.time()
        try:
            cfg = QincoConfig(D=D, M=1, K=256, L=1, de=None, dh=256 if D != 32 else 64, A=0, B=1, qinco1_mode=True, ivf_K=K)
            sd = synth_state_dict(cfg, 100 + i)
            c, x, mean, std = make_case(rs, kind, D, K, n)
            sd["steps.0.ivf_centroids.weight"] = c
            sd["data_mean"], sd["data_std"] = mean, std
            eng = QincoEngine(cfg, sd, max_batch=1024)
            got = eng.encode(x)[:, 0]
            st = eng.ivf_last_stats()
            rec.update(describe=eng.describe().split("ivf=")[1], candidates_per_vector=round(st["candidates"] / min(n, 1024), 2),
                       fell_back=st["fell_back"])
            xn = ((x.astype(np.float32) - mean) / std).astype(np.float32)
            d = (xn * xn).sum(1, dtype=np.float32)[:, None] + (c * c).sum(1, dtype=np.float32)[None] - np.float32(2) * (xn @ c.T)
            want = d.argmin(1)
            dmin = d[np.arange(n), want]
            dgot = d[np.arange(n), got]
            diff = np.nonzero(got != want)[0]
            # a differing row must be a tie of the reference's own table: within a relative 2e-5 of the minimum (the bar of
            # tests/test_hip_parity.py), or -- where the distance itself is ~0 (a vector on a centroid) -- within the fp32
            # evaluation error of the expression, 4 D 2^-24 of its terms' scale |x|^2 + |c|^2
            terms = (xn * xn).sum(1) + (c[want] * c[want]).sum(1) + 1e-30
            rel_d = (dgot - dmin) / np.maximum(np.abs(dgot), 1e-30)
            rel_terms = (dgot - dmin) / terms
            tie = (rel_d < 2e-5) | (rel_terms < 4 * D * 2.0 ** -24)
            rec.update(rows_differing=int(len(diff)), worst_rel_gap=float(rel_d[diff].max()) if len(diff) else 0.0,
                       worst_gap_over_terms=float(rel_terms[diff].max()) if len(diff) else 0.0)
            assert tie[diff].all(), f"rows {diff[~tie[diff]][:5]} are not ties: {rel_d[diff].max():.3e} / {rel_terms[diff].max():.3e}"
            if kind == "exact_dup":     # exact ties: argmin keeps the first (lower) id of a duplicated pair
                half = K // 2
                assert not ((got >= half) & (got < 2 * half)).any(), "a duplicate's higher id was returned"
            assert got.max() < K
            if kind == "beyond_fp16":
                assert "fp32" in rec["describe"] and "fp16" not in rec["describe"]
            if kind == "x_beyond_fp16":
                assert st["fell_back"], "inputs beyond the fp16 range did not raise the filter's flag"
            eng.close()
            rec["ok"] = True
        except Exception as e:  # noqa: BLE001
            rec.update(ok=False, error=f"{type(e).__name__}: {str(e)[:300]}")
            failures += 1
        rec["seconds"] = round(time.time() - t0, 2)
        line = json.dumps(rec)
        print(line, flush=True)
        if log:
            log.write(line + "\n")
            log.flush()
    print(f"{len(cases) - failures} of {len(cases)} IVF cases agree with the oracle", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
