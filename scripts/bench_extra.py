#!/usr/bin/env python
"""Extra measurements next to bench.py (not part of the driver contract): encode AND decode throughput for each
BASELINE.json workload (C1..C4), greedy and beam, with the fused-MLP kernel's roofline fraction.  One JSON line
per (workload, mode).  Usage: python scripts/bench_extra.py [C1 C2 ...] [--batch N] [--steps K]"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["C1", "C2", "C3", "C4"])
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--beams", type=int, nargs="*", default=None, help="override B (e.g. 1 8)")
    ap.add_argument("--host", action="store_true", help="also time the host-buffer entry points (PCIe inclusive)")
    ap.add_argument("--decode-rows", type=int, default=0, help="rows per decode call (default 16 x batch: the 128-row kernels; e.g. 12288: "
                    "the small-launch form at the reference's re-rank batch)")
    args = ap.parse_args()
    import torch
    from qinco_amd import QincoEngine, synth_codes, synth_state_dict, synth_vectors
    from qinco_amd.config import BASELINE_CONFIGS
    dev = torch.device("cuda", 0)
    for wl in args.workloads:
        cfg0 = BASELINE_CONFIGS[wl]
        sd = synth_state_dict(cfg0, 1236)
        beams = args.beams or [cfg0.B]
        eng = QincoEngine(cfg0, sd, max_batch=args.batch)
        x = torch.from_numpy(synth_vectors(cfg0, sd, args.batch, seed=42)).to(dev)
        for B in beams:
            if cfg0.A == 0 and B > 1:
                continue
            eng.set_beam(B=B)
            for mode in ("encode", "decode"):
                if mode == "decode" and B != beams[0]:
                    continue
                nvec = args.batch if mode == "encode" else (args.decode_rows or args.batch * 16)
                codes = torch.from_numpy(synth_codes(cfg0, nvec, seed=9).T.copy()).to(dev)
                cdt = np.int32 if cfg0.ivf else np.uint8
                fn = (lambda: eng.encode(x, code_dtype=cdt)) if mode == "encode" else (lambda: eng.decode(codes))
                fn()
                torch.cuda.synchronize()
                eng.profile_enable(True)
                eng.profile_read()
                t0 = time.perf_counter()
                reps = args.steps if mode == "encode" else max(args.steps, 12)   # (a decode call is ~5 ms: a loop shorter than
                for _ in range(reps):                                                 # ~50 ms measures the clock ramp)
                    fn()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                pr = eng.profile_read()
                eng.profile_enable(False)
                tf = pr["mlp_flops"] / (pr["mlp_ms"] * 1e-3) / 1e12 if pr["mlp_ms"] else 0.0
                tfx = pr["mlp_flops_executed"] / (pr["mlp_ms"] * 1e-3) / 1e12 if pr["mlp_ms"] else 0.0
                rec = {"workload": wl, "mode": mode, "A": eng.A, "B": B, "vectors_per_step": nvec,
                       "vectors_per_s": reps * nvec / dt, "us_per_vector": dt / (reps * nvec) * 1e6,
                       "gflop_per_vector": eng.flops_per_vector(mode) / 1e9,
                       "frac": tfx / PEAK, "mlp_executed_tflops": tfx,        # executed FLOPs / time / peak: <= 1 (bench.py roofline.frac)
                       "frac_algorithmic": tf / PEAK, "mlp_algorithmic_tflops": tf,
                       "mlp_share_of_time": pr["mlp_ms"] * 1e-3 / dt}
                if cfg0.ivf and mode == "encode":
                    st = eng.ivf_last_stats()
                    rec["ivf_exact_candidates_per_vector"] = st["candidates"] / nvec
                    rec["ivf_fell_back"] = st["fell_back"]
                print(json.dumps(rec), flush=True)
        if args.host:   # numpy in / numpy out through qinco_encode_host: H2D of x and D2H of the codes are inside the time
            eng.set_beam(B=beams[-1])
            xh = x.cpu().numpy()
            eng.encode(xh)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.encode(xh)
            dt = time.perf_counter() - t0
            print(json.dumps({"workload": wl, "mode": "encode_host(PCIe inclusive)", "B": beams[-1],
                              "vectors_per_s": args.steps * args.batch / dt}), flush=True)
        eng.close()


def lut_bandwidth():
    """f4: HBM-side rate of the look-up decoders (bytes = codes in + J gathered rows + row out, per vector)."""
    import torch
    from qinco_amd.lut import LutDecoder
    rs = np.random.RandomState(0)
    dev = torch.device("cuda", 0)
    for name, J, Kt, D, mul, n in (("AQ 8x256 d128", 8, 256, 128, 1, 1 << 22), ("pairwise 16x65536 d128", 16, 65536, 128, 256, 1 << 21)):
        tab = rs.randn(J, Kt, D).astype(np.float32)
        if mul == 1:
            dec, Mc = LutDecoder(tab, a=np.arange(J)), J
        else:
            dec, Mc = LutDecoder(tab, a=rs.randint(0, 13, J), b=rs.randint(0, 13, J), mul=mul), 13
        codes = torch.from_numpy(rs.randint(0, 256, (n, Mc)).astype(np.uint8)).to(dev)
        dec(codes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dec(codes)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        byts = n * (Mc + J * D * 4 + D * 4)
        print(json.dumps({"workload": "lut_decode " + name, "vectors_per_s": n / dt, "GB_per_s": byts / dt / 1e9,
                          "bytes_per_vector": byts / n}), flush=True)
        dec.close()


def knn_rate():
    """f3: brute-force top-100 at the reference's small-db scale (bigann1M: N = 1e6, Q = 1e4, D = 128) and D = 768, through the
    filtered form (no distance table in HBM; round 5) and the unfiltered table + radix-select form; ids must be equal."""
    import torch
    from qinco_amd.search import KnnSearcher
    dev = torch.device("cuda", 0)
    for D, N, Q in ((128, 1_000_000, 10_000), (768, 1_000_000, 2_048)):
        g = torch.Generator(device=dev).manual_seed(0)
        db = torch.randn(N, D, device=dev, generator=g)
        q = torch.randn(Q, D, device=dev, generator=g)
        got = {}
        for form, filtered, roles in (("filtered", True, False), ("filtered, two roles (opt-in)", True, True), ("table", False, False)):
            if roles and D > 128:
                continue
            knn = KnnSearcher(D, filtered=filtered, roles=roles)
            if os.environ.get("KNN_QUERY_BYTES"):
                knn.lib.qinco_knn_set_option(knn._h, 2, int(os.environ["KNN_QUERY_BYTES"]))
            knn.search(db, q, k=100)   # warm-up at the full shape (allocates the scratch)
            torch.cuda.synchronize()
            knn.roles_stats()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ids = knn.search(db, q, k=100)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            got[form] = ids
            print(json.dumps({"workload": f"knn top-100 N={N} Q={Q} D={D}", "form": form,
                              "seconds": best, "queries_per_s": Q / best, "table_tflops": 2.0 * D * N * Q / best / 1e12,
                              "frac_fp32_mfma": 2.0 * D * N * Q / best / 1e12 / PEAK, "pairs_per_s": N * Q / best,
                              **knn.last_stats(), "role_workgroups": knn.roles_stats()}), flush=True)
            knn.close()
            assert ids.shape == (Q, 100)
        for form in got:
            assert torch.equal(got[form], got["table"]), f"{form} differs from the table form"
        del db, q, got


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lut":
        lut_bandwidth()
    elif len(sys.argv) > 1 and sys.argv[1] == "knn":
        knn_rate()
    else:
        main()
