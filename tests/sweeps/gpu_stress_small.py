"""Stress of the small-launch form of the fused MLP (csrc/mlp_small_kernel.hpp): many distinct rows through every NT and both
kernels (decode in one launch, the greedy encode step), against the 128-row kernels bit for bit and run to run.  A ring-protocol
or barrier race shows up as a handful of differing rows in hundreds of thousands (the fence that keeps hosted gathers behind their
DMA was found that way).    python tests/sweeps/gpu_stress_small.py [--rows 200000] [--reps 3]  -> one JSON line per workload"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def main():
    import torch
    from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
    from qinco_amd.config import BASELINE_CONFIGS, QincoConfig
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workloads", default="tiny,S,C1,C2,M,C4,S_d96,S_d768")
    args = ap.parse_args()
    for wl in args.workloads.split(","):
        cfg = QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=4) if wl == "tiny" else BASELINE_CONFIGS[wl]
        sd = synth_state_dict(cfg, 99)
        n = args.rows if cfg.mlp_flops_per_row() < 3e6 else args.rows // 4
        rs = np.random.RandomState(5)
        codes = torch.from_numpy(np.stack([rs.randint(0, k, size=n) for k in cfg.K_vals], axis=1).astype(np.int32)).cuda()
        small = QincoEngine(cfg, sd, max_batch=16384)
        big = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"no_small_launch": True, "decode_folded": True})
        want = big.decode(codes, check=False).cpu().numpy().view(np.uint32)      # one call: the 128-row kernels, step by step
        rec = {"workload": wl, "describe": small.describe(), "decode_rows": n, "decode": {}}
        for call in (16, 1000, 4096, 8192, 12288, 16384, 20000):
            bad = rerun = 0
            first = None
            for rep in range(args.reps):
                got = torch.cat([small.decode(codes[i:i + call], check=False) for i in range(0, n, call)]).cpu().numpy().view(np.uint32)
                bad = max(bad, int((got != want).any(axis=1).sum()))
                if first is None:
                    first = got
                else:
                    rerun = max(rerun, int((got != first).any(axis=1).sum()))
            rec["decode"][str(call)] = {"rows_differing_from_128_row_kernels": bad, "rows_differing_run_to_run": rerun}
        if cfg.A > 0:       # greedy encode: n x A rows per step through the small form's encode-step kernel in calls of <= 1024 vectors
            ne = min(n // 4, 40000)
            x = torch.from_numpy(synth_vectors(cfg, sd, ne, seed=6)).cuda()
            small.set_beam(cfg.A, 1)
            big.set_beam(cfg.A, 1)
            cb, hb = big.encode(x, return_xhat=True)
            worst = 0
            for call in (64, 1000, 1024):
                for rep in range(args.reps):
                    parts = [small.encode(x[i:i + call], return_xhat=True) for i in range(0, ne, call)]
                    cs, hs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
                    worst = max(worst, int(((cs != cb).any(dim=1) | (hs.view(torch.int32) != hb.view(torch.int32)).any(dim=1)).sum()))
            rec["greedy_encode"] = {"vectors": ne, "rows_differing_from_128_row_kernels": worst}
        print(json.dumps(rec), flush=True)
        small.close()
        big.close()


if __name__ == "__main__":
    main()
