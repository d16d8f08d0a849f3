"""Small-launch form of the fused MLP (csrc/mlp_small_kernel.hpp) against the 128-row kernels: same bits?  how fast?

For each workload and call size: decode and encode through a handle with the small-launch form and through one created with
no_small_launch; compares the outputs bit for bit, times both (device-resident inputs, HIP events via torch on the call's stream)
and prints one JSON line per case.  Run on the GPU box: python scripts/exp_small_launch.py [--workloads C2,S,C1] [--sizes 1024,...]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    import torch
    from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
    from qinco_amd.config import BASELINE_CONFIGS, QincoConfig
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="tiny,S,C1,C2")
    ap.add_argument("--sizes", default="1024,4096,12288,16384")
    ap.add_argument("--modes", default="decode,encode1")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    PEAK = 157.3e12
    for wl in args.workloads.split(","):
        if wl == "tiny":
            cfg = QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=4)
        elif wl == "tinyid":
            cfg = QincoConfig(D=32, M=4, K=256, L=2, de=None, dh=64, A=8, B=4)
        else:
            cfg = BASELINE_CONFIGS[wl]
        sd = synth_state_dict(cfg, 1234 + len(wl))
        nmax = max(int(v) for v in args.sizes.split(","))
        engs = {}
        # bigf: the 128-row kernels with decode through the folded encode instance -- the association the small form uses
        for name, diag in (("small", {}), ("big", {"no_small_launch": True}), ("bigf", {"no_small_launch": True, "decode_folded": True})):
            engs[name] = QincoEngine(cfg, sd, max_batch=min(nmax, 16384), diagnostics=diag)
        print(json.dumps({"workload": wl, "describe": engs["small"].describe()}), flush=True)
        x = synth_vectors(cfg, sd, nmax, seed=7)
        xd = torch.from_numpy(x).cuda()
        rs = np.random.RandomState(3)
        codes = np.stack([rs.randint(0, k, size=nmax) for k in cfg.K_vals], axis=1).astype(np.int32)
        cd = torch.from_numpy(codes).cuda()
        for n in (int(v) for v in args.sizes.split(",")):
            for mode in args.modes.split(","):
                rec = {"workload": wl, "mode": mode, "n": n}
                outs = {}
                for name, eng in engs.items():
                    if mode == "decode":
                        fn = lambda: eng.decode(cd[:n], check=False)
                        flops = cfg.decode_flops_per_vector()
                    else:
                        B = 1 if mode == "encode1" else cfg.B
                        eng.set_beam(cfg.A, B)
                        fn = lambda: eng.encode(xd[:n], code_dtype=np.int32)
                        flops = cfg.with_search(cfg.A, B).encode_flops_per_vector()
                    out = fn()
                    torch.cuda.synchronize()
                    outs[name] = out.cpu().numpy()
                    ms = timed(fn, args.reps)
                    rec[name + "_ms"] = round(ms, 4)
                    rec[name + "_vec_per_s"] = round(n / ms * 1e3, 1)
                    rec[name + "_frac_algorithmic"] = round(n * flops / (ms * 1e-3) / PEAK, 4)
                a, b = outs["small"], outs["bigf"]
                if mode == "decode":
                    rec["bit_identical"] = bool((a.view(np.uint32) == b.view(np.uint32)).all())
                    rec["max_abs_diff"] = float(np.abs(a - b).max())
                    rec["max_abs"] = float(np.abs(b).max())
                    rec["finite"] = bool(np.isfinite(a).all())
                else:
                    rec["rows_differing"] = int((a != b).any(axis=1).sum())
                rec["speedup"] = round(rec["big_ms"] / rec["small_ms"], 3)
                print(json.dumps(rec), flush=True)
        for e in engs.values():
            e.close()


if __name__ == "__main__":
    main()
