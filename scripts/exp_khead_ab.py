"""KHEAD (VAR bit 4096: the head's per-group rows on the matrix pipe + K-outer first down-projection) against the production
two-workgroups-per-CU instance: codes and candidate distances must be the same bits; throughput at 16 384 and 1024 vectors per call.
    python scripts/exp_khead_ab.py [S C1 ...] [--A 8 16 ...]"""
import sys, time, json, argparse
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
import dataclasses

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["S", "C1"])
ap.add_argument("--beam", type=int, nargs=2, action="append", default=None, help="extra (A, B) overrides to check for identity")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--variants", nargs="*", default=["48:4476", "48:380"], help="P:VAR kernel instances")
args = ap.parse_args()
for wl in args.workloads:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    n = 16384
    x = torch.from_numpy(synth_vectors(cfg, sd, n * args.reps, seed=7)).cuda()
    codes = {}
    for var in args.variants:
        for mb in (16384, 1024):
            eng = QincoEngine(cfg, sd, max_batch=mb, diagnostics={"mlp_variant": tuple(int(v) for v in var.split(":"))})
            if mb == 16384:
                print(wl, var, eng.describe() if hasattr(eng, "describe") else "", flush=True)
            eng.encode(x[:mb], code_dtype=np.uint8); torch.cuda.synchronize()
            best = 0.0
            for _ in range(2):
                t0 = time.perf_counter()
                for r in range(args.reps):
                    for i in range(0, n, mb):
                        c = eng.encode(x[r * n + i: r * n + i + mb], code_dtype=np.uint8)
                torch.cuda.synchronize()
                best = max(best, args.reps * n / (time.perf_counter() - t0))
            print(json.dumps({"workload": wl, "var": var, "batch": mb, "vec_per_s": round(best)}), flush=True)
            if mb == 16384:
                codes[var] = eng.encode(x[:n], code_dtype=np.uint8).cpu().numpy()
            for (A, B) in (args.beam or []):
                eng.set_beam(A, B)
                codes[(var, A, B)] = eng.encode(x[:2048], code_dtype=np.uint8).cpu().numpy() if mb == 16384 else codes.get((var, A, B))
                eng.set_beam(cfg.A, cfg.B)
            eng.close()
    v0 = args.variants[0]
    for var in args.variants[1:]:
        print(json.dumps({"workload": wl, "rows_differing": int((codes[v0] != codes[var]).any(axis=1).sum()), "of": n, "between": [v0, var]}))
        for (A, B) in (args.beam or []):
            print(json.dumps({"workload": wl, "A": A, "B": B, "rows_differing": int((codes[(v0, A, B)] != codes[(var, A, B)]).any(axis=1).sum()), "of": 2048}))
