#!/usr/bin/env python
"""Round-3 stress of the ring-groups-of-8 production kernels: many distinct large batches through the production instance and
through the round-2 instance (groups of 4) of the same shape, codes and tracked reconstructions compared bit for bit; plus
run-to-run repeats.  Any ordering hole in the hand-counted vmcnt / barrier protocol shows up as a differing row.
    python scripts/gpu_stress_g8.py [C2 C4 M Q1_768] [--batches 12]"""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["C2", "C4", "M", "Q1_768"])
ap.add_argument("--batches", type=int, default=12)
a = ap.parse_args()
OLD = {"C2": (48, 124), "C4": None, "M": (48, 124), "Q1_768": (48, 196), "C3": (48, 124)}   # C4's round-2 instance is not compiled in
for wl in a.workloads:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    new = QincoEngine(cfg, sd, max_batch=16384)
    old = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"mlp_variant": OLD[wl]}) if OLD.get(wl) else None
    rows = bad = rep_bad = 0
    t0 = time.time()
    for b in range(a.batches):
        x = torch.from_numpy(synth_vectors(cfg, sd, 16384, seed=9000 + b)).cuda()
        c1, h1 = new.encode(x, return_xhat=True)
        c2, h2 = new.encode(x, return_xhat=True)
        rep_bad += int((c1 != c2).any(dim=1).sum()) + int((h1 != h2).any(dim=1).sum())
        if old is not None:
            c3, h3 = old.encode(x, return_xhat=True)
            bad += int((c1 != c3).any(dim=1).sum()) + int((h1 != h3).any(dim=1).sum())
        rows += 16384
    print(f"{wl}: {new.describe().split('decode')[0]} {rows} vectors in {a.batches} batches: run-to-run differing rows {rep_bad}, "
          f"rows differing from the groups-of-4 instance {bad if old else 'n/a'}  ({time.time() - t0:.0f} s)", flush=True)
    new.close()
    if old: old.close()
