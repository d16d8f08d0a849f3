#!/usr/bin/env python
"""Round-3 stress of the folded 16-row tile kernel on its production shape (QINCo1 at D = 768): distinct large batches, run-to-run
repeats bit for bit, the rows that differ from the per-row-head instance (VAR 1220) judged by the oracle's tie rule, and scattered
rows of every batch against the oracle.
    python tests/sweeps/gpu_stress_fold16.py [--batches 6]"""
import argparse, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import assert_only_near_ties, make_oracle
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=6)
a = ap.parse_args()
cfg = BASELINE_CONFIGS["Q1_768"]
sd = synth_state_dict(cfg, 1236)
oracle = make_oracle(cfg, sd)
new = QincoEngine(cfg, sd, max_batch=16384)
old = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"mlp_variant": (48, 1220)})
rows = rep_bad = differ = judged = 0
t0 = time.time()
rs = np.random.RandomState(1)
for b in range(a.batches):
    xh = synth_vectors(cfg, sd, 16384, seed=9100 + b)
    x = torch.from_numpy(xh).cuda()
    c1, h1 = new.encode(x, return_xhat=True)
    c2, h2 = new.encode(x, return_xhat=True)
    rep_bad += int((c1 != c2).any(dim=1).sum()) + int((h1 != h2).any(dim=1).sum())
    c3 = old.encode(x)
    c1n, c3n = c1.cpu().numpy(), c3.cpu().numpy()
    d = np.nonzero((c1n != c3n).any(axis=1))[0]
    differ += len(d)
    pick = np.unique(np.concatenate([d[:6], rs.choice(16384, 10, replace=False)]))      # differing rows + scattered rows
    want = oracle(xh[pick], step="encode").T
    assert_only_near_ties(oracle, xh[pick], c1n[pick], want, 2e-5, f"folded, batch {b}")
    assert_only_near_ties(oracle, xh[pick], c3n[pick], want, 2e-5, f"per-row head, batch {b}")
    judged += len(pick)
    rows += 16384
    print(f"batch {b}: run-to-run differing rows so far {rep_bad}; {len(d)} rows differ between the folded and the per-row head; "
          f"{len(pick)} rows judged by the oracle ({time.time() - t0:.0f} s)", flush=True)
print(f"Q1_768 {new.describe().split('decode')[0]}: {rows} vectors: run-to-run differing rows {rep_bad}; {differ} rows differ from VAR 1220 "
      f"(all judged ones are ties of the reference algorithm); {judged} rows checked against the oracle")
assert rep_bad == 0
