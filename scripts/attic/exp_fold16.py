#!/usr/bin/env python
"""Round 3: the folded head on the 16-row tile kernel (VAR 1236) against the head computed per row (VAR 1220): QINCo1 at D = 768
(De = D = 768, Dh = 256, L = 12... the reference's qinco1 preset), encode and decode; codes by the tie rule are the tests' business,
here: vec/s and how many code rows differ between the two forms."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_codes, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
cfg = BASELINE_CONFIGS["Q1_768"]
sd = synth_state_dict(cfg, 1236)
n = 16384
x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=2)).cuda()
codes_in = torch.from_numpy(synth_codes(cfg, 262144, seed=9).T.copy().astype(np.uint8)).cuda()
res = {}
for variant in ((48, 1220), None, (48, 1220), None):
    eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"mlp_variant": variant} if variant else None)
    c = eng.encode(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): c = eng.encode(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for _ in range(2): eng.decode(codes_in, check=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(4): eng.decode(codes_in, check=False)
    torch.cuda.synchronize(); dd = time.perf_counter() - t1
    fl = eng.flops_per_vector("encode")
    print(f"{eng.describe().split(' form')[0]}  encode {2*n/dt/1e3:7.3f} k vec/s = {2*n/dt*fl/1e12:6.1f} algorithmic TFLOP/s   "
          f"decode {4*262144/dd/1e6:6.3f} M vec/s", flush=True)
    res[variant] = c.cpu().numpy()
    eng.close()
d = (res[None] != res[(48, 1220)]).any(axis=1)
print(f"rows differing between the folded and the per-row head: {int(d.sum())} of {n}")
