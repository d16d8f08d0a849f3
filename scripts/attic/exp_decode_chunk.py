#!/usr/bin/env python
"""Decode throughput against rows per call (round 3): is the launch-level ramp / tail what keeps the short-MLP decode at 0.77?"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_codes, synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS
for wl in sys.argv[1:] or ["S", "C2"]:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    eng = QincoEngine(cfg, sd, max_batch=16384)
    for n in (98304, 262144, 1048576, 4194304):
        codes = torch.from_numpy(synth_codes(cfg, n, seed=9).T.copy().astype(np.uint8)).cuda()
        eng.decode(codes, check=False); torch.cuda.synchronize()
        eng.profile_enable(True); eng.profile_read()
        t0 = time.perf_counter()
        for _ in range(2): eng.decode(codes, check=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = eng.profile_read(); eng.profile_enable(False)
        print(f"{wl} decode {n:8d} rows/call: {2*n/dt/1e6:8.2f} M vec/s   mlp {pr['mlp_flops']/pr['mlp_ms']/1e9:6.1f} TFLOP/s = {pr['mlp_flops']/pr['mlp_ms']/1e9/157.3:.3f} of peak", flush=True)
    eng.close()
