#!/usr/bin/env python
"""Round 3: decode through the shape's un-folded instance against decode through the folded encode instance (vec/s decides)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_codes, synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS, preset
CFG = {"L_d96": preset("qinco2-L", D=96, M=8, B=8), "L_d256": preset("qinco2-L", D=256, M=8, B=8), "Q1_d256": preset("qinco1", D=256, M=8),
       "C2": BASELINE_CONFIGS["C2"], "C4": BASELINE_CONFIGS["C4"], "M": BASELINE_CONFIGS["M"], "S": BASELINE_CONFIGS["S"],
       "C1": BASELINE_CONFIGS["C1"], "S_d768": BASELINE_CONFIGS["S_d768"], "S_d96": BASELINE_CONFIGS["S_d96"]}
for wl in sys.argv[1:] or list(CFG):
    cfg = CFG[wl]
    sd = synth_state_dict(cfg, 1236)
    n = 524288 if cfg.De > 128 else 2097152
    codes = torch.from_numpy(synth_codes(cfg, n, seed=9).T.copy().astype(np.uint8)).cuda()
    for folded in (False, True):
        eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"decode_folded": folded})
        for _ in range(2): eng.decode(codes, check=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6): eng.decode(codes, check=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{wl:8s} {eng.describe().split('form')[0]} decode_folded={folded!s:5s} {6*n/dt/1e6:7.3f} M vec/s", flush=True)
        eng.close()
