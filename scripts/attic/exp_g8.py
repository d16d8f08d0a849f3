#!/usr/bin/env python
"""Round-3 experiment: shared weight ring with a barrier / refill every 8 fragments (VAR bit 1024) against every 4."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS, preset
EXTRA = {"L_d96": preset("qinco2-L", D=96, M=8, B=8), "L_d256": preset("qinco2-L", D=256, M=8, B=8), "Q1_d256": preset("qinco1", D=256, M=8),
         "M": BASELINE_CONFIGS["M"]}
def parse(w):   # "C2" (its ring-groups-of-8 variant) or "C2@96,1148" (an explicit P,VAR)
    if "@" in w:
        return w.split("@")[0], tuple(int(v) for v in w.split("@")[1].split(","))
    return w, (48, 1220) if w == "Q1_768" else (48, 1148)


WL = [("C2", (48, 1148)), ("S", (48, 1404)), ("C1", (48, 1404))] if len(sys.argv) < 2 else [parse(w) for w in sys.argv[1:]]
for wl, var in WL:
    cfg = EXTRA.get(wl) or BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    ref = None
    for v in (None, var, None, var):
        eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"mlp_variant": v} if v else None)
        x = torch.from_numpy(synth_vectors(cfg, sd, 16384, seed=1)).cuda()
        c, h = eng.encode(x, return_xhat=True); torch.cuda.synchronize()
        eng.profile_enable(True); eng.profile_read()
        reps = 2 if wl.startswith("Q1_768") else 4 if wl != "S" else 12
        t0 = time.perf_counter()
        for _ in range(reps): eng.encode(x, code_dtype=np.uint8)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = eng.profile_read()
        cc = (c.cpu().numpy(), h.cpu().numpy())
        if ref is None: ref = cc
        print(f"{wl} {eng.describe().split('decode')[0]} {reps*16384/dt/1e3:9.2f} k vec/s  mlp {pr['mlp_ms']/pr['mlp_launches']*1e3:9.1f} us/launch  "
              f"bits equal: {np.array_equal(ref[0], cc[0]) and np.array_equal(ref[1], cc[1])}", flush=True)
        eng.close()
