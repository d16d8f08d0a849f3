#!/usr/bin/env python
"""Round 3 (needs scripts/patches/r03_side_stream.patch applied; rejected, see profiles/r03_exp_side_stream.log): a large step's pre-selection (table + top-A: latency chains, matrix pipe ~40 % busy) on a second stream beside
xproj (matrix-pipe bound) -- QINCO_CREATE_PRESEL_SIDE_STREAM against the single-stream order; vec/s decides, codes must not change."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
for wl in sys.argv[1:] or ["S", "IVF_S", "M", "C2"]:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    n = 16384 if cfg.De > 128 else 262144
    x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=2)).cuda()
    res = {}
    for side in (False, True, False, True):
        eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"presel_side_stream": side})
        for _ in range(2): c = eng.encode(x)
        torch.cuda.synchronize()
        reps = 3 if cfg.De > 128 else 8
        t0 = time.perf_counter()
        for _ in range(reps): c = eng.encode(x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault(side, c.cpu().numpy())
        assert np.array_equal(res[side], c.cpu().numpy())
        print(f"{wl:6s} side_stream={side!s:5s} {reps*n/dt/1e3:9.2f} k vec/s", flush=True)
        eng.close()
    assert np.array_equal(res[False], res[True]), "codes changed"
