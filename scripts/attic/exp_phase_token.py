#!/usr/bin/env python
"""Round-3 experiment: the per-CU phase token of the two-workgroups-per-CU fused-MLP kernels (QINCO_CREATE_PHASE_TOKEN) against
the plain kernel: same codes, encode vec/s at large and small batches."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
for wl in sys.argv[1:] or ["S", "C1", "S_d768"]:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    ref = None
    for tok in (False, True, False, True):
        eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"phase_token": tok})
        out = []
        for n, reps in ((16384, 6), (1024, 60)):
            x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1)).cuda()
            c = eng.encode(x, code_dtype=np.uint8); torch.cuda.synchronize()
            eng.profile_enable(True); eng.profile_read()
            t0 = time.perf_counter()
            for _ in range(reps): c = eng.encode(x, code_dtype=np.uint8)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            pr = eng.profile_read(); eng.profile_enable(False)
            out.append(f"batch {n}: {reps*n/dt/1e3:8.1f} k vec/s (mlp {pr['mlp_ms']/pr['mlp_launches']*1e3:7.1f} us/launch)")
            if n == 16384:
                cc = c.cpu().numpy()
                if ref is None: ref = cc
                same = bool(np.array_equal(ref, cc))
        print(f"{wl} phase_token={tok!s:5s} codes_equal={same}  " + "  ".join(out), flush=True)
        eng.close()
