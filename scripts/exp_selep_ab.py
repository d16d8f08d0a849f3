import sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
cfg = BASELINE_CONFIGS["S"]
sd = synth_state_dict(cfg, 1236)
x = torch.from_numpy(synth_vectors(cfg, sd, 16384 * 3, seed=7)).cuda()
for name, diag in (("selep", {"epilogue_select": True}), ("two_kernels", {"no_epilogue_select": True})):
    for mb in (16384, 1024):
        eng = QincoEngine(cfg, sd, max_batch=mb, diagnostics=diag)
        eng.encode(x[:mb], code_dtype=np.uint8); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(3):
            for i in range(0, 16384, mb):
                c = eng.encode(x[r * 16384 + i: r * 16384 + i + mb], code_dtype=np.uint8)
        torch.cuda.synchronize()
        print(name, mb, round(3 * 16384 / (time.perf_counter() - t0)), "vec/s")
        eng.close()
