import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from qinco_amd import QincoEngine, synth_codes, synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS
cfg = BASELINE_CONFIGS["S"]
sd = synth_state_dict(cfg, 1236)
eng = QincoEngine(cfg, sd, max_batch=16384)
for n, reps in ((262144, 2), (262144, 8), (262144, 32), (1048576, 8), (65536, 64)):
    codes = torch.from_numpy(synth_codes(cfg, n, seed=9).T.copy().astype(np.uint8)).cuda()
    eng.decode(codes, check=False); torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_read()
    t0 = time.perf_counter()
    for _ in range(reps): eng.decode(codes, check=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = eng.profile_read(); eng.profile_enable(False)
    print(f"S decode {n:8d} rows/call x {reps:2d}: {reps*n/dt/1e6:8.2f} M vec/s   mlp {pr['mlp_flops']/pr['mlp_ms']/1e9/157.3:.3f} of peak", flush=True)
