import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS
wl = sys.argv[1] if len(sys.argv) > 1 else "IVF_S"
cfg = BASELINE_CONFIGS[wl]
sd = synth_state_dict(cfg, 1236)
eng = QincoEngine(cfg, sd, max_batch=8192)
x = torch.from_numpy(synth_vectors(cfg, sd, 8192, seed=42)).cuda()
for _ in range(2): eng.encode(x, code_dtype=np.int32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.encode(x, code_dtype=np.int32)
torch.cuda.synchronize()
print("ms per encode", (time.perf_counter() - t0) / 5 * 1e3)
