import sys, time
sys.path.insert(0, ".")
import torch
from qinco_amd import QincoEngine, synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS
torch.cuda.init()
for wl in ("C2", "C3", "C1", "S", "IVF_S"):
    cfg = BASELINE_CONFIGS[wl]
    t0 = time.perf_counter(); sd = synth_state_dict(cfg, 1236); t1 = time.perf_counter()
    eng = QincoEngine(cfg, sd, max_batch=16384); t2 = time.perf_counter()
    eng.close()
    e2 = QincoEngine(cfg, sd, max_batch=16384, split_f16=True) if wl in ("C2", "S") else None
    t3 = time.perf_counter()
    print(f"{wl}: synth {t1-t0:.2f} s, create {t2-t1:.2f} s" + (f", create split (with calibration twin) {t3-t2:.2f} s" if e2 else ""))
