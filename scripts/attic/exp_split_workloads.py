import json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT))
from qinco_amd import synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS
from qinco_amd.engine import QincoEngine
for wl, n in (("C1", 16384), ("S", 16384), ("C4", 16384), ("M", 16384), ("IVF_S", 16384)):
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((n, cfg.D)).astype(np.float32) * np.float32(sd["data_std"]) + sd["data_mean"]).cuda()
    res = {}
    for mode in ("fp32", "split"):
        eng = QincoEngine(cfg, sd, max_batch=n, split_f16=(mode == "split"))
        c = eng.encode(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            c = eng.encode(x)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        res[mode] = (n / dt, c.cpu().numpy())
        eng.close()
    diff = int((res["fp32"][1] != res["split"][1]).any(axis=1).sum())
    print(json.dumps({"workload": wl, "fp32_vec_s": res["fp32"][0], "split_vec_s": res["split"][0], "speedup": res["split"][0] / res["fp32"][0], "rows_differing": diff, "rows": n}), flush=True)
