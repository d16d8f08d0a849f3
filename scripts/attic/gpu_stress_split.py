"""Split-fp16 form at scale: 262 144 distinct vectors through qinco2-S, C1 and C2, twice (bitwise repeatable), against the fp32 path
(rows that differ, MSE of both)."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict
from qinco_amd.config import BASELINE_CONFIGS
from qinco_amd.evaluate import sqerr_sum

for wl, n in (("S", 262144), ("C1", 131072), ("C2", 65536)):
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.randn((n, cfg.D), generator=g, device="cuda") * float(sd["data_std"]) + torch.from_numpy(np.asarray(sd["data_mean"])).cuda()
    res = {}
    for mode in ("fp32", "split"):
        eng = QincoEngine(cfg, sd, max_batch=16384, split_f16=(mode == "split"))
        t0 = time.time(); c1 = eng.encode(x, code_dtype=np.uint8); torch.cuda.synchronize(); dt = time.time() - t0
        c2 = eng.encode(x, code_dtype=np.uint8)
        dec = eng.decode(c1, check=False)
        res[mode] = {"codes": c1, "repeatable": bool(torch.equal(c1, c2)), "mse": sqerr_sum(x, dec) / n, "vec_s": n / dt}
        eng.close()
    differ = int((res["fp32"]["codes"] != res["split"]["codes"]).any(dim=1).sum().item())
    print(json.dumps({"workload": wl, "vectors": n, "rows_differing": differ, "fraction": differ / n,
                      **{f"{m}_{k}": res[m][k] for m in res for k in ("repeatable", "mse", "vec_s")}}), flush=True)
    assert res["fp32"]["repeatable"] and res["split"]["repeatable"]
    assert differ <= n // 500 and abs(res["fp32"]["mse"] - res["split"]["mse"]) / res["fp32"]["mse"] < 1e-5
