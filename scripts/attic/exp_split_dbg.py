import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qinco_amd import synth_state_dict
from qinco_amd.config import QincoConfig
from qinco_amd.engine import QincoEngine
cases = [("tiny L1", QincoConfig(D=32, M=3, K=256, L=1, de=64, dh=128, A=8, B=4)),
         ("tiny L2", QincoConfig(D=32, M=3, K=256, L=2, de=64, dh=128, A=8, B=4)),
         ("tiny L4", QincoConfig(D=32, M=3, K=256, L=4, de=64, dh=128, A=8, B=4)),
         ("C2 L1", QincoConfig(D=128, M=3, K=256, L=1, de=384, dh=384, A=16, B=8)),
         ("C2 L2", QincoConfig(D=128, M=3, K=256, L=2, de=384, dh=384, A=16, B=8)),
         ("C2 L16", QincoConfig(D=128, M=3, K=256, L=16, de=384, dh=384, A=16, B=8))]
rng = np.random.default_rng(0)
for name, cfg in cases:
    sd = synth_state_dict(cfg, 7)
    codes = rng.integers(0, cfg.K, (512, cfg.M))
    outs = {}
    for mode in ("fp32", "split"):
        eng = QincoEngine(cfg, sd, max_batch=1024, split_f16=(mode == "split"))
        outs[mode] = np.asarray(eng.decode(codes))
        eng.close()
    e = np.abs(outs["split"] - outs["fp32"])
    print(json.dumps({"case": name, "max_rel": float(e.max() / np.abs(outs["fp32"]).max()), "rms_rel": float(np.sqrt((e**2).mean()) / np.sqrt((outs["fp32"]**2).mean())),
                      "worst_row": int(e.max(axis=1).argmax()), "rows_bad": int((e.max(axis=1) > 1e-4 * np.abs(outs["fp32"]).max()).sum())}), flush=True)
