"""GPU: split-fp16 FFN blocks against the fp32 kernel and the reference goldens (codes), and throughput of both."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from conftest import golden_cases, load_golden, ref_codes
from qinco_amd import synth_state_dict
from qinco_amd.engine import QincoEngine

names = sys.argv[1:] or ["C2_qinco2L_8x8_b8", "C2_qinco2L_8x8_b1", "C3_qinco2L_16x8_b8", "C2_qinco2L_8x8_b32"]
for name in names:
    cfg, seed = golden_cases()[name]
    sd = synth_state_dict(cfg, seed)
    g = load_golden(name)
    x = g["x"]
    want = ref_codes(g)
    out = {"case": name, "n": len(x)}
    for mode in ("fp32", "split"):
        eng = QincoEngine(cfg, sd, max_batch=1024, split_f16=(mode == "split"))
        codes = eng.encode(x)[0] if isinstance(eng.encode(x), tuple) else eng.encode(x)
        codes = np.asarray(codes)
        bad = int((codes != want).any(axis=1).sum())
        dec = np.asarray(eng.decode(want))
        ref = g["decoded"] if "decoded" in g else None
        out[mode] = {"rows_differing_from_reference": bad}
        if ref is not None:
            out[mode]["decode_max_rel_err"] = float(np.abs(dec - ref).max() / np.abs(ref).max())
        dec2 = np.asarray(eng.decode(codes))
        out[mode]["mse"] = float(((x - dec2) ** 2).sum(-1).mean())
        eng.close()
    print(json.dumps(out), flush=True)

# throughput at the bench shape
from qinco_amd.config import BASELINE_CONFIGS
cfg = BASELINE_CONFIGS["C2"]
sd = synth_state_dict(cfg, 1236)
n = 16384
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((n, cfg.D)).astype(np.float32) * sd["data_std"] + sd["data_mean"]).cuda()
for mode in ("fp32", "split"):
    eng = QincoEngine(cfg, sd, max_batch=n, split_f16=(mode == "split"))
    eng.encode(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        c = eng.encode(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(json.dumps({"bench": "C2 encode 16384", "mode": mode, "vec_per_s": n / dt, "ms": dt * 1e3}), flush=True)
    eng.close()
