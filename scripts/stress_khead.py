"""Race detector for KHEAD's hand-counted waits (csrc/mlp_kernel.hpp): the production instance against its twin (VAR 380) over many
large launches -- codes and tracked reconstructions must be the same bits every time.   python scripts/stress_khead.py [S C1 S_d96 ...]"""
import sys, json, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
from qinco_amd.config import BASELINE_CONFIGS

wls = sys.argv[1:] or ["S", "C1", "S_d96", "S_d768", "IVF_S"]
for wl in wls:
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    n = 16384
    reps = 12 if wl in ("S", "IVF_S", "S_d96") else 3
    eng = QincoEngine(cfg, sd, max_batch=n)     # (the default: KHEAD, and the selection in its epilogue where the shape has that instance)
    assert "var=4476" in eng.describe(), eng.describe()
    twin = QincoEngine(cfg, sd, max_batch=n, diagnostics={"mlp_variant": (48, 380)})
    bad = rows = 0
    t0 = time.time()
    for r in range(reps):
        x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1000 + r)).cuda()
        for m in (n, 8192 + 37 * r, 1024):      # full launches, a partly filled last workgroup, the reference's batch
            ck, hk = eng.encode(x[:m], return_xhat=True)
            ct, ht = twin.encode(x[:m], return_xhat=True)
            bad += int((ck != ct).any(dim=1).sum()) + int((hk != ht).any(dim=1).sum())
            rows += m
    print(json.dumps({"workload": wl, "rows_compared": rows, "rows_differing_in_codes_or_xhat_bits": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
    eng.close(); twin.close()
