#!/usr/bin/env python
"""Experiment (GPU box): the LDS-DMA ring kernels with several workgroups per CU.  QINCO_RING_PAD_KIB sets the dummy
dynamic LDS the launches ask for on top of the 48 KiB ring: 36 -> 1 workgroup per CU (production), 24 / 8 -> 2, 0 -> 3.
For each setting: 3 encodes of a small model at a large batch, rows that differ run to run and from the exclusive run.
    python scripts/exp_coresidency.py [pads ...]
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

MODELS = {
    "id_qinco1": dict(D=32, M=3, K=256, L=2, de=None, dh=64, A=0, B=1, qinco1_mode=True),
    "id_A32": dict(D=32, M=3, K=256, L=2, de=None, dh=64, A=32, B=4, qinco1_mode=False),
    "proj_A8": dict(D=32, M=3, K=256, L=2, de=64, dh=96, A=8, B=4, qinco1_mode=False),
    "S_128": dict(D=128, M=4, K=256, L=2, de=128, dh=256, A=16, B=8, qinco1_mode=False),
}


def worker(model, n, out):
    from qinco_amd import QincoConfig, QincoEngine, synth_state_dict, synth_vectors
    cfg = QincoConfig(**MODELS[model])
    sd = synth_state_dict(cfg, 13)
    x = synth_vectors(cfg, sd, n, seed=999)
    eng = QincoEngine(cfg, sd, max_batch=n)
    runs = [eng.encode(x, return_xhat=True) for _ in range(3)]
    np.savez(out, **{f"c{i}": r[0] for i, r in enumerate(runs)}, **{f"h{i}": r[1] for i, r in enumerate(runs)})


if __name__ == "__main__":
    if sys.argv[1:2] == ["--worker"]:
        worker(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    pads = [int(a) for a in sys.argv[1:]] or [36, 24, 8, 0]
    import tempfile
    tmp = tempfile.mkdtemp()
    for variant in ("", "48,196"):
        for model in MODELS:
            if variant and model == "S_128":
                continue
            n = 4096 if model != "S_128" else 8192
            ref = None
            for pad in pads:
                env = dict(os.environ, QINCO_RING_PAD_KIB=str(pad))
                if variant:
                    env["QINCO_MLP_VARIANT"] = variant
                out = os.path.join(tmp, f"{model}_{variant.replace(',', '_')}_{pad}.npz")
                r = subprocess.run([sys.executable, __file__, "--worker", model, str(n), out], env=env, capture_output=True, text=True)
                if r.returncode:
                    print(json.dumps({"model": model, "variant": variant, "pad": pad, "error": r.stderr[-500:]}))
                    continue
                d = dict(np.load(out))
                if ref is None:
                    ref = d
                rec = {"model": model, "variant": variant or "production", "pad_kib": pad, "n": n,
                       "rows_differ_run_to_run": int(((d["h0"] != d["h1"]).any(1) | (d["h0"] != d["h2"]).any(1)).sum()),
                       "rows_differ_from_first_pad": [int((d[f"h{i}"] != ref["h0"]).any(1).sum()) for i in range(3)]}
                print(json.dumps(rec), flush=True)
