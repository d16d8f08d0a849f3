#!/usr/bin/env python
"""Split of the pre-selection kernel's time into table and selection: the same launches with T = A in {1, 2, 8, 16, 32}
(run under rocprofv3 --kernel-trace; scripts/rocpd_summary.py groups by launch size)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors  # noqa: E402
from qinco_amd.config import preset  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for A in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 8, 16, 32)):
    cfg = preset("qinco2-S", D=128, M=3, A=A, B=8)
    sd = synth_state_dict(cfg, 5)
    eng = QincoEngine(cfg, sd, max_batch=n)
    x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1)).cuda()
    for _ in range(3):
        eng.encode(x, code_dtype=np.uint8)
    torch.cuda.synchronize()
    eng.close()
