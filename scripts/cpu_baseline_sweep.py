#!/usr/bin/env python
"""Host-side sweep behind bench.py's cpu_baseline: the oracle restatement (codeword MLP on torch CPU ops) at several
ATen thread counts and batch sizes on this box's cores.  One JSON line per setting."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from oracle.qinco_oracle import OracleQINCo  # noqa: E402
from qinco_amd import synth_state_dict, synth_vectors  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = BASELINE_CONFIGS[wl]
sd = synth_state_dict(cfg, 1236)
x = synth_vectors(cfg, sd, 1024, seed=4242)
for backend in ("torch", "numpy"):
    o = OracleQINCo.from_config(cfg, sd, backend=backend)
    for threads in (16, 32, 64, 128):
        if backend == "numpy" and threads != 32:
            continue
        torch.set_num_threads(threads)
        for batch in (128, 256, 1024):
            o(x[:32], step="encode")
            t0 = time.perf_counter()
            done = 0
            while done < 1024 and time.perf_counter() - t0 < 12:
                o(x[done:done + batch], step="encode")
                done += batch
            dt = time.perf_counter() - t0
            print(json.dumps({"workload": wl, "backend": backend, "threads": threads, "batch": batch, "vectors": done,
                              "vectors_per_s": done / dt, "gflops": done / dt * cfg.encode_flops_per_vector() / 1e9}), flush=True)
