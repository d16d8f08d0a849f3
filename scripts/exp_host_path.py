"""Host-pointer encode (qinco_encode_host: pinned two-deep pipeline) against the device-pointer path on the same rows.
    python scripts/exp_host_path.py [workload] [rows_per_call] [max_batch]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from qinco_amd import QincoEngine, apply_regime, regime_vectors, synth_state_dict  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
cfg = BASELINE_CONFIGS[wl]
sd = apply_regime(cfg, synth_state_dict(cfg, 1236), "bigann", 1236)
eng = QincoEngine(cfg, sd, max_batch=mb)
x = regime_vectors(cfg, sd, n, "bigann", seed=99)          # uint8 (n, D)
xd = torch.from_numpy(x).cuda()
for name, fn in (("device", lambda: eng.encode(xd, code_dtype=np.uint8)), ("host", lambda: eng.encode(x, code_dtype=np.uint8)),
                 ("host_int64", lambda: eng.encode(x, code_dtype=np.int64))):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{wl} {name:10s} rows/call {n}: median {np.median(ts):.2f} ms  min {ts.min():.2f}  -> {n / np.median(ts) * 1e3:.0f} vec/s")
