"""encode_database on a uint8 .bvecs file (the bench's encode_db_bvecs leg) with its knobs varied: where does the host side lose time?
    python scripts/exp_encode_db.py [workload] [n_db]"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from qinco_amd import apply_regime, regime_vectors, synth_state_dict  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402
from qinco_amd.encode_db import encode_database, get_data_memmap  # noqa: E402
from qinco_amd.model import QINCoHIP  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S"
n_db = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
cfg = BASELINE_CONFIGS[wl]
sd = apply_regime(cfg, synth_state_dict(cfg, 1236), "bigann", 1236)
model = QINCoHIP(cfg, sd, max_batch=16384)
block = regime_vectors(cfg, sd, 65536, "bigann", seed=99)
with tempfile.TemporaryDirectory(prefix="qinco_exp_") as tmp:
    path = os.path.join(tmp, "db.bvecs")
    rec = np.empty((len(block), cfg.D + 4), np.uint8)
    rec[:, :4] = np.frombuffer(np.int32(cfg.D).tobytes(), np.uint8)
    with open(path, "wb") as f:
        for i in range(0, n_db, len(block)):
            rec[:, 4:] = np.roll(block, i // len(block), axis=1)
            f.write(rec[: min(len(block), n_db - i)].tobytes())
    db = get_data_memmap(path)
    model(np.ascontiguousarray(db[:16384]), step="encode")
    ram = np.ascontiguousarray(db[:])        # the same rows already in memory (no memmap, no stride)
    for label, src, kw in (("memmap batch 65536", db, dict(batch=65536)), ("memmap batch 262144", db, dict(batch=262144)),
                           ("in-memory batch 262144", ram, dict(batch=262144)),
                           ("memmap batch 262144, no part-file writer threads (numpy at the end)", db, dict(batch=262144, writer_threads=0)),
                           ("memmap batch 262144, keep=False", db, dict(batch=262144, keep=False)),
                           ("memmap batch 262144, compact codes", db, dict(batch=262144, code_dtype="compact"))):
        out = os.path.join(tmp, f"enc_{abs(hash(label))}", "db.npz")
        t0 = time.perf_counter()
        st = {}
        encode_database(model, src, out, K=cfg.K, M=cfg.M, D=cfg.D, stats=st, **kw)
        dt = time.perf_counter() - t0
        print(f"{wl} {label}: {n_db / dt:.0f} vec/s ({dt:.2f} s) " + " ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in st.items()), flush=True)
    xd = torch.from_numpy(ram[:262144]).cuda()
    model.engine.encode(xd[:16384], code_dtype=np.uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        model.engine.encode(xd, code_dtype=np.uint8)
    torch.cuda.synchronize()
    print(f"{wl} resident: {3 * 262144 / (time.perf_counter() - t0):.0f} vec/s")
