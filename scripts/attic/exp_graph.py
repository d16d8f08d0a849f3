#!/usr/bin/env python
"""Round-3 experiment: does replaying one encode call as a HIP graph (captured through torch.cuda.CUDAGraph on the stream the
engine enqueues on) shorten the gaps between its ~25 dependent small launches?  qinco2-S at the reference's batch of 1024."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = BASELINE_CONFIGS[wl]
sd = synth_state_dict(cfg, 1236)
eng = QincoEngine(cfg, sd, max_batch=n)
x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1)).cuda()
for _ in range(3):
    ref = eng.encode(x, code_dtype=np.uint8)
torch.cuda.synchronize()


def timed(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_eager = timed(lambda: eng.encode(x, code_dtype=np.uint8))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.encode(x, code_dtype=np.uint8)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = eng.encode(x, code_dtype=np.uint8)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
assert torch.equal(out, ref)
t_graph = timed(g.replay)
print(f"{wl} batch {n}: eager {t_eager * 1e6:.1f} us/call = {n / t_eager:.0f} vec/s; graph replay {t_graph * 1e6:.1f} us/call = {n / t_graph:.0f} vec/s")
