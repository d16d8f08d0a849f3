#!/usr/bin/env python
"""The lane-pair selection (csrc/select.hpp pair_top_t) alone, on the pre-selection distances of a qinco2-S-shaped model: time per
32-group tile and the share of groups that went to the exact rounds, per T.  python scripts/exp_pair_select.py [groups=131072]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import _lib, synth_state_dict, synth_vectors  # noqa: E402
from qinco_amd.config import preset  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cfg = preset("qinco2-S", D=128, M=3, A=16, B=8)
sd = synth_state_dict(cfg, 5)
x = torch.from_numpy(synth_vectors(cfg, sd, G // 8, seed=1)).cuda()
cb0 = torch.from_numpy(np.asarray(sd["steps.0.codebook.weight"])).cuda()
sub = torch.from_numpy(np.asarray(sd["steps.1.substep.codebook.weight"])).cuda()
xn = (x - torch.from_numpy(np.asarray(sd["data_mean"])).cuda()) / float(sd["data_std"])
d0 = (xn * xn).sum(1, keepdim=True) + (cb0 * cb0).sum(1)[None] - 2 * xn @ cb0.T
top = d0.topk(8, largest=False).indices                                   # step 0: B = 8 beams
r = (xn[:, None, :] - cb0[top]).reshape(-1, cfg.D)                        # residuals of the G groups
d = ((r * r).sum(1, keepdim=True) + (sub * sub).sum(1)[None] - 2 * r @ sub.T).contiguous()
lib = _lib.load()
lib.qinco_debug_pair_select.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
ids = torch.empty((G, 64), dtype=torch.int32, device="cuda")
rounds = torch.empty(G, dtype=torch.int32, device="cuda")
order = torch.sort(d, dim=1, stable=True).indices
import os
variants = {"model": d, "uniform": torch.rand_like(d),
            "first16small": torch.rand_like(d) + (torch.arange(256, device="cuda")[None] >= 16).float()}   # every survivor in lane 0's registers
for name, dv in variants.items():
    dv = dv.contiguous()
    order = torch.sort(dv, dim=1, stable=True).indices
    print("data:", name)
    for coop in [int(v) for v in os.environ.get("COOP", "0,1").split(",")]:
        for T in [int(v) for v in os.environ.get("TS", "1,8,12,13,14,15,16,17").split(",")]:
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.qinco_debug_pair_select(dv.data_ptr(), G, T, coop, ids.data_ptr(), rounds.data_ptr(), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rounds.zero_()
            for _ in range(5):
                _lib.check(lib.qinco_debug_pair_select(dv.data_ptr(), G, T, coop, ids.data_ptr(), rounds.data_ptr(), st))
            e1.record()
            torch.cuda.synchronize()
            got = ids.view(-1)[: G * T].view(G, T)
            wrong = int((got.long() != order[:, :T]).any(1).sum())
            lo, hi = (rounds & 1) != 0, (rounds & 2) != 0
            s_lo, s_hi = (rounds >> 8) & 255, (rounds >> 16) & 255
            tiles = (lo | hi).view(-1, 8 if coop else 32).any(1).float().mean()
            print(f"coop={coop} T={T:2d}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per {G} groups; to the rounds: lane 0 says {float(lo.float().mean()) * 100:.2f} %, "
                  f"lane 1 {float(hi.float().mean()) * 100:.2f} % of the groups = {float(tiles) * 100:.1f} % of the tiles; survivors mean {float(s_lo.float().mean()):.1f}, "
                  f"lanes disagree on S in {int((s_lo != s_hi).sum())} groups; {wrong} rows differ from a stable sort")
