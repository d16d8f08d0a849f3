#!/usr/bin/env python
"""Where a fused-MLP tile's time goes (experiment build with -DQINCO_TIMELINE, scripts/build_exp_lib.py): per-wave cycle
stamps of the LAST launch of an encode -- entry, ring prologue, head operands assembled, FFN blocks done, epilogue issued,
everything retired -- averaged over the waves, next to the launch's makespan.
    QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_timeline.py S [n]"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors, synth_codes  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg = BASELINE_CONFIGS[wl]
sd = synth_state_dict(cfg, 1236)
import os
eng = QincoEngine(cfg, sd, max_batch=n, split_f16=bool(int(os.environ.get("QINCO_SPLIT_F16", "0"))),
                  diagnostics=({"mlp_variant": (48, int(os.environ["QINCO_VARIANT"]))} if os.environ.get("QINCO_VARIANT") else
                               {"epilogue_select": True} if not os.environ.get("QINCO_NO_SELEP") else None))
x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1)).cuda()
lib = eng.lib
lib.qinco_debug_timeline.restype = C.c_long
lib.qinco_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_long]


def report(label):
    tiles = lib.qinco_debug_timeline(eng._h, None, 0)
    buf = np.zeros((tiles, 8), np.uint64)
    lib.qinco_debug_timeline(eng._h, buf.ctypes.data, tiles)
    t = buf[buf[:, 5] != 0].astype(np.int64)
    d = {"launch": label, "workload": wl, "waves": int(len(t)),
         "ring_prologue": float((t[:, 1] - t[:, 0]).mean()), "head_operands_ready": float((t[:, 2] - t[:, 0]).mean()),
         "ffn_blocks": float((t[:, 3] - t[:, 2]).mean()), "epilogue_issue": float((t[:, 4] - t[:, 3]).mean()),
         "retire": float((t[:, 5] - t[:, 4]).mean()),
         "first_down_phase": float((t[:, 6] - t[:, 2]).mean()) if t[:, 6].any() else None,
         "selep_keys_published": float((t[:, 7] - t[:, 3]).mean()) if t[:, 7].any() else None, "wave_total": float((t[:, 5] - t[:, 0]).mean()),
         "makespan": float(t[:, 5].max() - t[:, 0].min()), "unit": "cycles of s_memtime (100 MHz constant clock x ? -- compare ratios)"}
    # workgroup turn-around on a CU slot: sort waves of (approximately) one slot is unknown; report the distribution of start times
    starts = np.sort(t[:, 0] - t[:, 0].min())
    d["start_time_quantiles"] = [float(np.quantile(starts, q)) for q in (0.1, 0.5, 0.9, 1.0)]
    print(json.dumps(d), flush=True)


for _ in range(2):
    eng.encode(x, code_dtype=np.uint8)
torch.cuda.synchronize()
report("encode, last step (A candidates per group)")
codes = torch.from_numpy(synth_codes(cfg, 16 * n, seed=9).T.copy()).cuda()
eng.decode(codes, check=False)
eng.decode(codes, check=False)
torch.cuda.synchronize()
report("decode, last step (one row per group)")
