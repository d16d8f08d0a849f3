#!/usr/bin/env python
"""Per-CU schedule of the LAST fused-MLP launch of an encode (experiment build with -DQINCO_TIMELINE): every tile's cycle stamps and
the CU / SIMD / wave slot it ran on -> for each SIMD the two residents' phases: how long both are outside their FFN blocks (matrix
pipe idle), both inside (sharing), the gap between a wave's retirement and the next wave's entry on the same slot.
    QINCO_VARIANT=380 QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_cu_schedule.py S [n]"""
import ctypes as C, json, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from qinco_amd import QincoEngine, synth_state_dict, synth_vectors  # noqa: E402
from qinco_amd.config import BASELINE_CONFIGS  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg = BASELINE_CONFIGS[wl]
sd = synth_state_dict(cfg, 1236)
diag = {"mlp_variant": (48, int(os.environ["QINCO_VARIANT"]))} if os.environ.get("QINCO_VARIANT") else None
eng = QincoEngine(cfg, sd, max_batch=n, diagnostics=diag)
x = torch.from_numpy(synth_vectors(cfg, sd, n, seed=1)).cuda()
lib = eng.lib
lib.qinco_debug_timeline.restype = C.c_long
lib.qinco_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
for _ in range(2):
    eng.encode(x, code_dtype=np.uint8)
torch.cuda.synchronize()
tiles = lib.qinco_debug_timeline(eng._h, None, 0)
buf = np.zeros((tiles, 8), np.uint64)
lib.qinco_debug_timeline(eng._h, buf.ctypes.data, tiles)
t = buf[buf[:, 5] != 0]
where = t[:, 7]
simd_key = (where >> 32) * (1 << 20) + ((where >> 4) & 0xfff)          # XCC, SE/SH, CU, SIMD
slot = where & 0xf
T = t[:, :7].astype(np.int64)
out = {"workload": wl, "variant": os.environ.get("QINCO_VARIANT", "production"), "tiles": int(len(T)), "simds": int(len(np.unique(simd_key)))}
idle_both, share_both, alone, gaps, periods, span = [], [], [], [], [], []
ffn_lo = 6 if False else 2   # FFN blocks = stamps 2 .. 3 (KHEAD: the streaming first down-projection starts at stamp 2 as well)
for k in np.unique(simd_key):
    idx = np.nonzero(simd_key == k)[0]
    tt = T[idx]
    o = np.argsort(tt[:, 0])
    tt, sl = tt[o], slot[idx][o]
    t0, t1 = tt[:, 0].min(), tt[:, 5].max()
    span.append(t1 - t0)
    # per wave slot: chain of tiles, gaps between retire and next entry
    for s in np.unique(sl):
        c = tt[sl == s]
        if len(c) > 1:
            gaps.extend((c[1:, 0] - c[:-1, 5]).tolist())
            periods.extend((c[1:, 0] - c[:-1, 0]).tolist())
    # sweep: number of waves inside FFN at each instant (events)
    ev = [(a, +1) for a in tt[:, ffn_lo]] + [(b, -1) for b in tt[:, 3]]
    ev.sort()
    cur, last, acc = 0, t0, {0: 0, 1: 0, 2: 0, 3: 0}
    for tm, dv in ev:
        acc[min(cur, 3)] += tm - last
        last = tm
        cur += dv
    acc[0] += t1 - last
    idle_both.append(acc[0]); alone.append(acc[1]); share_both.append(acc[2] + acc[3])
tot = float(np.sum(span))
out.update({"pipe_has_no_ffn_wave": float(np.sum(idle_both)) / tot, "one_ffn_wave": float(np.sum(alone)) / tot,
            "two_ffn_waves": float(np.sum(share_both)) / tot, "slot_gap_cycles_mean": float(np.mean(gaps)), "slot_gap_cycles_p90": float(np.quantile(gaps, 0.9)),
            "slot_period_mean": float(np.mean(periods)), "span_mean": float(np.mean(span)), "span_min": float(np.min(span)), "span_max": float(np.max(span)),
            "tile_cycles_mean": float((T[:, 5] - T[:, 0]).mean()), "ffn_cycles_mean": float((T[:, 3] - T[:, 2]).mean()),
            "tiles_per_simd_mean": float(len(T) / len(np.unique(simd_key)))})
print(json.dumps(out), flush=True)
np.save(os.environ.get("QINCO_SCHEDULE_OUT", "/tmp/schedule.npy"), buf[buf[:, 5] != 0])
