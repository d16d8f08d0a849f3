"""Feasibility study (CPU, numpy): how far from the fp32 reference does a split-fp16 evaluation of the residual blocks land?

x = xhi + xlo (two fp16 values), W = Whi + Wlo; x.W ~= xhi.Whi + xlo.Whi + xhi.Wlo  (the three products an fp16 MFMA with
fp32 accumulation would form; products are exact in fp32, emulated here in float64 and rounded once).  Prints the error
of f(c, xhat) against a float64 evaluation for (a) numpy fp32, (b) the split form, and the number of code rows of the
golden that change when the oracle's blocks use the split form."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle.qinco_oracle as O
from conftest import golden_cases, load_golden, ref_codes, selection_margins
from qinco_amd import synth_state_dict

F32 = np.float32
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16x3"

def split(x, dt, n):
    parts, r = [], x.astype(np.float64)
    for _ in range(n):
        p = r.astype(F32).astype(dt).astype(np.float64)
        parts.append(p); r = r - p
    return parts

def bf16(x):   # round-to-nearest-even bf16 kept in float32
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(F32)

class BF:  # dtype-like shim for split()
    pass

def split_bf(x, n):
    parts, r = [], x.astype(np.float64)
    for _ in range(n):
        p = bf16(r.astype(F32)).astype(np.float64)
        parts.append(p); r = r - p
    return parts

def mm(x, W):
    if MODE == "f32":
        return x @ W.T
    if MODE == "f64":
        return (x.astype(np.float64) @ W.T.astype(np.float64)).astype(F32)
    if MODE.startswith("f16"):
        xs, ws = split(x, np.float16, 2), split(W, np.float16, 2)
        acc = xs[0] @ ws[0].T + xs[1] @ ws[0].T + xs[0] @ ws[1].T
        if MODE == "f16x4": acc = acc + xs[1] @ ws[1].T
        return acc.astype(F32)
    if MODE == "bf16x6" or MODE == "bf16x3":
        xs, ws = split_bf(x, 3), split_bf(W, 3)
        acc = xs[0] @ ws[0].T + xs[1] @ ws[0].T + xs[0] @ ws[1].T
        if MODE == "bf16x6": acc = acc + xs[2] @ ws[0].T + xs[0] @ ws[2].T + xs[1] @ ws[1].T
        return acc.astype(F32)
    raise SystemExit(MODE)

def step_forward_split(w, c, xhat, qinco1_mode):
    z = c if w.in_proj is None else c @ w.in_proj.T
    cc = np.concatenate([z, np.broadcast_to(xhat, z.shape[:-1] + xhat.shape[-1:])], axis=-1)
    z = z + (cc @ w.cat_w.T + w.cat_b)
    for up, down in zip(w.up, w.down):
        h = np.maximum(mm(z, up), F32(0))
        z = z + mm(h, down)
    out = z if w.out_proj is None else z @ w.out_proj.T
    return out if qinco1_mode else out + c

name = sys.argv[1] if len(sys.argv) > 1 else "C2_qinco2L_8x8_b8"
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cfg, seed = golden_cases()[name]
sd = synth_state_dict(cfg, seed)
g = load_golden(name)
x = np.concatenate([g["x"][: nrows // 2], g["x"][-(nrows - nrows // 2):]])
want = np.concatenate([ref_codes(g)[: nrows // 2], ref_codes(g)[-(nrows - nrows // 2):]])

# (1) error of one step's f(c, xhat) on random candidates
o = O.OracleQINCo.from_config(cfg, sd)
w = o.steps[1] if hasattr(o, "steps") else None
rng = np.random.default_rng(0)
if w is not None:
    c = w.codebook[rng.integers(0, cfg.K, 512)]
    xh = (x[:1] - o.data_mean) / o.data_std * 0.7
    def f64_forward():
        global MODE
        m0, MODE = MODE, "f64"
        # full float64 evaluation
        z = c.astype(np.float64) if w.in_proj is None else c.astype(np.float64) @ w.in_proj.T.astype(np.float64)
        cc = np.concatenate([z, np.broadcast_to(xh.astype(np.float64), z.shape[:-1] + xh.shape[-1:])], axis=-1)
        z = z + (cc @ w.cat_w.T.astype(np.float64) + w.cat_b)
        for up, down in zip(w.up, w.down):
            z = z + np.maximum(z @ up.T.astype(np.float64), 0) @ down.T.astype(np.float64)
        out = z if w.out_proj is None else z @ w.out_proj.T.astype(np.float64)
        MODE = m0
        return out if cfg.qinco1_mode else out + c
    ref = f64_forward()
    for mode in ("f32", "f16x3", "f16x4", "bf16x3", "bf16x6"):
        MODE = mode
        out = step_forward_split(w, c, xh, cfg.qinco1_mode)
        e = np.abs(out - ref)
        print(f"{mode:7s} max abs err {e.max():.3e}  rms {np.sqrt((e**2).mean()):.3e}  (|f| rms {np.sqrt((ref**2).mean()):.3f})")

# (2) code rows that change
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
O.step_forward = step_forward_split
o2 = O.OracleQINCo.from_config(cfg, sd)
codes, _ = o2.encode((x - o2.data_mean) / o2.data_std)
bad = np.nonzero((codes.T != want).any(axis=1))[0] if codes.shape[0] == want.shape[1] else np.nonzero((codes != want).any(axis=1))[0]
print(MODE, "rows differing from the reference:", len(bad), "of", len(x))
if len(bad):
    O.step_forward = step_forward_split.__globals__["O"].step_forward  # (unchanged module attr; margins from the plain oracle below)
