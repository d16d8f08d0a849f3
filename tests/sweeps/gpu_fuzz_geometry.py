#!/usr/bin/env python
"""Randomised geometry / search-width sweep of the HIP path against the oracle (a one-off hunt for bugs the hand-picked test
cases miss; the log goes to profiles/).

    python tests/sweeps/gpu_fuzz_geometry.py --seed 7 --count 24 --prebuild     # build container: compile the on-demand instances
    python tests/sweeps/gpu_fuzz_geometry.py --seed 7 --count 24 --out gpurun_out/fuzz.jsonl      # GPU box

Every configuration the reference's constructor accepts is fair game (qinco_base.py:229-260, utils.py:166-172): D not a
multiple of 32, De = D or not, any hidden width, L from 0, M from 1, K from 16 to 1024 (other than 256: VALU tables), A = 0 (QINCo1 mode) to
A = K, B from 1 to beyond K, an IVF coarse step in front, a batch that is not a multiple of anything.  For each: encode codes
by the tie rule of tests/conftest.py (a differing row must sit on a reference margin < 2e-5 AND be reproduced exactly by the
oracle when nudged that way), tracked reconstructions and decode of random codes within 1e-5 relative.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def draw(rs: np.random.RandomState) -> dict:
    D = int(rs.choice([8, 24, 40, 64, 96, 100, 128, 160, 200, 256, 300, 384, 512]))
    qinco1 = rs.rand() < 0.25
    K = int(rs.choice([16, 64, 100, 256, 256, 256, 300, 512, 1024]))
    if qinco1:
        de, A = None, 0
        B = int(rs.choice([1, 1, 4]))
    else:
        de = None if rs.rand() < 0.3 else int(rs.choice([32, 64, 72, 128, 192, 256, 320, 384, 512]))
        A = int(rs.choice([1, 2, 4, 8, 16, 32, K]))
        B = int(rs.choice([1, 2, 4, 8, 16, 32, 64, 128]))     # (beyond K for the small codebooks: beam_0 = min(B, K))
    A = min(A, K)
    dh = int(rs.choice([32, 48, 64, 128, 200, 256, 384, 512]))
    L = int(rs.choice([0, 1, 2, 2, 3, 5]))
    M = int(rs.choice([1, 2, 3, 3, 5]))
    ivf_K = int(rs.choice([64, 1000, 4096])) if rs.rand() < 0.2 else None
    n = int(rs.choice([1, 7, 65, 130, 200, 200, 333]))
    max_batch = int(rs.choice([64, 128, 4096]))
    rebeam = None
    if rs.rand() < 0.3 and A > 0:      # search widths changed after creation (the CLI override, utils.py:166-172): qinco_set_beam
        rebeam = (int(rs.choice([1, 2, 4, 8, 16, min(32, K)])), int(rs.choice([1, 2, 4, 8, 16, 32])))
    u8 = bool(rs.rand() < 0.25)        # byte rows (bvecs datasets): the library converts them itself
    return dict(cfg=dict(D=D, M=M, K=K, L=L, de=de, dh=dh, A=A, B=B, qinco1_mode=qinco1, ivf_K=ivf_K), n=n, max_batch=max_batch, u8=u8,
                rebeam=rebeam)


def configs(seed: int, count: int):
    from qinco_amd import QincoConfig
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < count:
        c = draw(rs)
        try:
            cfg = QincoConfig(**c["cfg"])
        except (ValueError, AssertionError):
            continue
        if cfg.ivf and cfg.A and cfg.B > cfg.K:      # the reference's topk(max(A, B)) over K codewords raises (qinco_base.py:108-125)
            continue
        # keep the oracle's share of a case to seconds (n x beams x candidates x per-row FLOPs, fp32 numpy)
        if c["n"] * cfg.encode_flops_per_vector() > 6e10:
            continue
        out.append((cfg, c))
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--count", type=int, default=24)
    ap.add_argument("--prebuild", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cases = configs(a.seed, a.count)
    if a.prebuild:
        from qinco_amd.build import ensure_instance
        from qinco_amd import _lib
        lib = _lib.load()
        for cfg, c in cases:
            if lib.qinco_shape_supported(cfg.D, cfg.De, cfg.dh) == 1:
                continue
            t0 = time.time()
            p = ensure_instance(cfg.D, cfg.De, cfg.dh)
            print(f"{c['cfg']}: {p}  ({time.time() - t0:.0f} s)", flush=True)
        return 0

    from conftest import assert_only_near_ties, make_oracle
    NEAR_TIE, REL_TOL = 2e-5, 1e-5             # the bars of tests/test_hip_parity.py

    def rel_err(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    from qinco_amd import QincoEngine, apply_regime, regime_vectors, synth_codes, synth_state_dict, synth_vectors
    log = open(a.out, "w") if a.out else None
    failures = 0
    for i, (cfg, c) in enumerate(cases):
        rec = dict(i=i, **c)
        t0 = time.time()
        try:
            sd = synth_state_dict(cfg, 5000 + 31 * a.seed + i)
            x = synth_vectors(cfg, sd, c["n"], seed=6000 + i)
            if c.get("u8"):     # data_mean / data_std of a byte dataset's magnitude, rows clipped to [0, 255]
                sd = apply_regime(cfg, sd, "bigann", 7000 + i)
                x = regime_vectors(cfg, sd, c["n"], "bigann", seed=6000 + i)
                assert x.dtype == np.uint8
            eng = QincoEngine(cfg, sd, max_batch=c["max_batch"])
            rec["describe"] = eng.describe()
            if c.get("rebeam"):     # the handle was created for (A, B); it now searches with (A2, B2)
                A2, B2 = c["rebeam"]
                if cfg.ivf and B2 > cfg.K:
                    B2 = cfg.K
                eng.set_beam(A=A2, B=B2)
                cfg = cfg.with_search(A=A2, B=B2)
            oracle = make_oracle(cfg, sd)
            want = oracle(x.astype(np.float32), step="encode").T
            got, xhat = eng.encode(x, return_xhat=True)
            if i % 3 == 0:      # the narrow code types carry the same codes
                dt = np.uint8 if max(cfg.K, cfg.ivf_K or 0) <= 256 else np.int32
                assert np.array_equal(eng.encode(x, code_dtype=dt), got.astype(dt))
            if i % 4 == 1:      # device path (torch tensors on the GPU, asynchronous) carries the same bits as the host path
                import torch
                cd, hd = eng.encode(torch.from_numpy(x).cuda(), return_xhat=True)
                assert np.array_equal(cd.cpu().numpy(), got) and np.array_equal(hd.cpu().numpy(), xhat)
                assert np.array_equal(eng.decode(torch.from_numpy(got).cuda()).cpu().numpy(), eng.decode(got))
            rec["rows_on_ties"] = int(assert_only_near_ties(oracle, x, got, want, NEAR_TIE, str(c)))
            ok = (got == want).all(axis=1)
            ref = oracle(want.T, step="decode")
            e1 = rel_err(eng.decode(want), ref)
            e2 = rel_err((xhat * sd["data_std"] + sd["data_mean"])[ok], ref[ok]) if ok.any() else 0.0
            rc = synth_codes(cfg, 33, seed=3).T.copy()
            e3 = rel_err(eng.decode(rc), oracle(rc.T, step="decode"))
            rec.update(decode_rel=float(e1), xhat_rel=float(e2), rand_decode_rel=float(e3))
            assert max(e1, e2, e3) < REL_TOL, (e1, e2, e3)
            eng.close()
            rec["ok"] = True
        except Exception as e:  # noqa: BLE001  (a fuzz run reports every failure, it does not stop at the first)
            rec.update(ok=False, error=f"{type(e).__name__}: {str(e)[:400]}")
            failures += 1
        rec["seconds"] = round(time.time() - t0, 2)
        line = json.dumps(rec)
        print(line, flush=True)
        if log:
            log.write(line + "\n")
            log.flush()
    print(f"{len(cases) - failures} of {len(cases)} configurations agree with the oracle", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
