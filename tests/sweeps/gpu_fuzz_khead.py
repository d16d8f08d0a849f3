#!/usr/bin/env python
"""Randomised sweep of the short-MLP production kernels (KHEAD, csrc/mlp_kernel.hpp) against the oracle AND against their twin: model depth,
steps, search widths (also changed after creation), call sizes large enough for the 128-row kernels (a partly filled last workgroup
included), byte inputs -- on the compiled-in two-workgroups-per-CU shapes (D in 96 / 128 / 256 / 768, de = 128 or D, dh = 256).

    python tests/sweeps/gpu_fuzz_khead.py --seed 1 --count 24 --out gpurun_out/fuzz_khead.jsonl
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


ODD = False    # --odd: the short geometries served by kernel instances built on demand (zero-padded to 32-feature blocks: 3 / 4 blocks of
               # De, 5 / 7 of Dh), which take the same plan


def draw(rs):
    if ODD:
        D, de, dh = [(100, None, 200), (64, 96, 160), (100, None, 200)][int(rs.randint(3))]
        qinco1 = de is None and rs.rand() < 0.25
        A = 0 if qinco1 else int(rs.choice([8, 16, 16, 32, 64]))
        B = 1 if qinco1 else int(rs.choice([1, 4, 8]))
        return dict(cfg=dict(D=D, M=int(rs.choice([2, 3])), K=256, L=int(rs.choice([1, 2, 3])), de=de, dh=dh, A=A, B=B, qinco1_mode=qinco1),
                    n=int(rs.choice([1500, 2100, 3001])), rebeam=None, u8=False)
    D = int(rs.choice([96, 128, 128, 128, 256, 768]))
    qinco1 = D in (96, 128) and rs.rand() < 0.25
    de = None if (qinco1 or (D in (96, 128) and rs.rand() < 0.5)) else 128
    if D == 128:
        de = None
    A = 0 if qinco1 else int(rs.choice([1, 4, 8, 16, 16, 16, 20, 32, 64, 256]))
    B = int(rs.choice([1, 1, 4])) if qinco1 else int(rs.choice([1, 2, 4, 8, 8, 16]))
    L = int(rs.choice([1, 2, 2, 3, 4]))
    M = int(rs.choice([2, 3, 4]))
    n = int(rs.choice([700, 1500, 2100, 3001]))
    rebeam = None
    if A > 0 and rs.rand() < 0.35:
        rebeam = (int(rs.choice([8, 16, 32, 20])), int(rs.choice([1, 4, 8])))
    return dict(cfg=dict(D=D, M=M, K=256, L=L, de=de, dh=256, A=A, B=B, qinco1_mode=qinco1), n=n, rebeam=rebeam, u8=bool(rs.rand() < 0.2))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--count", type=int, default=24)
    ap.add_argument("--out", default="")
    ap.add_argument("--odd", action="store_true")
    a = ap.parse_args()
    global ODD
    ODD = a.odd
    if ODD:
        from qinco_amd.build import ensure_instance
        for shape in ((100, 100, 200), (64, 96, 160)):
            ensure_instance(*shape)
    from conftest import assert_only_near_ties, make_oracle
    from qinco_amd import QincoConfig, QincoEngine, apply_regime, regime_vectors, synth_state_dict, synth_vectors
    NEAR_TIE, REL_TOL = 2e-5, 1e-5
    rs = np.random.RandomState(a.seed)
    log = open(a.out, "w") if a.out else None
    failures = done = 0
    while done < a.count:
        c = draw(rs)
        cfg = QincoConfig(**c["cfg"])
        rows = c["n"] * max(cfg.B, 1) * (cfg.A or cfg.K)
        if rows < 40000 or c["n"] * cfg.encode_flops_per_vector() > 2.5e11:   # the 128-row kernels' territory, an oracle of seconds
            continue
        rec = dict(i=done, **c)
        t0 = time.time()
        try:
            sd = synth_state_dict(cfg, 9000 + 17 * a.seed + done)
            x = synth_vectors(cfg, sd, c["n"], seed=9100 + done)
            if c["u8"]:
                sd = apply_regime(cfg, sd, "bigann", 9200 + done)
                x = regime_vectors(cfg, sd, c["n"], "bigann", seed=9100 + done)
            eng = QincoEngine(cfg, sd, max_batch=4096)
            rec["describe"] = eng.describe()
            assert "var=4476" in rec["describe"], rec["describe"]
            twin = QincoEngine(cfg, sd, max_batch=4096, diagnostics={"mlp_variant": (48, 380)})
            assert "var=380" in twin.describe(), twin.describe()
            if c["rebeam"]:
                A2, B2 = c["rebeam"]
                eng.set_beam(A=A2, B=B2)
                twin.set_beam(A=A2, B=B2)
                cfg = cfg.with_search(A=A2, B=B2)
            got, xhat = eng.encode(x, return_xhat=True)
            gt, xt = twin.encode(x, return_xhat=True)
            assert np.array_equal(got, gt) and np.array_equal(xhat.view(np.uint32), xt.view(np.uint32)), "KHEAD differs from its twin"
            oracle = make_oracle(cfg, sd)
            want = oracle(x.astype(np.float32), step="encode").T
            rec["rows_on_ties"] = int(assert_only_near_ties(oracle, x, got, want, NEAR_TIE, str(c)))
            ref = oracle(want.T, step="decode")
            dec = eng.decode(want)
            e1 = float(np.abs(dec - ref).max() / np.abs(ref).max())
            rec["decode_rel"] = e1
            assert e1 < REL_TOL, e1
            eng.close()
            twin.close()
            rec["ok"] = True
        except Exception as e:      # noqa: BLE001
            rec["ok"] = False
            rec["error"] = repr(e)[:400]
            failures += 1
        rec["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(rec), flush=True)
        if log:
            log.write(json.dumps(rec) + "\n")
            log.flush()
        done += 1
    print(json.dumps({"seed": a.seed, "cases": done, "failures": failures}))
    if log:
        log.write(json.dumps({"seed": a.seed, "cases": done, "failures": failures}) + "\n")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
