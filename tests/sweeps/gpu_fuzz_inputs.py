#!/usr/bin/env python
"""Adversarial INPUT rows through encode / decode against the oracle (the geometry sweep uses well-behaved Gaussian rows).

    python tests/sweeps/gpu_fuzz_inputs.py --out gpurun_out/fuzz_inputs.jsonl         # GPU box

Models: the golden cases' (tests/golden/cases.py: synthetic and reference-trained weights).  Rows per kind, all through the same
bars as the tests (tie rule with oracle replay; reconstructions within 1e-5):
  zeros / constant rows, every row the same, rows that ARE reconstructions of random codes (the last step's best distance is ~0:
  ties at rounding level), rows 1e4 x and 1e-6 x the data's scale, one-hot spikes, the corners of the byte cube (all 0 / all 255),
  rows equal to the data mean (normalised input exactly 0), and alternating-sign saw-teeth.
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
sys.path.insert(0, str(ROOT / "tests" / "golden"))

MODELS = ["tiny_proj_beam", "tiny_id_qinco1", "tiny_ivf_beam", "trained_qinco2S", "trained_qinco1", "trained_ivf_qinco2S",
          "trained_tiny_proj", "norm_bigann_u8", "C2_qinco2L_8x8_b8"]
KINDS = ["zeros", "constant", "all_same", "on_reconstruction", "huge", "tiny", "one_hot", "byte_corners", "data_mean", "sawtooth"]


def rows(kind, cfg, sd, oracle, rs, n):
    D = cfg.D
    mean, std = np.asarray(sd["data_mean"], np.float32), float(np.asarray(sd["data_std"]).reshape(-1)[0])
    if kind == "zeros":
        return np.zeros((n, D), np.float32)
    if kind == "constant":
        return np.full((n, D), 3.25 * std, np.float32) * rs.choice([-1.0, 1.0], (n, 1)).astype(np.float32)
    if kind == "all_same":
        return np.repeat((mean + std * rs.randn(1, D)).astype(np.float32), n, axis=0)
    if kind == "on_reconstruction":
        from qinco_amd import synth_codes
        codes = synth_codes(cfg, n, seed=int(rs.randint(1 << 30)))
        return np.asarray(oracle(codes, step="decode"), np.float32)
    if kind == "huge":
        return (mean + 1e4 * std * rs.randn(n, D)).astype(np.float32)
    if kind == "tiny":
        return (mean + 1e-6 * std * rs.randn(n, D)).astype(np.float32)
    if kind == "one_hot":
        x = np.repeat(mean[None], n, axis=0).astype(np.float32)
        x[np.arange(n), rs.randint(0, D, n)] += 40.0 * std
        return x
    if kind == "byte_corners":
        x = np.zeros((n, D), np.uint8)
        x[1::2] = 255
        x[2::4, ::2] = 255
        return x
    if kind == "data_mean":
        return np.repeat(mean[None], n, axis=0).astype(np.float32)
    if kind == "sawtooth":
        s = np.where(np.arange(D) % 2 == 0, 1.0, -1.0).astype(np.float32)
        return (mean + std * s[None] * (0.5 + rs.rand(n, 1))).astype(np.float32)
    raise KeyError(kind)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--split", action="store_true", help="the opt-in split-fp16 form (models without a split instance are skipped; a row "
                    "set that leaves the fp16 range must be REFUSED with an error, never encoded wrongly)")
    a = ap.parse_args()
    from cases import case_model
    from conftest import assert_only_near_ties, make_oracle
    from qinco_amd import QincoEngine
    rs = np.random.RandomState(a.seed)
    log = open(a.out, "w") if a.out else None
    failures = total = 0
    for name in MODELS:
        cfg, sd = case_model(name)
        oracle = make_oracle(cfg, sd)
        try:
            eng = QincoEngine(cfg, sd, max_batch=64, split_f16=a.split)
        except NotImplementedError as e:
            print(json.dumps(dict(model=name, skipped=str(e)[:120])), flush=True)
            continue
        n = 24 if name.startswith("C2") else 96
        for kind in KINDS:
            rec = dict(model=name, kind=kind, n=n)
            t0 = time.time()
            total += 1
            try:
                x = rows(kind, cfg, sd, oracle, rs, n)
                xf = x.astype(np.float32)
                want = oracle(xf, step="encode").T
                try:
                    got, xhat = eng.encode(x, return_xhat=True)
                except RuntimeError as e:      # the split form's overflow flag (qinco_check): a refusal, not a result
                    assert a.split and "fp16" in str(e), e
                    rec.update(ok=True, refused=str(e)[:100], seconds=round(time.time() - t0, 2))
                    print(json.dumps(rec), flush=True)
                    if log:
                        log.write(json.dumps(rec) + "\n")
                    continue
                assert np.isfinite(xhat).all()
                rec["rows_on_ties"] = int(assert_only_near_ties(oracle, x, got, want, 2e-5, f"{name}/{kind}"))
                ok = (got == want).all(axis=1)
                ref = np.asarray(oracle(want.T, step="decode"), np.float64)
                dec = np.asarray(eng.decode(want), np.float64)
                rec["decode_rel"] = float(np.abs(dec - ref).max() / max(np.abs(ref).max(), 1e-30))
                assert rec["decode_rel"] < 1e-5
                rec["rows_equal"] = int(ok.sum())
                rec["ok"] = True
            except Exception as e:  # noqa: BLE001
                rec.update(ok=False, error=f"{type(e).__name__}: {str(e)[:300]}")
                failures += 1
            rec["seconds"] = round(time.time() - t0, 2)
            line = json.dumps(rec)
            print(line, flush=True)
            if log:
                log.write(line + "\n")
                log.flush()
        eng.close()
    print(f"{total - failures} of {total} (model, kind) pairs agree with the oracle", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
