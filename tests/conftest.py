import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Golden cases: name -> (config kwargs, seed).  Must mirror tests/golden/make_golden.py::CASES.
def golden_cases():
    from qinco_amd.config import QincoConfig, preset
    return {
        "tiny_proj_beam": (QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=4), 11),
        "tiny_proj_greedyA": (QincoConfig(D=32, M=4, K=256, L=2, de=64, dh=96, A=8, B=1), 12),
        "tiny_id_qinco1": (QincoConfig(D=32, M=4, K=256, L=2, de=None, dh=64, A=0, B=1, qinco1_mode=True), 13),
        "tiny_id_A0_beam": (QincoConfig(D=32, M=3, K=256, L=1, de=None, dh=64, A=0, B=3, qinco1_mode=False), 14),
        "tiny_proj_dh128": (QincoConfig(D=32, M=4, K=256, L=3, de=64, dh=128, A=8, B=4), 18),
        "C1_qinco1_8x8": (preset("qinco1", D=128, M=8), 1235),
        "C2_qinco2L_8x8_b8": (preset("qinco2-L", D=128, M=8, B=8), 1236),
        "C2_qinco2L_8x8_b1": (preset("qinco2-L", D=128, M=8, B=1), 1236),
        "C4_qinco2L_d768_b8": (preset("qinco2-L", D=768, M=8, B=8), 1238),
        "C3_qinco2L_16x8_b8": (preset("qinco2-L", D=128, M=16, B=8), 1237),
        "C2_qinco2L_8x8_b32": (preset("qinco2-L", D=128, M=8, B=32), 1236),
        "tiny_smallK_wideB": (QincoConfig(D=32, M=4, K=64, L=2, de=64, dh=96, A=8, B=128), 17),
        "qinco1_d768": (preset("qinco1", D=768, M=3), 1241),
        "tiny_ivf_beam": (QincoConfig(D=32, M=3, K=256, L=2, de=64, dh=96, A=4, B=8, ivf_K=2048), 15),
        "tiny_ivf_greedy_id": (QincoConfig(D=32, M=3, K=256, L=2, de=None, dh=64, A=8, B=1, ivf_K=1024), 16),
        "ivf_qinco2S_d128": (preset("qinco2-S", D=128, M=4, B=8, ivf_K=65536), 1240),
    }


def load_golden(name):
    return dict(np.load(GOLDEN / f"{name}.npz"))


def make_oracle(cfg, sd):
    from oracle.qinco_oracle import OracleQINCo
    return OracleQINCo.from_config(cfg, sd)


def ref_codes(g):
    return g["codes_wrapper"] if "codes_wrapper" in g else g["codes_base"]


def selection_margins(oracle, x):
    """Per row and step m >= 1: the oracle's relative gap at the selection boundaries of that step -- between the last
    kept and the first dropped candidate distance, and (A > 0) between the A-th and (A+1)-th codeword of each beam's
    pre-selection table.  A code row may differ from the oracle only where this gap is at rounding level."""
    trace = {}
    xn = (np.asarray(x, np.float32) - oracle.data_mean) / oracle.data_std
    oracle.encode(xn, trace)
    out = []
    for m in range(1, oracle.M):
        d = np.sort(trace[f"dists{m}"], axis=-1)
        fo = min(oracle.B if m < oracle.M - 1 else 1, d.shape[1] - 1)
        mg = (d[:, fo] - d[:, fo - 1]) / np.maximum(np.abs(d[:, fo]), 1e-12)
        if f"dsub{m}" in trace:
            ds = np.sort(trace[f"dsub{m}"], axis=-1)
            a = trace[f"top{m}"].shape[-1]
            if a < ds.shape[-1]:
                mg = np.minimum(mg, ((ds[..., a] - ds[..., a - 1]) / np.maximum(np.abs(ds[..., a]), 1e-12)).min(axis=1))
        out.append(mg)
    return np.stack(out, axis=1) if out else np.zeros((len(x), 0), np.float32)


def assert_only_near_ties(oracle, x, got, want, near_tie, label=""):
    """got / want: (n, M) code rows.  Every mismatching row must have an oracle selection margin below `near_tie` at or
    after its first differing column (= step); returns the number of (legitimately) differing rows."""
    bad = np.nonzero((got != want).any(axis=1))[0]
    if len(bad) == 0:
        return 0
    mg = selection_margins(oracle, x[bad])
    for r, i in enumerate(bad):
        first = int(np.nonzero(got[i] != want[i])[0][0])
        assert first > 0, f"{label}: row {i} differs at step 0"
        m = float(mg[r, max(first - 1, 0):].min())
        print(f"{label}: row {i} differs from step {first}; oracle margin there {m:.3e}")
        assert m < near_tie, f"{label}: row {i} differs although the oracle margin is {m:.3e}"
    return len(bad)
