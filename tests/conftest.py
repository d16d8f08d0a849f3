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
        "C1_qinco1_8x8": (preset("qinco1", D=128, M=8), 1235),
        "C2_qinco2L_8x8_b8": (preset("qinco2-L", D=128, M=8, B=8), 1236),
        "C2_qinco2L_8x8_b1": (preset("qinco2-L", D=128, M=8, B=1), 1236),
        "C4_qinco2L_d768_b8": (preset("qinco2-L", D=768, M=4, B=8), 1238),
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
