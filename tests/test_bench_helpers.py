"""bench.py's host-side helpers (no GPU): the executed-FLOP count of the folded kernels, the roofline record, the timeout guard
around RCCL calls."""
import sys
import time

import pytest

from conftest import ROOT

sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_executed_flops_of_the_folded_kernels():
    from qinco_amd.config import BASELINE_CONFIGS
    c2, c1, c4, s, q768 = (BASELINE_CONFIGS[k] for k in ("C2", "C1", "C4", "S", "Q1_768"))

    def share(cfg, A, **kw):          # executed / algorithmic for rows in groups of A
        return bench.executed_flops(cfg, A, 1, **kw) / (A * cfg.mlp_flops_per_row())
    assert abs(share(c2, 16) - 0.9240) < 1e-3      # 92.4 % of the algorithmic FLOPs at C2
    assert abs(share(c1, 256) - 0.9396) < 1e-3
    assert abs(share(c4, 16) - 0.8510) < 1e-3
    assert 0.59 < share(s, 16) < 0.62
    assert share(c2, 1) > share(c2, 16)            # decode: nothing of the per-group GEMM is shared
    assert share(c2, 16, fold=False) == pytest.approx(1.0)
    assert share(q768, 256, fold2=False) > share(q768, 256)     # the 16-row tile form folds the head only


def test_roofline_record_is_consistent():
    from qinco_amd.config import BASELINE_CONFIGS
    cfg = BASELINE_CONFIGS["C2"]
    rows = 2_097_152
    prof = {"mlp_ms": 7 * 134.0, "mlp_launches": 7, "mlp_flops": 7 * rows * cfg.mlp_flops_per_row(),
            "mlp_flops_executed": 7 * bench.executed_flops(cfg, rows, rows // 16)}
    rf = bench.roofline_dict(prof, dt=0.95)
    assert rf["bound"] == "mfma" and rf["peak"] == bench.PEAK_FP32_MFMA_TFLOPS and rf["launches"] == 7
    assert abs(rf["algorithmic_tflops"] - rows * cfg.mlp_flops_per_row() / 0.134 / 1e12) < 1e-6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["frac"] < rf["frac_algorithmic"] and abs(rf["frac"] / rf["frac_algorithmic"] - 0.9240) < 1e-3
    assert rf["frac"] <= 1.0
    assert abs(rf["mlp_share_of_step_time"] - 7 * 0.134 / 0.95) < 1e-9


def test_timeout_guard_for_rccl_calls():
    assert bench.call_with_timeout(lambda: 41 + 1, 5) == 42
    with pytest.raises(ValueError, match="boom"):
        bench.call_with_timeout(lambda: (_ for _ in ()).throw(ValueError("boom")), 5)
    t0 = time.time()
    with pytest.raises(TimeoutError):
        bench.call_with_timeout(lambda: time.sleep(30), 0.3)
    assert time.time() - t0 < 5
