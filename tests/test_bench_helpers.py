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


def test_shard_checksum_sees_misplaced_and_reordered_rows():
    """The multi-rank line's `gathered_rows_verified_on_rank0`: a shard that lands in other rows of the gathered matrix, or in its
    rows in another order, has another checksum; blocks of rows add up to the whole."""
    import numpy as np
    rng = np.random.RandomState(3)
    codes = rng.randint(0, 256, (5000, 8)).astype(np.uint8)
    whole = bench.shard_checksum(codes, 1000)
    assert whole == bench.shard_checksum(codes.astype(np.int32), 1000)                      # the wire type does not matter
    assert whole != bench.shard_checksum(codes, 1001)                                       # same rows, one row further down
    swapped = codes.copy()
    swapped[[10, 11]] = swapped[[11, 10]]
    assert not np.array_equal(swapped[10], swapped[11]) and whole != bench.shard_checksum(swapped, 1000)
    parts = (bench.shard_checksum(codes[:1234], 1000) + bench.shard_checksum(codes[1234:], 2234)) % (1 << 64)
    assert parts == whole and bench.shard_checksum(codes[:0], 7) == 0
    assert np.array_equal(bench.fake_code_rows(100, 50, 8), bench.fake_code_rows(0, 150, 8)[100:])
    assert len(np.unique(bench.fake_code_rows(0, 4096, 8))) == 256


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dry_rccl_steps_on_gloo_without_a_gpu(world):
    """`bench.py --gpus N --dry-rccl --backend gloo` runs the communication steps of the multi-GPU bench alone -- payload group,
    one grouped 1-byte send / recv with every peer, the PRODUCT's gather_codes on fake rows sharded like encode_database shards them --
    over host buffers, so this tier executes the very lines the first 8-GPU run executes (there on RCCL, device buffers)."""
    import json
    import os
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = 100_003
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dry-rccl", "--backend", "gloo", "--dry-rows", str(n),
                        "--no-affinity", "--rccl-timeout", "120"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["all_ok"] and rec["rccl_ranks_seen"] == world and rec["rows"] == n
    assert rec["steps_ok_on_all_ranks"] == {"communicator": True, "p2p_1_byte_with_every_peer": True, "gather_codes": True}
    per = rec["per_rank"]
    assert [p["rank"] for p in per] == list(range(world)) and sum(p["rows"] for p in per) == n
    assert per[-1]["rows"] == n - (n // world) * (world - 1)                                 # the last rank takes the remainder
    for p in per:
        g = p["steps"]["gather_codes"]
        assert p["steps"]["p2p_1_byte_with_every_peer"]["peers"] == world - 1 and g["wire_dtype"] == "uint8" and g["ranks"] == world
        assert g["bytes_sent"] == (0 if p["rank"] == 0 else p["rows"] * 8)
    assert per[0]["steps"]["gather_codes"]["bytes_received"] == (n - per[0]["rows"]) * 8
