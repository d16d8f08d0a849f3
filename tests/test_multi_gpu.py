"""N > 1 on the GPU box: the HIP engine behind encode_database with one process per rank (reference
search_tasks.py:85-137 + run.sh:8), and bench.py's own multi-rank launch.  Codes from P ranks must equal the 1-process
HIP codes bitwise."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_model

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, outdir, backend, n, dev_input=False):
    port = str(_free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "rank_worker.py"), str(r), str(world), port, str(outdir),
                               backend, str(n)] + (["dev"] if dev_input else []), env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


def _single_process_codes(n):
    from qinco_amd import synth_state_dict, synth_vectors
    from qinco_amd.model import QINCoHIP
    cfg, sd = golden_model("tiny_proj_beam")
    model = QINCoHIP(cfg, sd, max_batch=256)
    want = model(synth_vectors(cfg, sd, n, seed=4), step="encode").T
    model.engine.close()
    return cfg, np.asarray(want)


def _check_outputs(tmp_path, world, n):
    from qinco_amd.encode_db import EncodedDBIterator, shard_bounds
    cfg, want = _single_process_codes(n)
    got = np.load(tmp_path / "gathered.npy")
    assert got.dtype == np.int64 and np.array_equal(got, want)
    it = EncodedDBIterator(str(tmp_path / "db.npz"), K=cfg.K, M=cfg.M, D=cfg.D)
    assert it.n_parts == world and np.array_equal(it.load_all(), want)
    for r in range(world):
        s, e = shard_bounds(n, world, r)
        part = np.load(tmp_path / f"db.part_{r}.npz")["codes"]
        assert part.dtype == np.int64 and np.array_equal(part, want[s:e])


@pytest.mark.parametrize("world,n,dev_input", [(2, 1203, False), (3, 700, True)])
def test_ranks_sharing_one_gpu_gloo(tmp_path, world, n, dev_input):
    """Each rank owns a QincoEngine on GPU 0 (the 1-GPU box); collectives on gloo."""
    _run_ranks(world, tmp_path, "gloo", n, dev_input)
    _check_outputs(tmp_path, world, n)


def test_two_ranks_rccl(tmp_path):
    """One GPU per rank, backend nccl (= RCCL): barriers and the gather of the codes run on device buffers."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    _run_ranks(2, tmp_path, "nccl", 1203, True)
    _check_outputs(tmp_path, 2, 1203)


def test_native_gather_entry_point_single_rank():
    """qinco_gather_codes with world = 1 (no communicator): the root's own shard lands in the output; argument errors are
    status codes.  (Two ranks over RCCL: test_two_ranks_native_rccl_gather, needs two GPUs.)"""
    import ctypes as C
    import torch
    from qinco_amd import _lib
    from qinco_amd.comm import gather_codes_native
    codes = torch.arange(37 * 8, dtype=torch.int32, device="cuda").reshape(37, 8) % 256
    for dt in (torch.uint8, torch.int32, torch.int64):
        c = codes.to(dt)
        out = gather_codes_native(c, [37], rank=0)
        assert out.dtype == dt and torch.equal(out, c)
    lib = _lib.load()
    cnt = (C.c_int64 * 2)(37, 5)
    assert lib.qinco_gather_codes(codes.data_ptr(), 36, 8, _lib.CODE_I32, None, cnt, 2, 0, 0, None, None) == -1     # n_local != counts[rank]
    assert lib.qinco_gather_codes(codes.data_ptr(), 37, 8, _lib.CODE_I32, codes.data_ptr(), cnt, 2, 0, 0, None, None) == -1   # no communicator
    assert b"communicator" in lib.qinco_last_error()


def test_two_ranks_native_rccl_gather(tmp_path):
    """The C-ABI route of SURVEY 8(e): one process per GPU, RCCL communicator created through ctypes (no torch.distributed),
    qinco_gather_codes (grouped ncclSend / ncclRecv of uneven shards) -- gathered codes equal the 1-process codes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    n = 1203
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "rank_worker_native.py"), str(r), "2", str(tmp_path / "uid"),
                               str(tmp_path), str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    _, want = _single_process_codes(n)
    assert np.array_equal(np.load(tmp_path / "gathered_native.npy"), want)


def _bench(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_ranks_from_a_bare_python():
    """`python bench.py --gpus 2` without torchrun: the script re-executes itself under torch.distributed.run."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    rec = _bench("--gpus", "2", "--backend", backend, "--workload", "C1", "--batch", "512", "--steps", "2", "--warmup", "1")
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    mg = rec["multi_gpu"]
    assert mg["control_plane"] == "gloo" and mg["gather_ok_on_all_ranks"] and mg["per_rank_vectors"] == [1024, 1024]
    assert len(mg["per_rank_encode_vectors_per_s"]) == 2 and all(v > 0 for v in mg["per_rank_encode_vectors_per_s"])
    assert len(mg["per_rank_gather_s"]) == 2 and mg["gather_bytes_per_rank"] == 2 * 512 * 8
    assert abs(rec["value"] - 2 * 2 * 512 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 1e-6
    # the timed region is the product's path (encode_shard + gather_codes), and the line shows who ran where and what arrived
    assert "gather_codes" in mg["path"] and mg["gathered_rows_verified_on_rank0"] is True
    for r, info in enumerate(mg["per_rank"]):
        assert info["rank"] == r and info["vectors"] == 1024 and info["gather_ok"] and info["rows_on_rank0_equal_to_the_shard"]
        assert info["ranks_seen_by_payload_group"] == 2 and info["wire_dtype"] == "uint8" and info["encode_s"] > 0
        assert info["bytes_sent"] == (0 if r == 0 else 1024 * 8) and info["bytes_received"] == (1024 * 8 if r == 0 else 0)
        assert isinstance(info["gpu"], int) and "pci_bus_id" in info and "numa_node" in info
    if backend == "nccl":
        assert mg["rccl_ranks_seen"] == 2 and all(i["transport"] == "nccl" and i["buffers"] == "device" for i in mg["per_rank"])
        assert len({i["gpu"] for i in mg["per_rank"]}) == 2 and len({i["pci_bus_id"] for i in mg["per_rank"]}) == 2


def test_bench_strong_scaling_splits_one_database():
    """--scaling strong: ONE database (here 3 x 512 + 100 rows, so the last shard is longer) split over the ranks like
    encode_database does (search_tasks.py:103-104); `value` = database size / max-over-ranks time."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 3 else "gloo"
    rec = _bench("--gpus", "3", "--backend", backend, "--workload", "C1", "--batch", "512", "--steps", "3", "--warmup", "1",
                 "--scaling", "strong", "--db", "1636")
    assert rec["n_gpus"] == 3 and rec["scaling"] == "strong" and rec["config"]["distinct_vectors_encoded"] == 1636
    mg = rec["multi_gpu"]
    assert mg["per_rank_vectors"] == [545, 545, 546] and mg["gather_ok_on_all_ranks"] and mg["gathered_rows_verified_on_rank0"] is True
    assert mg["gather_bytes_per_rank"] == 546 * 8 and [i["bytes_sent"] for i in mg["per_rank"]] == [0, 545 * 8, 546 * 8]
    assert abs(rec["value"] - 1636 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-6


def test_rccl_world_of_one():
    """First contact with the real librccl on a 1-GPU box (search_tasks.py:85-137, run.sh:8): a communicator of ONE rank
    made through ctypes, ncclCommCount == 1, and qinco_gather_codes routing every code type -- odd byte counts, an empty shard --
    through a grouped ncclSend-to-self + ncclRecv-from-self ON that communicator; the library whose ncclSend the C ABI resolved
    (dladdr) is the one that made the communicator."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_world1_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])     # (RCCL prints its banner on stdout too)
    print("RCCL resolved by qinco_gather_codes:", rec["library"])
    assert rec["count"] == 1
    assert os.path.realpath(rec["library"]) == rec["made_by"] and "rccl" in os.path.basename(rec["library"])
    assert len(rec["cases"]) == 12 and all(c["equal"] for c in rec["cases"]), rec["cases"]
    assert rec["copy_route_equal"]


def test_torch_rccl_world_of_one():
    """`bench.py --gpus 1 --dry-rccl`: torch.distributed with a gloo control plane and an `nccl` (= RCCL) payload group of ONE rank:
    all_reduce, a grouped isend / irecv with itself (batch_isend_irecv), the product's gather_codes run through its transfer lines on
    device buffers (self_transfer: rank 0 in both roles; encode_db.gather_codes), then the C-ABI route (RcclComm.from_process_group
    -> qinco_gather_codes).  Every step on the real library."""
    rec = _bench("--gpus", "1", "--dry-rccl", "--rccl-timeout", "120", "--dry-rows", "300001")
    assert rec["n_gpus"] == 1 and rec["all_ok"], rec
    want = {"communicator", "p2p_1_byte_with_every_peer", "gather_codes", "native_qinco_gather_codes"}
    assert set(rec["steps_ok_on_all_ranks"]) == want and all(rec["steps_ok_on_all_ranks"].values()), rec
    st = rec["per_rank"][0]["steps"]
    assert st["communicator"]["backend"] == "rccl" and st["communicator"]["ranks"] == 1
    assert st["p2p_1_byte_with_every_peer"]["self"] is True
    g = st["gather_codes"]
    assert g["transport"] == "nccl" and g["buffers"] == "device" and g["wire_dtype"] == "uint8"
    assert g["bytes_sent"] == g["bytes_received"] == 300001 * 8                # (the whole shard went out and came back)
    assert rec["nccl_comm_count"] == 1 and "rccl" in os.path.basename(rec["rccl_library"])
    print("RCCL:", rec["rccl_library"], rec["torch_nccl_version"])


def test_gather_codes_self_transfer_on_rccl_every_wire_type():
    """encode_db.gather_codes on an `nccl` group of one rank with self_transfer: the three wire types (bytes; int32 with an IVF
    column; int64) and an empty shard, each through all_reduce + one grouped isend / irecv on device buffers."""
    code = r"""
import json, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from qinco_amd.encode_db import gather_codes
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:" + sys.argv[2])
g = dist.new_group(backend="nccl", device_id=torch.device("cuda", 0))
rng = np.random.default_rng(3)
out = []
for hi, dt, wire in ((256, np.int64, "uint8"), (2 ** 20, np.int32, "int32"), (2 ** 40, np.int64, "int64")):
    for n in (1, 4099, 0):
        a = rng.integers(0, hi, size=(n, 9)).astype(dt)
        if n: a[0, 0] = hi - 1
        st = {}
        got = gather_codes(a, n, dist, device=torch.device("cuda", 0), group=g, stats=st, self_transfer=True)
        out.append({"equal": bool(np.array_equal(got, a.astype(np.int64))) and got.dtype == np.int64, "wire": st["wire_dtype"],
                    "want_wire": wire if n else "uint8", "sent": st["bytes_sent"], "recv": st["bytes_received"], "n": n})
print(json.dumps(out))
dist.destroy_process_group()
"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code, str(ROOT), str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("[{")][-1])
    assert len(out) == 9
    for c in out:
        assert c["equal"] and c["wire"] == c["want_wire"] and c["sent"] == c["recv"], c


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dry_rccl_on_the_gpu_box(world):
    """`bench.py --gpus N --dry-rccl`: the communication steps of the multi-GPU bench alone (payload group, a grouped 1-byte send /
    recv with every peer, the product's gather_codes on a million fake rows, and -- on RCCL -- ncclCommInitRank through ctypes +
    qinco_gather_codes).  With one GPU per rank this is RCCL over xGMI; on a 1-GPU box the ranks share GPU 0 and the payload rides
    on gloo (host buffers): the same lines of bench.py and encode_db.py either way."""
    import torch
    rccl = torch.cuda.device_count() >= world
    rec = _bench("--gpus", str(world), "--dry-rccl", "--backend", "nccl" if rccl else "gloo", "--rccl-timeout", "120")
    assert rec["n_gpus"] == world and rec["all_ok"], rec
    assert rec["rccl_ranks_seen"] == world and rec["rows"] == 1_000_000
    want = {"communicator", "p2p_1_byte_with_every_peer", "gather_codes"} | ({"native_qinco_gather_codes"} if rccl else set())
    assert set(rec["steps_ok_on_all_ranks"]) == want and all(rec["steps_ok_on_all_ranks"].values())
    if rccl:
        assert rec["nccl_comm_count"] == world
        assert len({p["pci_bus_id"] for p in rec["per_rank"]}) == world
        assert all(p["steps"]["gather_codes"]["transport"] == "nccl" and p["steps"]["gather_codes"]["buffers"] == "device" for p in rec["per_rank"])
    assert sum(p["rows"] for p in rec["per_rank"]) == 1_000_000


@pytest.mark.parametrize("mode", ["1", "hang"], ids=["raises", "hangs"])
def test_bench_fails_soft_when_rccl_is_unavailable(mode):
    """--backend nccl with more ranks than GPUs is refused up front; but an RCCL that fails -- or hangs -- at run time must
    not lose the run: the ranks write part files and the line says so.  Simulated with a test hook read by bench.py only
    (QINCO_BENCH_FORCE_GATHER_ERROR = "1": the gather raises; "hang": it never returns and --rccl-timeout ends the wait)."""
    import os
    os.environ["QINCO_BENCH_FORCE_GATHER_ERROR"] = mode
    try:
        rec = _bench("--gpus", "2", "--backend", "gloo", "--workload", "C1", "--batch", "256", "--steps", "1", "--warmup", "0",
                     "--rccl-timeout", "3")
    finally:
        del os.environ["QINCO_BENCH_FORCE_GATHER_ERROR"]
    mg = rec["multi_gpu"]
    assert rec["value"] > 0 and mg["gather"].startswith("failed:") and "part files" in mg["gather"]
    assert ("TimeoutError" in mg["gather"]) == (mode == "hang")
    assert not mg["gather_ok_on_all_ranks"] and all(v > 0 for v in mg["per_rank_encode_vectors_per_s"])


def test_bench_survives_a_real_rccl_refusal():
    """The RCCL code path itself (communicator creation on a `nccl` sub-group next to the gloo control plane), on hardware: two
    ranks on ONE GPU is something RCCL refuses (or never completes) -- the bench must notice, fall back to part files and still
    print its line.  On a box with >= 2 GPUs the same command simply succeeds over RCCL."""
    import os
    import torch
    os.environ["QINCO_BENCH_NCCL_SHARED_GPU"] = "1"
    try:
        rec = _bench("--gpus", "2", "--backend", "nccl", "--workload", "C1", "--batch", "256", "--steps", "1", "--warmup", "0",
                     "--rccl-timeout", "60")
    finally:
        del os.environ["QINCO_BENCH_NCCL_SHARED_GPU"]
    mg = rec["multi_gpu"]
    print(mg["gather"], "|", mg["rccl_note"])
    assert rec["value"] > 0 and all(v > 0 for v in mg["per_rank_encode_vectors_per_s"])
    if torch.cuda.device_count() >= 2:
        assert mg["gather"] == "rccl" and mg["gather_ok_on_all_ranks"]
    else:
        assert mg["gather"].startswith("failed:") and mg["rccl_note"] and not mg["gather_ok_on_all_ranks"]


def test_bench_driver_line_carries_the_metric_grid():
    """The default legs of the driver's N = 1 line (BASELINE.json metric: beam in {1, 8} over its configs): c1 with the
    oracle code-identity count and its CPU sample, c3, c4, and the bvecs -> encode_database leg (here on a small file)."""
    rec = _bench("--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--bvecs-vectors", "40000")
    c1 = rec["c1"]
    assert "error" not in c1, c1
    n_same, n_all = (int(v) for v in c1["greedy_rows_equal_to_oracle"].split("/"))
    assert n_all == 256 and n_same >= 254 and c1["cpu_baseline"]["value"] > 0 and c1["gpu_over_cpu"] > 10
    assert 0.5 < c1["roofline"]["frac"] <= 1.0 and c1["roofline"]["frac"] <= c1["roofline"]["frac_algorithmic"] + 1e-9
    for rows in (1024, 12288, 16384):      # decode at the reference's call sizes rides on the c1 and qinco2_S legs
        d = c1[f"decode_batch_{rows}"]
        assert d["rows_per_call"] == rows and d["value"] > 0 and 0 < d["roofline"]["frac"] <= 1.0
    assert c1["decode_batch_12288"]["roofline"]["frac"] > 0.6 and "weight_stream_gb_per_s" in c1["decode_batch_1024"]
    par = rec["parity"]
    for key in ("C1", "C2", "C2_beam1", "C3", "C4"):
        assert "error" not in par[key], par[key]
        same, rows = (int(v) for v in par[key]["codes_equal_to_reference"].split("/"))
        assert rows > 0 and same >= rows - 1 and par[key]["decode_max_rel_err"] < 1e-5, par[key]
    for k in ("qinco2_S", "ivf_qinco2_S"):
        assert "error" not in rec[k] and rec[k]["value"] > 0, rec[k]
    assert rec["ivf_qinco2_S"]["ivf"]["ivf_K"] == 1 << 20 and not rec["ivf_qinco2_S"]["ivf"]["fell_back_to_fp32_table"]
    for k, M, D in (("c3", 16, 128), ("c4", 8, 768)):
        assert "error" not in rec[k], rec[k]
        assert rec[k]["M"] == M and rec[k]["D"] == D and rec[k]["value"] > 0 and 0.5 < rec[k]["roofline"]["frac"] <= 1.0
    for key in ("encode_db_bvecs", "encode_db_bvecs_qinco2S"):
        db = rec[key]
        assert "error" not in db, db
        assert db["vectors"] == 40000 and db["codes_equal_to_resident_path"] and db["value"] > 0 and db["resident_value"] > 0
    assert "kernel_instances" in rec["config"]


def test_bench_line_schema_single_gpu():
    """A short single-GPU run carries every field of the contract line plus roofline / decode / mse / batch_1024."""
    rec = _bench("--workload", "C1", "--batch", "2048", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "decode", "mse", "batch_1024"):
        assert k in rec, k
    rf = rec["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and rf["frac"] <= rf["frac_algorithmic"] + 1e-9
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rec["decode"]["value"] > rec["value"] and 0 < rec["decode"]["roofline"]["frac"] <= 1.0
    for rows in (1024, 12288, 16384):
        assert 0 < rec[f"decode_batch_{rows}"]["roofline"]["frac"] <= 1.0
    assert rec["mse"]["value"] > 0 and rec["mse"]["vectors"] == 2 * 2048
    assert rec["config"]["distinct_vectors_encoded"] == 2 * 2048
