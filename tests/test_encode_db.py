"""Host logic around the hot path: sharding, part files, readers, and the world_size-2 gather on gloo (CPU).
The model object here is an oracle-backed stand-in (tests only); the product path uses QINCoHIP."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_model, make_oracle


def test_shard_bounds_match_reference_formula():
    from qinco_amd.encode_db import shard_bounds
    for N in (0, 1, 7, 8, 1000, 1_000_003, 10**9):
        for P in (1, 2, 3, 4, 8):
            spans = [shard_bounds(N, P, r) for r in range(P)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            for r in range(P):
                assert spans[r] == ((N // P) * r, (N // P) * (r + 1) if r < P - 1 else N)   # search_tasks.py:103-104
                if r:
                    assert spans[r][0] == spans[r - 1][1]


def test_vecs_readers(tmp_path):
    from qinco_amd.encode_db import get_data_memmap
    rs = np.random.RandomState(0)
    d, n = 12, 37
    xb = rs.randint(0, 256, (n, d)).astype(np.uint8)
    raw = np.zeros((n, d + 4), np.uint8)
    raw[:, :4] = np.frombuffer(np.int32(d).tobytes(), np.uint8)
    raw[:, 4:] = xb
    raw.tofile(tmp_path / "a.bvecs")
    got = get_data_memmap(str(tmp_path / "a.bvecs"))
    assert got.shape == (n, d) and got.dtype == np.uint8 and got.strides == (d + 4, 1) and np.array_equal(got, xb)
    xf = rs.randn(n, d).astype(np.float32)
    rawf = np.zeros((n, d + 1), np.float32)
    rawf[:, 0] = np.frombuffer(np.int32(d).tobytes(), np.float32)[0]
    rawf[:, 1:] = xf
    rawf.tofile(tmp_path / "a.fvecs")
    assert np.array_equal(get_data_memmap(str(tmp_path / "a.fvecs")), xf)
    xi = rs.randint(0, 1000, (n, d)).astype(np.int32)
    np.concatenate([np.full((n, 1), d, np.int32), xi], axis=1).tofile(tmp_path / "a.ivecs")
    assert np.array_equal(get_data_memmap(str(tmp_path / "a.ivecs")), xi)
    np.save(tmp_path / "a.npy", xf)
    assert np.array_equal(get_data_memmap(str(tmp_path / "a.npy")), xf)
    with pytest.raises(ValueError):
        get_data_memmap(str(tmp_path / "a.bin"))


class OracleModel:
    """Stand-in with the reference model's call signature (tests only)."""

    def __init__(self, name):
        self.cfg, self.sd = golden_model(name)
        self.o = make_oracle(self.cfg, self.sd)

    def __call__(self, x, step):
        return self.o(np.asarray(x, np.float32), step=step)


def test_encode_database_single_process_files(tmp_path):
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import EncodedDBIterator, encode_database
    model = OracleModel("tiny_proj_greedyA")
    cfg = model.cfg
    db = synth_vectors(cfg, model.sd, 301, seed=4)
    out = str(tmp_path / "enc" / "db.npz")
    codes = encode_database(model, db, out, K=cfg.K, M=cfg.M, D=cfg.D, batch=64)
    assert codes.shape == (301, cfg.M) and codes.dtype == np.int64
    assert np.array_equal(codes, model(db, step="encode").T)              # batching does not change codes
    hdr = np.load(out)
    assert {k: int(hdr[k]) for k in hdr.files} == {"n_parts": 1, "K": cfg.K, "M": cfg.M, "D": cfg.D}
    part = np.load(out[:-4] + ".part_0.npz")["codes"]
    assert part.dtype == np.int64 and np.array_equal(part, codes)
    it = EncodedDBIterator(out, K=cfg.K, M=cfg.M, D=cfg.D)
    assert np.array_equal(it.load_all(), codes)
    chunks = list(it.iter(batch_size=100))
    assert [len(c) for c in chunks] == [100, 100, 100, 1] and it.batch_end_id == 301
    with pytest.raises(AssertionError):
        EncodedDBIterator(out, M=cfg.M + 1)


@pytest.mark.parametrize("threads", [4, 0])
def test_encode_database_resume_skips_finished_part_files(tmp_path, threads):
    """The reference's encode_database has no resume (SURVEY: a crashed rank loses the job).  Part files are written under a
    temporary name and renamed when whole; resume=True loads a finished part instead of encoding the shard again, and
    encodes it again when the file is missing, truncated or of another size."""
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import encode_database
    model = OracleModel("tiny_proj_greedyA")
    cfg = model.cfg
    db = synth_vectors(cfg, model.sd, 150, seed=4)
    calls = []

    def counting(x, step):
        calls.append(len(x))
        return model(x, step)
    out = str(tmp_path / "db.npz")
    part = out[:-4] + ".part_0.npz"
    kw = dict(K=cfg.K, M=cfg.M, D=cfg.D, batch=64, writer_threads=threads)
    first = encode_database(counting, db, out, **kw)
    assert sum(calls) == 150 and os.path.exists(part) and not [f for f in os.listdir(tmp_path) if ".tmp" in f]
    calls.clear()
    again = encode_database(counting, db, out, resume=True, **kw)
    assert not calls and np.array_equal(again, first)                     # nothing encoded, same codes
    blob = open(part, "rb").read()
    open(part, "wb").write(blob[: len(blob) // 2])                          # a truncated file is not a finished one
    redone = encode_database(counting, db, out, resume=True, **kw)
    assert sum(calls) == 150 and np.array_equal(redone, first) and np.array_equal(np.load(part)["codes"], first)
    calls.clear()
    encode_database(counting, db[:100], out, resume=True, **kw)            # another shard size: encoded again
    assert sum(calls) == 100 and len(np.load(part)["codes"]) == 100
    calls.clear()
    encode_database(counting, db[:100], out, **kw)                         # resume is opt-in
    assert sum(calls) == 100


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, n):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import encode_database
    model = OracleModel("tiny_proj_greedyA")
    cfg = model.cfg
    db = synth_vectors(cfg, model.sd, n, seed=4)
    full = encode_database(model, db, os.path.join(outdir, "db.npz"), K=cfg.K, M=cfg.M, D=cfg.D, batch=50,
                           dist=dist, gather=True)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 203), (3, 100)])
def test_sharded_encode_equals_single_process_gloo(tmp_path, world, n):
    """N>1 path on CPU: codes from P ranks (part files AND the gathered matrix) equal the 1-process codes bitwise."""
    import torch.multiprocessing as mp
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import EncodedDBIterator, shard_bounds
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    model = OracleModel("tiny_proj_greedyA")
    want = model(synth_vectors(model.cfg, model.sd, n, seed=4), step="encode").T
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), want)
    it = EncodedDBIterator(str(tmp_path / "db.npz"))
    assert it.n_parts == world and np.array_equal(it.load_all(), want)
    for r in range(world):
        s, e = shard_bounds(n, world, r)
        assert len(np.load(tmp_path / f"db.part_{r}.npz")["codes"]) == e - s


def _resume_worker(rank, world, port, outdir, n):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import encode_database
    model = OracleModel("tiny_proj_greedyA")
    cfg = model.cfg
    db = synth_vectors(cfg, model.sd, n, seed=4)
    rows = []

    def counting(x, step):
        rows.append(len(x))
        return model(x, step)
    full = encode_database(counting, db, os.path.join(outdir, "db.npz"), K=cfg.K, M=cfg.M, D=cfg.D, batch=50, dist=dist, gather=True,
                           resume=True)
    open(os.path.join(outdir, f"rows_{rank}.txt"), "w").write(str(sum(rows)))
    if rank == 0:
        np.save(os.path.join(outdir, "gathered_resume.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_resume_after_one_rank_was_lost_gloo(tmp_path):
    """Three ranks, the part file of rank 1 is gone (the rank died): a second run with resume=True encodes that shard only, every
    rank still meets the barriers and the gather, and rank 0 gets the same matrix."""
    import torch.multiprocessing as mp
    from qinco_amd.encode_db import shard_bounds
    world, n = 3, 160
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    os.remove(tmp_path / "db.part_1.npz")
    mp.spawn(_resume_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    s, e = shard_bounds(n, world, 1)
    assert [int(open(tmp_path / f"rows_{r}.txt").read()) for r in range(world)] == [0, e - s, 0]
    assert np.array_equal(np.load(tmp_path / "gathered_resume.npy"), np.load(tmp_path / "gathered.npy"))
    assert len(np.load(tmp_path / "db.part_1.npz")["codes"]) == e - s


def test_part_file_writer_is_the_reference_format(tmp_path):
    """PartFileWriter (parallel, incremental deflate) must produce what np.savez_compressed(path, codes=...) means to every
    reader: np.load gives the same int64 array, zipfile's CRC check passes, sizes beyond one chunk and exact chunk boundaries
    included; and encode_database writes the same part file with and without it."""
    import zipfile
    from qinco_amd.encode_db import EncodedDBIterator, PartFileWriter, encode_database
    rs = np.random.RandomState(0)
    hdr = 128
    for rows in (0, 1, 777, 70001, (2 * PartFileWriter.CHUNK - hdr) // 64):
        c = rs.randint(0, 256, (rows, 8)).astype(np.int64)
        p = str(tmp_path / f"w{rows}.npz")
        w = PartFileWriter(p, rows, 8, threads=3)
        for i in range(0, rows, 20000):
            w.add(c[i:i + 20000])
        w.close()
        got = np.load(p)["codes"]
        assert got.dtype == np.int64 and got.shape == c.shape and np.array_equal(got, c)
        assert zipfile.ZipFile(p).testzip() is None and zipfile.ZipFile(p).namelist() == ["codes.npy"]
    # an empty batch behind the last rows (a producer that flushes once more) leaves the closed deflate stream alone
    c = rs.randint(0, 256, (3000, 8)).astype(np.int64)
    p = str(tmp_path / "tail.npz")
    w = PartFileWriter(p, 3000, 8, threads=2)
    w.add(c)
    w.add(c[:0])
    w.close()
    raw = open(p, "rb").read()
    w2 = PartFileWriter(str(tmp_path / "tail2.npz"), 3000, 8, threads=2)
    w2.add(c)
    w2.close()
    assert raw == open(tmp_path / "tail2.npz", "rb").read() and np.array_equal(np.load(p)["codes"], c)
    assert zipfile.ZipFile(p).testzip() is None
    model = OracleModel("tiny_proj_beam")
    from qinco_amd import synth_vectors
    db = synth_vectors(model.cfg, model.sd, 130, seed=2)
    outs = []
    for threads in (0, 4):
        out = str(tmp_path / f"t{threads}" / "db.npz")
        encode_database(model, db, out, K=model.cfg.K, M=model.cfg.M, D=model.cfg.D, batch=50, writer_threads=threads)
        outs.append(EncodedDBIterator(out).load_all())
    assert np.array_equal(outs[0], outs[1])


def test_numa_cpu_set_of_a_gpu_from_sysfs(tmp_path, monkeypatch):
    """qinco_amd.affinity: PCI address -> sysfs numa_node -> node cpulist (the lookup bench.py's ranks bind themselves with);
    anything missing is "leave the affinity alone", never an error."""
    from qinco_amd import affinity
    assert affinity._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    (tmp_path / "bus/pci/devices/0000:c1:00.0").mkdir(parents=True)
    (tmp_path / "bus/pci/devices/0000:c1:00.0/numa_node").write_text("1\n")
    (tmp_path / "devices/system/node/node1").mkdir(parents=True)
    (tmp_path / "devices/system/node/node1/cpulist").write_text("64-127,192-255\n")
    monkeypatch.setattr(affinity, "gpu_pci_address", lambda i: "0000:c1:00.0" if i == 0 else None)
    node, cpus = affinity.numa_cpus_of_gpu(0, sysfs=str(tmp_path))
    assert node == 1 and len(cpus) == 128 and cpus[0] == 64 and cpus[-1] == 255
    assert affinity.numa_cpus_of_gpu(1, sysfs=str(tmp_path)) is None          # no PCI address
    (tmp_path / "bus/pci/devices/0000:c1:00.0/numa_node").write_text("-1\n")     # single-node box
    assert affinity.numa_cpus_of_gpu(0, sysfs=str(tmp_path)) is None
    assert affinity.bind_to_gpu_numa(0) is None or isinstance(affinity.bind_to_gpu_numa(0), dict)


def test_host_memory_of_a_50_million_vector_encode(tmp_path):
    """The north_star's job is 10^9 vectors x 8 codes: an 8 GB payload.  Codes must stay bytes on the host from the model call to the
    gathered matrix (the reference's int64 everywhere would be 64 GB per copy, and the round-3 code held two such copies on rank 0).
    A 50-million-vector encode with a stand-in model in a process of its own: peak resident memory stays near ONE copy of the
    byte payload (400 MB) + a few batches, far below the 3.2 GB of a single int64 copy; the part file still holds int64 codes."""
    import subprocess
    script = r'''
import os, resource, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from qinco_amd.encode_db import EncodedDBIterator, encode_database
N, M, B = 50_000_000, 8, 1_000_000
class DB:                      # (N, 4) uint8 rows made on demand: the input side holds one batch
    def __len__(self): return N
    def __getitem__(self, sl):
        i = np.arange(sl.start, sl.stop, dtype=np.int64)
        return ((i[:, None] * np.array([1, 3, 5, 7])) & 255).astype(np.uint8)
def model(x, step):            # (M, n) int64 like the reference's model object
    s = x.astype(np.int64).sum(axis=1)
    return ((s[None, :] + np.arange(M)[:, None] * 17) & 255)
base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
codes = encode_database(model, DB(), os.path.join(sys.argv[2], "db.npz"), K=256, M=M, D=4, batch=B, code_dtype="compact", gather=True)
peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
assert codes.dtype == np.uint8 and codes.shape == (N, M)
it = EncodedDBIterator(os.path.join(sys.argv[2], "db.npz"), K=256, M=M, D=4)
first = next(it.iter(1000))
assert first.dtype == np.int64 and np.array_equal(first, codes[:1000])
print("PEAK_MB", (peak - base) / 1024.0)
'''
    r = subprocess.run([sys.executable, "-c", script, str(ROOT), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    peak_mb = float(r.stdout.split("PEAK_MB")[1])
    assert peak_mb < 1200, f"peak resident growth {peak_mb:.0f} MB for a 400 MB byte payload"


def _payload_group_worker(rank, world, port, outdir, n):
    """gather_codes on a payload group that is NOT the default group (bench.py: gloo control plane + a sub-group for the codes)."""
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qinco_amd.encode_db import gather_codes, shard_bounds
    payload = dist.new_group(backend="gloo")
    s, e = shard_bounds(n, world, rank)
    rows = np.arange(s, e, dtype=np.int64)[:, None]
    mine = ((rows * 7 + np.arange(5)[None, :]) % 256).astype(np.uint8)
    if n == 3 * 41 + 2:                       # one job also carries a wide column (an IVF id): the wire type widens on every rank
        mine = mine.astype(np.int32)
        mine[:, 0] = rows[:, 0] * 1000
    stats = {}
    full = gather_codes(mine, n, dist, code_dtype="compact", group=payload, stats=stats)
    np.save(os.path.join(outdir, f"stats_{rank}.npy"), np.array([stats["ranks"], stats["bytes_sent"], stats["bytes_received"]]))
    if rank == 0:
        np.save(os.path.join(outdir, "full.npy"), full)
        open(os.path.join(outdir, "wire.txt"), "w").write(stats["wire_dtype"] + " " + stats["transport"] + " " + stats["buffers"])
    else:
        assert full is mine
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 101), (3, 3 * 41 + 2), (8, 8 * 13 + 5), (4, 2)])
def test_gather_codes_on_a_payload_group_gloo(tmp_path, world, n):
    """The end-of-job transfer at 2, 3 and 8 ranks (uneven last shard; N < world: empty shards send nothing): rank 0's matrix is the
    database's codes in file order, and the stats say what moved."""
    import torch.multiprocessing as mp
    from qinco_amd.encode_db import shard_bounds
    mp.spawn(_payload_group_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    rows = np.arange(n, dtype=np.int64)[:, None]
    want = (rows * 7 + np.arange(5)[None, :]) % 256
    wide = n == 3 * 41 + 2
    if wide:
        want[:, 0] = rows[:, 0] * 1000
    full = np.load(tmp_path / "full.npy")
    assert full.dtype == (np.int32 if wide else np.uint8) and np.array_equal(full, want)
    assert open(tmp_path / "wire.txt").read() == ("int32" if wide else "uint8") + " gloo host"
    isz = 4 if wide else 1
    for r in range(world):
        s, e = shard_bounds(n, world, r)
        ranks, sent, recv = np.load(tmp_path / f"stats_{r}.npy")
        assert ranks == world and sent == (0 if r == 0 else (e - s) * 5 * isz)
        assert recv == ((n - (shard_bounds(n, world, 0)[1])) * 5 * isz if r == 0 else 0)


def _keep_false_worker(rank, world, port, outdir, n):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import encode_database
    model = OracleModel("tiny_proj_greedyA")
    cfg = model.cfg
    db = synth_vectors(cfg, model.sd, n, seed=4)
    assert encode_database(model, db, os.path.join(outdir, "db.npz"), K=cfg.K, M=cfg.M, D=cfg.D, batch=50, dist=dist, keep=False) is None
    dist.barrier()
    dist.destroy_process_group()


def test_empty_shards_write_their_part_files_with_keep_false_gloo(tmp_path):
    """N < world: the ranks in front hold EMPTY shards (search_tasks.py:103-104 gives the remainder to the last one).  The header
    says n_parts = world, so each of them owes the readers a part file -- also in the part-files-only mode (keep=False), where
    round 4 wrote none and EncodedDBIterator failed on the missing file."""
    import torch.multiprocessing as mp
    from qinco_amd import synth_vectors
    from qinco_amd.encode_db import EncodedDBIterator
    world, n = 3, 2
    mp.spawn(_keep_false_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    model = OracleModel("tiny_proj_greedyA")
    want = model(synth_vectors(model.cfg, model.sd, n, seed=4), step="encode").T
    it = EncodedDBIterator(str(tmp_path / "db.npz"))
    assert it.n_parts == world and np.array_equal(it.load_all(), want)
    assert [len(np.load(tmp_path / f"db.part_{r}.npz")["codes"]) for r in range(world)] == [0, 0, 2]
    assert all(np.load(tmp_path / f"db.part_{r}.npz")["codes"].dtype == np.int64 for r in range(world))


def test_rccl_id_file_is_matched_by_nonce_not_by_clock(tmp_path):
    """RcclComm's id exchange (qinco_amd/comm.py), the waiting ranks' side -- no RCCL call is made here.  With a job nonce a file is
    this job's iff it ends in the nonce: a day-old file with the right nonce is taken (a rank may start arbitrarily late, clocks
    may disagree), a brand-new file of another job is not.  Without a nonce the single-host rule applies: not older than
    max_skew_s (30 s) before this rank's start, whatever timeout_s the rank is prepared to wait."""
    import time
    from qinco_amd.comm import RcclComm
    c = RcclComm.__new__(RcclComm)
    f = str(tmp_path / "uid")
    blob = bytes(range(128))
    open(f, "wb").write(blob + b"job-41")
    old = time.time() - 86400
    os.utime(f, (old, old))
    uid = c._exchange_through_file(1, f, 0.3, "job-41")
    assert bytes(uid.internal) == blob
    with pytest.raises(TimeoutError, match="job-42"):
        c._exchange_through_file(1, f, 0.3, "job-42")                       # another job's file, however fresh
    open(f, "wb").write(blob)                                               # no nonce: the mtime rule, max_skew_s of slack
    assert bytes(c._exchange_through_file(2, f, 5.0, None).internal) == blob
    minute_old = time.time() - 60                                           # a job that died a minute ago left it: not ours,
    os.utime(f, (minute_old, minute_old))                                   # although this rank would wait 120 s for rank 0
    with pytest.raises(TimeoutError):
        c._exchange_through_file(2, f, 0.3, None)
    assert bytes(c._exchange_through_file(2, f, 0.3, None, max_skew_s=90.0).internal) == blob
    os.utime(f, (old, old))
    with pytest.raises(TimeoutError):
        c._exchange_through_file(2, f, 0.3, None)
    with pytest.raises(ValueError):
        RcclComm.__new__(RcclComm).__init__(0, 1)                           # neither an id nor a way to get one


def test_gather_codes_self_transfer_needs_rccl():
    """gather_codes(self_transfer=True): a job of one rank sends its shard to itself through the function's own transfer lines -- the
    way a 1-GPU box executes them on RCCL (tests/test_multi_gpu.py).  gloo has no connection from a rank to itself: a clear error, not
    gloo's 'Pair is not connected'; without a process group, or without the flag, one rank keeps returning its shard."""
    import subprocess
    import sys
    from qinco_amd.encode_db import gather_codes
    a = np.arange(40, dtype=np.int64).reshape(5, 8)
    assert gather_codes(a, 5, None, self_transfer=True) is a                       # no process group: nothing to run through
    code = """
import sys, numpy as np, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from qinco_amd.encode_db import gather_codes
dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:" + sys.argv[2])
a = np.arange(40, dtype=np.int64).reshape(5, 8)
assert gather_codes(a, 5, dist) is a
try:
    gather_codes(a, 5, dist, self_transfer=True)
    print("NO ERROR")
except ValueError as e:
    print("ValueError:", e)
dist.destroy_process_group()
"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-c", code, str(ROOT), str(port)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ValueError: gather_codes(self_transfer=True) needs a payload group on RCCL" in r.stdout, r.stdout
