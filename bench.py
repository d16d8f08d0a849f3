#!/usr/bin/env python
"""bench.py -- encode throughput of the MI355X QINCo2 engine on BASELINE.json's metric.

    python bench.py [--gpus N --steps K --warmup W] [--workload C2] [--batch 16384] [--scaling weak|strong]

`--gpus N` with N > 1 works both ways: pre-launched (the driver's `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) and from a bare `python bench.py --gpus N`, which re-executes itself under torch.distributed.run with one
rank per GPU (like the reference's `accelerate launch --multi_gpu`, run.sh:8).

A "step" = one pass of the hot path (model(x, step="encode")) over one batch of `--batch` synthetic fp32 vectors per
GPU.  Every step (and every warm-up step) encodes a DIFFERENT batch; all batches are generated on the device before the
timed region (inputs resident in HBM).  The timed region is the PRODUCT's database encode at every N: the rank's contiguous range
(search_tasks.py:103-104) goes through qinco_amd.encode_db.encode_shard -- QINCoHIP model calls in passes of `--batch` rows, codes
narrowed to bytes and copied to the host under the next pass -- and, for N > 1, through encode_db.gather_codes: no data-path
collective, one send of the byte shard per rank > 0, rank 0 receives each into its rows of one (N, M) matrix (RCCL; inside the
timed region).  multi_gpu.per_rank says, per rank: GPU index, PCI bus id, NUMA node, encode s, gather s, bytes, the ranks the
payload group saw, and whether the rows that arrived on rank 0 are the shard's (a position-weighted checksum).
`--dry-rccl` runs only the communication steps on fake rows (seconds: first contact with a multi-GPU node).
  --scaling weak   (default, the driver's form): every rank encodes K batches of its own -> work grows with N;
  --scaling strong: ONE database of K x batch vectors (or --db N) is split over the ranks like encode_database does
                    (the north_star's "1B-vector encode" statement); a step is then 1/K of the whole job.
Control plane (barriers, timing exchange) runs on a gloo group, the payload gather on RCCL (`--backend nccl`); if RCCL
cannot be initialised or the gather raises, the ranks fall back to per-rank part files (the reference's own output format)
and the line says so in multi_gpu.gather -- a scaling run still yields per-rank vectors/s.  Each rank binds itself to the
CPU set of its GPU's NUMA node (qinco_amd/affinity.py).  Rank 0 prints ONE JSON line.

metric/unit: encode vectors/s (BASELINE.json "metric"); workload C2 = qinco2-L 8x8, D=128, A=16, B=8
(BASELINE.json configs[1]) with seeded synthetic weights (no trained checkpoints offline).
roofline: the fused codeword-MLP kernel (99.9 % of the FLOPs), fp32 MFMA bound.  The kernels take the row-independent head of
the MLP out of the per-row work, so the matrix pipe EXECUTES fewer FLOPs than the reference's algorithm counts.
  `frac`             = executed FLOPs per launch / mean launch duration / peak -- a pipe utilisation, <= 1 by construction; the
                       library counts the executed FLOPs per launch from the kernel form that ran (qinco_profile_read2), the
                       duration is HIP events on the launch stream; profiles/ holds the SQ_INSTS_VALU_MFMA_MOPS_F32 passes that
                       check the count;
  `achieved`         = `frac` x peak (TFLOP/s through the pipe);
  `frac_algorithmic` = ALGORITHMIC FLOPs (rows x R_mlp) / duration / peak: the figure comparable with the
                       reference's own FLOP count -- it exceeds `frac`, and 1 on short models.
Also in the line (N = 1, measured after the timed region):
  `decode` -- every timed code row in one call; `decode_batch_1024`, `decode_batch_12288` -- the call sizes the reference
          decodes at (compute_MSE: cfg.batch = 1024 rows per call, qinco_tasks.py:112-125; the search re-rank:
          cfg.search.batch_size = 12 288, search_tasks.py:475-486), through the small-launch form of the fused MLP;
  `parity` -- rows of the committed reference fixtures (tests/golden: inputs + the imported reference's own codes) that this
          build reproduces, C1 .. C4, run here on the GPU;
  `mse`, `beam1`, `batch_1024`, `split_f16` -- as in round 2;
  `c1` -- BASELINE configs[0] / the north_star's literal target (qinco1 8x8, greedy): vectors/s, roofline, the number of
          code rows equal to the oracle's on a 256-vector sample, and the oracle's own rate on that sample (CPU baseline);
  `c3`, `c4` -- BASELINE configs[2] (M = 16) and configs[3] (D = 768) at their bench batch, with roofline;
  `qinco2_S`, `ivf_qinco2_S` -- the small preset and its IVF form (ivf_K = 2^20: the reference's billion-scale family);
  `encode_db_bvecs` (C2) and `encode_db_bvecs_qinco2S` -- the actual `task=encode` data path (search_tasks.py:85-137): a uint8 .bvecs file on disk ->
          get_data_memmap -> encode_database(QINCoHIP) -> part file, i.e. host buffers / PCIe included, next to the
          HBM-resident rate of the same model.
  `small_db_search` -- f3 at bigann1M's shape: both forms of the brute-force top-100, bit-equality asserted on the line;
  `roofline.traffic` (round 6) -- L2<->fabric bytes per launch of the dominant kernel, MEASURED for this run's command by two
          rocprofv3 --pmc child passes (FETCH_SIZE x2, WRITE_SIZE) that run on the idle GPU while the CPU baseline is timed
          (`--no-pmc`: fall back to the committed pass's bytes per row, and say so);
  `rccl_world1`, `rccl_world1_ok` (round 6) -- `bench.py --gpus 1 --dry-rccl` as a child process: every RCCL step of the
          multi-GPU bench on the real library with a communicator of ONE rank (`--no-rccl-check` skips it).
cpu_baseline: the oracle restatement with its codeword MLP on torch CPU ops (oracle/qinco_oracle.py, backend "torch":
same op sequence as the reference's CPU path, codes equal to the numpy oracle and to the imported reference) timed on
this box's host cores at the reference's batch of 1024 on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
PEAK_F16_MFMA_TFLOPS = 2516.6   # dense fp16 / bf16 MFMA peak at 2.4 GHz (16 x the fp32-in rate; MI355X_MICROARCH.md)
# measured in the build container (8 cores, C2, 256 vectors): imported reference wrapper 46.5 vec/s,
# this port (torch backend) 42.3 vec/s, numpy oracle 9.3 vec/s -- all three give identical codes
REF_OVER_PORT_CONTAINER = 1.10
WORKLOAD_NAMES = {"C1": "C1: qinco1 8x8 greedy encode", "C2": "C2: qinco2-L 8x8 encode", "C3": "C3: qinco2-L 16x8 encode",
                  "C4": "C4: qinco2-L 8x8 encode, D=768", "S": "qinco2-S 8x8 encode",
                  "IVF_S": "IVF-qinco2-S 8x8 encode, ivf_K = 2^20 (the reference's billion-scale model family)"}


def pmc_traffic_per_row():
    """L2<->fabric bytes per MLP row from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE).  Counters cannot be read from inside this process, so the figure of the separate PMC run of this
    same command is scaled to this run's rows per launch."""
    for name in ("r05f_c2_traffic.json", "r04_c2_traffic.json", "r03_c2_traffic.json", "r02_c2_traffic.json", "r01_c2_traffic.json"):
        try:
            with open(ROOT / "profiles" / name) as f:
                return float(json.load(f)["bytes_per_row"]), name
        except Exception:
            continue
    return None, None


def pmc_traffic_live(args, kernel_substr: str = "mlp_kernel<") -> dict:
    """`roofline.traffic` measured for THIS run's command: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
    pass -- MI355X_MICROARCH.md, PMC slots) of `bench.py --steps 2 --warmup 1` with this run's workload and batch, as child
    processes (counters cannot be read from inside the timed process), untimed.  Per pass: the dispatches of the dominant kernel
    at its largest grid (the full-size launches of the timed loop), mean counter value and mean duration.  FETCH_SIZE / WRITE_SIZE
    are KiB; FETCH_SIZE counts half of the bytes of wide coalesced reads on gfx950 (same guide): x2.  Returns {"ok": False, "error"}
    when rocprofv3 is not there or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"ok": False, "error": "rocprofv3 not found"}
    child = [sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--workload", args.workload, "--batch", str(args.batch),
             "--no-cpu-baseline", "--no-extras", "--no-legs", "--no-rccl-check", "--no-pmc", "--no-affinity"]
    if getattr(args, "split_f16", False):
        child.append("--split-f16")
        kernel_substr = "mlp_split_kernel<"
    env = dict(os.environ, TMPDIR="/tmp")
    out = {"ok": True, "passes": {}}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="qinco_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "t", "--", *child], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return {"ok": False, "error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(dbs)} db; {r.stderr[-200:]}"}
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, grid_size, value, duration from counters_collection where counter_name = ? and "
                               "kernel_name like ?", (counter, f"%{kernel_substr}%")).fetchall()
            if not rows:
                return {"ok": False, "error": f"no {kernel_substr} dispatch with {counter} in the pass"}
            gmax = max(g for _, g, _, _ in rows)
            full = [(v, du, k) for k, g, v, du in rows if g == gmax]
            out["passes"][counter] = {"kernel": full[0][2][:80], "dispatches": len(full), "grid_threads": int(gmax),
                                      "mean_kib": sum(v for v, _, _ in full) / len(full),
                                      "mean_duration_us": sum(du for _, du, _ in full) / len(full) / 1e3}
        except Exception as e:                                # noqa: BLE001 -- never the headline's problem
            return {"ok": False, "error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w = out["passes"]["FETCH_SIZE"], out["passes"]["WRITE_SIZE"]
    out["fetch_bytes_x2"] = f["mean_kib"] * 1024 * 2
    out["write_bytes"] = w["mean_kib"] * 1024
    out["bytes_per_launch"] = out["fetch_bytes_x2"] + out["write_bytes"]
    out["seconds"] = time.perf_counter() - t0
    return out


def cpu_threads(torch) -> int:
    # The reference's own CPU protocol runs 32 ATen threads (qinco_tasks.py:492).  On this box (128 cores) more threads are
    # slower: 16 / 32 / 64 / 128 threads gave 39 / 41 / 33 / 20 vectors/s at batch 1024 (profiles/r02_cpu_sweep.jsonl).
    threads = min(32, int(torch.get_num_threads()))
    torch.set_num_threads(threads)
    return threads


def cpu_baseline(cfg, sd, budget_s: float = 20.0, chunk: int = 1024):
    """Oracle encode throughput on the host cores (rank 0, N=1 only): bounded sample of the same workload at the
    reference's default batch (qinco_cfg.yaml:38), ATen threads = the cores torch picked (stated)."""
    import torch
    from oracle.qinco_oracle import OracleQINCo
    from qinco_amd import synth_vectors
    threads = cpu_threads(torch)
    oracle = OracleQINCo.from_config(cfg, sd, backend="torch")
    x = synth_vectors(cfg, sd, 16 * chunk, seed=4242)
    oracle(x[:64], step="encode")  # warm-up
    done, t0 = 0, time.perf_counter()
    while done < len(x):
        oracle(x[done:done + chunk], step="encode")
        done += chunk
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    v = done / dt
    return {"value": v, "unit": "vectors/s", "cores": threads, "threads": threads, "kind": "port",
            "gflops": v * cfg.encode_flops_per_vector() / 1e9,
            "reference_over_port_in_build_container": REF_OVER_PORT_CONTAINER,
            "sample": f"{done} vectors of the same workload in {dt:.1f} s (oracle restatement, codeword MLP on torch CPU "
                      f"fp32 ops, batches of {chunk} like qinco_cfg.yaml:38, {threads} ATen threads like qinco_tasks.py:492, "
                      f"{os.cpu_count()} logical cores on the box)"}


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int) -> int:
    """A bare `python bench.py --gpus N`: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rccl_world_of_one(timeout_s: float) -> dict:
    """The N = 1 line's `rccl_world1`: `bench.py --gpus 1 --dry-rccl` in a child process (so that an RCCL that hangs or aborts
    cannot take the bench line with it) -- a communicator of ONE rank on the real librccl: torch.distributed `nccl` group +
    all_reduce + grouped isend / irecv to self + the product's gather_codes through its transfer lines, then ncclCommInitRank through
    ctypes + qinco_gather_codes (grouped ncclSend / ncclRecv to self).  What a 1-GPU lease can execute of SURVEY 8(e)'s RCCL path."""
    cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--dry-rccl", "--dry-rows", "200000", "--no-affinity",
           "--rccl-timeout", str(min(timeout_s, 120.0))]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(timeout_s, 120.0) * 3 + 120)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        steps = rec["per_rank"][0]["steps"]
        return {"all_ok": bool(rec["all_ok"]), "steps": {k: (True if v["ok"] else v.get("error")) for k, v in steps.items()},
                "rccl_ranks_seen": rec.get("rccl_ranks_seen"), "nccl_comm_count": rec.get("nccl_comm_count"),
                "rccl_library": rec.get("rccl_library"), "torch_nccl_version": rec.get("torch_nccl_version"),
                "seconds": time.perf_counter() - t0, "command": "bench.py --gpus 1 --dry-rccl (child process, untimed)"}
    except Exception as e:                                    # noqa: BLE001 -- never the headline's problem
        return {"all_ok": False, "error": f"{type(e).__name__}: {e}"[:300], "seconds": time.perf_counter() - t0}


def call_with_timeout(fn, seconds):
    """Run fn() on a helper thread; TimeoutError if it has not returned after `seconds` (a wedged RCCL call must not take
    the whole run with it: the caller falls back to part files and the process leaves through os._exit at the end)."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:                            # noqa: BLE001 -- re-raised on the caller's thread
            box["error"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        raise TimeoutError(f"no answer after {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box.get("value")


def shard_checksum(codes: np.ndarray, first_row: int) -> int:
    """Position-weighted 64-bit sum of a block of code rows (row index in the DATABASE, so a shard that lands in the wrong rows of
    the gathered matrix -- or in the right rows in the wrong order -- changes it)."""
    total = np.uint64(0)
    if codes.size == 0:
        return 0
    cols = (np.arange(codes.shape[1], dtype=np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(1))[None, :]
    with np.errstate(over="ignore"):
        for i0 in range(0, len(codes), 1 << 20):           # (blocks: the uint64 temporaries of 10^8 rows would be gigabytes)
            blk = codes[i0:i0 + (1 << 20)]
            rows = (np.arange(first_row + i0, first_row + i0 + len(blk), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))[:, None]
            total = total + ((blk.astype(np.uint64) + np.uint64(1)) * rows * cols).sum(dtype=np.uint64)
    return int(total)


def fake_code_rows(first_row: int, rows: int, M: int) -> np.ndarray:
    """Deterministic byte codes of database rows [first_row, first_row + rows): every rank can make its shard and rank 0 the whole
    matrix it must receive (--dry-rccl)."""
    r = np.arange(first_row, first_row + rows, dtype=np.uint64)[:, None]
    c = np.arange(M, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        return (((r * np.uint64(2654435761) + c * np.uint64(40503)) >> np.uint64(7)) & np.uint64(255)).astype(np.uint8)


def dry_rccl(torch, dist, args, rank, world, dev, dev_index, data_group, data_note, affinity, wedged, json_fd):
    """--dry-rccl: first contact with a multi-GPU node without a model in the way.  Per rank, each step under --rccl-timeout and
    each recorded with its seconds or its error:
      1. communicator: the payload group was created (and probed with one all-reduce) in main() -- RCCL with --backend nccl;
      2. p2p: one grouped exchange of 1 byte with EVERY peer (batch_isend_irecv: the ncclGroupStart / ncclSend / ncclRecv /
         ncclGroupEnd shape of qinco_gather_codes and of gather_codes' transfers);
      3. gather_codes: the product's end-of-job gather (qinco_amd.encode_db) of --dry-rows fake byte rows x 8 codes, sharded like
         encode_database shards (the last rank takes the remainder); rank 0 compares every received row with what it must be;
      4. native: RcclComm.from_process_group (ncclCommInitRank through ctypes, the id broadcast on the control plane), ncclCommCount,
         and qinco_gather_codes (include/qinco_hip.h) on the same rows -- only with --backend nccl (it needs one GPU per rank).
    Rank 0 prints one JSON line; exit status 0 even when steps failed (the line says which)."""
    from qinco_amd.affinity import gpu_pci_address
    from qinco_amd.encode_db import gather_codes, shard_bounds
    M = 8
    N = args.dry_rows
    start, end = shard_bounds(N, world, rank)
    mine_np = fake_code_rows(start, end - start, M)
    steps = {}
    use_rccl = data_group is not None
    pdev = dev if use_rccl else torch.device("cpu")

    def step(name, fn):
        nonlocal wedged
        t0 = time.perf_counter()
        try:
            if wedged:
                raise RuntimeError("skipped: an earlier RCCL call on this rank never returned")
            val = call_with_timeout(fn, args.rccl_timeout)
            steps[name] = {"ok": True, "s": time.perf_counter() - t0, **(val or {})}
        except BaseException as e:                            # noqa: BLE001
            wedged = wedged or isinstance(e, TimeoutError)
            steps[name] = {"ok": False, "s": time.perf_counter() - t0, "error": f"{type(e).__name__}: {e}"[:300]}

    steps["communicator"] = {"ok": use_rccl or args.backend == "gloo", "backend": "rccl" if use_rccl else ("gloo" if args.backend == "gloo" else None),
                             "error": data_note, "ranks": int(dist.get_world_size(data_group)) if use_rccl else int(dist.get_world_size()) if world > 1 else 1}

    def set_device():
        if torch.cuda.is_available():
            torch.cuda.set_device(dev_index)

    def p2p():
        set_device()
        send = torch.full((1,), rank + 7 * (world == 1), dtype=torch.uint8, device=pdev)
        recv = [torch.full((1,), 255, dtype=torch.uint8, device=pdev) for _ in range(world)]
        peers = [p for p in range(world) if p != rank] or [rank]      # (a world of one: the grouped send / recv with ITSELF)
        ops = []
        for peer in peers:
            ops.append(dist.P2POp(dist.isend, send, peer, group=data_group))
            ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group=data_group))
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        if use_rccl:
            torch.cuda.synchronize(dev)
        got = [int(recv[p].item()) for p in peers]
        if got != ([7] if world == 1 else peers):
            raise RuntimeError(f"wrong bytes from the peers: {got}")
        return {"peers": len(got), "self": world == 1}

    gst: dict = {}

    def gather():
        set_device()
        full = gather_codes(mine_np, N, dist, device=dev if use_rccl else None, code_dtype="compact", group=data_group, stats=gst,
                            self_transfer=world == 1)
        if rank == 0:
            if full.shape != (N, M) or not np.array_equal(full, fake_code_rows(0, N, M)):
                raise RuntimeError("the gathered matrix is not the database's codes")
        return {k: gst.get(k) for k in ("ranks", "wire_dtype", "bytes_sent", "bytes_received", "transport", "buffers")} | {"transfer_s": gst.get("seconds")}

    def native():
        from qinco_amd.comm import RcclComm, gather_codes_native
        torch.cuda.set_device(dev_index)
        comm = RcclComm.from_process_group()                  # the id travels on the gloo control plane
        seen = comm.count()
        counts = [shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0] for r in range(world)]
        out = gather_codes_native(torch.from_numpy(mine_np).to(dev), counts, rank, root=0, comm=comm)
        ok = True
        if rank == 0:
            ok = bool(np.array_equal(out.cpu().numpy(), fake_code_rows(0, N, M)))
        lib_path = comm.library_path()                        # (dladdr of the ncclSend qinco_gather_codes called)
        made_by = os.path.realpath(comm.rccl._name)
        comm.close()
        if not ok:
            raise RuntimeError("qinco_gather_codes: the gathered matrix is not the database's codes")
        if os.path.realpath(lib_path) != made_by:
            raise RuntimeError(f"qinco_gather_codes called {lib_path} but the communicator was made by {made_by}")
        return {"nccl_comm_count": seen, "rccl_library": lib_path}

    if (world > 1 and (use_rccl or args.backend == "gloo")) or (world == 1 and use_rccl):
        step("p2p_1_byte_with_every_peer", p2p)
        step("gather_codes", gather)
        if use_rccl:
            step("native_qinco_gather_codes", native)
    me = {"rank": rank, "gpu": dev_index, "pci_bus_id": gpu_pci_address(dev_index), "numa_node": (affinity or {}).get("numa_node"),
          "rows": end - start, "steps": steps}
    allr = [me]
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, me)                      # gloo control plane
    if rank == 0:
        names = [k for k in steps]
        out = {"metric": "dry-rccl: communicator + 1-byte grouped send/recv per peer + gather_codes of fake rows", "n_gpus": world,
               "backend": args.backend, "rows": N, "codes_per_row": M,
               "all_ok": all(r["steps"][k]["ok"] for r in allr for k in r["steps"]),
               "steps_ok_on_all_ranks": {k: all(r["steps"].get(k, {}).get("ok", False) for r in allr) for k in names},
               "rccl_ranks_seen": steps.get("gather_codes", {}).get("ranks"),
               "nccl_comm_count": steps.get("native_qinco_gather_codes", {}).get("nccl_comm_count"),
               "rccl_library": steps.get("native_qinco_gather_codes", {}).get("rccl_library"),
               "torch_nccl_version": list(torch.cuda.nccl.version()) if use_rccl else None,
               "per_rank": allr}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if wedged:
        sys.stderr.flush()
        os._exit(0)
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()
    return 0


def dry_rccl_on_gloo_without_gpu(torch, dist, args, rank, world, json_fd):
    """`--dry-rccl --backend gloo` on a machine without a GPU: the same steps over host buffers (the CPU test tier runs them at 2, 3
    and 8 ranks)."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dry_rccl(torch, dist, args, rank, world, torch.device("cpu"), 0, None, None, None, False, json_fd)


def synth_batch_device(torch, cfg, mean_t, std, n, seed, dev):
    """x = mean + std * N(0, I) (the S0 inputs of SURVEY.md 8d), drawn on the device from a seeded generator."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    z = torch.randn((n, cfg.D), generator=g, device=dev, dtype=torch.float32)
    return z * std + mean_t


def executed_flops(cfg, rows, groups, fold=True, fold2=True):
    """The library's own count (csrc/qinco_hip.hip mlp_flops_executed, reported by qinco_profile_read2), restated for the tests:
    FLOPs the matrix pipe executes for `rows` MLP rows in `groups` (vector, beam) groups.  Un-folded: the algorithmic count.  FOLD:
    the concat Linear and in_proj leave the per-row work (a table per codeword + U = W_x xhat once per group); FOLD2: the first
    up-projection too (Q = W_up[0] U per group).  The 16-row tile form (De > 384) has FOLD only."""
    L = max(cfg.L, 1)
    if not fold:
        return rows * (2.0 * (cfg.De + cfg.D) * cfg.De + 4.0 * L * cfg.De * cfg.dh + (4.0 * cfg.D * cfg.De if cfg.De != cfg.D else 0.0))
    per_row = 4.0 * L * cfg.De * cfg.dh + (2.0 * cfg.De * cfg.D if cfg.De != cfg.D else 0.0)
    per_group = 2.0 * cfg.D * cfg.De
    if fold2:
        per_row -= 2.0 * cfg.De * cfg.dh
        per_group += 2.0 * cfg.De * cfg.dh
    return rows * per_row + groups * per_group


def roofline_dict(prof, dt=None, kernel="qinco::mlp_kernel (+ its xproj pre-GEMM)"):
    """prof = QincoEngine.profile_read(): event time, launches, algorithmic and executed FLOPs of the fused-MLP launches."""
    launches = max(prof["mlp_launches"], 1)
    avg_ms = prof["mlp_ms"] / launches
    sec = prof["mlp_ms"] * 1e-3
    alg = prof["mlp_flops"] / sec / 1e12 if sec > 0 else 0.0
    exe = prof["mlp_flops_executed"] / sec / 1e12 if sec > 0 else 0.0
    d = {"bound": "mfma", "kernel": kernel, "achieved": exe, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": exe / PEAK_FP32_MFMA_TFLOPS, "frac_algorithmic": alg / PEAK_FP32_MFMA_TFLOPS, "algorithmic_tflops": alg,
         "executed_over_algorithmic_flops": (prof["mlp_flops_executed"] / prof["mlp_flops"]) if prof["mlp_flops"] else None,
         "avg_launch_ms": avg_ms, "launches": prof["mlp_launches"], "flops_per_launch": prof["mlp_flops"] / launches,
         "executed_flops_per_launch": prof["mlp_flops_executed"] / launches}
    if dt:
        d["mlp_share_of_step_time"] = sec / dt
    return d


# ------------------------------------------------------------------------------------------------------------------
# extra legs (N = 1, after the timed region)
# ------------------------------------------------------------------------------------------------------------------
def leg_workload(torch, dev, name, steps, batch, oracle_sample=0, decode_sizes=(), split_compare=True):
    """One of BASELINE.json's other configurations at its bench batch: K distinct resident batches, timed like the headline,
    with the fused-MLP roofline.  oracle_sample > 0 (C1): that many vectors are also encoded by the oracle on the host --
    the count of identical code rows (the north_star's "bit-exact greedy codes") and the oracle's rate on the sample."""
    from qinco_amd import QincoEngine, synth_state_dict
    from qinco_amd.config import BASELINE_CONFIGS
    cfg = BASELINE_CONFIGS[name]
    sd = synth_state_dict(cfg, 1236)
    eng = QincoEngine(cfg, sd, max_batch=batch)
    mean_t = torch.from_numpy(np.asarray(sd["data_mean"])).to(dev)
    std = float(sd["data_std"])
    xs = [synth_batch_device(torch, cfg, mean_t, std, batch, 7_000_003 * s + 11, dev) for s in range(steps + 1)]
    cdt = np.int32 if cfg.ivf else np.uint8        # (an IVF id does not fit a byte)
    eng.encode(xs[0], code_dtype=cdt)
    torch.cuda.synchronize(dev)
    eng.profile_enable(True)
    eng.profile_read()
    t0 = time.perf_counter()
    codes = [eng.encode(xs[1 + s], code_dtype=cdt) for s in range(steps)]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    out = {"workload": WORKLOAD_NAMES[name], "value": steps * batch / dt, "unit": "vectors/s", "steps": steps,
           "vectors_per_step": batch, "ms_per_step": dt / steps * 1e3, "A": cfg.A, "B": cfg.B, "M": cfg.M, "D": cfg.D,
           "gflop_per_vector": eng.flops_per_vector("encode") / 1e9,
           "roofline": roofline_dict(prof, dt, kernel="qinco::mlp_kernel (+ xproj)" if cfg.De <= 384 else "qinco::mlp16_kernel")}
    if decode_sizes:      # the reference's default host batch (qinco_cfg.yaml:38): one encode call per 1024 distinct vectors
        small = [xs[1][i:i + 1024] for i in range(0, batch, 1024)]
        eng.encode(small[0], code_dtype=cdt)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for xb in small:
            eng.encode(xb, code_dtype=cdt)
        torch.cuda.synchronize(dev)
        out["batch_1024"] = {"value": batch / (time.perf_counter() - t1), "unit": "vectors/s", "calls": len(small)}
    if decode_sizes:      # decode of these codes at the reference's call sizes (leg_decode_calls)
        codes_all = torch.cat(codes)
        for rows in decode_sizes:
            out[f"decode_batch_{rows}"] = leg_decode_calls(torch, dev, eng, cfg, codes_all, rows)
        del codes_all
    if split_compare and not cfg.ivf:
        # the opt-in split-fp16 form on the same timed batches (NOT part of `value`): rate, and the code rows it changes
        try:
            eng2 = QincoEngine(cfg, sd, max_batch=batch, split_f16=True)
            eng2.encode(xs[0], code_dtype=cdt)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            codes2 = [eng2.encode(xs[1 + s], code_dtype=cdt) for s in range(steps)]
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - t2
            differ = int(sum(int((a != b).any(dim=1).sum().item()) for a, b in zip(codes, codes2)))
            out["split_f16"] = {"value": steps * batch / dt2, "unit": "vectors/s", "speedup_vs_f32_path": (steps * batch / dt2) / out["value"],
                                "rows_differing_from_f32_path": differ, "rows": steps * batch,
                                "note": "opt-in QincoEngine(split_f16=True); fixture counts: parity.*.split_f16_codes_equal_to_reference"}
            eng2.close()
            del codes2
        except Exception as e:                                # noqa: BLE001 -- no split instance for the shape
            out["split_f16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if cfg.ivf:
        st = eng.ivf_last_stats()
        out["ivf"] = {"ivf_K": cfg.ivf_K, "exact_candidates_per_vector": st["candidates"] / batch, "fell_back_to_fp32_table": st["fell_back"],
                      "note": "coarse assignment: fp16 matrix-core filter (1/8 sample, then all centroids) + exact fp32 decision (DESIGN.md 3.4)"}
    if oracle_sample:
        from oracle.qinco_oracle import OracleQINCo
        threads = cpu_threads(torch)
        oracle = OracleQINCo.from_config(cfg, sd, backend="torch")
        x = xs[1][:oracle_sample].cpu().numpy()
        got = codes[0][:oracle_sample].cpu().numpy().astype(np.int64)
        oracle(x[:16], step="encode")
        t1 = time.perf_counter()
        # (the reference's protocol: batches of cfg.batch = 1024, qinco_tasks.py:99-125; a 256-vector sample is one call)
        want = np.concatenate([oracle(x[i:i + 1024], step="encode").T for i in range(0, len(x), 1024)])
        dt_o = time.perf_counter() - t1
        same = int((got == want).all(axis=1).sum())
        out["greedy_rows_equal_to_oracle"] = f"{same}/{oracle_sample}"
        if same < oracle_sample:      # every differing row must sit on a rounding-level tie of the oracle's own selection (tests/conftest.py)
            try:
                sys.path.insert(0, str(ROOT / "tests"))
                from conftest import assert_only_near_ties
                out["differing_rows_on_oracle_ties_below_2e-5"] = int(assert_only_near_ties(oracle, x, got, want, 2e-5, "c1 leg"))
            except AssertionError as e:
                out["differing_rows_on_oracle_ties_below_2e-5"] = f"NO: {e}"[:300]
        out["cpu_baseline"] = {"value": oracle_sample / dt_o, "unit": "vectors/s", "cores": threads, "kind": "port",
                               "gflops": oracle_sample / dt_o * cfg.encode_flops_per_vector() / 1e9,
                               "sample": f"the same {oracle_sample} vectors in oracle calls of <= 1024 ({dt_o:.1f} s, {threads} ATen threads; "
                                         f"--c1-cpu-vectors 10000 = BASELINE.md section 3's protocol)"}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    eng.close()
    del xs, codes
    torch.cuda.empty_cache()
    return out


def leg_encode_db_bvecs(torch, dev, n_db, batch, workload="C2"):
    """`task=encode` end to end (search_tasks.py:85-137): BigANN-style uint8 .bvecs on disk -> memmap (strided rows) ->
    encode_database(QINCoHIP) in host batches (qinco_encode_host: H2D of the bytes, uint8 -> fp32 and normalisation on the
    GPU, D2H of the codes) -> `<out>.npz` + part file.  Model: the C2 network with BigANN-magnitude normalisation constants
    (qinco_amd.synth.apply_regime).  Reported beside the HBM-resident rate of the same model on the same bytes."""
    from qinco_amd import apply_regime, regime_vectors, synth_state_dict
    from qinco_amd.config import BASELINE_CONFIGS
    from qinco_amd.encode_db import EncodedDBIterator, encode_database, get_data_memmap
    from qinco_amd.model import QINCoHIP
    cfg = BASELINE_CONFIGS[workload]
    sd = apply_regime(cfg, synth_state_dict(cfg, 1236), "bigann", 1236)
    model = QINCoHIP(cfg, sd, max_batch=batch)
    block = regime_vectors(cfg, sd, 65536, "bigann", seed=99)           # uint8 (65536, D); the file repeats it with a roll
    with tempfile.TemporaryDirectory(prefix="qinco_bench_") as tmp:
        path = os.path.join(tmp, "db.bvecs")
        rec = np.empty((len(block), cfg.D + 4), np.uint8)
        rec[:, :4] = np.frombuffer(np.int32(cfg.D).tobytes(), np.uint8)
        t0 = time.perf_counter()
        with open(path, "wb") as f:
            for i in range(0, n_db, len(block)):
                rec[:, 4:] = np.roll(block, i // len(block), axis=1)   # distinct rows per block
                f.write(rec[: min(len(block), n_db - i)].tobytes())
        t_write = time.perf_counter() - t0
        db = get_data_memmap(path)
        assert db.shape == (n_db, cfg.D) and db.dtype == np.uint8
        model(np.ascontiguousarray(db[:batch]), step="encode")        # warm-up (page cache, staging buffers)
        out_path = os.path.join(tmp, "enc", "db.npz")
        t0 = time.perf_counter()
        codes = encode_database(model, db, out_path, K=cfg.K, M=cfg.M, D=cfg.D, batch=4 * batch, code_dtype="compact")   # (as a 10^9-vector job keeps them: bytes)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()          # the reference's writer on the same codes: one core, at the end of the job
        np.savez_compressed(os.path.join(tmp, "ref_way.npz"), codes=codes)
        dt_ref_writer = time.perf_counter() - t0
        it = EncodedDBIterator(out_path, K=cfg.K, M=cfg.M, D=cfg.D)
        part_bytes = os.path.getsize(out_path[:-4] + ".part_0.npz")
        assert it.n_parts == 1 and codes.shape == (n_db, cfg.M)
        # the same bytes resident in HBM (device-pointer path)
        n_res = min(n_db, 8 * batch)
        xd = torch.from_numpy(np.ascontiguousarray(db[:n_res])).to(dev)
        model.engine.encode(xd[:batch], code_dtype=np.uint8)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        cres = model.engine.encode(xd, code_dtype=np.uint8)
        torch.cuda.synchronize(dev)
        dt_res = time.perf_counter() - t0
        same = bool(np.array_equal(cres.cpu().numpy().astype(np.int64), codes[:n_res].astype(np.int64)))
    model.engine.close()
    torch.cuda.empty_cache()
    return {"workload": workload, "value": n_db / dt, "unit": "vectors/s", "vectors": n_db, "seconds": dt, "host_batch": 4 * batch,
            "part_file_writer": "parallel deflate while encoding (qinco_amd.encode_db.PartFileWriter, 8 threads); "
                                f"np.savez_compressed of the same codes at the end would add {dt_ref_writer:.2f} s",
            "np_savez_compressed_s": dt_ref_writer,
            "resident_value": n_res / dt_res, "host_over_resident": (n_db / dt) / (n_res / dt_res),
            "codes_equal_to_resident_path": same, "input": "uint8 .bvecs",
            "file_bytes": n_db * (cfg.D + 4), "file_write_s": t_write, "part_file_bytes": part_bytes,
            "path": "np.memmap (strided uint8 rows) -> encode_database -> QINCoHIP.__call__ -> qinco_encode_host -> part file in the "
                    "reference's format (deflated int64 codes, search_tasks.py:125-131)"}


def leg_decode_calls(torch, dev, eng, cfg, codes_all, rows, min_calls=8):
    """Decode in calls of `rows` code rows each (distinct rows per call, resident in HBM), the way the reference's loops call
    model(codes, step="decode").  Below ~one 128-row workgroup per CU a call runs on the small-launch form of the fused MLP
    (csrc/mlp_small_kernel.hpp): every step of a row tile in ONE launch.  Its bound stays the fp32 matrix pipe down to one 16-row
    tile per CU (4096 rows); below that the launch cannot fill the chip -- 1024 rows are 64 workgroups on 256 CUs, which all
    stream the step's weights from L2 / Infinity Cache: weight_stream_gb_per_s says how fast."""
    n_all = int(codes_all.shape[0])
    calls = max(min_calls, min(64, n_all // rows))
    chunks = [codes_all[(i * rows) % max(n_all - rows + 1, 1):][:rows] for i in range(calls)]
    eng.decode(chunks[0], check=False)
    torch.cuda.synchronize(dev)
    eng.profile_enable(True)
    eng.profile_read()
    t0 = time.perf_counter()
    for c in chunks:
        eng.decode(c, check=False)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    eng.check_codes()
    rf = roofline_dict(prof, dt, kernel="qinco::mlp_small_kernel (all steps of a row tile in one launch)" if rows < 24576
                       else "qinco::mlp_kernel per step (+ xproj)")
    tiles = (rows + 15) // 16
    weights_mb = (cfg.M_total - 1) * (cfg.mlp_flops_per_row() / 2.0) * 4 / 1e6      # fp32 weights of every step's MLP
    wgs = min(tiles, 256) if tiles <= 256 else None
    out = {"value": calls * rows / dt, "unit": "vectors/s", "rows_per_call": rows, "calls": calls, "us_per_call": dt / calls * 1e6,
           "us_per_vector": dt / (calls * rows) * 1e6, "roofline": rf}
    if wgs is not None and rf["avg_launch_ms"] > 0:
        out["bound_note"] = (f"{tiles} row tiles of 16 on 256 CUs: at most {tiles / 256:.2f} of the matrix pipes have rows to work on; every one "
                             f"of the {wgs} workgroups streams all {weights_mb:.0f} MB of MLP weights through L2")
        out["weight_stream_gb_per_s"] = wgs * weights_mb * 1e6 / (rf["avg_launch_ms"] * 1e-3) / 1e9
        out["frac_of_occupied_pipes"] = rf["frac"] / min(1.0, tiles / 256)
    return out


def leg_small_db_search(torch, dev, N=1_000_000, Q=4096, D=128, k=100):
    """f3 (run_search_full_direct_small_db, search_tasks.py:551-603) at bigann1M's shape: top-100 of 10^6 reconstructions for 4096
    queries through the filtered form (no distance table in HBM; csrc/knn_kernel.hpp) and the table form; ids and distance bits must
    be equal.  FLOPs = 2 D per pair (the distance table the reference computes with approx_pairwise_distance)."""
    from qinco_amd.search import KnnSearcher
    g = torch.Generator(device=dev).manual_seed(0)
    db = torch.randn(N, D, device=dev, generator=g)
    q = torch.randn(Q, D, device=dev, generator=g)
    out = {"workload": f"top-{k} of {N} x {Q} queries, D={D}", "unit": "queries/s"}
    got = {}
    for name, filtered in (("filtered", True), ("table", False)):
        knn = KnnSearcher(D, filtered=filtered)
        knn.search(db, q, k=k)
        torch.cuda.synchronize(dev)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            got[name] = knn.search(db, q, k=k, return_dist=True)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        out[name] = {"value": Q / best, "ms": best * 1e3, "tflops": 2.0 * D * N * Q / best / 1e12,
                     "frac_fp32_mfma": 2.0 * D * N * Q / best / 1e12 / PEAK_FP32_MFMA_TFLOPS, **knn.last_stats()}
        knn.close()
    out["value"] = out["filtered"]["value"]
    out["forms_equal_bit_for_bit"] = bool(torch.equal(got["filtered"][0], got["table"][0])
                                          and torch.equal(got["filtered"][1].view(torch.int32), got["table"][1].view(torch.int32)))
    return out


def parity_counts(torch, dev):
    """Rows of the committed reference fixtures this build reproduces, on this GPU: tests/golden/<case>.npz holds the inputs and the
    codes / reconstructions the IMPORTED REFERENCE produced for them (tests/golden/make_golden.py ran /root/reference in the build
    container; nothing of it is needed here).  Codes: rows equal / rows; decode: max |xhat - reference| / max |reference| of the
    reference's own codes."""
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    from cases import case_model
    from qinco_amd import QincoEngine
    out = {}
    for key, name in (("C1", "C1_qinco1_8x8"), ("C2", "C2_qinco2L_8x8_b8"), ("C2_beam1", "C2_qinco2L_8x8_b1"),
                      ("C3", "C3_qinco2L_16x8_b8"), ("C4", "C4_qinco2L_d768_b8")):
        try:
            g = np.load(ROOT / "tests" / "golden" / f"{name}.npz")
            cfg, sd = case_model(name)
            eng = QincoEngine(cfg, sd, max_batch=256)
            want = g["codes_wrapper"] if "codes_wrapper" in g.files else g["codes_base"]
            got = eng.encode(torch.from_numpy(g["x"]).to(dev), code_dtype=np.int64).cpu().numpy()
            same = int((got == want).all(axis=1).sum())
            dec = eng.decode(torch.from_numpy(want).to(dev)).cpu().numpy()
            ref = g["decoded"]
            out[key] = {"codes_equal_to_reference": f"{same}/{len(want)}",
                        "decode_max_rel_err": float(np.abs(dec - ref).max() / np.abs(ref).max()), "fixture": f"tests/golden/{name}.npz"}
            eng.close()
            try:                                              # the opt-in split-fp16 form against the same reference codes
                eng2 = QincoEngine(cfg, sd, max_batch=256, split_f16=True)
                got2 = eng2.encode(torch.from_numpy(g["x"]).to(dev), code_dtype=np.int64).cpu().numpy()
                out[key]["split_f16_codes_equal_to_reference"] = f"{int((got2 == want).all(axis=1).sum())}/{len(want)}"
                eng2.close()
            except Exception as e:                            # noqa: BLE001
                out[key]["split_f16_codes_equal_to_reference"] = f"unavailable: {type(e).__name__}"
        except Exception as e:                                # noqa: BLE001 -- the headline must not depend on a fixture
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4", "S", "IVF_S"],
                    help="C2 = the driver's configuration (BASELINE.json configs[1]); S / IVF_S: the reference's billion-scale family, e.g. "
                         "--scaling strong --db 80000000 --workload S as the stand-in for 10 M vectors per GPU of the 1 B-vector job")
    ap.add_argument("--batch", type=int, default=16384, help="vectors per step per GPU")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: per-GPU work fixed (K batches per rank); strong: one database split over the ranks")
    ap.add_argument("--db", type=int, default=0, help="strong scaling: database size (default steps x batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the decode / mse / beam1 / batch_1024 / split legs")
    ap.add_argument("--no-legs", action="store_true", help="skip the c1 / c3 / c4 / encode_db_bvecs legs")
    ap.add_argument("--bvecs-vectors", type=int, default=1_000_000, help="size of the encode_db_bvecs leg's file")
    ap.add_argument("--no-affinity", action="store_true", help="do not bind the rank to its GPU's NUMA node")
    ap.add_argument("--split-f16", action="store_true",
                    help="NOT the driver's configuration: run the whole bench (any --gpus) on the opt-in split-fp16 form; the line says so in dtype / config")
    ap.add_argument("--rccl-timeout", type=float, default=180.0,
                    help="seconds an RCCL call (communicator creation, the gather) may take before the run falls back to part files")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="payload gather: nccl = RCCL over xGMI (one GPU per rank). gloo is a test hook: ranks may share a GPU.")
    ap.add_argument("--dry-rccl", action="store_true",
                    help="no model, no encode: build the communicator(s), one 1-byte grouped send / recv with every peer, then the product's "
                         "gather_codes (and the C-ABI qinco_gather_codes) on --dry-rows fake code rows, each step under --rccl-timeout; "
                         "prints one JSON line saying which steps worked on which rank (first contact with a multi-GPU node in seconds)")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not measure roofline.traffic with rocprofv3 --pmc child passes (it then "
                                                          "falls back to the committed pass's bytes per row, and says so)")
    ap.add_argument("--no-rccl-check", action="store_true", help="N = 1: skip the world-of-one RCCL check (`rccl_world1` on the line)")
    ap.add_argument("--dry-rows", type=int, default=1_000_000, help="--dry-rccl: rows of the fake database")
    ap.add_argument("--c1-cpu-vectors", type=int, default=256,
                    help="vectors of the c1 leg's CPU sample (BASELINE.md 3 / SURVEY 8d quote C1's CPU protocol at 10 000: minutes of host time)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries exactly ONE line (rank 0's JSON): native libraries print there too (gloo's "[Gloo] Rank 0 is connected
    # to 7 peer ranks" comes from C++ std::cout), so file descriptor 1 points at stderr until that line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N` bare)")
    ndev = torch.cuda.device_count()
    cpu_dry = ndev == 0 and args.dry_rccl and args.backend == "gloo"    # (the communication steps alone, host buffers: runs anywhere)
    if cpu_dry:
        return dry_rccl_on_gloo_without_gpu(torch, dist, args, rank, world, json_fd)
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (qinco_amd has no CPU path)")
    if args.backend == "nccl" and world > ndev and not os.environ.get("QINCO_BENCH_NCCL_SHARED_GPU"):   # (test hook: let RCCL refuse it)
        raise SystemExit(f"{world} ranks but only {ndev} GPU(s): RCCL needs one GPU per rank (--backend gloo lets ranks share one)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    affinity = None
    if not args.no_affinity:
        from qinco_amd.affinity import bind_to_gpu_numa
        affinity = bind_to_gpu_numa(dev_index)

    # control plane on gloo (always works on one node), payload on RCCL when asked for and available
    data_group, data_note = None, None
    wedged = False            # an RCCL call timed out: its thread is still stuck, so the process must leave through os._exit
    if world > 1 or args.dry_rccl:
        # (--dry-rccl at --gpus 1: a job of ONE rank still builds both groups -- gloo control plane, RCCL payload group -- so that
        # a 1-GPU box executes every communication line of the multi-GPU bench on the real library, rank 0 in both roles)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
        if args.backend == "nccl":
            ok = 1
            try:
                data_group = dist.new_group(backend="nccl", device_id=dev)

                def probe_rccl():                             # forces communicator creation now, outside the timed region
                    torch.cuda.set_device(dev_index)
                    probe = torch.ones(1, device=dev)
                    dist.all_reduce(probe, group=data_group)
                    # ... and the gather's own point-to-point pattern once with one byte per rank: whatever the backend builds lazily
                    # for a (rank, 0) pair is built here, not inside the timed region
                    # (one batch per rank, as gather_codes issues them; a world of one talks to itself)
                    box = torch.zeros(world, dtype=torch.uint8, device=dev)
                    ops = [dist.P2POp(dist.irecv, box[r:r + 1], r, group=data_group) for r in range(1 if world > 1 else 0, world)] if rank == 0 else []
                    if rank != 0 or world == 1:
                        ops.append(dist.P2POp(dist.isend, torch.full((1,), rank, dtype=torch.uint8, device=dev), 0, group=data_group))
                    for q in dist.batch_isend_irecv(ops):
                        q.wait()
                    torch.cuda.synchronize(dev)
                call_with_timeout(probe_rccl, args.rccl_timeout)
            except BaseException as e:                        # noqa: BLE001 -- any failure or a hang: part files instead
                wedged = wedged or isinstance(e, TimeoutError)
                ok, data_note = 0, f"RCCL unavailable on rank {rank}: {type(e).__name__}: {e}"[:300]
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # gloo: everybody agrees on the payload path
            if int(flag.item()) == 0:
                data_group = None
                data_note = data_note or "RCCL unavailable on another rank"

    if args.dry_rccl:
        return dry_rccl(torch, dist, args, rank, world, dev, dev_index, data_group, data_note, affinity, wedged, json_fd)

    from qinco_amd import QincoEngine, synth_state_dict
    from qinco_amd.config import BASELINE_CONFIGS
    from qinco_amd.encode_db import compact_code_dtype, encode_shard, gather_codes, shard_bounds
    from qinco_amd.evaluate import sqerr_sum
    from qinco_amd.model import QINCoHIP

    cfg = BASELINE_CONFIGS[args.workload]
    sd = synth_state_dict(cfg, 1236)
    # the reference-shaped model object (model(x, step="encode") -> (M, n) int64 on x's device): what encode_database drives
    model = QINCoHIP(cfg, sd, max_batch=args.batch, device=dev_index, split_f16=args.split_f16)
    eng = model.engine
    K, W = args.steps, args.warmup
    strong = args.scaling == "strong"

    # The synthetic database.  weak: rank r owns batches r K .. r K + K - 1 of one seeded stream -- the reference's contiguous range
    # (search_tasks.py:103-104) of a database of world x K x batch rows.  strong: the database is batches 0 .. ceil(db / batch) - 1
    # of that stream and rank r owns its range of it.  The rank's whole shard is ONE tensor resident in HBM before the clock starts.
    mean_t = torch.from_numpy(np.asarray(sd["data_mean"])).to(dev)
    std = float(sd["data_std"])

    def stream_batch(i):
        return synth_batch_device(torch, cfg, mean_t, std, args.batch, 1_000_003 * i + 42, dev)

    warm = [stream_batch(10_000 + rank * max(W, 1) + s) for s in range(W)]
    if strong:
        db_size = args.db or K * args.batch
        start, end = shard_bounds(db_size, world, rank)
        b0, b1 = start // args.batch, (end + args.batch - 1) // args.batch
        shard = torch.cat([stream_batch(i) for i in range(b0, b1)])[start - b0 * args.batch: end - b0 * args.batch].contiguous() \
            if end > start else torch.empty((0, cfg.D), device=dev)
    else:
        db_size = K * args.batch * world
        start, end = shard_bounds(db_size, world, rank)
        shard = torch.cat([stream_batch(rank * K + s) for s in range(K)]) if K else torch.empty((0, cfg.D), device=dev)
    my_vecs = end - start
    assert len(shard) == my_vecs
    batches = [shard[i:i + args.batch] for i in range(0, len(shard), args.batch)]     # (views: the extras' legs re-use them)

    def device_sync():        # (a wedged RCCL kernel would make a device-wide synchronize wait forever: the compute stream only)
        if wedged:
            torch.cuda.current_stream(dev).synchronize()
        else:
            torch.cuda.synchronize(dev)

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()                                    # gloo
        device_sync()

    for xb in warm:
        model(xb, step="encode")
    barrier()
    eng.profile_enable(True)
    eng.profile_read()
    barrier()
    t0 = time.perf_counter()
    # ---- the timed region IS the product's database encode (qinco_amd.encode_db, search_tasks.py:85-137 minus the files):
    # encode_shard over the rank's range in passes of `batch` rows (the model call asynchronous on this thread's stream, the codes
    # of the previous pass narrowed to bytes and copied to the host on a helper thread), then gather_codes: every rank > 0 sends its
    # (n_r, M) byte shard once, rank 0 receives each straight into its rows of ONE (N, M) matrix.
    wire_np = compact_code_dtype(cfg.K, cfg.M_total, cfg.M)          # uint8; int32 with an IVF column (an IVF id does not fit a byte)
    mine_np = encode_shard(model, shard, 0, my_vecs, batch=args.batch, code_dtype="compact", K=cfg.K, M=cfg.M)
    if mine_np.size == 0:
        mine_np = np.zeros((0, cfg.M_total), wire_np)
    assert mine_np.shape == (my_vecs, cfg.M_total) and mine_np.dtype == wire_np
    t_enc = t_gather = 0.0
    gather_how = None
    gstats: dict = {}
    gathered = None
    if world > 1:
        torch.cuda.synchronize(dev)
        t_enc = time.perf_counter() - t0
        try:
            hook = os.environ.get("QINCO_BENCH_FORCE_GATHER_ERROR")   # test hook (tests/test_multi_gpu.py): the fail-soft paths
            if hook == "hang":
                call_with_timeout(lambda: time.sleep(1e6), args.rccl_timeout)
            elif hook:
                raise RuntimeError("forced by QINCO_BENCH_FORCE_GATHER_ERROR")
            if data_group is not None:
                def gather_rccl():
                    torch.cuda.set_device(dev_index)
                    return gather_codes(mine_np, db_size, dist, device=dev, code_dtype="compact", group=data_group, stats=gstats)
                gathered = call_with_timeout(gather_rccl, args.rccl_timeout)
                gather_how = "rccl"
            elif args.backend == "gloo":
                gathered = gather_codes(mine_np, db_size, dist, code_dtype="compact", stats=gstats)
                gather_how = "gloo (test hook: host buffers)"
            else:
                raise RuntimeError(data_note or "no RCCL group")
            if rank == 0:
                assert gathered.shape == (db_size, cfg.M_total) and gathered.dtype == wire_np
        except BaseException as e:                            # noqa: BLE001 -- fail soft: the reference's own part files
            wedged = wedged or isinstance(e, TimeoutError)
            outdir = os.environ.get("QINCO_BENCH_PARTS", tempfile.gettempdir())
            np.savez_compressed(os.path.join(outdir, f"qinco_bench_codes.part_{rank}.npz"), codes=mine_np)
            gather_how = f"failed: {type(e).__name__}: {e}"[:300] + " -> part files"
            gathered = None
        t_gather = time.perf_counter() - t0 - t_enc
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)
    mine = torch.from_numpy(mine_np).to(dev)                  # (the extras decode / compare these codes on the device)

    per_rank = None
    if world > 1:   # timing and identity exchange on the control plane (gloo, host objects) -- after the clock has stopped
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        from qinco_amd.affinity import gpu_pci_address
        me = {"rank": rank, "gpu": dev_index, "pci_bus_id": gpu_pci_address(dev_index), "numa_node": (affinity or {}).get("numa_node"),
              "vectors": int(my_vecs), "encode_s": t_enc, "gather_s": t_gather,
              "gather_ok": not (gather_how or "").startswith("failed"), "gather": gather_how,
              "bytes_sent": gstats.get("bytes_sent"), "bytes_received": gstats.get("bytes_received"),
              "ranks_seen_by_payload_group": gstats.get("ranks"), "wire_dtype": gstats.get("wire_dtype"),
              "transport": gstats.get("transport"), "buffers": gstats.get("buffers"),
              # 64-bit sum of the shard's code bytes, weighted by position: rank 0 recomputes it from the gathered matrix
              "shard_checksum": shard_checksum(mine_np, start)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)
        if rank == 0 and gathered is not None:
            for r, info in enumerate(per_rank):
                s_r, e_r = shard_bounds(db_size, world, r)
                info["rows_on_rank0_equal_to_the_shard"] = shard_checksum(gathered[s_r:e_r], s_r) == info["shard_checksum"]

    if rank == 0:
        total_vecs = db_size
        value = total_vecs / dt if dt > 0 else 0.0
        mlp_row = cfg.mlp_flops_per_row()
        rf = roofline_dict(prof, dt, kernel=("qinco::mlp_kernel (+ its xproj pre-GEMM)" if not args.split_f16 else
                                             "qinco::mlp_split_kernel (+ xproj_split): runs on the fp16 pipe, quoted against the fp32-MFMA peak"))
        bpr, bpr_src = pmc_traffic_per_row() if args.workload == "C2" else (None, None)
        rows_per_launch = rf["flops_per_launch"] / mlp_row
        rf["traffic"] = bpr * rows_per_launch if bpr else None
        rf["traffic_unit"] = (f"bytes per launch, L2<->fabric: NOT measured in this process (counters cannot be read from inside it) -- "
                              f"bytes per MLP row of the separate rocprofv3 --pmc pass of this command (profiles/{bpr_src}) x this "
                              f"run's rows per launch") if bpr else None
        out = {
            "metric": ("encode vectors/sec (BigANN-shaped d=128 8x8, beam=%d)" % cfg.B) if args.workload in ("C1", "C2") else
                      ("encode vectors/sec (d=%d %dx8, beam=%d)" % (cfg.D, cfg.M, cfg.B)),
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3 if K else 0.0, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32" if not args.split_f16 else "f32 operands as fp16 hi+lo on the fp16 MFMA, fp32 accumulate (--split-f16)",
            "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.workload],
                       "D": cfg.D, "M": cfg.M, "K": cfg.K, "L": cfg.L, "de": cfg.De, "dh": cfg.dh, "A": cfg.A, "B": cfg.B,
                       "vectors_per_step_per_gpu": args.batch if not strong else None,
                       "parallelism": f"shard{world}", "distinct_vectors_encoded": total_vecs,
                       "inputs": "every step encodes a different seeded batch, all resident in HBM before the clock starts",
                       "weights": "seeded synthetic (RandomState 1236)",
                       "gflop_per_vector": eng.flops_per_vector("encode") / 1e9,
                       "kernel_instances": eng.describe(), "cpu_affinity": affinity},
            "roofline": rf,
        }
        if world > 1:
            out["multi_gpu"] = {
                "backend": args.backend, "control_plane": "gloo", "gather": gather_how,
                "path": "qinco_amd.encode_db.encode_shard + gather_codes (the product's encode_database path: one send per rank > 0, "
                        "irecv straight into rank 0's (N, M) matrix; search_tasks.py:95-134)",
                "gather_ok_on_all_ranks": all(t["gather_ok"] for t in per_rank),
                "gathered_rows_verified_on_rank0": (all(t.get("rows_on_rank0_equal_to_the_shard", False) for t in per_rank)
                                                    if gathered is not None else None),
                "rccl_note": data_note,
                "rccl_ranks_seen": gstats.get("ranks") if gather_how == "rccl" else None,
                "per_rank_vectors": [t["vectors"] for t in per_rank],
                "per_rank_encode_vectors_per_s": [t["vectors"] / t["encode_s"] if t["encode_s"] > 0 else 0.0 for t in per_rank],
                "per_rank_encode_s": [t["encode_s"] for t in per_rank],
                "per_rank_gather_s": [t["gather_s"] for t in per_rank],
                "gather_bytes_per_rank": int(max(t["vectors"] for t in per_rank) * cfg.M_total * wire_np.itemsize),
                "per_rank": per_rank,
                "note": "gather time of a rank includes waiting for the slowest rank's encode",
            }
        if world == 1 and not args.no_rccl_check:
            out["rccl_world1"] = rccl_world_of_one(args.rccl_timeout)
            out["rccl_world1_ok"] = bool(out["rccl_world1"].get("all_ok", False))
        if world == 1 and not args.no_extras and K > 0 and not strong and not cfg.ivf:
            extras(torch, dev, args, cfg, sd, eng, out, mine, batches, warm, value, sqerr_sum, QincoEngine)
        if world == 1 and not args.no_legs and not args.split_f16:
            out["parity"] = parity_counts(torch, dev)
            for key, fn in (("c1", lambda: leg_workload(torch, dev, "C1", 3, 16384, oracle_sample=args.c1_cpu_vectors, decode_sizes=(1024, 12288, 16384))),
                            ("c3", lambda: leg_workload(torch, dev, "C3", 2, 16384)),
                            ("c4", lambda: leg_workload(torch, dev, "C4", 2, 16384)),
                            ("qinco2_S", lambda: leg_workload(torch, dev, "S", 6, 16384, decode_sizes=(1024, 12288, 16384))),
                            ("ivf_qinco2_S", lambda: leg_workload(torch, dev, "IVF_S", 6, 16384)),
                            ("encode_db_bvecs", lambda: leg_encode_db_bvecs(torch, dev, args.bvecs_vectors, 16384)),
                            ("encode_db_bvecs_qinco2S", lambda: leg_encode_db_bvecs(torch, dev, args.bvecs_vectors, 16384, "S")),
                            ("small_db_search", lambda: leg_small_db_search(torch, dev))):
                try:        # a leg can never take the headline down with it
                    out[key] = fn()
                except Exception as e:                        # noqa: BLE001
                    out[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
        pmc_box, pmc_thread = {}, None
        if world == 1 and not args.no_pmc and K > 0:
            # roofline.traffic of this command, measured now: the two PMC passes run as child processes on the (idle) GPU while
            # this process times the CPU baseline on the host cores
            import threading
            pmc_thread = threading.Thread(target=lambda: pmc_box.update(pmc_traffic_live(args)), daemon=True)
            pmc_thread.start()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if pmc_thread is not None:
            pmc_thread.join(600)
            if pmc_box.get("ok"):
                rf["traffic"] = pmc_box["bytes_per_launch"]
                rf["traffic_unit"] = ("bytes per launch of the dominant kernel, L2<->fabric (Infinity-Cache hits included): MEASURED for this command by "
                                      "two rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; child processes of this run, "
                                      "untimed, same workload and batch, mean over the full-size launches)")
                rf["traffic_passes"] = pmc_box["passes"]
                # algorithmic bytes of one launch (SURVEY 8d / DESIGN 5): every (vector, beam) group's xhat row read once, every
                # vector's x once, the step's weights once (2 FLOP per weight per row -> bytes = 2 x FLOP per row), the B winners
                # (xhat row + history) written once
                groups, vecs = rows_per_launch / max(cfg.A, 1), rows_per_launch / max(cfg.A * cfg.B, 1)
                alg = groups * cfg.D * 4 + vecs * cfg.D * 4 + 2.0 * mlp_row + groups * (cfg.D * 4 + 4 * cfg.M_total)
                rf["algorithmic_bytes_per_launch"] = alg
                rf["traffic_over_algorithmic_bytes"] = pmc_box["bytes_per_launch"] / alg
                rf["traffic_GBps"] = pmc_box["bytes_per_launch"] / (pmc_box["passes"]["FETCH_SIZE"]["mean_duration_us"] * 1e-6) / 1e9
                rf["traffic_measure_seconds"] = pmc_box.get("seconds")
            else:
                rf["traffic_live_error"] = pmc_box.get("error", "the PMC passes did not finish")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if wedged:                # a helper thread is still inside RCCL: skip every destructor that could wait for it
        sys.stderr.flush()
        os._exit(0)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def extras(torch, dev, args, cfg, sd, eng, out, mine, batches, warm, value, sqerr_sum, QincoEngine):
    """decode / mse / batch_1024 / beam1 / split_f16 on the timed batches (N = 1)."""
    K = len(batches)
    mlp_row = cfg.mlp_flops_per_row()
    # ---- decode of the codes just produced + MSE of encode -> decode (qinco_tasks.py:87-148) ----
    codes_all = mine.reshape(-1, cfg.M_total)
    eng.decode(codes_all[:args.batch], check=False)
    torch.cuda.synchronize(dev)
    eng.profile_enable(True)
    eng.profile_read()
    t1 = time.perf_counter()
    dec = eng.decode(codes_all, check=False)
    torch.cuda.synchronize(dev)
    dt_dec = time.perf_counter() - t1
    prd = eng.profile_read()
    eng.profile_enable(False)
    eng.check_codes()
    out["decode"] = {"value": codes_all.shape[0] / dt_dec, "unit": "vectors/s", "vectors": int(codes_all.shape[0]),
                     "gflop_per_vector": eng.flops_per_vector("decode") / 1e9,
                     "roofline": roofline_dict(prd, dt_dec, kernel="qinco::mlp_kernel per step (+ xproj)")}
    # ---- decode at the reference's own call sizes: compute_MSE decodes cfg.batch = 1024 rows per call (qinco_tasks.py:112-125,
    # qinco_cfg.yaml:38), the search re-rank cfg.search.batch_size = 12 288 (search_tasks.py:475-486, qinco_cfg.yaml:137) ----
    for key, rows in (("decode_batch_1024", 1024), ("decode_batch_12288", 12288), ("decode_batch_16384", 16384)):
        out[key] = leg_decode_calls(torch, dev, eng, cfg, codes_all, rows)
    xs = torch.cat(batches)
    out["mse"] = {"value": sqerr_sum(xs, dec) / xs.shape[0], "vectors": int(xs.shape[0]),
                  "definition": "sum_i |x_i - decode(encode(x_i))|^2 / N (mse_scale 1; metrics.py:51-58)"}
    del dec, xs

    def timed_encode(xb_list):
        for xb in xb_list[:1]:
            eng.encode(xb, code_dtype=np.uint8)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for xb in xb_list:
            eng.encode(xb, code_dtype=np.uint8)
        torch.cuda.synchronize(dev)
        return sum(len(xb) for xb in xb_list) / (time.perf_counter() - t2)
    # ---- the reference's default host batch (qinco_cfg.yaml:38): 1024 vectors per call ----
    small = [batches[0][i:i + 1024] for i in range(0, min(args.batch, 16384), 1024)]
    out["batch_1024"] = {"value": timed_encode(small), "unit": "vectors/s", "calls": len(small),
                         "note": "same engine, one encode call per 1024 distinct vectors"}
    # ---- greedy search (beam = 1, BASELINE metric: beam in {1, 8}) ----
    if cfg.B > 1:
        eng.set_beam(B=1)
        out["beam1"] = {"value": timed_encode(batches[:min(K, 2)]), "unit": "vectors/s", "A": eng.A, "B": 1,
                        "gflop_per_vector": eng.flops_per_vector("encode") / 1e9}
        # ... at the reference's host batch: 1024 x A = 16 384 MLP rows per step, the small-launch form's encode-step kernel
        out["beam1_batch_1024"] = {"value": timed_encode(small), "unit": "vectors/s", "calls": len(small)}
        out["beam1_batch_1024"]["over_beam1"] = out["beam1_batch_1024"]["value"] / out["beam1"]["value"]
        eng.set_beam(B=cfg.B)
    # ---- the opt-in split-fp16 form of the FFN blocks (include/qinco_hip.h QINCO_CREATE_SPLIT_F16): NOT the headline --
    # `value` above is the fp32 path; this leg re-encodes the same timed batches and counts the code rows that change
    try:
        if args.split_f16:
            raise RuntimeError("the whole line is the split form (--split-f16); roofline.frac is quoted against the fp32-MFMA peak")
        eng2 = QincoEngine(cfg, sd, max_batch=args.batch, split_f16=True)
    except Exception as e:      # no split instance for this shape (or any other failure): the headline must not depend on it
        out["split_f16"] = {"error": f"{type(e).__name__}: {e}"}
        return
    try:
        eng2.encode(warm[0] if warm else batches[0], code_dtype=np.uint8)
        torch.cuda.synchronize(dev)
        eng2.profile_enable(True)
        eng2.profile_read()
        t3 = time.perf_counter()
        codes2 = torch.cat([eng2.encode(xb, code_dtype=np.uint8) for xb in batches])
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t3
        pr2 = eng2.profile_read()
        eng2.profile_enable(False)
        s_launches = max(pr2["mlp_launches"], 1)
        s_ms, s_fpl = pr2["mlp_ms"] / s_launches, pr2["mlp_flops"] / s_launches
        s_ach = s_fpl / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0
        differ = int((codes2 != codes_all).any(dim=1).sum().item())
        dec2 = eng2.decode(codes2, check=False)
        xs = torch.cat(batches)
        blocks = 4.0 * cfg.L * cfg.De * cfg.dh - 2.0 * cfg.De * cfg.dh          # per row, block 0's up-projection folded
        if cfg.De != cfg.D and (cfg.D // 32) % 2 == 0:
            blocks += 2.0 * cfg.D * cfg.De                                      # out_proj in the split form too
        blocks += (2.0 * cfg.D * cfg.De + 2.0 * cfg.De * cfg.dh) / (cfg.A or cfg.K)   # xproj, per group
        f16_tflops = 3.0 * blocks * (s_fpl / mlp_row) / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0
        n_all = int(codes_all.shape[0])
        out["split_f16"] = {
            "value": n_all / dt2, "unit": "vectors/s", "ms_per_step": dt2 / K * 1e3,
            "speedup_vs_f32_path": (n_all / dt2) / value if value > 0 else None,
            "rows_differing_from_f32_path": differ, "rows": n_all,
            "mse": sqerr_sum(xs, dec2) / xs.shape[0],
            "roofline": {"bound": "mfma", "kernel": "qinco::mlp_split_kernel (+ xproj)", "avg_launch_ms": s_ms,
                         "achieved_algorithmic_fp32_equivalent_tflops": s_ach,
                         "f16_mfma_tflops_executed": f16_tflops, "peak_f16_mfma_tflops": PEAK_F16_MFMA_TFLOPS,
                         "frac_of_f16_peak": f16_tflops / PEAK_F16_MFMA_TFLOPS},
            "arithmetic": "FFN blocks, out_proj, xproj: fp32 operands as fp16 hi + lo, 3 fp16 MFMAs per product, fp32 accumulate; tables, distances, selection fp32",
            "note": "opt-in (QincoEngine(split_f16=True) / qinco_create_ex); parity tests: tests/test_hip_parity.py::test_split_f16_*"}
        del dec2, xs, codes2
        eng2.close()
        torch.cuda.empty_cache()
    except Exception as e:
        out["split_f16"] = {"error": f"{type(e).__name__}: {e}"}


if __name__ == "__main__":
    main()
