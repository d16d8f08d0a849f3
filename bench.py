#!/usr/bin/env python
"""bench.py -- encode throughput of the MI355X QINCo2 engine on BASELINE.json's metric.

    python bench.py [--gpus N --steps K --warmup W] [--workload C2] [--batch 16384]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one pass of the hot path (model(x, step="encode")) over one batch of `--batch` synthetic fp32
vectors per GPU, inputs already resident in HBM.  Weak scaling: every rank encodes its own shard of the
database (contiguous range sharding like search_tasks.py:103-104, no data-path collective); the uint8 codes
of all timed steps are gathered to rank 0 over RCCL inside the timed region.  Rank 0 prints ONE JSON line.

metric/unit: encode vectors/s (BASELINE.json "metric"); workload C2 = qinco2-L 8x8, D=128, A=16, B=8
(BASELINE.json configs[1]) with seeded synthetic weights (no trained checkpoints offline).
roofline: the fused codeword-MLP kernel (99.9 % of the FLOPs), fp32 MFMA bound: achieved = algorithmic FLOPs
per launch (rows x R_mlp) / mean launch duration measured with HIP events on the launch stream.
cpu_baseline: the numpy oracle (oracle/qinco_oracle.py, shown equal to the imported reference in
tests/golden) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)


def pmc_traffic_per_row():
    """L2<->fabric bytes per MLP row from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE; profiles/r01_c2_traffic.json).  Counters cannot be read from inside this process, so the figure
    of the separate PMC run of this same command is scaled to this run's rows per launch."""
    try:
        with open(ROOT / "profiles" / "r01_c2_traffic.json") as f:
            return float(json.load(f)["bytes_per_row"])
    except Exception:
        return None


def cpu_baseline(cfg, sd, budget_s: float = 12.0, chunk: int = 32):
    """Oracle encode throughput on the host cores (rank 0, N=1 only): bounded sample of the same workload."""
    from oracle.qinco_oracle import OracleQINCo
    from qinco_amd import synth_vectors
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    oracle = OracleQINCo.from_config(cfg, sd)
    x = synth_vectors(cfg, sd, 4096, seed=4242)
    oracle(x[:8], step="encode")  # warm-up
    done, t0 = 0, time.perf_counter()
    while done < len(x):
        oracle(x[done:done + chunk], step="encode")
        done += chunk
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "vectors/s", "cores": int(threads), "kind": "port",
            "sample": f"{done} vectors of the same workload in {dt:.1f} s (numpy fp32 oracle, batches of {chunk})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4"])
    ap.add_argument("--batch", type=int, default=16384, help="vectors per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank). gloo is a test hook: ranks may share a GPU.")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPU(s): RCCL needs one GPU per rank")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
    from qinco_amd.config import BASELINE_CONFIGS

    cfg = BASELINE_CONFIGS[args.workload]
    sd = synth_state_dict(cfg, 1236)
    eng = QincoEngine(cfg, sd, max_batch=args.batch)

    # this rank's shard of the synthetic database: rows [rank*batch, (rank+1)*batch) of one seeded stream per step
    x_host = synth_vectors(cfg, sd, args.batch, seed=42 + rank)
    x = torch.from_numpy(x_host).to(dev)
    K = args.steps

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.encode(x, code_dtype=np.uint8)
    barrier()
    eng.profile_enable(True)
    eng.profile_read()
    codes_steps = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        codes_steps.append(eng.encode(x, code_dtype=np.uint8))
    mine = torch.stack(codes_steps) if K else torch.empty(0, dtype=torch.uint8, device=dev)
    if world > 1:  # the end-of-job gather of the uint8 codes over RCCL / xGMI
        mine_c = mine.to(comm_dev)
        bucket = [torch.empty_like(mine_c) for _ in range(world)] if rank == 0 else None
        dist.gather(mine_c, bucket, dst=0)
        if rank == 0:
            assert len(bucket) == world and all(b.shape == mine_c.shape for b in bucket)
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_vecs = K * args.batch * world
        value = total_vecs / dt if dt > 0 else 0.0
        launches = max(prof["mlp_launches"], 1)
        avg_ms = prof["mlp_ms"] / launches
        flops_per_launch = prof["mlp_flops"] / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        bpr = pmc_traffic_per_row() if args.workload == "C2" else None
        rows_per_launch = flops_per_launch / cfg.mlp_flops_per_row()
        # FOLD (mlp_kernel.hpp): the row-independent head of the MLP is not recomputed per row, so the MFMA executes
        # fewer FLOPs than the reference's algorithm counts; `achieved` stays algorithmic, this is the executed share.
        Ae = cfg.A or cfg.K
        head = (2.0 * cfg.D * cfg.De if cfg.De != cfg.D else 0.0) + 2.0 * (cfg.De + cfg.D) * cfg.De   # FOLD
        head += 2.0 * cfg.De * cfg.dh if cfg.L > 0 else 0.0                                            # FOLD2
        per_group = 2.0 * cfg.D * cfg.De + (2.0 * cfg.De * cfg.dh if cfg.L > 0 else 0.0)               # xproj
        executed = 1.0 - (head - per_group / Ae) / cfg.mlp_flops_per_row()
        out = {
            "metric": "encode vectors/sec (BigANN-shaped d=128 8x8, beam=%d)" % cfg.B,
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3 if K else 0.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: qinco2-L 8x8 encode" if args.workload == "C2" else args.workload,
                       "D": cfg.D, "M": cfg.M, "K": cfg.K, "L": cfg.L, "de": cfg.De, "dh": cfg.dh, "A": cfg.A, "B": cfg.B,
                       "vectors_per_step_per_gpu": args.batch, "parallelism": f"shard{world}",
                       "weights": "seeded synthetic (RandomState 1236)",
                       "gflop_per_vector": eng.flops_per_vector("encode") / 1e9},
            "roofline": {"bound": "mfma", "kernel": "qinco::mlp_kernel (+ its xproj pre-GEMM)", "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": (bpr * rows_per_launch if bpr else None),
                         "traffic_unit": "bytes per launch (L2<->fabric, PMC pass in profiles/r01_c2_traffic.json)",
                         "mfma_flops_executed_frac": executed, "mfma_pipe_tflops": achieved * executed,
                         "avg_launch_ms": avg_ms, "launches": prof["mlp_launches"],
                         "flops_per_launch": flops_per_launch,
                         "mlp_share_of_step_time": prof["mlp_ms"] * 1e-3 / dt if dt > 0 else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
