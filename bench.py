#!/usr/bin/env python
"""bench.py -- encode throughput of the MI355X QINCo2 engine on BASELINE.json's metric.

    python bench.py [--gpus N --steps K --warmup W] [--workload C2] [--batch 16384]

`--gpus N` with N > 1 works both ways: pre-launched (the driver's `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) and from a bare `python bench.py --gpus N`, which re-executes itself under torch.distributed.run with one
rank per GPU (like the reference's `accelerate launch --multi_gpu`, run.sh:8).

A "step" = one pass of the hot path (model(x, step="encode")) over one batch of `--batch` synthetic fp32 vectors per
GPU.  Every step (and every warm-up step) encodes a DIFFERENT batch; all batches are generated on the device before the
timed region (inputs resident in HBM).  Weak scaling: every rank encodes its own shard of the synthetic database
(contiguous range sharding like search_tasks.py:103-104, no data-path collective); the uint8 codes of all timed steps are
gathered to rank 0 over RCCL inside the timed region.  Rank 0 prints ONE JSON line.

metric/unit: encode vectors/s (BASELINE.json "metric"); workload C2 = qinco2-L 8x8, D=128, A=16, B=8
(BASELINE.json configs[1]) with seeded synthetic weights (no trained checkpoints offline).
roofline: the fused codeword-MLP kernel (99.9 % of the FLOPs), fp32 MFMA bound.  `achieved` = ALGORITHMIC FLOPs per
launch (rows x R_mlp) / mean launch duration measured with HIP events on the launch stream; `frac` =
achieved / peak.  The kernel folds the row-independent head of the MLP out, so the matrix pipe executes
fewer FLOPs than the algorithm counts: `frac_executed` = executed FLOPs / duration / peak is the pipe-utilisation figure
(it cannot exceed 1; the algorithmic one can on short models).
Also in the line (N = 1, measured after the timed region): `decode` (vectors/s + its roofline), `mse` of
encode -> decode over the timed batches (AnyVectMSE, metrics.py:51-58), `beam1` (greedy encode), `batch_1024` (encode at
the reference's default batch, qinco_cfg.yaml:38), `split_f16` (the opt-in split-fp16 form of the FFN blocks on the same
batches: vectors/s, the code rows that differ from the fp32 path's, MSE; never the headline `value`).
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


def pmc_traffic_per_row():
    """L2<->fabric bytes per MLP row from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE).  Counters cannot be read from inside this process, so the figure of the separate PMC run of this
    same command is scaled to this run's rows per launch."""
    for name in ("r02_c2_traffic.json", "r01_c2_traffic.json"):
        try:
            with open(ROOT / "profiles" / name) as f:
                return float(json.load(f)["bytes_per_row"]), name
        except Exception:
            continue
    return None, None


def cpu_baseline(cfg, sd, budget_s: float = 20.0, chunk: int = 1024):
    """Oracle encode throughput on the host cores (rank 0, N=1 only): bounded sample of the same workload at the
    reference's default batch (qinco_cfg.yaml:38), ATen threads = the cores torch picked (stated)."""
    import torch
    from oracle.qinco_oracle import OracleQINCo
    from qinco_amd import synth_vectors
    # The reference's own CPU protocol runs 32 ATen threads (qinco_tasks.py:492).  On this box (128 cores) more threads are
    # slower: 16 / 32 / 64 / 128 threads gave 39 / 41 / 33 / 20 vectors/s at batch 1024 (profiles/r02_cpu_sweep.jsonl).
    threads = min(32, int(torch.get_num_threads()))
    torch.set_num_threads(threads)
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


def synth_batch_device(torch, cfg, mean_t, std, n, seed, dev):
    """x = mean + std * N(0, I) (the S0 inputs of SURVEY.md 8d), drawn on the device from a seeded generator."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    z = torch.randn((n, cfg.D), generator=g, device=dev, dtype=torch.float32)
    return z * std + mean_t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4"])
    ap.add_argument("--batch", type=int, default=16384, help="vectors per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the decode / mse / beam1 / batch_1024 legs")
    ap.add_argument("--split-f16", action="store_true",
                    help="NOT the driver's configuration: run the whole bench (any --gpus) on the opt-in split-fp16 form; the line says so in dtype / config")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank). gloo is a test hook: ranks may share a GPU.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N` bare)")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (qinco_amd has no CPU path)")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPU(s): RCCL needs one GPU per rank (--backend gloo lets ranks share one)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")

    from qinco_amd import QincoEngine, synth_state_dict
    from qinco_amd.config import BASELINE_CONFIGS
    from qinco_amd.evaluate import sqerr_sum

    cfg = BASELINE_CONFIGS[args.workload]
    sd = synth_state_dict(cfg, 1236)
    eng = QincoEngine(cfg, sd, max_batch=args.batch, split_f16=args.split_f16)
    K, W = args.steps, args.warmup

    # This rank's shard of the synthetic database: step s of rank r encodes rows [(r (W + K) + s) batch, ... + batch) of one
    # seeded stream; every batch is distinct and resident in HBM before the clock starts.
    mean_t = torch.from_numpy(np.asarray(sd["data_mean"])).to(dev)
    std = float(sd["data_std"])
    batches = [synth_batch_device(torch, cfg, mean_t, std, args.batch, 1_000_003 * (rank * (W + K) + s) + 42, dev)
               for s in range(W + K)]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    for s in range(W):
        eng.encode(batches[s], code_dtype=np.uint8)
    barrier()
    eng.profile_enable(True)
    eng.profile_read()
    codes_steps = []
    barrier()
    t0 = time.perf_counter()
    for s in range(K):
        codes_steps.append(eng.encode(batches[W + s], code_dtype=np.uint8))
    mine = torch.stack(codes_steps) if K else torch.empty(0, dtype=torch.uint8, device=dev)
    t_enc = t_gather = 0.0
    if world > 1:  # the end-of-job gather of the uint8 codes over RCCL / xGMI
        torch.cuda.synchronize(dev)
        t_enc = time.perf_counter() - t0
        mine_c = mine.to(comm_dev)
        bucket = [torch.empty_like(mine_c) for _ in range(world)] if rank == 0 else None
        dist.gather(mine_c, bucket, dst=0)
        if rank == 0:
            assert len(bucket) == world and all(b.shape == mine_c.shape for b in bucket)
        torch.cuda.synchronize(dev)
        t_gather = time.perf_counter() - t0 - t_enc
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_enable(False)

    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine_t = torch.tensor([t_enc, t_gather], dtype=torch.float64, device=comm_dev)
        allt = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        per_rank = [[float(a[0]), float(a[1])] for a in allt]

    if rank == 0:
        total_vecs = K * args.batch * world
        value = total_vecs / dt if dt > 0 else 0.0
        mlp_row = cfg.mlp_flops_per_row()

        def mlp_roofline(pr):
            launches = max(pr["mlp_launches"], 1)
            avg_ms = pr["mlp_ms"] / launches
            fpl = pr["mlp_flops"] / launches
            ach = fpl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            return launches, avg_ms, fpl, ach

        launches, avg_ms, flops_per_launch, achieved = mlp_roofline(prof)
        bpr, bpr_src = pmc_traffic_per_row() if args.workload == "C2" else (None, None)
        rows_per_launch = flops_per_launch / mlp_row
        # FOLD / FOLD2 (mlp_kernel.hpp): the row-independent head of the MLP is not recomputed per row, so the MFMA
        # executes fewer FLOPs than the reference's algorithm counts.
        def executed_share(Ae):
            head = (2.0 * cfg.D * cfg.De if cfg.De != cfg.D else 0.0) + 2.0 * (cfg.De + cfg.D) * cfg.De   # FOLD
            head += 2.0 * cfg.De * cfg.dh if cfg.L > 0 else 0.0                                            # FOLD2
            per_group = 2.0 * cfg.D * cfg.De + (2.0 * cfg.De * cfg.dh if cfg.L > 0 else 0.0)               # xproj
            return 1.0 - (head - per_group / Ae) / mlp_row
        executed = executed_share(cfg.A or cfg.K)
        out = {
            "metric": "encode vectors/sec (BigANN-shaped d=128 8x8, beam=%d)" % cfg.B,
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3 if K else 0.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not args.split_f16 else "f32 operands as fp16 hi+lo on the fp16 MFMA, fp32 accumulate (--split-f16)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: qinco2-L 8x8 encode" if args.workload == "C2" else args.workload,
                       "D": cfg.D, "M": cfg.M, "K": cfg.K, "L": cfg.L, "de": cfg.De, "dh": cfg.dh, "A": cfg.A, "B": cfg.B,
                       "vectors_per_step_per_gpu": args.batch, "parallelism": f"shard{world}",
                       "distinct_vectors_encoded": total_vecs,
                       "inputs": "every step encodes a different seeded batch, all resident in HBM before the clock starts",
                       "weights": "seeded synthetic (RandomState 1236)",
                       "gflop_per_vector": eng.flops_per_vector("encode") / 1e9},
            "roofline": {"bound": "mfma", "kernel": ("qinco::mlp_kernel (+ its xproj pre-GEMM)" if not args.split_f16 else
                                                      "qinco::mlp_split_kernel (+ xproj_split): fp16 pipe, so frac against the fp32-MFMA peak exceeds 1"),
                         "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "frac_executed": achieved * executed / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": (bpr * rows_per_launch if bpr else None),
                         "traffic_unit": f"bytes per launch (L2<->fabric, PMC pass in profiles/{bpr_src})" if bpr else None,
                         "mfma_flops_executed_frac": executed, "mfma_pipe_tflops": achieved * executed,
                         "avg_launch_ms": avg_ms, "launches": prof["mlp_launches"],
                         "flops_per_launch": flops_per_launch,
                         "mlp_share_of_step_time": prof["mlp_ms"] * 1e-3 / dt if dt > 0 else None},
        }
        if world > 1:
            out["multi_gpu"] = {
                "backend": args.backend,
                "per_rank_encode_vectors_per_s": [K * args.batch / t[0] if t[0] > 0 else 0.0 for t in per_rank],
                "per_rank_encode_s": [t[0] for t in per_rank],
                "per_rank_gather_s": [t[1] for t in per_rank],
                "gather_bytes_per_rank": int(mine.numel()),
                "note": "gather time of a rank includes waiting for the slowest rank's encode",
            }
        if world == 1 and not args.no_extras and K > 0:
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
            _, d_ms, _, d_ach = mlp_roofline(prd)
            d_exec = executed_share(1)
            out["decode"] = {"value": codes_all.shape[0] / dt_dec, "unit": "vectors/s", "vectors": int(codes_all.shape[0]),
                             "gflop_per_vector": eng.flops_per_vector("decode") / 1e9,
                             "roofline": {"bound": "mfma", "achieved": d_ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                          "frac": d_ach / PEAK_FP32_MFMA_TFLOPS,
                                          "frac_executed": d_ach * d_exec / PEAK_FP32_MFMA_TFLOPS, "avg_launch_ms": d_ms}}
            xs = torch.cat(batches[W:W + K])
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
            small = [batches[W][i:i + 1024] for i in range(0, min(args.batch, 16384), 1024)]
            out["batch_1024"] = {"value": timed_encode(small), "unit": "vectors/s", "calls": len(small),
                                 "note": "same engine, one encode call per 1024 distinct vectors"}
            # ---- greedy search (beam = 1, BASELINE metric: beam in {1, 8}) ----
            if cfg.B > 1:
                eng.set_beam(B=1)
                out["beam1"] = {"value": timed_encode(batches[W:W + min(K, 2)]), "unit": "vectors/s", "A": eng.A, "B": 1,
                                "gflop_per_vector": eng.flops_per_vector("encode") / 1e9}
                eng.set_beam(B=cfg.B)
            # ---- the opt-in split-fp16 form of the FFN blocks (include/qinco_hip.h QINCO_CREATE_SPLIT_F16): NOT the headline --
            # `value` above is the fp32 path; this leg re-encodes the same timed batches and counts the code rows that change
            try:
                if args.split_f16:
                    raise RuntimeError("the whole line is the split form (--split-f16); roofline.frac is quoted against the fp32-MFMA peak")
                eng2 = QincoEngine(cfg, sd, max_batch=args.batch, split_f16=True)
            except Exception as e:      # no split instance for this shape (or any other failure): the headline must not depend on it
                eng2 = None
                out["split_f16"] = {"error": f"{type(e).__name__}: {e}"}
            if eng2 is not None:
                try:
                    eng2.encode(batches[0], code_dtype=np.uint8)
                    torch.cuda.synchronize(dev)
                    eng2.profile_enable(True)
                    eng2.profile_read()
                    t3 = time.perf_counter()
                    codes2 = torch.stack([eng2.encode(batches[W + s], code_dtype=np.uint8) for s in range(K)])
                    torch.cuda.synchronize(dev)
                    dt2 = time.perf_counter() - t3
                    pr2 = eng2.profile_read()
                    eng2.profile_enable(False)
                    _, s_ms, s_fpl, s_ach = mlp_roofline(pr2)
                    differ = int((codes2.reshape(-1, cfg.M_total) != codes_all).any(dim=1).sum().item())
                    dec2 = eng2.decode(codes2.reshape(-1, cfg.M_total), check=False)
                    xs = torch.cat(batches[W:W + K])
                    blocks = 4.0 * cfg.L * cfg.De * cfg.dh - 2.0 * cfg.De * cfg.dh          # per row, block 0's up-projection folded
                    if cfg.De != cfg.D and (cfg.D // 32) % 2 == 0:
                        blocks += 2.0 * cfg.D * cfg.De                                      # out_proj in the split form too
                    blocks += (2.0 * cfg.D * cfg.De + 2.0 * cfg.De * cfg.dh) / (cfg.A or cfg.K)   # xproj, per group
                    f16_tflops = 3.0 * blocks * (s_fpl / mlp_row) / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0
                    out["split_f16"] = {
                        "value": K * args.batch / dt2, "unit": "vectors/s", "ms_per_step": dt2 / K * 1e3,
                        "speedup_vs_f32_path": (K * args.batch / dt2) / value if value > 0 else None,
                        "rows_differing_from_f32_path": differ, "rows": int(codes_all.shape[0]),
                        "mse": sqerr_sum(xs, dec2) / xs.shape[0],
                        "roofline": {"bound": "mfma", "kernel": "qinco::mlp_split_kernel (+ xproj)", "avg_launch_ms": s_ms,
                                     "achieved_algorithmic_fp32_equivalent_tflops": s_ach,
                                     "f16_mfma_tflops_executed": f16_tflops, "peak_f16_mfma_tflops": PEAK_F16_MFMA_TFLOPS,
                                     "frac_of_f16_peak": f16_tflops / PEAK_F16_MFMA_TFLOPS},
                        "arithmetic": "FFN blocks, out_proj, xproj: fp32 operands as fp16 hi + lo, 3 fp16 MFMAs per product, fp32 accumulate; tables, distances, selection fp32",
                        "note": "opt-in (QincoEngine(split_f16=True) / qinco_create_ex); parity tests: tests/test_hip_parity.py::test_split_f16_*"}
                    del dec2, xs, codes2
                    eng2.close()
                except Exception as e:
                    out["split_f16"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
