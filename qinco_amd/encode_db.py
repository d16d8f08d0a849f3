"""Database encoding: the caller of the hot path (reference `encode_database`, qinco/search/search_tasks.py:85-137)
and the on-disk format either side of it (qinco/search/search_utils.py:33-78, qinco/datasets.py:102-120).

One process per GPU; rank r encodes the contiguous range [ (N//P) r, (N//P)(r+1) ) (the last rank also takes the
remainder, search_tasks.py:103-104); there is no collective on the data path.  Output, two interchangeable ways:
  * part files exactly like the reference (`<out>.npz` header with n_parts, K, M, D written by rank 0 and
    `<out>.part_<rank>.npz` holding codes (N_r, M) int64 per rank) -- restartable per shard;
  * one gather of the uint8/int64 codes to rank 0 over torch.distributed (backend "nccl" = RCCL over xGMI on
    MI355X, "gloo" in the CPU tests): M bytes per vector, so a single collective at the end of the job.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Optional

import numpy as np


# ---------------------------------------------------------------------------------------------
# input files: *.bvecs / *.fvecs / *.ivecs / *.npy   (datasets.py:102-120: faiss.contrib.vecs_io mmaps)
# ---------------------------------------------------------------------------------------------
def _vecs_mmap(path: str, dtype, header_items: int):
    raw = np.memmap(path, dtype=dtype, mode="r")
    d = int(np.memmap(path, dtype=np.int32, mode="r", shape=(1,))[0])
    if d <= 0 or raw.size % (d + header_items):
        raise ValueError(f"{path}: not a vecs file (d={d})")
    return raw.reshape(-1, d + header_items)[:, header_items:]


def get_data_memmap(path: str) -> np.ndarray:
    """Memory-mapped (N, D) view: record = int32 d followed by d x {u8, f32, i32}.  bvecs rows stay strided
    uint8 (d+4 bytes apart) and are converted to fp32 on the GPU (search_tasks.py:109-110)."""
    if path.endswith(".bvecs"):
        return _vecs_mmap(path, np.uint8, 4)
    if path.endswith(".fvecs"):
        return _vecs_mmap(path, np.float32, 1)
    if path.endswith(".ivecs"):
        return _vecs_mmap(path, np.int32, 1)
    if path.endswith(".npy"):
        return np.load(path, mmap_mode="r")
    raise ValueError(f"unknown vector file type: {path}")


# ---------------------------------------------------------------------------------------------
# sharding
# ---------------------------------------------------------------------------------------------
def shard_bounds(db_size: int, nproc: int, proc_id: int) -> tuple[int, int]:
    """search_tasks.py:103-104."""
    per = db_size // nproc
    start = per * proc_id
    end = per * (proc_id + 1) if proc_id < nproc - 1 else db_size
    return start, end


def _dist_info(dist):
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def encode_shard(model: Callable, db_vecs, start: int, end: int, batch: int = 65536,
                 to_device: Optional[Callable] = None) -> np.ndarray:
    """Encode rows [start, end) in batches: codes (end-start, M) int64 in file order (search_tasks.py:107-116).
    `model(x, step="encode")` returns (M, n) like the reference's model object."""
    parts = []
    for i0 in range(start, end, batch):
        i1 = min(end, i0 + batch)
        xb = db_vecs[i0:i1]
        if to_device is not None:
            xb = to_device(xb)
        codes = model(xb, step="encode")
        codes = codes.T
        if hasattr(codes, "cpu"):
            codes = codes.cpu().numpy()
        parts.append(np.ascontiguousarray(codes, dtype=np.int64))
    if not parts:
        return np.zeros((0, 0), np.int64)
    return np.concatenate(parts)


def _comm_device(dist, device=None):
    """Device the collectives' tensors must live on: RCCL (backend "nccl") moves GPU buffers only, gloo CPU ones."""
    if device is not None:
        return device
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        import torch
        return torch.device("cuda", torch.cuda.current_device())
    return None


def _barrier(dist, device=None):
    dev = _comm_device(dist, device)
    if dev is not None and getattr(dev, "type", None) == "cuda":
        dist.barrier(device_ids=[dev.index if dev.index is not None else 0])
    else:
        dist.barrier()


def encode_database(model: Callable, db_vecs, output: str, *, K: int, M: int, D: int, batch: int = 65536,
                    dist=None, gather: bool = False, to_device: Optional[Callable] = None, device=None):
    """Reference-compatible database encode.  Returns this rank's codes; with gather=True rank 0 returns the
    whole (N, M) code matrix collected with one collective (other ranks: their own shard).  `device`: where the
    collective's buffers live (default: this rank's current GPU under backend "nccl", the host under "gloo").

    Files: `<output>` = np.savez_compressed(n_parts, K, M, D) by rank 0; `<base>.part_<rank>.npz` = codes
    (search_tasks.py:119-134; the reference logs `.{rank}.npz` but writes `.part_{rank}.npz`)."""
    assert output.endswith(".npz")
    base = output[:-4]
    rank, world = _dist_info(dist)
    if world > 1:
        _barrier(dist, device)
    start, end = shard_bounds(len(db_vecs), world, rank)
    codes = encode_shard(model, db_vecs, start, end, batch, to_device)
    if codes.size == 0:
        codes = np.zeros((0, M), np.int64)
    if world > 1:
        _barrier(dist, device)
    if rank == 0:
        d = os.path.dirname(output)
        if d:
            os.makedirs(d, exist_ok=True)
        np.savez_compressed(output, n_parts=world, K=K, M=M, D=D)
    np.savez_compressed(base + f".part_{rank}.npz", codes=codes)
    if world > 1:
        _barrier(dist, device)
    if gather:
        return gather_codes(codes, len(db_vecs), dist, device=device)
    return codes


def gather_codes(codes_local: np.ndarray, db_size: int, dist, device=None) -> Optional[np.ndarray]:
    """One gather of the per-rank code shards to rank 0.  Shards are padded to the longest one
    (the last rank holds the remainder) so a single fixed-size collective moves everything; codes travel as
    uint8 when K <= 256 (M bytes per vector).  Returns (N, M) int64 on rank 0, the local shard elsewhere."""
    rank, world = _dist_info(dist)
    if world == 1:
        return codes_local
    import torch
    device = _comm_device(dist, device)
    M = codes_local.shape[1]
    small = codes_local.size == 0 or int(codes_local.max()) < 256
    flag = torch.tensor([1 if small else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    wire = np.uint8 if int(flag.item()) == 1 else np.int64
    longest = max(shard_bounds(db_size, world, r)[1] - shard_bounds(db_size, world, r)[0] for r in range(world))
    buf = np.zeros((longest, M), wire)
    buf[: len(codes_local)] = codes_local.astype(wire)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return codes_local
    parts = []
    for r in range(world):
        s, e = shard_bounds(db_size, world, r)
        parts.append(out[r][: e - s].cpu().numpy().astype(np.int64))
    return np.concatenate(parts)


# ---------------------------------------------------------------------------------------------
# reader of the encoded database (search_utils.py:33-78)
# ---------------------------------------------------------------------------------------------
class EncodedDBIterator:
    def __init__(self, base_path: str, K: Optional[int] = None, M: Optional[int] = None, D: Optional[int] = None):
        assert base_path.endswith(".npz")
        self.part_base_path = base_path[:-4]
        info = np.load(base_path)
        self.n_parts = int(info["n_parts"])
        for name, want in (("K", K), ("M", M), ("D", D)):
            got = int(info[name])
            assert want is None or want == got, f"{name}: header has {got}, expected {want}"
            setattr(self, name, got)
        self.batch_start_id = self.batch_end_id = None
        self.n_samples = None

    def iter(self, batch_size: Optional[int] = None):
        self.batch_start_id = 0
        for i_part in range(self.n_parts):
            db_codes = np.load(self.part_base_path + f".part_{i_part}.npz")["codes"]
            bs = batch_size or len(db_codes)
            self.part_n_batches = math.ceil(len(db_codes) / bs) if bs else 0
            self.n_samples = self.n_parts * len(db_codes)
            for ib in range(0, len(db_codes), bs or 1):
                batch = db_codes[ib: ib + bs]
                self.batch_end_id = self.batch_start_id + len(batch)
                yield batch
                self.batch_start_id += len(batch)

    def load_all(self) -> np.ndarray:
        return np.concatenate(list(self.iter()), axis=0)
