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


def compact_code_dtype(K: int, cols: int, M: int):
    """The narrowest integer type that holds a code row: a byte per code when K <= 256 and there is no IVF column (cols == M),
    else int32 (an IVF id goes up to 2^24).  The reference keeps int64 everywhere (8 B per code): at 10^9 vectors x 8 codes that is
    64 GB per copy for an 8 GB payload."""
    return np.dtype(np.uint8) if (K <= 256 and cols == M) else np.dtype(np.int32)


def encode_shard(model: Callable, db_vecs, start: int, end: int, batch: int = 65536,
                 to_device: Optional[Callable] = None, sink: Optional[Callable] = None, pipeline: bool = True,
                 code_dtype=np.int64, keep: bool = True, K: Optional[int] = None, M: Optional[int] = None,
                 stats: Optional[dict] = None) -> Optional[np.ndarray]:
    """Encode rows [start, end) in batches: codes (end-start, M) in file order (search_tasks.py:107-116).
    `model(x, step="encode")` returns (M, n) like the reference's model object.  sink(codes_batch): called with every
    batch's codes as they arrive (the part-file writer compresses them while the GPU encodes the next batch).

    Memory: every batch is narrowed as it arrives and stored into ONE preallocated (rows, cols) array -- uint8 when K <= 256 (int32
    with an IVF column; K, M given) -- so a shard costs `cols` bytes per vector on the host while it is encoded; `code_dtype` is
    what is RETURNED (np.int64 = the reference's type, widened once at the end; "compact" = as stored).  keep=False: nothing is
    retained, every batch goes to `sink` only (a part-file-only encode of a billion vectors holds one batch), returns None.

    pipeline: the host work either side of the model call -- paging the next batch in from the memmap, and the transpose /
    narrowing / sink of the previous batch's codes -- runs on two helper threads while the model call (which releases
    the GIL inside libqinco_hip) keeps the GPU busy.  The reference does all of it serially (search_tasks.py:107-116), which
    is free next to its CPU encode and 15 % of the wall clock next to a qinco2-S encode on the GPU.  Same results, same order."""
    import time as _time
    bounds = [(i0, min(end, i0 + batch)) for i0 in range(start, end, batch)]
    state = {"out": None}
    tm = {"model_s": 0.0, "wait_load_s": 0.0, "wait_finish_s": 0.0, "batches": len(bounds)}   # where the driving thread's time went (stats)
    if not bounds:
        return np.zeros((0, 0), np.int64) if keep else None

    def load(k):
        i0, i1 = bounds[k]
        xb = db_vecs[i0:i1]
        if to_device is not None:
            return to_device(xb)
        if isinstance(xb, np.memmap) or (isinstance(xb, np.ndarray) and not xb.flags.owndata and pipeline):
            xb = np.array(xb, copy=True)           # page the rows in HERE (a contiguous memmap slice is a view until it is copied),
        return xb                                  # not inside the timed model call

    def finish(k, codes, ready):
        if ready is not None:
            ready.synchronize()                    # the encode that produced `codes` (enqueued on the CALLER's stream) has finished
        codes = codes.T
        if hasattr(codes, "cpu"):
            codes = codes.cpu().numpy()
        if state["out"] is None:
            cols = codes.shape[1]
            narrow = compact_code_dtype(K, cols, M) if (K is not None and M is not None) else np.dtype(np.int64)
            state["narrow"] = narrow
            if keep:
                state["out"] = np.empty((end - start, cols), narrow)
        narrow = state["narrow"]
        if narrow.itemsize < 8 and codes.size and (int(codes.max()) > np.iinfo(narrow).max or int(codes.min()) < 0):
            raise ValueError(f"a code does not fit {narrow} (K={K})")
        small = np.ascontiguousarray(codes, dtype=narrow)
        if keep:
            i0, i1 = bounds[k]
            state["out"][i0 - start: i1 - start] = small
        if sink is not None:
            sink(small)

    def event_of(codes):
        """Device-path encodes are asynchronous on the caller's current stream; the helper thread that copies them to the host
        must wait for THAT stream (its own current stream is the default one)."""
        if hasattr(codes, "is_cuda") and codes.is_cuda:
            import torch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(codes.device))
            return ev
        return None

    if not pipeline or len(bounds) == 1:
        for k in range(len(bounds)):
            finish(k, model(load(k), step="encode"), None)
    else:
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max_workers=1) as loader, cf.ThreadPoolExecutor(max_workers=1) as finisher:   # one each: order kept
            # (to_device runs on the caller's thread: the current CUDA device is per-thread state, and `.cuda()` in a helper
            # thread would land on device 0)
            prefetch = to_device is None
            nxt = loader.submit(load, 0) if prefetch else None
            done = None
            for k in range(len(bounds)):
                t0 = _time.perf_counter()
                xb = nxt.result() if prefetch else load(k)
                if prefetch and k + 1 < len(bounds):
                    nxt = loader.submit(load, k + 1)
                t1 = _time.perf_counter()
                codes = model(xb, step="encode")
                t2 = _time.perf_counter()
                if done is not None:
                    done.result()                       # (surfaces exceptions of the previous batch's post-processing)
                done = finisher.submit(finish, k, codes, event_of(codes))
                t3 = _time.perf_counter()
                tm["wait_load_s"] += t1 - t0
                tm["model_s"] += t2 - t1
                tm["wait_finish_s"] += t3 - t2
            t3 = _time.perf_counter()
            done.result()
            tm["wait_finish_s"] += _time.perf_counter() - t3
    if stats is not None:
        stats.update(tm)
    if not keep:
        return None
    out = state["out"]
    if isinstance(code_dtype, str) and code_dtype == "compact":
        return out
    return out if out.dtype == np.dtype(code_dtype) else out.astype(code_dtype)


# ---------------------------------------------------------------------------------------------
# part-file writer: the reference's format, written as fast as the GPU produces codes
# ---------------------------------------------------------------------------------------------
def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, m) for m in mat]


_CRC_SHIFT = {}     # len -> the 32 x 32 GF(2) operator that advances a CRC over `len` zero bytes


def _crc_shift_operator(n: int):
    op = _CRC_SHIFT.get(n)
    if op is None:
        ident = [1 << i for i in range(32)]
        op = ident
        power = _gf2_square(_gf2_square(_gf2_square([0xEDB88320] + [1 << i for i in range(31)])))   # one zero BYTE
        k = n
        while k:
            if k & 1:
                op = [_gf2_times(power, col) for col in op]
            power = _gf2_square(power)
            k >>= 1
        _CRC_SHIFT[n] = op
    return op


def crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """CRC-32 of A + B from crc32(A), crc32(B), len(B) (what zlib's crc32_combine does; Python's zlib does not export it): lets
    every chunk's CRC be computed on the worker thread that deflates it.  The operator of a length is cached (all chunks
    but the last have the same one)."""
    if len2 <= 0:
        return crc1
    return _gf2_times(_crc_shift_operator(len2), crc1) ^ crc2


class PartFileWriter:
    """`np.savez_compressed(path, codes=<(N, M) int64>)` (search_tasks.py:125-131) without its cost.

    numpy deflates the 8 N M bytes of a part file on one core at the end of the job: 2.4-4.6 s per million vectors -- nothing
    next to the reference's CPU encode, but MORE than the whole GPU encode of a qinco2-S model (1.5 s per million).  Same file
    format here -- a zip with one deflated member `codes.npy`, readable by np.load, by the reference's EncodedDBIterator
    (search_utils.py:33-78) and by zipfile's CRC check -- produced pigz-style: the .npy byte stream is cut into 1 MiB chunks (one batch of 65 536 x 8 codes = 4 of them: the
    tail behind the last batch deflates on 4 threads, not on one),
    each deflated independently on a thread pool (zlib releases the GIL) and closed with a sync flush, so that their
    concatenation is ONE valid raw-deflate stream; chunks are compressed while later batches are still being encoded.
    rows / M must be known up front (the .npy header comes first in the stream); zip64 records are always written (numpy
    does the same: force_zip64).  zlib level 4 (numpy: 6): on int64 code rows 11 ms instead of 70 ms per MiB for files 1.6 % larger
    (0.2113 against 0.2080 of the raw size) -- what remains to deflate behind the last batch is what a job waits for at the end, and
    the last rows are cut into one piece per thread (round 5: the tail of a 10^6-vector qinco2-S job 38-71 ms -> see HISTORY)."""

    CHUNK = 1 << 20

    def __init__(self, path: str, rows: int, M: int, threads: int = 8, level: int = 4):
        import concurrent.futures as cf
        import io
        import struct
        import time as _t
        self.path, self.rows, self.M, self.level = path, int(rows), int(M), level
        hdr = io.BytesIO()
        np.lib.format.write_array_header_1_0(hdr, {"descr": "<i8", "fortran_order": False, "shape": (self.rows, self.M)})
        self._buf = bytearray(hdr.getvalue())
        self._usize = self._csize = 0
        self._crc = 0
        self._seen = 0
        self._pool = cf.ThreadPoolExecutor(max_workers=max(1, threads))
        self._futs = []          # compressed chunks not yet on disk, in stream order
        self._depth = 4 * max(1, threads)
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        lt = _t.localtime()
        self._dos = ((lt.tm_hour << 11) | (lt.tm_min << 5) | (lt.tm_sec // 2),
                     ((max(lt.tm_year, 1980) - 1980) << 9) | (lt.tm_mon << 5) | lt.tm_mday)
        self._name = b"codes.npy"
        self._tmp = path + ".tmp"     # renamed onto `path` by close(): a part file that exists is a complete one (resume relies on it)
        self._f = open(self._tmp, "wb")
        z64 = struct.pack("<HHQQ", 1, 16, 0, 0)                     # patched at close: uncompressed, compressed size
        self._f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 45, 0, 8, self._dos[0], self._dos[1], 0, 0xFFFFFFFF, 0xFFFFFFFF,
                                  len(self._name), len(z64)))
        self._f.write(self._name)
        self._z64_at = self._f.tell()
        self._f.write(z64)

    def _deflate(self, data: bytes, last: bool):
        import zlib
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        return co.compress(data) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(data), len(data)

    def _drain(self, everything: bool):
        """Finished chunks at the head of the queue go to disk (in order); at most `_depth` chunks are kept in flight."""
        while self._futs and (everything or self._futs[0].done() or len(self._futs) > self._depth):
            blob, crc, n = self._futs.pop(0).result()
            self._f.write(blob)
            self._csize += len(blob)
            self._crc = crc32_combine(self._crc, crc, n)

    def _submit(self, last: bool):
        while len(self._buf) >= self.CHUNK or (last and self._buf):
            take = bytes(self._buf[: self.CHUNK])
            del self._buf[: self.CHUNK]
            fin = last and not self._buf
            self._usize += len(take)
            self._futs.append(self._pool.submit(self._deflate, take, fin))
            self._finished = fin
        self._drain(False)

    def add(self, codes: np.ndarray):
        codes = np.ascontiguousarray(codes, dtype="<i8")
        assert codes.ndim == 2 and codes.shape[1] == self.M and self._seen + len(codes) <= self.rows
        if len(codes) == 0:          # (nothing to append: in particular the deflate stream a last add() has closed stays closed)
            return
        self._seen += len(codes)
        self._buf += codes.tobytes()
        self._finished = False
        if self._seen == self.rows and len(self._buf) > 0:
            # the last rows: everything still buffered is cut into one piece per thread (>= 64 KiB) and closed now, so that close()
            # finds the stream finished instead of waiting for whole 1 MiB chunks
            piece = max(64 << 10, -(-len(self._buf) // self._pool._max_workers))
            piece = min(piece, self.CHUNK)
            chunk, self.CHUNK = self.CHUNK, piece
            try:
                self._submit(True)
            finally:
                self.CHUNK = chunk
            return
        self._submit(False)

    def close(self):
        import struct
        assert self._seen == self.rows, f"{self._seen} of {self.rows} rows written"
        if not getattr(self, "_finished", False):
            self._submit(True)
        if not getattr(self, "_finished", False):       # nothing was left to flush (an empty file, or the data ended on a chunk boundary): terminate the stream
            self._futs.append(self._pool.submit(self._deflate, b"", True))
        self._drain(True)
        f, name = self._f, self._name
        cd_at = f.tell()
        z64c = struct.pack("<HHQQQ", 1, 24, self._usize, self._csize, 0)   # + local header offset
        f.write(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 45, 45, 0, 8, self._dos[0], self._dos[1], self._crc, 0xFFFFFFFF,
                            0xFFFFFFFF, len(name), len(z64c), 0, 0, 0, 0o600 << 16, 0xFFFFFFFF))
        f.write(name)
        f.write(z64c)
        cd_size = f.tell() - cd_at
        eocd64_at = f.tell()
        f.write(struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, 1, 1, cd_size, cd_at))
        f.write(struct.pack("<IIQI", 0x07064B50, 0, eocd64_at, 1))
        f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, 1, 1, min(cd_size, 0xFFFFFFFF), min(cd_at, 0xFFFFFFFF), 0))
        f.seek(14)
        f.write(struct.pack("<I", self._crc))
        f.seek(self._z64_at + 4)
        f.write(struct.pack("<QQ", self._usize, self._csize))
        f.close()
        os.replace(self._tmp, self.path)
        self._pool.shutdown()


def _comm_device(dist, device=None, group=None):
    """Device the collectives' tensors must live on: RCCL (backend "nccl") moves GPU buffers only, gloo CPU ones."""
    if device is not None:
        return device
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
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
                    dist=None, gather: bool = False, to_device: Optional[Callable] = None, device=None,
                    writer_threads: int = 8, resume: bool = False, code_dtype=np.int64, keep: bool = True, stats: Optional[dict] = None):
    """Reference-compatible database encode.  Returns this rank's codes; with gather=True rank 0 returns the
    whole (N, M) code matrix collected with one collective (other ranks: their own shard).  `device`: where the
    collective's buffers live (default: this rank's current GPU under backend "nccl", the host under "gloo").

    Files: `<output>` = np.savez_compressed(n_parts, K, M, D) by rank 0; `<base>.part_<rank>.npz` = codes
    (search_tasks.py:119-134; the reference logs `.{rank}.npz` but writes `.part_{rank}.npz`).  The part file is deflated on
    `writer_threads` threads while the shard is still being encoded (PartFileWriter; 0 = numpy's single-threaded
    np.savez_compressed at the end, the reference's way -- same format either way).

    Host memory (the north_star's 10^9-vector job): codes are held as uint8 (int32 with an IVF column) from the GPU to the gather
    and widened to the file format's int64 only chunk by chunk inside the part-file writer.  code_dtype = what is RETURNED:
    np.int64 (default, the reference's type: 8 B per code, fine at 10^6), "compact" (as held: M bytes per vector; the gathered
    matrix on rank 0 too).  keep=False: nothing is returned or retained (part files only; not with gather or writer_threads=0).

    resume=True: a rank whose part file is already there -- right number of rows AND code columns, next to an `<output>` header
    written for the same n_parts, K, M, D -- loads it instead of encoding its shard again (the reference's encode_database has no
    resume: a crashed rank costs the whole job; part files are written under a temporary name and renamed when complete, so one
    that exists is whole).  Every rank still takes part in the barriers and in the gather."""
    assert output.endswith(".npz")
    assert keep or (not gather and writer_threads > 0), "keep=False writes part files only"
    base = output[:-4]
    rank, world = _dist_info(dist)
    if world > 1:
        _barrier(dist, device)
    start, end = shard_bounds(len(db_vecs), world, rank)
    state = {"writer": None}
    part = base + f".part_{rank}.npz"
    done = _load_part(part, end - start, (M, M + 1), _header_matches(output, world, K, M, D)) if resume else None
    compact = isinstance(code_dtype, str) and code_dtype == "compact"

    def sink(c):   # created with the first batch: the number of code columns is the model's business (M + 1 with an IVF column)
        if state["writer"] is None:
            state["writer"] = PartFileWriter(part, end - start, c.shape[1], writer_threads)
        state["writer"].add(c)
    if done is not None:
        codes = done
        if keep and (compact or gather):
            codes = codes.astype(compact_code_dtype(K, codes.shape[1], M))
        elif not keep:
            codes = None
    else:
        codes = encode_shard(model, db_vecs, start, end, batch, to_device, sink if (writer_threads > 0 and end > start) else None,
                             code_dtype="compact", keep=keep, K=K, M=M, stats=stats)
    writer = state["writer"]
    if codes is not None and codes.size == 0:
        codes = np.zeros((0, M), compact_code_dtype(K, M, M))
    if world > 1:
        _barrier(dist, device)
    if rank == 0:
        d = os.path.dirname(output)
        if d:
            os.makedirs(d, exist_ok=True)
        np.savez_compressed(output, n_parts=world, K=K, M=M, D=D)
    if writer is not None:
        import time as _time
        t0 = _time.perf_counter()
        writer.close()
        if stats is not None:
            stats["writer_close_s"] = _time.perf_counter() - t0
    elif done is None and (keep or end == start):
        # (no writer: writer_threads = 0, or an EMPTY shard -- N < world -- which never reaches the sink: the header says
        # n_parts = world, so every rank owes the readers a part file, also with keep=False)
        tmp = part + ".tmp.npz"      # (np.savez appends .npz to other suffixes)
        np.savez_compressed(tmp, codes=(codes if codes is not None else np.zeros((0, M))).astype(np.int64))
        os.replace(tmp, part)
    if world > 1:
        _barrier(dist, device)
    if not keep:
        return None
    if gather:
        codes = gather_codes(codes, len(db_vecs), dist, device=device, code_dtype="compact")
    if compact or codes.dtype == np.dtype(code_dtype):
        return codes
    return codes.astype(code_dtype)


def _header_matches(output: str, n_parts: int, K: int, M: int, D: int) -> bool:
    """resume: the `<output>` header of an earlier run describes the same job (a part file of another model or another world size
    with the same shard length must not be taken for this run's)."""
    try:
        info = np.load(output)
        return all(int(info[k]) == v for k, v in (("n_parts", n_parts), ("K", K), ("M", M), ("D", D)))
    except Exception:
        return False


def _load_part(path: str, rows: int, cols=None, header_ok: bool = True) -> Optional[np.ndarray]:
    """The codes of a finished part file, or None (missing, unreadable, of another shard size or column count, or without a
    matching header)."""
    if not header_ok or not os.path.exists(path):
        return None
    try:
        codes = np.load(path)["codes"]
    except Exception:      # a truncated / foreign file: encode the shard again
        return None
    if codes.ndim != 2 or len(codes) != rows or (cols is not None and codes.shape[1] not in cols):
        return None
    return codes


def gather_codes(codes_local: np.ndarray, db_size: int, dist, device=None, code_dtype=np.int64, group=None,
                 stats: Optional[dict] = None, self_transfer: bool = False) -> Optional[np.ndarray]:
    """The per-rank code shards to rank 0: every rank sends its shard once, rank 0 receives each one straight into
    its rows of ONE preallocated (N, M) matrix of the wire type -- uint8 when every code is below 256 (M bytes per vector), else
    the shards' own type -- so the collective costs rank 0 one copy of the payload (8 GB at 10^9 x 8 codes), not a padded bucket
    per rank plus int64 copies of all of them.  Shards may differ in length (the last rank holds the remainder): point-to-point
    transfers (grouped on RCCL) instead of a fixed-size gather.  Returns (N, M) on rank 0 -- `code_dtype` np.int64 (the reference's
    type) or "compact" (the wire type) -- and the local shard elsewhere.

    group: the process group that carries the payload (default: the default group) -- a job whose default group is a gloo control
    plane passes its "nccl" sub-group here and the transfers run on RCCL from device buffers.  Call sequence on RCCL, per rank:
    all_reduce(MAX) of one int32 (the wire type), then ONE batch_isend_irecv (ncclGroupStart ... ncclGroupEnd) -- rank r > 0: one
    isend of its (n_r, M) shard; rank 0: world - 1 irecv, each into its row range of the (N, M) matrix -- wait, and on rank 0 one
    D2H copy.  stats (optional dict) receives wire_dtype, bytes_sent, bytes_received, ranks (the group's size as torch.distributed
    sees it) and the seconds spent in the transfers.

    self_transfer: a job of ONE rank normally has nothing to move and returns its shard.  With self_transfer=True (and an
    initialised process group whose payload group is on RCCL -- gloo cannot send to itself: ValueError) it runs the very same lines with rank 0 in both roles -- the all_reduce, then one grouped isend to
    itself + irecv from itself -- which is how a 1-GPU box executes this function on the real RCCL
    (tests/test_multi_gpu.py::test_torch_rccl_world_of_one, bench.py --gpus 1 --dry-rccl)."""
    import time
    rank, world = _dist_info(dist)
    compact = isinstance(code_dtype, str) and code_dtype == "compact"
    loopback = world == 1 and self_transfer and dist is not None and dist.is_available() and dist.is_initialized()
    if world == 1 and not loopback:
        return codes_local if compact else codes_local.astype(code_dtype, copy=False)
    import torch
    if group is not None and dist.get_world_size(group) != world:
        raise ValueError("gather_codes: the payload group must span every rank of the job")
    if loopback and str(dist.get_backend(group)) != "nccl":
        raise ValueError("gather_codes(self_transfer=True) needs a payload group on RCCL (backend 'nccl'): gloo has no connection from a rank to itself")
    device = _comm_device(dist, device, group)
    t0 = time.perf_counter()
    M = codes_local.shape[1]
    small = codes_local.size == 0 or int(codes_local.max()) < 256
    wide = 0 if small else (1 if codes_local.dtype.itemsize <= 4 and int(codes_local.max()) < 2 ** 31 else 2)
    flag = torch.tensor([wide], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    wire = (np.uint8, np.int32, np.int64)[int(flag.item())]
    mine = torch.from_numpy(np.ascontiguousarray(codes_local, dtype=wire))
    if device is not None:
        mine = mine.to(device)
    if stats is not None:
        stats.update(wire_dtype=np.dtype(wire).name, ranks=int(dist.get_world_size(group)), bytes_sent=0, bytes_received=0,
                     transport=str(dist.get_backend(group)), buffers="device" if device is not None else "host")
    # this rank's transfers, issued as ONE batch (RCCL: one ncclGroupStart / ncclGroupEnd; gloo: the ops one by one)
    ops, full, received = [], None, 0
    sender = rank != 0 or loopback
    if sender and len(mine):
        ops.append(dist.P2POp(dist.isend, mine, 0, group=group))
    if rank == 0:
        full = torch.empty((db_size, M), dtype=mine.dtype, device=mine.device)
        if not loopback:                             # rank 0's own shard: a copy into its rows
            s0, e0 = shard_bounds(db_size, world, 0)
            full[s0:e0] = mine
        for r in range(0 if loopback else 1, world):
            s, e = shard_bounds(db_size, world, r)
            if e > s:
                ops.append(dist.P2POp(dist.irecv, full[s:e], r, group=group))
                received += (e - s) * M * mine.element_size()
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    if device is not None and ops:
        torch.cuda.synchronize(device)               # (RCCL's calls return when they are enqueued: the seconds below are the transfer's)
    if stats is not None:
        stats.update(bytes_sent=int(mine.numel() * mine.element_size()) if sender else 0, bytes_received=int(received),
                     seconds=time.perf_counter() - t0)
    if rank != 0:
        return codes_local
    out = full.cpu().numpy()
    return out if compact else out.astype(code_dtype, copy=False)


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
            # (the reference's own estimate, on purpose: n_parts x the length of the part being read, search_utils.py:62 -- exact
            # only for equal shards; callers that need the count use load_all() or the header)
            self.n_samples = self.n_parts * len(db_codes)
            for ib in range(0, len(db_codes), bs or 1):
                batch = db_codes[ib: ib + bs]
                self.batch_end_id = self.batch_start_id + len(batch)
                yield batch
                self.batch_start_id += len(batch)

    def load_all(self) -> np.ndarray:
        return np.concatenate(list(self.iter()), axis=0)
