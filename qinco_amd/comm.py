"""RCCL communicator for hosts that do not go through torch.distributed: a thin ctypes view of librccl (the one already in
the process -- PyTorch's copy when torch is imported, else /opt/rocm's) and the native gather of include/qinco_hip.h
(qinco_gather_codes).  One rank per GPU; the unique id travels through any side channel the launcher has (a file here).
The product's default multi-GPU path is qinco_amd.encode_db (torch.distributed); this is the C-ABI route of SURVEY.md 8(e).
"""
from __future__ import annotations

import ctypes as C
import os
import time
from pathlib import Path
from typing import Optional, Sequence

from . import _lib


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _librccl() -> C.CDLL:
    import importlib.util
    cands = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cands.append(Path(list(spec.submodule_search_locations)[0]) / "lib" / "librccl.so")
    except (ImportError, ValueError):
        pass
    cands += [Path("/opt/rocm/lib/librccl.so.1"), Path("librccl.so.1")]
    last = None
    for c in cands:
        try:
            return C.CDLL(str(c), mode=C.RTLD_GLOBAL)
        except OSError as e:
            last = e
    raise _lib.QincoLibraryError(f"no RCCL library found: {last}")


class RcclComm:
    """ncclCommInitRank over a unique id exchanged through `id_file` (rank 0 writes it, the others wait for it).  max_skew_s: how
    much older than this rank's own start the file may be (ranks of one job start within seconds of each other)."""

    def __init__(self, rank: int, world: int, id_file: str, timeout_s: float = 120.0, max_skew_s: float = 30.0):
        _lib.load()                      # one HIP runtime first
        self.rccl = _librccl()
        self.rank, self.world = rank, world
        uid = NcclUniqueId()
        # The id file carries a start time next to the id: a file left behind by an EARLIER job (same path) is older than this
        # process and is ignored by the waiting ranks -- they would otherwise hang in ncclCommInitRank on a dead id.  Rank 0
        # replaces the file atomically and removes it once every rank has joined.
        born = time.time()
        if rank == 0:
            self._ok(self.rccl.ncclGetUniqueId(C.byref(uid)))
            tmp = id_file + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(bytes(uid.internal))
            os.replace(tmp, id_file)
        else:
            t0 = time.time()
            while True:
                try:
                    fresh = os.path.getmtime(id_file) >= born - max_skew_s
                    blob = open(id_file, "rb").read() if fresh else b""
                except OSError:
                    blob = b""
                if len(blob) == 128:
                    break
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"no fresh RCCL unique id at {id_file}")
                time.sleep(0.05)
            C.memmove(C.byref(uid), blob, 128)
        self.comm = C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        self._ok(self.rccl.ncclCommInitRank(C.byref(self.comm), world, uid, rank))
        if rank == 0:               # every rank has joined (the call is collective): the id has served
            try:
                os.remove(id_file)
            except OSError:
                pass

    def _ok(self, rc: int):
        if rc != 0:
            self.rccl.ncclGetErrorString.restype = C.c_char_p
            raise RuntimeError(f"RCCL: {self.rccl.ncclGetErrorString(rc).decode()}")

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def gather_codes_native(codes_dev, counts: Sequence[int], rank: int, root: int = 0, comm: Optional[RcclComm] = None, stream=None):
    """codes_dev: torch CUDA tensor (n_local, M) of uint8 / int32 / int64 on this rank.  Returns the (N, M) tensor on `root`
    (None elsewhere).  Runs qinco_gather_codes on `stream` (default: the current torch stream) and synchronises it."""
    import torch
    lib = _lib.load()
    world = len(counts)
    M = int(codes_dev.shape[1])
    dt = {torch.int64: _lib.CODE_I64, torch.int32: _lib.CODE_I32, torch.uint8: _lib.CODE_U8}[codes_dev.dtype]
    codes_dev = codes_dev.contiguous()
    out = torch.empty((int(sum(counts)), M), dtype=codes_dev.dtype, device=codes_dev.device) if rank == root else None
    st = stream if stream is not None else torch.cuda.current_stream(codes_dev.device).cuda_stream
    cnt = (C.c_int64 * world)(*[int(c) for c in counts])
    _lib.check(lib.qinco_gather_codes(codes_dev.data_ptr(), int(codes_dev.shape[0]), M, dt,
                                      out.data_ptr() if out is not None else None, cnt, world, rank, root,
                                      comm.comm if comm is not None else None, st))
    torch.cuda.synchronize(codes_dev.device)
    return out
