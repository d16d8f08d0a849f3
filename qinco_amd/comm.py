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
    # (c_ubyte, not c_char: ctypes hands a c_char array back as bytes CUT AT THE FIRST NUL -- an id is a socket address and a
    # magic number, full of zero bytes; rounds 2-4 wrote such a truncated id to the file and no 2-GPU box ever ran it)
    _fields_ = [("internal", C.c_ubyte * 128)]


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
    """ncclCommInitRank over a unique id that reaches the ranks one of three ways:

    * `uid` -- the 128 bytes themselves, for launchers with a channel of their own (`RcclComm.from_process_group`: rank 0's id
      broadcast over an existing torch.distributed control plane, no file at all);
    * `id_file` + `nonce` -- rank 0 writes `id || nonce` and the others accept only a file that ends in THEIR nonce (a job id, the
      launcher's run id, a random token passed on the command line): a file left behind by an earlier job under the same path can
      never be taken for this job's, however late a rank starts and whatever the clocks of the hosts / the file server say;
    * `id_file` alone -- single-host fall-back: a file is fresh when it was written at most `max_skew_s` (30 s) before this rank
      started, measured on this host's clock -- short on purpose: a file left by a job that died a minute ago under the same path
      must not be taken for this job's (the ranks would block in ncclCommInitRank on a dead id).  Launchers whose ranks start
      further apart pass a nonce.

    Rank 0 replaces the file atomically and removes it once every rank has joined (ncclCommInitRank is collective)."""

    def __init__(self, rank: int, world: int, id_file: Optional[str] = None, timeout_s: float = 120.0, nonce: Optional[str] = None,
                 uid: Optional[bytes] = None, max_skew_s: float = 30.0):
        _lib.load()                      # one HIP runtime first
        self.rccl = _librccl()
        self.rank, self.world = rank, world
        if uid is None and id_file is None:
            raise ValueError("RcclComm needs the unique id (uid=) or a file to exchange it through (id_file=)")
        if uid is not None:
            if len(uid) != 128:
                raise ValueError("an RCCL unique id is 128 bytes")
            uid_s = NcclUniqueId()
            C.memmove(C.byref(uid_s), bytes(uid), 128)
        else:
            uid_s = self._exchange_through_file(rank, id_file, timeout_s, nonce, max_skew_s)
        self.comm = C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        self._ok(self.rccl.ncclCommInitRank(C.byref(self.comm), world, uid_s, rank))
        if rank == 0 and uid is None:               # every rank has joined: the id has served
            try:
                os.remove(id_file)
            except OSError:
                pass

    def new_unique_id(self) -> bytes:
        uid = NcclUniqueId()
        self._ok(self.rccl.ncclGetUniqueId(C.byref(uid)))
        return bytes(uid.internal)

    @classmethod
    def from_process_group(cls, group=None) -> "RcclComm":
        """The communicator of the ranks of an initialised torch.distributed group (any backend: gloo is the control plane of
        qinco_amd.encode_db): rank 0's unique id travels in one broadcast_object_list, nothing touches the file system."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            probe = cls.__new__(cls)
            _lib.load()
            probe.rccl = _librccl()
            box[0] = probe.new_unique_id()
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, uid=box[0])

    @classmethod
    def world_of_one(cls) -> "RcclComm":
        """A communicator with this process as its only rank (ncclGetUniqueId -> ncclCommInitRank(nranks = 1)): what a 1-GPU
        box can create on the real library; qinco_gather_codes routes its shard through it (send-to-self + recv-from-self)."""
        probe = cls.__new__(cls)
        _lib.load()
        probe.rccl = _librccl()
        return cls(0, 1, uid=probe.new_unique_id())

    def library_path(self) -> str:
        """The shared object whose ncclSend / ncclRecv qinco_gather_codes calls in this process (dladdr, qinco_rccl_library)."""
        buf = C.create_string_buffer(4096)
        _lib.check(_lib.load().qinco_rccl_library(buf, len(buf)))
        return buf.value.decode()

    def count(self) -> int:
        """ncclCommCount: the ranks RCCL itself sees in this communicator."""
        n = C.c_int(0)
        self.rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._ok(self.rccl.ncclCommCount(self.comm, C.byref(n)))
        return n.value

    def _exchange_through_file(self, rank: int, id_file: str, timeout_s: float, nonce: Optional[str],
                               max_skew_s: float = 30.0) -> NcclUniqueId:
        uid = NcclUniqueId()
        tag = nonce.encode() if nonce is not None else b""
        born = time.time()
        if rank == 0:
            self._ok(self.rccl.ncclGetUniqueId(C.byref(uid)))
            tmp = id_file + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(bytes(uid.internal) + tag)
            os.replace(tmp, id_file)
            return uid
        t0 = time.time()
        while True:
            try:
                blob = open(id_file, "rb").read()
                if nonce is not None:
                    ok = len(blob) == 128 + len(tag) and blob[128:] == tag       # this job's file: no clock is consulted
                else:
                    ok = len(blob) == 128 and os.path.getmtime(id_file) >= born - max_skew_s
            except OSError:
                ok = False
            if ok:
                break
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no RCCL unique id of this job at {id_file}" + (f" (nonce {nonce!r})" if nonce is not None else ""))
            time.sleep(0.05)
        C.memmove(C.byref(uid), blob[:128], 128)
        return uid

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
