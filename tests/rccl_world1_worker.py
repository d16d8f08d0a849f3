"""A world of ONE rank on the real RCCL, no torch.distributed: RcclComm through ctypes (ncclGetUniqueId -> ncclCommInitRank(nranks = 1)
-> ncclCommCount) and qinco_gather_codes WITH that communicator -- grouped ncclSend to self + ncclRecv from self (csrc/comm_hip.hip)
-- for every code type and an uneven byte count.  Prints one JSON line.  Run by tests/test_multi_gpu.py::test_rccl_world_of_one
as a child process (a hanging RCCL must not hang pytest)."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import numpy as np
    import torch
    from qinco_amd.comm import RcclComm, gather_codes_native
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = RcclComm.world_of_one()
    rec = {"count": comm.count(), "library": comm.library_path(), "made_by": os.path.realpath(comm.rccl._name), "cases": []}
    rng = np.random.default_rng(5)
    for dt, hi in ((np.uint8, 256), (np.int32, 2 ** 24), (np.int64, 2 ** 40)):
        for n, M in ((1, 1), (1001, 7), (65537, 9), (0, 8)):          # (odd byte counts; an empty shard)
            a = rng.integers(0, hi, size=(n, M)).astype(dt)
            got = gather_codes_native(torch.from_numpy(a).to(dev), [n], 0, root=0, comm=comm)
            rec["cases"].append({"dtype": np.dtype(dt).name, "rows": n, "M": M, "bytes": int(a.nbytes),
                                 "equal": bool(np.array_equal(got.cpu().numpy(), a))})
    # the copy route (no communicator) must still be there for hosts without RCCL
    a = rng.integers(0, 256, size=(33, 8)).astype(np.uint8)
    rec["copy_route_equal"] = bool(np.array_equal(gather_codes_native(torch.from_numpy(a).to(dev), [33], 0).cpu().numpy(), a))
    comm.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
