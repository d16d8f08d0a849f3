"""One rank of the native (C-ABI) multi-GPU path: its own GPU, its own QINCoHIP, an RCCL communicator made through ctypes,
qinco_gather_codes for the end-of-job collective.  argv: rank world id_file outdir n"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    rank, world, id_file, outdir, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], Path(sys.argv[4]), int(sys.argv[5])
    import torch
    from conftest import golden_model
    from qinco_amd import synth_vectors
    from qinco_amd.comm import RcclComm, gather_codes_native
    from qinco_amd.encode_db import shard_bounds
    from qinco_amd.model import QINCoHIP
    torch.cuda.set_device(rank)
    cfg, sd = golden_model("tiny_proj_beam")
    model = QINCoHIP(cfg, sd, max_batch=256, device=rank)
    x = synth_vectors(cfg, sd, n, seed=4)
    s, e = shard_bounds(n, world, rank)
    codes = model.engine.encode(torch.from_numpy(x[s:e]).cuda(), code_dtype=np.uint8)
    comm = RcclComm(rank, world, id_file, nonce=outdir.name)      # (the job's nonce: pytest's per-test directory)
    counts = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
    out = gather_codes_native(codes, counts, rank, root=0, comm=comm)
    if rank == 0:
        np.save(outdir / "gathered_native.npy", out.cpu().numpy().astype(np.int64))
    comm.close()


if __name__ == "__main__":
    main()
