"""One rank of a multi-process database encode (launched by tests/test_multi_gpu.py, one process per rank).

    python tests/rank_worker.py RANK WORLD PORT OUTDIR BACKEND N [device-input]

Drives the product path end to end: QINCoHIP (HIP engine) -> encode_database (range sharding, part files, one gather).
BACKEND nccl = RCCL, one GPU per rank; gloo = ranks may share GPU 0 (CPU collectives)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    rank, world, port, outdir, backend, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
    dev_input = len(sys.argv) > 7
    import torch
    import torch.distributed as dist
    from conftest import golden_model
    from qinco_amd import synth_state_dict, synth_vectors
    from qinco_amd.encode_db import encode_database
    from qinco_amd.model import QINCoHIP
    ndev = torch.cuda.device_count()
    dev_index = rank % ndev if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd = golden_model("tiny_proj_beam")
    model = QINCoHIP(cfg, sd, max_batch=256)
    db = synth_vectors(cfg, sd, n, seed=4)
    to_device = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev_input else None
    full = encode_database(model, db, os.path.join(outdir, "db.npz"), K=cfg.K, M=cfg.M, D=cfg.D, batch=100, dist=dist,
                           gather=True, to_device=to_device)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full)
    if backend == "nccl":
        dist.barrier(device_ids=[dev_index])
    else:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
