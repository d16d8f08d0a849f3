#!/usr/bin/env python
"""Timing experiment (QINCO_HIP_LIB = an experiment build whose knn_table_kernel<128, true> prints cycle stamps of four workgroups):
one filtered search of 2048 queries x 10^6 rows at D = 128."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from qinco_amd.search import KnnSearcher  # noqa: E402
g = torch.Generator(device="cuda").manual_seed(0)
db = torch.randn(1_000_000, 128, device="cuda", generator=g)
q = torch.randn(2048, 128, device="cuda", generator=g)
knn = KnnSearcher(128)
for _ in range(2):
    knn.search(db, q, k=100)
    torch.cuda.synchronize()
print(knn.last_stats())
