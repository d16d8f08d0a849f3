#!/usr/bin/env python
"""Rebuild every cached kernel-instance module under qinco_amd/_instances/ that is older than the kernel sources (after a change to
csrc/*.hpp), so that the GPU box finds them fresh instead of compiling them itself.   python scripts/refresh_instances.py"""
import re
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qinco_amd import build as b  # noqa: E402

shapes = sorted({tuple(int(v) for v in m.groups()) for so in b.INST.glob("inst_*.so")
                 if (m := re.match(r"inst_(\d+)_(\d+)_(\d+)_\d+_\d+\.so", so.name))})
t0 = time.time()
for s in shapes:
    b.ensure_instance(*s, verbose=True)
print(f"{len(shapes)} shapes checked in {time.time() - t0:.0f} s")
