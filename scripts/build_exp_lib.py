#!/usr/bin/env python
"""Experiment builds of libqinco_hip.so (never shipped): the production objects with a few translation units recompiled
under extra -D flags.    python scripts/build_exp_lib.py NAME "-DQINCO_TIMELINE" 128,128,256,48,124 [more shapes]
-> scripts/exp_libs/lib_NAME.so  (select it with QINCO_HIP_LIB=...)"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qinco_amd import build as B  # noqa: E402

name, defs = sys.argv[1], ["-DQINCO_EXPERIMENT"] + sys.argv[2].split()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
B.build()
out_dir = ROOT / "scripts" / "exp_libs"
obj_dir = out_dir / f"obj_{name}"
obj_dir.mkdir(parents=True, exist_ok=True)
cc = B.hipcc()
objs = {o.name: o for o in B.OBJ.glob("*.o")}
procs = []
for (d, de, dh, p, var) in shapes:
    o = obj_dir / f"mlp_{d}_{de}_{dh}_{p}_{var}.o"
    procs.append(subprocess.Popen([cc, *B.FLAGS, *defs, f"-DQD={d}", f"-DQDE={de}", f"-DQDH={dh}", f"-DQP={p}", f"-DQVAR={var}", "-c",
                                   str(B.CSRC / "mlp_inst.hip"), "-o", str(o)]))
    objs[o.name] = o
o = obj_dir / "qinco_hip.o"
procs.append(subprocess.Popen([cc, *B.FLAGS, *defs, "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", str(B.CSRC / "qinco_hip.hip"), "-o", str(o)]))
objs[o.name] = o
for p in procs:
    assert p.wait() == 0
lib = out_dir / f"lib_{name}.so"
subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", str(lib), *map(str, objs.values())])
print(lib)
