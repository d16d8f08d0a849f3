"""Build libqinco_hip.so (gfx950) in-tree with hipcc.

One object per fused-MLP kernel instance listed in csrc/shapes.def (each is a ~10k-instruction fully
unrolled MFMA chain and takes ~40 s to compile, so they are built in parallel) plus the C-ABI translation
unit; everything is linked into qinco_amd/libqinco_hip.so, which travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "_build"
LIB = PKG / "libqinco_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]


def shapes() -> list[tuple[int, int, int, int, int]]:
    txt = (CSRC / "shapes.def").read_text()
    return [tuple(int(v) for v in m.groups())
            for m in re.finditer(r"^QINCO_SHAPE\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", txt, re.M)]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libqinco_hip.so")
    return exe


def _newer(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(d.stat().st_mtime <= t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> Path:
    OBJ.mkdir(exist_ok=True)
    cc = hipcc()
    mlp_deps = [CSRC / n for n in ("mlp_kernel.hpp", "mlp16_kernel.hpp", "mlp_split_kernel.hpp", "mlp_args.hpp", "mlp_launch.hpp", "mlp_inst.hip")]
    headers = sorted(CSRC.glob("*.hpp")) + [CSRC / "shapes.def", PKG.parent / "include" / "qinco_hip.h"]
    tasks: list[tuple[Path, list[str]]] = []
    objs: list[Path] = []
    for (d, de, dh, p, var) in shapes():
        o = OBJ / f"mlp_{d}_{de}_{dh}_{p}_{var}.o"
        objs.append(o)
        if force or not _newer(o, mlp_deps):
            tasks.append((o, [cc, *FLAGS, f"-DQD={d}", f"-DQDE={de}", f"-DQDH={dh}", f"-DQP={p}", f"-DQVAR={var}", "-c",
                              str(CSRC / "mlp_inst.hip"), "-o", str(o)]))
    o = OBJ / "qinco_hip.o"
    objs.append(o)
    if force or not _newer(o, headers + [CSRC / "qinco_hip.hip"]):
        # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs.  The table / filter kernels post-process every accumulator on the VALU
        # (arg-min, max, compare), which cannot read AGPRs: with AGPR accumulators the IVF filter spent 3 of 4 VALU instructions
        # on v_accvgpr_read / write (csrc/ivf_f16_kernel.hpp).  The fused-MLP instances are separate objects and keep their plan.
        tasks.append((o, [cc, *FLAGS, "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", str(CSRC / "qinco_hip.hip"), "-o", str(o)]))
    o = OBJ / "search_hip.o"
    objs.append(o)
    if force or not _newer(o, [CSRC / n for n in ("search_hip.hip", "knn_kernel.hpp", "abi_util.hpp", "mlp_args.hpp")]
                           + [PKG.parent / "include" / "qinco_hip.h"]):
        tasks.append((o, [cc, *FLAGS, "-c", str(CSRC / "search_hip.hip"), "-o", str(o)]))
    if tasks:
        if verbose:
            print(f"[qinco_amd.build] compiling {len(tasks)} object(s) for {ARCH}", file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=jobs or min(len(tasks), os.cpu_count() or 4)) as ex:
            for f in [ex.submit(_run, cmd) for _, cmd in tasks]:
                f.result()
    for stale in OBJ.glob("*.o"):
        if stale not in objs:
            stale.unlink()
    if force or tasks or not _newer(LIB, objs):
        _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)])
        if verbose:
            print(f"[qinco_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
