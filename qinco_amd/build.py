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
INST = PKG / "_instances"      # kernel instances built on demand (ensure_instance); travels to the GPU box, not into git
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]


def shapes() -> list[tuple[int, int, int, int, int]]:
    txt = (CSRC / "shapes.def").read_text()
    return [tuple(int(v) for v in m.groups())
            for m in re.finditer(r"^QINCO_SHAPE\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", txt, re.M)]


def small_shapes() -> list[tuple[int, int, int, int]]:
    """(D, De, Dh, FOLD2) of the small-launch fused-MLP kernels (csrc/small_shapes.def, csrc/mlp_small_kernel.hpp)."""
    txt = (CSRC / "small_shapes.def").read_text()
    return [tuple(int(v) for v in m.groups())
            for m in re.finditer(r"^QINCO_SMALL_SHAPE\((\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", txt, re.M)]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libqinco_hip.so")
    return exe


def _deps_of(obj: Path) -> list[Path] | None:
    """Prerequisites hipcc recorded for `obj` (-MD -MF obj.d: every file the translation unit really included)."""
    d = obj.with_suffix(".d")
    if not d.exists():
        return None
    txt = d.read_text().replace("\\\n", " ")
    body = txt.split(":", 1)[1] if ":" in txt else ""
    root = str(PKG.parent)   # (toolchain / system headers are not tracked)
    return [Path(t) for t in body.split() if t.startswith(root) and not t.endswith(":")]


def _fresh(obj: Path, cmd: list[str]) -> bool:
    """obj exists, is newer than every recorded prerequisite, and was built by the same command line."""
    deps = _deps_of(obj)
    stamp = obj.with_suffix(".cmd")
    if not obj.exists() or deps is None or not stamp.exists() or stamp.read_text() != " ".join(cmd):
        return False
    t = obj.stat().st_mtime
    return all(d.exists() and d.stat().st_mtime <= t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)


def _compile(obj: Path, cmd: list[str]) -> None:
    _run([*cmd, "-MD", "-MF", str(obj.with_suffix(".d"))])
    obj.with_suffix(".cmd").write_text(" ".join(cmd))


def instance_cmd(cc: str, shape: tuple, out: Path, extra: tuple = ()) -> list[str]:
    d, de, dh, p, var = shape
    return [cc, *FLAGS, *extra, f"-DQD={d}", f"-DQDE={de}", f"-DQDH={dh}", f"-DQP={p}", f"-DQVAR={var}", "-c",
            str(CSRC / "mlp_inst.hip"), "-o", str(out)]


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> Path:
    """Incremental: an object is rebuilt when its command line changed or any file it included (hipcc -MD depfile) is newer."""
    OBJ.mkdir(exist_ok=True)
    cc = hipcc()
    tasks: list[tuple[Path, list[str]]] = []
    objs: list[Path] = []

    def want(o: Path, cmd: list[str]):
        objs.append(o)
        if force or not _fresh(o, cmd):
            tasks.append((o, cmd))

    for shape in shapes():
        o = OBJ / ("mlp_" + "_".join(map(str, shape)) + ".o")
        want(o, instance_cmd(cc, shape, o))
    for d, de, dh, f2 in small_shapes():
        o = OBJ / f"small_{d}_{de}_{dh}_{f2}.o"
        want(o, [cc, *FLAGS, f"-DQD={d}", f"-DQDE={de}", f"-DQDH={dh}", f"-DQF2={f2}", "-c", str(CSRC / "mlp_small_inst.hip"), "-o", str(o)])
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs.  The table / filter kernels post-process every accumulator on the VALU
    # (arg-min, max, compare), which cannot read AGPRs: with AGPR accumulators the IVF filter spent 3 of 4 VALU instructions
    # on v_accvgpr_read / write (csrc/ivf_f16_kernel.hpp).  The fused-MLP instances are separate objects and keep their plan.
    o = OBJ / "qinco_hip.o"
    want(o, [cc, *FLAGS, "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", str(CSRC / "qinco_hip.hip"), "-o", str(o)])
    for name in ("search_hip", "comm_hip"):
        o = OBJ / f"{name}.o"
        want(o, [cc, *FLAGS, "-c", str(CSRC / f"{name}.hip"), "-o", str(o)])
    if tasks:
        if verbose:
            print(f"[qinco_amd.build] compiling {len(tasks)} object(s) for {ARCH}", file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=jobs or min(len(tasks), os.cpu_count() or 4)) as ex:
            for f in [ex.submit(_compile, o, cmd) for o, cmd in tasks]:
                f.result()
    for stale in [*OBJ.glob("mlp_*.o"), *OBJ.glob("small_*.o")]:
        if stale not in objs:
            for suf in (".o", ".d", ".cmd"):
                stale.with_suffix(suf).unlink(missing_ok=True)
    if force or tasks or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs), "-ldl"])
        if verbose:
            print(f"[qinco_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


def instance_plan(Dp: int, Dep: int, Dhp: int) -> tuple[int, int]:
    """(P, VAR) of the kernel form that serves a padded shape (csrc/shapes.def explains the VAR bits): the 32-row kernel with the
    folded head -- two workgroups per CU with KHEAD on the short shapes -- while its activations fit the register file (z = De / 2 VGPRs,
    y = Dh / 2 AGPRs per lane: De <= 384, Dh <= 512, De + Dh <= 768; (256, 512) compiles to 236 VGPRs + 256 AGPRs, no
    scratch), else the 16-row tile kernel."""
    if Dp > 1024:
        raise NotImplementedError(f"no kernel form for D={Dp}: the per-group pre-GEMM keeps a group's D / 32 input blocks in registers "
                                  "(D <= 1024; the reference's widest dataset is 768)")
    if Dep <= 384 and Dhp <= 512 and Dep + Dhp <= 768:
        return (48, 4476) if (Dep <= 128 and Dhp <= 256) else (48, 124)     # (short shapes: two workgroups per CU + KHEAD)
    if Dep <= 768 and max(Dep, Dhp) <= 1024:
        return (48, 1236)     # 16-row tile form, ring groups of 8, folded head
    raise NotImplementedError(f"no kernel form for De={Dep}, Dh={Dhp}: the 16-row tile kernel holds De/4 + max(De, Dh)/4 registers "
                              "of activations per lane (De <= 768, Dh <= 1024)")


def ensure_instance(D: int, De: int, Dh: int, verbose: bool = False) -> Path | None:
    """Make sure the loaded libqinco_hip.so has a fused-MLP kernel instance for a model geometry, building one on demand.

    The library is asked for the padded shape (qinco_padded_shape) and whether it is served already (shapes.def lists every
    preset of the reference; qinco_shape_supported).  If not, ONE translation unit of csrc/mlp_inst.hip is compiled for that
    shape into qinco_amd/_instances/inst_<shape>.so (about a minute; cached by shape, rebuilt when the kernel sources change) and
    registered with qinco_load_instance.  Returns the module's path, or None when nothing had to be loaded.
    Needs hipcc on the machine the first time a shape is seen; raises RuntimeError with that message otherwise."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    out3 = (C.c_int32 * 3)()
    _lib.check(lib.qinco_padded_shape(D, De, Dh, out3))
    if lib.qinco_shape_supported(D, De, Dh):
        return None
    Dp, Dep, Dhp = (int(v) for v in out3)
    P, var = instance_plan(Dp, Dep, Dhp)
    inst_dir = INST
    try:
        inst_dir.mkdir(exist_ok=True)
        probe = inst_dir / ".writable"
        probe.touch()
        probe.unlink()
    except OSError:       # a read-only installation: the user's cache directory instead of the package directory
        inst_dir = Path(os.environ.get("XDG_CACHE_HOME", Path.home() / ".cache")) / "qinco_amd" / "instances"
        inst_dir.mkdir(parents=True, exist_ok=True)
    # the encode instance, and -- for the two-workgroups-per-CU forms -- the un-folded twin decode runs on (DESIGN.md 3.1:
    # faster there, slower on the wide shapes); compiled side by side
    # (KHEAD needs its twin without the bit for the group sizes it does not take; the un-folded decode twin pays on the OCC2 shapes only)
    variants = [var] + ([var & ~4096] if var & 4096 else []) + ([var & ~(16 | 32 | 4096)] if (var & 16) and (var & 256) else [])
    jobs = []
    for v in variants:
        so = inst_dir / f"inst_{Dp}_{Dep}_{Dhp}_{P}_{v}.so"
        cmd = [c for c in instance_cmd("hipcc", (Dp, Dep, Dhp, P, v), so, extra=("-DQINCO_INSTANCE_MODULE", "-shared")) if c != "-c"]
        jobs.append((so, cmd))
    # One process per GPU: eight ranks meeting a new geometry at the same moment must not write the same files at once, and none
    # may load a module another rank is still writing -- the freshness check runs UNDER the lock, and a module is compiled under a
    # temporary name and renamed into place (a .so that exists is a complete one).
    import fcntl
    with open(inst_dir / f".lock_{Dp}_{Dep}_{Dhp}", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        todo = [(so, cmd) for so, cmd in jobs if not _fresh(so, cmd[1:])]
        if todo:
            cc = hipcc()          # raises when there is no compiler: a new geometry cannot be served on this machine
            if verbose:
                print(f"[qinco_amd.build] compiling {len(todo)} kernel instance(s) for ({Dp}, {Dep}, {Dhp}) [model ({D}, {De}, {Dh})]",
                      file=sys.stderr)

            def compile_one(job):
                so, cmd = job
                tmp = so.with_suffix(f".tmp{os.getpid()}.so")
                real = [cc, *[str(tmp) if c == str(so) else c for c in cmd[1:]]]
                _run([*real, "-MD", "-MF", str(so.with_suffix(".d"))])
                os.replace(tmp, so)
                so.with_suffix(".cmd").write_text(" ".join(cmd[1:]))
            with cf.ThreadPoolExecutor(max_workers=len(todo)) as ex:
                list(ex.map(compile_one, todo))
    for so, _ in jobs:                       # the encode instance first: the first entry of a shape is its production instance
        _lib.check(lib.qinco_load_instance(str(so).encode()))
    return jobs[0][0]


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
