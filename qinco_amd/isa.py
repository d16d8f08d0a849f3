"""Read the gfx950 code objects back out of a built library: kernel metadata (registers, scratch, LDS) and disassembly.

Build-time tooling (tests/test_isa.py, scripts/isa_report.py): the fused-MLP kernels' speed AND correctness lean on properties
the compiler happens to give them -- no scratch, hand-counted `s_waitcnt vmcnt(n)` around LDS-DMA rings that hipcc must not
turn into `vmcnt(0)`, an exact number of MFMAs per tile.  Nothing here runs on a GPU; it needs llvm-objdump / llvm-readelf /
llvm-objcopy of the ROCm toolchain (/opt/rocm/lib/llvm/bin)."""
from __future__ import annotations

import re
import shutil
import struct
import subprocess
import tempfile
from dataclasses import dataclass, field
from pathlib import Path

LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def tool(name: str) -> str:
    p = LLVM_BIN / name
    if p.exists():
        return str(p)
    w = shutil.which(name)
    if w is None:
        raise RuntimeError(f"{name} not found (ROCm LLVM tools expected under {LLVM_BIN})")
    return w


def code_objects(so_path) -> list[bytes]:
    """Every gfx950 ELF embedded in a host shared object: hipcc leaves one uncompressed offload bundle per translation unit in the
    `.hip_fatbin` section (magic, entry count, then {offset, size, triple-length, triple} per entry)."""
    with tempfile.TemporaryDirectory() as d:
        fat = Path(d) / "fat.bin"
        subprocess.run([tool("llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(so_path), str(Path(d) / "ignored.so")], check=True,
                       capture_output=True)
        data = fat.read_bytes()
    out = []
    for m in re.finditer(re.escape(_BUNDLE_MAGIC), data):
        off = m.start()
        p = off + len(_BUNDLE_MAGIC)
        (n,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(n):
            o, s, ln = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ln].decode()
            p += ln
            if "gfx950" in triple and s:
                out.append(data[off + o: off + o + s])
    return out


@dataclass
class Kernel:
    name: str                      # mangled symbol
    meta: dict                     # the kernel's entry of the AMDGPU metadata note
    text: list = field(default_factory=list)     # disassembly, one instruction per entry ("mnemonic operands")
    addr: list = field(default_factory=list)     # byte offset of each instruction from the kernel's first one
    target: list = field(default_factory=list)   # branches: byte offset of the destination (None for everything else)

    def loops(self) -> list:
        """[(first, last)] instruction index ranges of the backward branches (a loop body with its closing branch), outermost first."""
        at = {a: i for i, a in enumerate(self.addr)}
        out = [(at[t], i) for i, t in enumerate(self.target) if t is not None and t <= self.addr[i] and t in at]
        return sorted(out, key=lambda r: r[0] - r[1])


def _metadata(elf_path: str) -> list[dict]:
    import yaml
    txt = subprocess.run([tool("llvm-readelf"), "--notes", elf_path], check=True, capture_output=True, text=True).stdout
    a = txt.index("---")
    b = txt.index("...", a) if "..." in txt[a:] else len(txt)
    doc = yaml.safe_load(txt[a + 3:b])
    return doc["amdhsa.kernels"]


def kernels(elf: bytes) -> list[Kernel]:
    with tempfile.TemporaryDirectory() as d:
        p = Path(d) / "co.elf"
        p.write_bytes(elf)
        metas = {m[".name"]: m for m in _metadata(str(p))}
        dis = subprocess.run([tool("llvm-objdump"), "-d", "--no-show-raw-insn", str(p)], check=True, capture_output=True, text=True).stdout
    out, cur, start = {}, None, 0
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:$", line.strip())
        if m:
            cur, start = m.group(2), int(m.group(1), 16)
            if cur in metas:
                out[cur] = Kernel(cur, metas[cur])
            continue
        if cur in out and "//" in line:
            ins, _, com = line.partition("//")
            ins = re.sub(r"\s+", " ", ins.strip())
            am = re.match(r"\s*([0-9A-Fa-f]+):", com)
            if not ins or not am:
                continue
            k = out[cur]
            k.text.append(ins)
            k.addr.append(int(am.group(1), 16) - start)
            tm = re.search(r"<[^>+]+\+0x([0-9a-f]+)>\s*$", com) if ins.startswith(("s_cbranch", "s_branch")) else None
            k.target.append(int(tm.group(1), 16) if tm else (0 if ins.startswith(("s_cbranch", "s_branch")) and com.rstrip().endswith(f"<{cur}>") else None))
    return list(out.values())


def short_name(mangled: str) -> str:
    """_ZN5qinco10mlp_kernelILi128ELi384E...EEvNS_7MlpArgsE -> mlp_kernel<128,384,...> (enough of a demangler for this library's
    kernels: integer and bool template arguments)."""
    m = re.match(r"_ZN5qinco(\d+)", mangled)
    if not m:
        return mangled
    n = int(m.group(1))
    start = m.end()
    base = mangled[start:start + n]
    rest = mangled[start + n:]
    args = []
    if rest.startswith("I"):
        for t, neg, v in re.findall(r"L([ib])(n?)(\d+)E", rest.split("EEv")[0] + "E"):
            args.append(("-" if neg else "") + v if t == "i" else ("true" if v == "1" else "false"))
    return base + ("<" + ",".join(args) + ">" if args else "")


def find(kernel_list, base: str, *targs) -> Kernel:
    want = base + "<" + ",".join(str(t).lower() if isinstance(t, bool) else str(t) for t in targs) + ">"
    hits = [k for k in kernel_list if short_name(k.name) == want]
    if len(hits) != 1:
        raise KeyError(f"{want}: {len(hits)} kernels match")
    return hits[0]


_WAIT = re.compile(r"^s_waitcnt\b(.*)$")


def is_mfma(ins: str) -> bool:
    return ins.startswith("v_mfma_")


def vmcnt_of(ins: str):
    """The vmcnt field of an s_waitcnt, or None when the instruction does not wait on it."""
    m = _WAIT.match(ins)
    if not m:
        return None
    v = re.search(r"vmcnt\((\d+)\)", m.group(1))
    return int(v.group(1)) if v else None


def stats(k: Kernel) -> dict:
    """Counts over a kernel's instruction stream.  `mfma span` = first to last MFMA.  vmcnt0_before_ds_read: `s_waitcnt vmcnt(0)`
    inside the span whose next memory / matrix instruction is an LDS read -- the signature of hipcc ordering ring reads behind every
    LDS-DMA in flight (it cost two rounds)."""
    t = k.text
    mf = [i for i, ins in enumerate(t) if is_mfma(ins)]
    lo, hi = (mf[0], mf[-1]) if mf else (0, -1)
    vm0 = vm0_ds = 0
    for i in range(lo, hi + 1):
        if vmcnt_of(t[i]) == 0:
            vm0 += 1
            for j in range(i + 1, min(hi + 1, i + 40)):
                if t[j].startswith("ds_read") or t[j].startswith("ds_load"):
                    vm0_ds += 1
                    break
                if is_mfma(t[j]) or t[j].startswith(("global_", "buffer_", "flat_", "scratch_", "ds_write", "ds_store", "s_barrier")):
                    break
    # the hot loop: the backward-branch region with the most MFMAs (the FFN-block loop of the fused-MLP kernels)
    loop = max(((a, b) for a, b in k.loops()), key=lambda r: sum(is_mfma(x) for x in t[r[0]:r[1] + 1]), default=None)
    body = t[loop[0]:loop[1] + 1] if loop else []
    return {"insts": len(t), "mfma": len(mf), "vmcnt0_in_mfma_span": vm0, "vmcnt0_before_ds_read": vm0_ds,
            "scratch_insts": sum(ins.startswith("scratch_") for ins in t),
            "lds_dma": sum("global_load_lds" in ins for ins in t),
            "loop_insts": len(body), "loop_mfma": sum(is_mfma(x) for x in body),
            "loop_vmcnt0": sum(vmcnt_of(x) == 0 for x in body), "loop_scratch": sum(x.startswith("scratch_") for x in body),
            "loop_lgkmcnt0": sum(lgkmcnt_of(x) == 0 for x in body)}


def lgkmcnt_of(ins: str):
    m = _WAIT.match(ins)
    if not m:
        return None
    v = re.search(r"lgkmcnt\((\d+)\)", m.group(1))
    return int(v.group(1)) if v else None


def hoisted_loads_in_front_of_ring_dmas(k: Kernel) -> list:
    """Indices of plain vector-memory loads that sit between a counted `s_waitcnt vmcnt(N)` and the LDS-DMA (global_load_lds) issued
    behind it.  In the small-launch kernel a ring read is  wait -> ds_read -> DMA of the refill  and every other load of the wave is
    HOSTED behind that DMA: the wait counts of later reads are raised by exactly the number of such loads (hosted_extra), which is only
    right while they stay younger than the DMA.  HISTORY.md round 4: without the fence hipcc hoisted them and one wave in thousands
    consumed a fragment that had not landed."""
    bad, t = [], k.text
    for i, ins in enumerate(t):
        if "global_load_lds" not in ins:
            continue
        seen, ring_read = [], False
        for j in range(i - 1, max(i - 60, -1), -1):
            if "global_load_lds" in t[j]:
                break                                # the ring prologue's back-to-back DMAs: no read in between
            if t[j].startswith(("ds_read", "ds_load")):
                ring_read = True
            if vmcnt_of(t[j]) is not None:
                if ring_read:
                    bad += seen                      # a ring read: wait -> ds_read -> this DMA, with plain loads in between
                break
            if t[j].startswith(("global_load", "buffer_load", "flat_load")):
                seen.append(j)
    return bad

