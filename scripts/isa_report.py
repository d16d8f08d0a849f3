#!/usr/bin/env python
"""Per-kernel ISA facts of libqinco_hip.so (or a kernel-instance module): registers, scratch, LDS from the code objects' metadata
notes, instruction counts from the disassembly.  No GPU needed (llvm-objdump / llvm-readelf of the ROCm toolchain).

    python scripts/isa_report.py [path.so] [--kernel SUBSTRING] [--dump DIR]

The CPU test tier runs the same extraction (tests/test_isa.py -> qinco_amd/isa.py) and asserts the properties the kernels'
schedules rely on."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from qinco_amd import isa  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("so", nargs="?", default=str(ROOT / "qinco_amd" / "libqinco_hip.so"))
    ap.add_argument("--kernel", help="only kernels whose mangled name contains this")
    ap.add_argument("--dump", help="write each listed kernel's disassembly to DIR/<name>.s")
    a = ap.parse_args()
    so, sub, dump = Path(a.so), a.kernel, (Path(a.dump) if a.dump else None)
    rows = []
    for co in isa.code_objects(so):
        for k in isa.kernels(co):
            if sub and sub not in k.name:
                continue
            st = isa.stats(k)
            st["hoisted"] = len(isa.hoisted_loads_in_front_of_ring_dmas(k)) if "mlp_small" in k.name else 0
            rows.append((k.name, k.meta, st))
            if dump:
                dump.mkdir(parents=True, exist_ok=True)
                (dump / (isa.short_name(k.name).replace("<", "_").replace(">", "").replace(",", "_").replace(" ", "") + ".s")).write_text("\n".join(k.text))
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'scr':>4s} {'spill':>5s} {'lds':>7s} {'insts':>7s} {'mfma':>6s} {'vm0':>4s} {'vm0@ds':>6s} {'loop':>6s} {'l.mfma':>6s} {'l.vm0':>5s} {'l.scr':>5s} {'hoist':>5s}")
    for name, m, st in sorted(rows):
        print(f"{isa.short_name(name)[:70]:70s} {m['.vgpr_count']:5d} {m['.agpr_count']:5d} {m['.private_segment_fixed_size']:4d} "
              f"{m['.vgpr_spill_count']:5d} {m['.group_segment_fixed_size']:7d} {st['insts']:7d} {st['mfma']:6d} {st['vmcnt0_in_mfma_span']:4d} "
              f"{st['vmcnt0_before_ds_read']:6d} {st['loop_insts']:6d} {st['loop_mfma']:6d} {st['loop_vmcnt0']:5d} {st['loop_scratch']:5d} {st['hoisted']:5d}")


if __name__ == "__main__":
    main()
