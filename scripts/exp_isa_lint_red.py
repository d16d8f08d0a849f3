#!/usr/bin/env python
"""Show that the ISA lint (tests/test_isa.py) turns RED on the regression that cost rounds 3-4: the epilogue selection's keys in a
second __shared__ object (QINCO_EXP_SELEP_SECOND_LDS) instead of the tail of the weight ring's array -- hipcc's alias analysis then
orders every LDS read of the ring behind the LDS-DMAs in flight.  Compiles instance (128, 128, 256, 48, VAR) both ways into /tmp
(no GPU) and prints the lint's counters.

    python scripts/exp_isa_lint_red.py [VAR=6524]"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qinco_amd import build, isa  # noqa: E402


def main():
    var = int(sys.argv[1]) if len(sys.argv) > 1 else 6524
    shape = (128, 128, 256, 48, var)
    with tempfile.TemporaryDirectory() as d:
        for label, extra in (("production layout (keys in the tail of the ring's array)", ()),
                             ("second __shared__ object (rounds 3-4)", ("-DQINCO_EXPERIMENT", "-DQINCO_EXP_SELEP_SECOND_LDS"))):
            obj = Path(d) / f"inst_{len(extra)}.o"
            subprocess.run(build.instance_cmd(build.hipcc(), shape, obj, extra=extra), check=True)
            k = [k for co in isa.code_objects(obj) for k in isa.kernels(co) if isa.short_name(k.name) == f"mlp_kernel<{','.join(map(str, shape))}>"][0]
            st = isa.stats(k)
            loops = [(a, b) for a, b in k.loops() if sum(isa.is_mfma(x) for x in k.text[a:b + 1]) == 1024]
            a, b = min(loops, key=lambda r: r[1] - r[0])
            body = k.text[a:b + 1]
            red = st["vmcnt0_before_ds_read"] > 0 or any(isa.vmcnt_of(x) == 0 for x in body)
            print(f"{label}:\n    vmcnt(0) in front of a ring read (first..last MFMA): {st['vmcnt0_before_ds_read']}, vmcnt(0) inside the FFN loop: "
                  f"{sum(isa.vmcnt_of(x) == 0 for x in body)}, counted waits in the loop: {sum((isa.vmcnt_of(x) or 0) > 0 for x in body)}, "
                  f"scratch {k.meta['.private_segment_fixed_size']} B  ->  lint {'RED' if red else 'green'}")


if __name__ == "__main__":
    main()
