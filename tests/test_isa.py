"""ISA lint of the built library (CPU tier: llvm-objdump + the code objects' metadata notes, no GPU).

The fused-MLP kernels' speed AND correctness depend on properties hipcc 7.2 happens to give them:
hand-counted `s_waitcnt vmcnt(n)` around asm-issued gathers and LDS-DMA rings, an empty asm fence as the only thing that keeps hosted
loads behind their DMA (without it: one wave in thousands consumed a fragment that had not landed), "a second __shared__ object
makes hipcc emit vmcnt(0) in front of 142 ring reads" (it cost two rounds).  A toolchain bump or an innocent edit can undo any
of them without a single parity test noticing on a lucky day.  These tests read the properties back out of the binary."""
import re

import pytest

from conftest import ROOT

from qinco_amd import isa
from qinco_amd.build import shapes, small_shapes

LIB = ROOT / "qinco_amd" / "libqinco_hip.so"

FOLD, FOLD2, OCC2, SELEP, KHEAD, T16, SPLIT = 16, 32, 256, 2048, 4096, 128, 512


@pytest.fixture(scope="module")
def kernels():
    if not LIB.exists():
        pytest.skip("libqinco_hip.so not built")
    out = {}
    for co in isa.code_objects(LIB):
        for k in isa.kernels(co):
            out.setdefault(isa.short_name(k.name), k)      # (the same template instance in several translation units: same code)
    return out


def production_instances():
    """shapes.def: the first entry of a (D, De, Dh) is its production instance; plus what production launches beside it -- the KHEAD
    instances' twins (380: group sizes KHEAD does not take), the epilogue-selection form of the qinco2-S shape (6524, the default
    there), and the un-folded decode instance (332)."""
    first, prod = set(), []
    for d, de, dh, p, var in shapes():
        if var & SPLIT:
            continue
        if (d, de, dh) not in first:
            first.add((d, de, dh))
            prod.append((d, de, dh, p, var))
    all_ = set(shapes())
    for d, de, dh, p, var in list(prod):
        if var & KHEAD:
            for v in (var & ~KHEAD, var | SELEP):
                if (d, de, dh, p, v) in all_:
                    prod.append((d, de, dh, p, v))
    for p in (64, 48):
        if (128, 128, 256, p, 332) in all_:
            prod.append((128, 128, 256, p, 332))
            break
    return prod


def mlp_kernel_of(kernels, inst):
    d, de, dh, p, var = inst
    if var & T16:     # 16-row tile form: mlp16_kernel<D, De, Dh, P, ring-group form, folded, mode>
        return kernels[f"mlp16_kernel<{d},{de},{dh},{p},{2 if var & 1024 else 1},{'true' if var & FOLD else 'false'},0>"]
    return kernels[f"mlp_kernel<{d},{de},{dh},{p},{var}>"]


def expected_mfma(inst):
    """Static MFMA count of a production instance = the library's executed-FLOP model (bench.executed_flops / qinco_profile_read2)
    for one 32-row (16-row) tile with the FFN loop body counted once: (total, loop body)."""
    d, de, dh, p, var = inst
    proj = de != d
    if var & T16:                               # v_mfma_f32_16x16x4_f32: 2048 FLOP, 16 rows; FOLD only
        per_row_loop = 4 * de * dh
        head = 0 if var & FOLD else (2 * (de + d) * de + (2 * de * d if proj else 0))
        tail = 2 * de * d if proj else 0
        return (16 * (head + per_row_loop + tail)) // 2048, 16 * per_row_loop // 2048
    loop = 32 * 4 * de * dh // 4096             # v_mfma_f32_32x32x2_f32: 4096 FLOP, 32 rows
    out = 32 * 2 * de * d // 4096 if proj else 0
    if var & FOLD2:
        head = 32 * 2 * de * dh // 4096         # block 0's down-projection (its up-projection is the table P + the per-group Q)
    elif var & FOLD:
        head = 0
    else:
        head = 32 * (2 * (de + d) * de + (2 * de * d if proj else 0)) // 4096
    extra = 0
    if var & KHEAD:                             # one-hot MFMAs: Q per hidden block, U per embedding block; identity projections: xhat and x per block
        extra = dh // 32 + de // 32 + (0 if proj else 2 * (d // 32))
    return head + loop + out + extra, loop


@pytest.mark.parametrize("inst", production_instances(), ids=lambda i: "x".join(map(str, i)))
def test_production_mlp_instances(kernels, inst):
    d, de, dh, p, var = inst
    k = mlp_kernel_of(kernels, inst)
    m, st = k.meta, isa.stats(k)
    # (ii) the register plan: OCC2 = two workgroups per CU = at most 256 registers per lane (VGPR + AGPR); LDS within the CU's 160 KiB
    assert m[".vgpr_count"] <= (256 if var & OCC2 else 512), (m[".vgpr_count"], m[".agpr_count"])
    assert m[".group_segment_fixed_size"] <= 160 * 1024 // (2 if var & OCC2 else 1)
    assert m[".wavefront_size"] == 64 and not m[".uses_dynamic_stack"]
    # (iv) MFMAs per tile == the executed-FLOP model (what roofline.frac is computed from)
    want_total, want_loop = expected_mfma(inst)
    assert st["mfma"] == want_total, (st["mfma"], want_total)
    loops = [(a, b) for a, b in k.loops() if sum(isa.is_mfma(x) for x in k.text[a:b + 1]) == want_loop]
    assert loops, f"no loop with the FFN block's {want_loop} MFMAs: {[(a, b) for a, b in k.loops()][:8]}"
    a, b = min(loops, key=lambda r: r[1] - r[0])
    body = k.text[a:b + 1]
    # (i) no scratch in the FFN loop, ever; none at all on the one-workgroup-per-CU plan.  The two-workgroups-per-CU instances keep a
    # few loop-invariant address registers in scratch from the head to the epilogue (128 VGPRs): stores before the first MFMA, loads
    # behind the loop -- bounded here so that growth is a decision, not an accident.
    assert not any(x.startswith("scratch_") for x in body)
    if var & OCC2:
        # (P = 64 with its fourth ring register set: 6524 keeps 18 registers = 76 B there, 4476 11 = 48 B; P = 48: 15 / 5)
        assert m[".private_segment_fixed_size"] <= 80 and m[".vgpr_spill_count"] <= 20, (m[".private_segment_fixed_size"], m[".vgpr_spill_count"])
    else:
        assert m[".private_segment_fixed_size"] == 0 and m[".vgpr_spill_count"] == 0
    # (iii) the ring's look-ahead: inside the FFN loop every vector-memory wait is a COUNTED one -- a vmcnt(0) there drains the whole
    # LDS-DMA ring -- and nowhere between the first and the last MFMA does a vmcnt(0) sit in front of a ring read
    assert sum(isa.vmcnt_of(x) == 0 for x in body) == 0
    assert st["vmcnt0_before_ds_read"] == 0
    if p in (48, 64) and not (var & T16):
        # shared ring: one raw s_barrier + one refill DMA per group of 4 (8) fragments, the waits in front of them counted
        # (fragments of the up- and the down-projection, each section padded to a multiple of the ring depth: mlp_args.hpp stream_dims)
        section = -(-(want_loop // 4 // 2) // p) * p
        frags = 2 * section
        group = 8 if var & 1024 else 4
        assert sum(x == "s_barrier" for x in body) == frags // group
        assert sum("global_load_lds_dwordx4" in x for x in body) == frags // 4
        counted = [isa.vmcnt_of(x) for x in body if isa.vmcnt_of(x) is not None]
        assert len(counted) == frags // group and min(counted) >= p // 4 - 5


def small_instances():
    return [(d, de, dh, nt, dec) for d, de, dh, f2 in small_shapes() if f2 for nt in (1, 2, 3, 4) for dec in (False, True)]


def test_small_launch_kernels(kernels):
    """mlp_small_kernel (csrc/mlp_small_kernel.hpp): per instance that exists (small_plan decides which NT fit in LDS / registers):
    MFMA count = fragments per wave x 4 NT; no vmcnt(0) in front of a ring read; and (v) the hosted-load discipline: in every ring read
    (counted wait -> ds_read -> refill DMA) no plain global load sits between the wait and the DMA -- hosted_extra counts the
    hosted gathers as YOUNGER than that DMA, which holds only while hipcc leaves them behind it."""
    checked = 0
    for d, de, dh, nt, dec in small_instances():
        name = f"mlp_small_kernel<{d},{de},{dh},{nt},true,{'true' if dec else 'false'}>"
        if name not in kernels:
            continue
        k = kernels[name]
        st = isa.stats(k)
        nw = 8
        ndb, neb, nhb = d // 16, de // 16, dh // 16
        ndw, new, nhw = -(-ndb // nw), -(-neb // nw), -(-nhb // nw)
        proj = d != de
        f_hx, f_hq, f_up, f_down, f_out = ndb * new, neb * nhw, neb * nhw, nhb * new, (neb * ndw if proj else 0)
        frags = f_down + (f_up + f_down) + f_out + ((f_hx + f_hq) if dec else 0)      # FOLD2: block 0 = its down-projection only
        assert st["mfma"] == frags * 4 * nt, (name, st["mfma"], frags * 4 * nt)
        # (LATE decode -- wide models with many row tiles gather a step's table rows BEHIND the GEMM they meet, no registers to hold
        # them across it: one drained wait per step in front of the next GEMM's first ring read, DESIGN.md 3.2 / mlp_small_kernel.hpp)
        late = dec and ((new + nhw) * nt > 16 or ndw * nt > 8)     # (D = 768 with two row tiles: the c rows too)
        assert st["vmcnt0_before_ds_read"] <= (1 if late else 0), name
        assert isa.hoisted_loads_in_front_of_ring_dmas(k) == [], name
        assert k.meta[".vgpr_count"] <= 256                                            # 8 waves per workgroup: two per SIMD
        # scratch: none up to two row tiles; three tiles of the 256-wide decode keep 6 registers there (28 B); four tiles of the wide
        # shapes spill by the dozen -- known, and small_nt's cost model only picks NT = 4 where it still wins (profiles/r05_isa_report.txt)
        if nt <= 2:
            assert k.meta[".private_segment_fixed_size"] == 0, name
        elif nt == 3:
            assert k.meta[".private_segment_fixed_size"] <= 32, (name, k.meta[".private_segment_fixed_size"])
        checked += 1
    assert checked >= 40


def test_support_kernels_have_no_scratch(kernels):
    """Everything that is not a fused-MLP instance: no scratch, and the selection / table kernels within one wave's register file."""
    for name, k in kernels.items():
        if name.startswith(("mlp_kernel", "mlp16_kernel", "mlp_small_kernel", "mlp_split_kernel", "xproj_kernel<768,384,384>")):
            continue
        if name.startswith("dist_topk_mfma_kernel"):
            # two workgroups per CU (256 registers) with the tile's 128 distances live to the end: what the exact arg-min ROUNDS need
            # (T = 1, T > 32, degenerate groups) sits in scratch across the selection; nothing of it inside the table's MFMA loop
            assert k.meta[".private_segment_fixed_size"] <= 64 and k.meta[".vgpr_count"] <= 256, (name, k.meta[".private_segment_fixed_size"])
            mf = [i for i, x in enumerate(k.text) if isa.is_mfma(x)]
            if not name.startswith("dist_topk_mfma_kernel<32,"):      # (D = 32, the tiny test models: one unrolled feature block, spills inside it)
                assert not any(x.startswith("scratch_") for x in k.text[mf[0]:mf[-1] + 1]), name
            continue
        assert k.meta[".private_segment_fixed_size"] == 0, (name, k.meta[".private_segment_fixed_size"])
        assert k.meta[".vgpr_count"] <= 512


def test_filtered_search_table_kernel_keeps_its_plan(kernels):
    """knn_table_kernel<D, true> (the small-db search's filtered form): two waves per SIMD up to D = 128 (<= 256 registers, two
    workgroups' LDS within a CU's 160 KiB); the survivors go to the wave's LDS list by ballot + prefix count, so the only global
    atomics are the flush's (one inlined copy per 8 query rows of a block + the final one: 21), never one per (query, 32 rows) slice, and the MFMA loop holds
    no returning atomic in front of every slice -- the first version's 34 `s_waitcnt vmcnt(0)` in the loop."""
    def find(prefix):
        hits = [k for name, k in kernels.items() if name.startswith(prefix)]
        assert len(hits) == 1, (prefix, [n for n in kernels if n.startswith(prefix)])
        return hits[0]
    for D in (32, 64, 96, 128):
        k = find(f"knn_table_kernel<{D},true,2>")                  # (third argument: waves per SIMD)
        assert k.meta[".vgpr_count"] <= 256 and k.meta[".group_segment_fixed_size"] <= 80 * 1024, (D, k.meta)
        atomics = sum(x.startswith("global_atomic_add") for x in k.text)
        assert 1 <= atomics <= 24, (D, atomics)                    # (one per slice would be 16 per inlined block: 80)
        assert sum(x.startswith(("ds_write_b64", "ds_write2_b32", "ds_write2st64_b32")) for x in k.text) >= 16    # the wave-list appends
    k = find("knn_table_kernel<128,true,2>")
    mf = [i for i, x in enumerate(k.text) if isa.is_mfma(x)]
    assert len(mf) == 64 * 4                                       # first block, two pipelined blocks per trip, remainder block
    # Round 6 (every non-MFMA instruction of a wave costs its chain the issue time: scripts/ubench/mfma_valu.hip) -- the loop's diet:
    # the ring is fed by buffer loads with scalar offsets (no 64-bit VALU address per load), the filter is one burst per block with
    # the threshold test as ONE float compare per pair (16 per burst, 4 inlined bursts + the tail's) and no ballot through a VGPR
    ops = [x.split()[0] for x in k.text]
    assert sum(o == "buffer_load_dwordx4" for o in ops) >= 48 and sum(o == "global_load_dwordx4" for o in ops) <= 32, \
        (sum(o == "buffer_load_dwordx4" for o in ops), sum(o == "global_load_dwordx4" for o in ops))
    assert sum(o.startswith("v_cmp_le_f32") for o in ops) == 16 * 5      # (two bursts in the loop, the remainder block's, and the two tails)
    body = k.text[mf[64]:mf[192]]                                  # the two pipelined blocks of the loop
    steady = [x.split()[0] for x in body]
    assert sum(o in ("v_add_co_u32_e32", "v_addc_co_u32_e32", "v_add_co_u32_e64", "v_addc_co_u32_e64") for o in steady) == 0, "a VALU address per ring load is back"
    assert sum(o == "v_cmp_ne_u32_e32" for o in steady) == 0, "the ballot goes through a VGPR again"
    assert sum(o == "ds_read_b128" for o in steady) >= 16          # thresholds: 8 wide LDS reads per burst, not one narrow read per slice
    # the two-role form (opt-in): eight waves, one workgroup per CU by its LDS, the MFMA role's stream holds no VALU arithmetic
    r = find("knn_table_roles_kernel<128>")
    assert r.meta[".vgpr_count"] <= 256 and r.meta[".private_segment_fixed_size"] == 0 and r.meta[".group_segment_fixed_size"] > 80 * 1024
    assert sum(isa.is_mfma(x) for x in r.text) == 64 * 4
    k = find("knn_table_kernel<128,false")
    assert sum(x.startswith("global_atomic") for x in k.text) == 0


def test_pair_selection_runs_on_min_max_not_on_compare_and_select(kernels):
    """select.hpp pair_top_t: the 63 + 32 comparators of the bucket-minimum sort and the 63 + 32 of the 32-key sort are
    v_min / v_max pairs -- f32 for the threshold, f64 for the (distance, index) keys packed into a double's mantissa -- and the
    compaction pass is five VALU instructions and one LDS write per distance.  Read back from the large-launch table kernel."""
    k = kernels["dist_topk_mfma_kernel<128,8>"]
    ops = [x.split()[0] for x in k.text]
    assert sum(o == "v_min_f64" for o in ops) >= 111 and sum(o == "v_max_f64" for o in ops) >= 111
    assert sum(o.startswith("v_cmp_le_f32") for o in ops) == 128             # one threshold test per distance
    assert sum(o == "v_permlane32_swap_b32_e32" for o in ops) <= 64
    mf = [i for i, x in enumerate(k.text) if isa.is_mfma(x)]
    assert len(mf) == 128                                                    # feature-block loop body: 4 q x 8 codeword blocks x 4


def test_lint_sees_a_drained_ring():
    """The lint's own test: a kernel body in which hipcc ordered the ring reads behind the DMAs in flight (the round-3/4 SELEP build:
    `s_waitcnt vmcnt(0)` in front of ring ds_reads) is reported; the counted form is not."""
    good = ["v_mfma_f32_32x32x2_f32 a[0:15], v1, v2, a[0:15]", "s_waitcnt vmcnt(9)", "s_barrier", "global_load_lds_dwordx4 v[0:1], off",
            "ds_read_b128 v[4:7], v3", "v_mfma_f32_32x32x2_f32 a[0:15], v1, v2, a[0:15]"]
    bad = good[:4] + ["s_waitcnt vmcnt(0)"] + good[4:]       # what hipcc put in front of 142 ring reads of the round-3/4 SELEP build
    mk = lambda t: isa.Kernel("k", {}, t, list(range(0, 4 * len(t), 4)), [None] * len(t))   # noqa: E731
    assert isa.stats(mk(good))["vmcnt0_before_ds_read"] == 0 and isa.stats(mk(bad))["vmcnt0_before_ds_read"] == 1
    hoisted = ["s_waitcnt vmcnt(5)", "ds_read_b128 v[4:7], v3", "global_load_dwordx4 v[8:11], v[0:1], off", "global_load_lds_dwordx4 v[0:1], off"]
    kept = ["s_waitcnt vmcnt(5)", "ds_read_b128 v[4:7], v3", "global_load_lds_dwordx4 v[0:1], off", "global_load_dwordx4 v[8:11], v[0:1], off"]
    assert isa.hoisted_loads_in_front_of_ring_dmas(mk(hoisted)) == [2] and isa.hoisted_loads_in_front_of_ring_dmas(mk(kept)) == []
    assert isa.short_name("_ZN5qinco10mlp_kernelILi128ELi384ELi384ELi48ELi1148EEEvNS_7MlpArgsE") == "mlp_kernel<128,384,384,48,1148>"
    assert isa.short_name("_ZN5qinco16mlp_small_kernelILi32ELi64ELi96ELi2ELb1ELb0EEEvNS_9SmallArgsE") == "mlp_small_kernel<32,64,96,2,true,false>"
