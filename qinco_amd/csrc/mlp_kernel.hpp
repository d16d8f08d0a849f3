// Fused QINCo codeword-MLP kernel for gfx950 (MI355X / CDNA4).
//
// Computes, for R rows (row = one candidate codeword of one beam of one vector),
//     f(c, xhat) = out_proj( FFN_L(...FFN_1( concat(in_proj(c), xhat) )) ) + coeff*c
// followed by   cand = f(c, xhat) + xhat   and   dist = |x|^2 + |cand|^2 - 2 x.cand
// i.e. the body of QINCoInferenceStep.forward (reference qinco/model/qinco_inference.py:31-40,
// = QINCoStep.forward qinco/model/qinco_base.py:262-280, QConcat :60-64, QBlockFFN :93-97)
// plus the candidate / distance epilogue of QINCoInferenceStepEncoder.forward (:190-199).
//
// MI355X design (not a translation of the reference's ATen op sequence; measurements in DESIGN.md 3.1):
//  * The GEMMs are evaluated TRANSPOSED: Out^T[feat x rows] = W[feat_out x feat_in] . In^T[feat_in x rows]
//    on v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain numerics).  A wave owns 32 rows and ALL features.
//    In the 32x32 C/D layout lane l holds row (l&31) and features {(r&3)+8(r>>2)+4(l>>5)}; that is
//    exactly a legal B-operand layout for the next layer once the K-order of the weights is permuted
//    the same way (the host packs the weights).  Activations therefore never leave the register file:
//    no LDS round trip and no HBM traffic between the 2L+3 GEMMs.
//  * Weights are a single sequential stream of 1 KiB "fragments" (64 lanes x float4 = the A operands
//    of 4 consecutive MFMAs) packed on the host in exactly the order the kernel consumes them, fetched through
//    an LDS-DMA ring (global_load_lds_dwordx4) that the 4 waves of a workgroup share.
//  * 1 wave per SIMD: z (De/2 VGPRs) + y/h (Dh/2 AGPRs) + two chain accumulators; chain epilogues (ReLU,
//    residual add) run one chain late, under MFMAs that do not depend on them.
//
// Template: D, De, Dh model geometry; P ring depth in fragments; VAR feature bits (shapes.def):
//    4 LDSR   weight stream through an LDS-DMA ring (else: P-deep register ring of plain global loads)
//    8 PINNED explicit VGPR/AGPR plan + one-chain-late epilogues (else: eager epilogues, compiler-placed registers)
//   64 SHR    one ring per workgroup, wave w DMAs the fragments = w (mod 4), raw s_barrier every 4 fragments
//   16 FOLD   the part of the MLP head that does not depend on the row is not recomputed per row:
//             z_k = in_proj(c_k) and W_cat[:, :De] z_k + b depend only on the codeword k (a (K, De) table T built at
//             qinco_create), W_cat[:, De:] xhat depends only on the (vector, beam) group (U, one small MFMA GEMM per
//             step: xproj_kernel).  The kernel starts from z = T[cid] + U[group]: -4.9 % MFMA work at C2.
//             Same real-number result as QConcat.forward; the fp32 association differs ((b + Wz z) + Wx xhat).
//  256 OCC2   two workgroups per CU (launch bounds (256, 2): <= 256 registers per lane): one chain accumulator and eager chain
//             epilogues instead of the one-chain-late plan (two waves per SIMD fill each other's pipe drains), so that a
//             wave's prologue gathers and epilogue stores run under the other wave's MFMAs.  For short MLPs (qinco2-S, QINCo1:
//             De = 128, Dh = 256), where those phases are 24 % of a tile's time (profiles/r02_timeline.jsonl).
//   32 FOLD2  (with FOLD) the first FFN block's up-projection is linear in z = T + U as well:
//             W_up[0] z = P[cid] + Q[group] with P = W_up[0] T (table) and Q = W_up[0] U (same xproj launch), so the
//             kernel starts with y = relu(P[cid] + Q[group]) and the first down-projection: -2.9 % more at C2, -20 %
//             for the two-block qinco2-S.
// 4096 KHEAD  (OCC2 + FOLD2, round 4) the head's per-group rows are added ON THE MATRIX PIPE and the first down-projection runs
//             K-outer, so that the gathers of the head stream under its MFMAs:
//               z = T[cid] + U[g],  y = relu(P[cid] + Q[g]):  the per-codeword rows are gathered into registers as before, the
//               per-group rows -- the same bytes for every row of a group -- arrive as ONE MFMA per 32-feature block with a one-hot B
//               operand:  acc[f][row] += sum_k U[g0 + k][f] * [group(row) == g0 + k], k = 0, 1  (A: lane (f, k) holds U[g0 + k][f], one
//               coalesced dword instead of four 16-byte gathers;  fmaf(U, 1, T) rounds like T + U and adding 0 * U' is exact: same bits).
//               The first down-projection takes its fragments in (input block, q, output block) order with NEB accumulators, so
//               only ONE y block has to exist at a time: y block ib + 2 is gathered while block ib is on the matrix pipe.  Per
//               output element the products are added in the order of the ob-outer form: same bits.
//             Before: under OCC2's 128-VGPR budget hipcc drained the head's 96 gathers in 12 batches by vmcnt(0): 68 k of a
//             short-MLP tile's 228 k cycles (profiles/r03_timeline.jsonl) with nothing on the matrix pipe.
// 2048 SELEP  (identity projections, shared ring) the step's per-vector top-T in the epilogue, when MlpArgs::sel_T > 0 and a vector's
//             F * A candidates sit inside one workgroup (128 % (F A) == 0): the candidates stay in z's registers, their
//             (distance, index) keys meet in LDS, every row counts the keys below its own -- its rank in beam_select_kernel's order
//             -- and only the T winners are stored: rows straight into the next step's xhat, codes into its history.  No candidate / distance
//             write-back (512 + 4 B per row of which the selection used 1/16), no beam_select launch.  Same distances, same
//             selection: bit-identical codes.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "mlp_args.hpp"
#include "select.hpp"

namespace qinco {

#define QINCO_INL __device__ __forceinline__
#define QINCO_LAMBDA __attribute__((always_inline))

template <class F, int... Is>
QINCO_INL void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f.template operator()<Is>(), ...);
}
template <int N, class F>
QINCO_INL void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

QINCO_INL f32x16 load_block(const float* p) {
  // p already includes the half*4 lane offset and the 32-feature block offset.
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * q);
    v[4 * q + 0] = t[0];
    v[4 * q + 1] = t[1];
    v[4 * q + 2] = t[2];
    v[4 * q + 3] = t[3];
  }
  return v;
}

QINCO_INL f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  return v;
}

#define QINCO_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// relu as ONE integer VALU op (v_max_i32 0, bits): positive floats have non-negative bit patterns and keep their
// order, negative floats (and -0.0) have the sign bit set and clamp to 0.  fmaxf() would cost two v_max_f32
// here (the compiler canonicalises an MFMA result first), and inline asm would hide VALU->MFMA hazards from it.
QINCO_INL float relu1(float v) {
  int b = __builtin_bit_cast(int, v);
  b = b > 0 ? b : 0;
  return __builtin_bit_cast(float, b);
}
QINCO_INL void relu16(f32x16& v) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = relu1(v[i]);
}

// Register-class pins: empty asm statements that force a 16-register block into VGPRs / AGPRs at that point
// (clang cannot reference lambda captures from an asm operand, hence the helpers).
QINCO_INL void pin_v(f32x16& v) { asm volatile("" : "+v"(v)); }
QINCO_INL void pin4_v(f32x4& v) { asm volatile("" : "+v"(v)); }
QINCO_INL void pin_a(f32x16& v) { asm volatile("" : "+a"(v)); }

// A operand of a one-hot MFMA (KHEAD): the row of ANOTHER group rides in the same instruction and is multiplied by 0 for this wave's
// other rows -- which is only exact for finite values (0 * inf = NaN would poison the neighbouring group, i.e. another vector's rows,
// which the reference's row-by-row arithmetic never does).  v_med3_f32 against +-FLT_MAX returns every finite value unchanged and
// turns inf / NaN into +-FLT_MAX: a vector with non-finite inputs keeps garbage results of its own, its neighbours keep theirs.
QINCO_INL float one_hot_operand(float v) { return __builtin_amdgcn_fmed3f(v, -3.4028234663852886e38f, 3.4028234663852886e38f); }

// A 32-feature block of a table row requested by loads hipcc does not see as loads (KHEAD).  hipcc's wait-count pass treats every
// LDS-DMA of the weight ring as a "flat" access that may complete out of order, so any wait IT generates for a global load while
// a ring DMA is in flight is vmcnt(0): the whole ring and every other gather drained (that is what made the head's 12 batches 12
// exposed round trips).  These loads are waited for by hand (wait_block: a counted vmcnt that names the registers, so no use can
// move above it); the 16 registers are assembled from the four quads afterwards (the coalescer makes them one tuple).
struct AsmBlock {
  f32x4 q[4];
};
template <int OFF>
QINCO_INL void asm_load_block(AsmBlock& b, const float* p) {   // bytes OFF + {0, 32, 64, 96} from p (p includes the half * 4 lane offset)
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(b.q[0]) : "v"(p), "n"(OFF) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(b.q[1]) : "v"(p), "n"(OFF + 32) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(b.q[2]) : "v"(p), "n"(OFF + 64) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(b.q[3]) : "v"(p), "n"(OFF + 96) : "memory");
}
template <int N>
QINCO_INL f32x16 wait_block(AsmBlock& b) {   // "at most N vector-memory operations younger than this block's loads": then they have landed
  asm volatile("s_waitcnt vmcnt(%4)" : "+a"(b.q[0]), "+a"(b.q[1]), "+a"(b.q[2]), "+a"(b.q[3]) : "n"(N) : "memory");
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = b.q[q][e];
  return v;
}

#ifndef QINCO_KHEAD_LA
#define QINCO_KHEAD_LA 2
#endif
template <int D, int DE, int DH, int P, int VAR>
__global__ void __launch_bounds__(256, (VAR & 256) ? 2 : 1) mlp_kernel(MlpArgs a) {
  constexpr bool FOLD = (VAR & 16) != 0;
  constexpr bool FOLD2 = (VAR & 32) != 0;
  constexpr bool OCC2 = (VAR & 256) != 0;
  static_assert(!FOLD2 || (FOLD && (VAR & 8)), "FOLD2 needs FOLD and the pinned plan");
  constexpr StreamDims SL = stream_dims(D, DE, DH, P, FOLD, FOLD2);
  constexpr int NDB = SL.NDB, NEB = SL.NEB, NHB = SL.NHB;
  constexpr bool PROJ = SL.PROJ;
  constexpr int NYB = NHB > NEB ? NHB : NEB;
  constexpr bool LDSR = (VAR & 4) != 0;
  constexpr bool PINNED = (VAR & 8) != 0;
  constexpr bool SHR = (VAR & 64) != 0;
  constexpr bool G8 = (VAR & 1024) != 0;   // shared ring with a barrier / refill every 8 fragments instead of every 4 (A/B variant)
  constexpr bool SELEP = (VAR & 2048) != 0;
  constexpr bool KHEAD = (VAR & 4096) != 0;
  static_assert((VAR & ~(4 | 8 | 16 | 32 | 64 | 256 | 1024 | 2048 | 4096)) == 0, "unknown VAR bits");
  static_assert(!KHEAD || (OCC2 && FOLD2 && SHR && !G8), "KHEAD is a form of the two-workgroups-per-CU folded kernel");
  static_assert(!SELEP || (SHR && D == DE), "SELEP: every wave reaches the epilogue's barriers, candidates live in z's registers");
  static_assert(!G8 || (SHR && P % 24 == 0 && P / 4 >= 7), "G8 is a form of the shared ring");
  static_assert(!OCC2 || (VAR & 8), "OCC2 is a form of the pinned plan");
  // shared ring: 4 issuers, and the register rotation must restart in phase at every section (sections are multiples of P fragments):
  // P a multiple of 12 with 3 register sets, or (round 5) a multiple of 16 with 4 -- P = 64 divides the 128-fragment sections of the
  // short shapes (De = 128, Dh = 256), whose stream then carries no padding (P = 48 pads each to 144: 11 % dead ring groups)
  static_assert(!SHR || (LDSR && (P % 12 == 0 || P % 16 == 0) && P / 4 >= 5), "shared ring: P multiple of 12 (3 register sets) or of 16 (4)");
  static_assert(!G8 || P % 12 == 0, "");
  static_assert(!LDSR || SHR || (P % 3 == 0 && P >= 6 && P <= 39), "per-wave LDS rings: 4 x P KiB must fit 160 KiB");

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long tile = (long)blockIdx.x * 4 + wave;
  if constexpr (!SHR) {
    if (tile * 32 >= a.R) return;  // wave-uniform; waves are independent (no barriers)
  }                                // SHR: every wave runs (barriers); rows past R are clamped and never stored
  long row = tile * 32 + j;
  const bool valid = row < a.R;
  if (!valid) row = a.R - 1;
#ifdef QINCO_TIMELINE
  auto stamp = [&](int i) QINCO_LAMBDA {
    if (a.timeline && lane == 0) a.timeline[tile * 8 + i] = __builtin_readcyclecounter();
  };
  stamp(0);
  if (a.timeline && lane == 0 && !SELEP) {   // where the tile ran: XCC id << 32 | HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SE/SH [15:12])
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.timeline[tile * 8 + 7] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
#else
  auto stamp = [](int) QINCO_LAMBDA {};
#endif
#if defined(QINCO_EXPERIMENT) && defined(QINCO_STAGGER_EXP)
  // experiment builds: the second residents of the CUs in the launch's first round (workgroups 256 .. 511 share a CU with 0 .. 255,
  // scripts/ubench/residency.hip) start QINCO_STAGGER_EXP x 8128 cycles late
  if constexpr (OCC2) {
    if (blockIdx.x >= 256 && blockIdx.x < 512)
      for (int i = 0; i < QINCO_STAGGER_EXP; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const long g = row / a.A;
  int cid = a.cand_ids ? a.cand_ids[row] : (int)(row - g * a.A);
  // KHEAD: the candidate id has ARRIVED before the ring prologue's DMAs go out.  hipcc otherwise waits for it at its first use, behind
  // the DMAs -- and a wait it generates with ring DMAs in flight is vmcnt(0): the prologue's round trip in front of the head's burst
  if constexpr ((VAR & 4096) != 0) asm volatile("" : "+v"(cid)::"memory");
#if defined(QINCO_EXPERIMENT) && defined(QINCO_EXP_FEW_LINES)
  // experiment builds (WRONG results, timing only): every row gathers the same few table rows and stores to the same few candidate
  // rows -- a gather / store instruction then touches 1-4 cache lines instead of 64: what the vector L1's look-up rate costs
  cid &= (QINCO_EXP_FEW_LINES & 1) ? 1 : 0xffff;
#endif
  const float* cptr = a.codebook + (long)cid * D + half * 4;
  const float* xhptr = a.xhat + g * D + half * 4;

  // ---- weight stream ---------------------------------------------------------------------------------
  // take<T>() = "fragment T of the current section" (sections are padded to multiples of P).
  //  * register ring (!LDSR): P fragments prefetched into VGPRs by plain global loads (round-1 first kernel: 9.7 % of
  //    wave time parked in s_waitcnt -- every L2 miss of the lock-stepped stream stalls all CUs of an XCD);
  //  * LDS-DMA ring (LDSR): global_load_lds_dwordx4 fills LDS far ahead at no VGPR cost, a 3-set register ring is read
  //    from LDS two fragments ahead (ds_read_b128, lane-linear = conflict free).  hipcc does NOT order a ds_read
  //    behind an in-flight LDS-DMA, so that side is a counted vmcnt by hand; the LDS reads are ordinary loads, so
  //    hipcc counts lgkmcnt and no register is ever "in flight" behind its back.
  //      - per-wave rings (!SHR): no barriers, but every wave issues one DMA per fragment;
  //      - shared ring (SHR): issuing a global_load_lds costs the issuing wave ~20 cycles of matrix-pipe idle wherever
  //        it is placed (scripts/ubench/frag_loop.hip: 278 -> 264 cycles per fragment) and the four waves fetch the
  //        same bytes: wave w DMAs only the fragments = w (mod 4) and a raw s_barrier every 4 fragments -- after the
  //        issuers' counted vmcnt -- publishes the landed group.
  const f32x4* wp = a.wstream + lane;
  constexpr int NRING = LDSR ? ((SHR && P % 12 != 0) ? 4 : 3) : P;   // register sets of the ring (shared ring with P % 16 == 0: four)
  f32x4 ring[NRING];
  // (SELEP's keys and indices live in the tail of this array: a SECOND __shared__ object in the kernel makes hipcc order the ring's
  // LDS reads behind the LDS-DMAs in flight -- s_waitcnt vmcnt(0) in front of 142 of them in the SELEP instance of rounds 3-4)
  constexpr int RING4 = LDSR ? (SHR ? 1 : 4) * P * 64 : 1;
  __shared__ f32x4 lds_ring[RING4 + (((VAR & 2048) != 0) ? 96 : 0)];
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  [[maybe_unused]] f32x4* myring = lds_ring + ((LDSR && !SHR) ? wave_u * P * 64 : 0);
  // DMA of fragment T (relative to the current section origin wp) into its ring slot.  SHR: T is a multiple of 4
  // and wave w fetches fragment T + w (branch-free: the wave offset is folded into both base pointers).
  [[maybe_unused]] const int wofs = SHR ? wave_u * 64 : 0;
  auto dma = [&]<int T>() QINCO_LAMBDA {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + wofs + T * 64),
                                     (__attribute__((address_space(3))) void*)(myring + wofs + (T % P) * 64), 16, 0, 0);
  };
  // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields = "no wait"): the builtin keeps the wait visible to hipcc's
  // own counter bookkeeping; the empty asm fences pin the LDS reads / DMAs of the ring on their side of the wait.
  auto wait_vm = [&]<int N>() QINCO_LAMBDA {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
  };
  f32x16 z[NEB];
  f32x16 y[NYB];
  // KHEAD: state that lives from the head into the first down-projection
  constexpr int LA = QINCO_KHEAD_LA < SL.NHB ? QINCO_KHEAD_LA : SL.NHB;   // y blocks in flight ahead of the matrix pipe (a P row has NHB of them)
  constexpr int KHEAD_BURST = NEB + NHB + 4 * (LA + NEB);   // vector-memory loads of the head's burst
  // Gathers in flight at the ring wait of group G of the first down-projection (see take): the DMA that wait is for was issued at
  // group G + 2 - P / 4 (in the ring prologue for G <= P / 4 - 3).  Issued since then: the head's burst (NEB + NHB dwords, LA + NEB blocks of 4
  // loads; only while G <= 9) and the 4 loads of y block ib + LA, requested in front of group 4 NEB ib / 4 ... = block ib's first group
  // (NEB groups per block), for every ib < NHB - LA with  first <= G <= first + P / 4 - 3.
  constexpr auto khead_hosted = [](int G) constexpr {
    int n = G <= P / 4 - 3 ? KHEAD_BURST : 0;
    for (int ib = 0; ib + LA < NHB; ++ib)
      if (NEB * ib <= G && G <= NEB * ib + P / 4 - 3) n += 4;
    return n;
  };
  [[maybe_unused]] AsmBlock yb[KHEAD ? LA : 1];
  [[maybe_unused]] float ua[NEB], qa[NHB];   // (dead registers outside KHEAD)
  [[maybe_unused]] long gbase = 0;
  [[maybe_unused]] int dg = 0, dg_last = 0;
  [[maybe_unused]] const float* pptr = nullptr;
  auto khead_burst = [&]() QINCO_LAMBDA {
    // CONTRACT (the host launches this instance only then, launch_mlp): a wave's 32 rows span at most TWO groups -- A = 0 (K rows
    // per group), A = 16, or A a multiple of 32 -- so one MFMA with K = 2 adds the group rows (rows are clamped to R - 1: g is
    // monotone over the lanes).  Other A take the instance without KHEAD and its ob-outer stream.
    gbase = ((long)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)g);
    dg = (int)(g - gbase);
    dg_last = __builtin_amdgcn_readlane(dg, 31);
    const int k = half < dg_last ? half : dg_last;   // past the last group: its copy, times 0
    const float* up = a.uproj + (gbase + k) * DE + j;
    const float* qp = a.qproj + (gbase + k) * DH + j;
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { ua[ob] = up[ob * 32]; });
    static_for<NHB>([&]<int ob>() QINCO_LAMBDA { qa[ob] = qp[ob * 32]; });
    const float* tptr = a.ttab + (long)cid * DE + half * 4;
    pptr = a.ptab + (long)cid * DH + half * 4;
    static_for<LA>([&]<int i>() QINCO_LAMBDA { asm_load_block<i * 128>(yb[i], pptr); });
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = load_block(tptr + ob * 32); });
  };
  if constexpr (SHR && G8) {
    // fragments 0 .. P-9 are issued (P/4 - 2 per wave); "<= P/4 - 3 outstanding" = every wave's first one landed
    static_for<P / 4 - 2>([&]<int i>() QINCO_LAMBDA { dma.template operator()<4 * i>(); });
    wait_vm.template operator()<P / 4 - 3>();
    __builtin_amdgcn_s_barrier();
    ring[0] = myring[lane];
    ring[1] = myring[64 + lane];
  } else if constexpr (SHR) {
    // fragments 0 .. P-5 are issued (P/4 - 1 per wave); "<= P/4 - 2 outstanding" = every wave's first one landed
    static_for<P / 4 - 1>([&]<int i>() QINCO_LAMBDA { dma.template operator()<4 * i>(); });
    // KHEAD: the head's gathers go out behind the prologue's DMAs, so that the two round trips overlap; the wait for the first
    // fragment leaves them in flight (they are younger than every DMA issued so far)
    if constexpr (KHEAD) khead_burst();
    wait_vm.template operator()<P / 4 - 2 + (KHEAD ? KHEAD_BURST : 0)>();
    __builtin_amdgcn_s_barrier();
    ring[0] = myring[lane];
    ring[1] = myring[64 + lane];
  } else if constexpr (LDSR) {
    static_for<P - 1>([&]<int i>() QINCO_LAMBDA { dma.template operator()<i>(); });
    wait_vm.template operator()<P - 2>();  // fragment 0 has landed
    ring[0] = myring[lane];
    wait_vm.template operator()<P - 3>();  // fragment 1
    ring[1] = myring[64 + lane];
  } else {
#pragma unroll
    for (int i = 0; i < P; ++i) ring[i] = wp[i * 64];
  }
  stamp(1);   // ring prologue done (first fragments landed, first barrier passed)
  // EX: vector-memory loads OTHER than ring DMAs that this wave has issued since the DMA this wait is for and that may still be in
  // flight (KHEAD's head and block gathers): memory operations return in order, so the count is raised by exactly their number --
  // with EX = 0 the wait would also drain every gather issued in the last nine groups
  auto take = [&]<int T, int EX = 0>() QINCO_LAMBDA -> f32x4 {
    if constexpr (SHR && G8) {
      // Groups of 8.  Before fragment T = 8g every wave has issued 2g + P/4 - 2 DMAs; "<= P/4 - 5 outstanding" = its first
      // 2g + 3 landed, so past the barrier fragments <= 8g + 11 are in LDS: covers the reads (<= 8g + 9) of this group.  The
      // two refills overwrite fragments 8g - 8 .. 8g - 1, which every wave consumed before this barrier.
      if constexpr ((T & 7) == 0) {
        wait_vm.template operator()<P / 4 - 5>();
        __builtin_amdgcn_s_barrier();
        dma.template operator()<T + P - 8>();
        dma.template operator()<T + P - 4>();
      }
      ring[(T + 2) % 3] = myring[((T + 2) % P) * 64 + lane];
      asm volatile("" ::: "memory");
      return ring[T % 3];
    } else if constexpr (SHR) {
      // Before fragment T = 4g every wave has issued g + P/4 - 1 DMAs; "<= P/4 - 3 outstanding" = its first g + 2
      // landed, so past the barrier fragments <= 4g + 7 are in LDS: covers the reads (<= 4g + 5) of this group.
      // The refill overwrites fragments 4g - 4 .. 4g - 1, which every wave consumed before this barrier.
      if constexpr ((T & 3) == 0) {
        wait_vm.template operator()<P / 4 - 3 + EX>();
        __builtin_amdgcn_s_barrier();
        dma.template operator()<T + P - 4>();
      }
      ring[(T + 2) % NRING] = myring[((T + 2) % P) * 64 + lane];
      asm volatile("" ::: "memory");   // the ring reads keep their program order (see fragmm)
      return ring[T % NRING];
    } else if constexpr (LDSR) {
      // fragments T, T+1 are in ring[]; fragment T+2's DMA is P-4 DMAs old; refill the slot of fragment T-1.
      wait_vm.template operator()<P - 4>();
      ring[(T + 2) % 3] = myring[((T + 2) % P) * 64 + lane];
      dma.template operator()<T + P - 1>();
      return ring[T % 3];
    } else {
      f32x4 w = ring[T % P];
      ring[T % P] = wp[(T + P) * 64];
      return w;
    }
  };
  auto skip_pad = [&]<int FROM, int TO>() QINCO_LAMBDA {
    static_for<TO - FROM>([&]<int i>() QINCO_LAMBDA { (void)take.template operator()<FROM + i>(); });
  };
  // One fragment: acc += W[ob, 8 features (4q..4q+3 of each half) of block ib] . b  (4 dependent MFMAs), then `extra`
  // (a slice of a chain epilogue that the following MFMAs do not depend on).
  auto noop = []() QINCO_LAMBDA {};
  auto fragmm = [&]<int T, int q, int EX = 0>(f32x16& acc, const f32x16& b, auto&& extra) QINCO_LAMBDA {
    f32x4 w = take.template operator()<T, EX>();
    // Shared ring: the barrier in front of fragment T + 1 (T = 3 mod 4) licenses the refill of the slots of fragments <= T, so
    // this wave's LDS reads of those fragments must have COMPLETED when it arrives there.  Program order alone does not give
    // that -- hipcc moves MFMAs, and with them the s_waitcnt lgkmcnt that completes a ds_read, across s_barrier (found on the
    // 16-row kernel, where it corrupted ~1 wave in 100 at three workgroups per CU: mlp16_kernel.hpp).  The
    // pin makes the last fragment ahead of the barrier a register value at this point of the program; the ring reads are
    // issued in program order (memory fences in take) and LDS returns a wave's reads in order, so every earlier fragment has
    // arrived too; asm volatile does not cross the barrier's fences.  Costs 0.8 % at C2, gains 1.2 % at qinco2-S.
    if constexpr (SHR && (T & (G8 ? 7 : 3)) == (G8 ? 7 : 3)) pin4_v(w);
    static_for<4>([&]<int e>() QINCO_LAMBDA { acc = QINCO_MFMA(w[e], b[4 * q + e], acc); });
    extra();
  };


  if constexpr (KHEAD) {
    // (the head's burst was issued in front of the ring prologue's wait: khead_burst)
  } else if constexpr (FOLD) {
    // ---- A-C folded: z = T[cid] + U[group] -----------------------------------------------------------
    // (hipcc turns these gathers into batches of 8 loads drained by vmcnt(0): 5 exposed round trips per tile with 288
    // registers, 12 under OCC2's 128-VGPR budget -- 74 k of a short-MLP tile's 226 k cycles, profiles/r02_timeline.jsonl.
    // Hand-pipelined orders -- one block ahead; all table rows first, straight into z / y's AGPRs -- spill 74-136 registers
    // under OCC2 and were dropped.)
    const float* tptr = a.ttab + (long)cid * DE + half * 4;
    const float* uptr = a.uproj + g * DE + half * 4;
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = load_block(tptr + ob * 32) + load_block(uptr + ob * 32); });
    if constexpr (FOLD2) {
      const float* pptr = a.ptab + (long)cid * DH + half * 4;
      const float* qptr = a.qproj + g * DH + half * 4;
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
        y[ob] = load_block(pptr + ob * 32) + load_block(qptr + ob * 32);
        relu16(y[ob]);
        pin_a(y[ob]);
      });
    }
  } else {
    // ---- A: z = in_proj(c)  (K-outer; c blocks streamed from the codebook, next block prefetched) ---------
    if constexpr (PROJ) {
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = zero16(); });
      f32x16 cb = load_block(cptr);
      static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
        f32x16 cur = cb;
        if constexpr (ib + 1 < NDB) cb = load_block(cptr + (ib + 1) * 32);
        static_for<4>([&]<int q>() QINCO_LAMBDA {
          static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
            fragmm.template operator()<(ib * 4 + q) * NEB + ob, q>(z[ob], cur, noop);
          });
        });
      });
      skip_pad.template operator()<NEB * NDB * 4, SL.T_IN>();
      wp += SL.T_IN * 64;
    } else {
      static_for<NEB>([&]<int ib>() QINCO_LAMBDA { z[ib] = load_block(cptr + ib * 32); });
    }

    // ---- B: y = bias of the concat Linear ------------------------------------------------------
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        f32x4 w = take.template operator()<ob * 4 + q>();
        if constexpr (SHR && ((ob * 4 + q) & (G8 ? 7 : 3)) == (G8 ? 7 : 3)) pin4_v(w);   // (see fragmm)
        static_for<4>([&]<int e>() QINCO_LAMBDA { y[ob][4 * q + e] = w[e]; });
      });
    });
    skip_pad.template operator()<NEB * 4, SL.T_BIAS>();
    wp += SL.T_BIAS * 64;

    // ---- C: y += W_cat . [z ; xhat]   then z = z + y   (QConcat.forward) -------------------------
    {
      f32x16 xb = load_block(xhptr);
      static_for<NEB + NDB>([&]<int ib>() QINCO_LAMBDA {
        f32x16 b;
        if constexpr (ib < NEB) {
          b = z[ib];
        } else {
          b = xb;
          if constexpr (ib + 1 < NEB + NDB) xb = load_block(xhptr + (ib + 1 - NEB) * 32);
        }
        static_for<4>([&]<int q>() QINCO_LAMBDA {
          static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
            fragmm.template operator()<(ib * 4 + q) * NEB + ob, q>(y[ob], b, noop);
          });
        });
      });
    }
    skip_pad.template operator()<NEB*(NEB + NDB) * 4, SL.T_CAT>();
    wp += SL.T_CAT * 64;
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = z[ob] + y[ob]; });
  }

#ifdef QINCO_TIMELINE
  {   // all head operands have arrived (forces the waits here, where the production kernel lets them overlap the first chain)
    float chk = 0.f;
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { chk += z[ob][0]; });
    if (chk == 123.4567f) a.cand_out[0] = chk;
  }
#endif
  stamp(2);   // z (and y) assembled
  // ---- D: L residual FFN blocks: z = z + W_down . relu(W_up . z)   (QBlockFFN.forward) ---------
  if constexpr (PINNED) {
    // Register-file plan (the compiler is told, not asked): z lives in VGPRs (B operand of the up-projection,
    // VALU-updated by the residual add), y lives in AGPRs (B operand of the down-projection, never touched by
    // VALU after it is written), chain accumulators t[2] alternate.  Epilogues run one chain late, a quarter per
    // fragment, under MFMAs that do not depend on them (an eager epilogue drains the matrix pipe at each of the
    // 24 chain ends per layer: s_nop 15 + ~64 dependent VALU ops):
    //   up:   y[ob-1] = relu(t_prev)  (16 v_max_i32 + 16 v_accvgpr_write)
    //   down: z[ob-1] += t_prev       (16 v_add_f32)
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { pin_v(z[ob]); });
    f32x16 t[2];
    auto up_phase = [&]() QINCO_LAMBDA {
      if constexpr (OCC2) {   // one accumulator, epilogue at the chain end
        static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
          t[0] = zero16();
          static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
            static_for<4>([&]<int q>() QINCO_LAMBDA { fragmm.template operator()<(ob * NEB + ib) * 4 + q, q>(t[0], z[ib], noop); });
          });
          relu16(t[0]);
          y[ob] = t[0];
          pin_a(y[ob]);
        });
        skip_pad.template operator()<NHB * NEB * 4, SL.T_UP>();
        wp += SL.T_UP * 64;
        return;
      }
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
        t[ob & 1] = zero16();
        static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            constexpr int TI = (ob * NEB + ib) * 4 + q;
            if constexpr (ib == 0 && ob > 0) {
              fragmm.template operator()<TI, q>(t[ob & 1], z[ib], [&]() QINCO_LAMBDA {
                static_for<4>([&]<int e>() QINCO_LAMBDA { y[ob - 1][4 * q + e] = relu1(t[(ob - 1) & 1][4 * q + e]); });
              });
              if constexpr (q == 3) pin_a(y[ob - 1]);
            } else {
              fragmm.template operator()<TI, q>(t[ob & 1], z[ib], noop);
            }
          });
        });
      });
      relu16(t[(NHB - 1) & 1]);
      y[NHB - 1] = t[(NHB - 1) & 1];
      pin_a(y[NHB - 1]);
      skip_pad.template operator()<NHB * NEB * 4, SL.T_UP>();
      wp += SL.T_UP * 64;
    };
    auto down_phase = [&]() QINCO_LAMBDA {
      if constexpr (OCC2) {
        static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
          t[0] = zero16();
          static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
            static_for<4>([&]<int q>() QINCO_LAMBDA { fragmm.template operator()<(ob * NHB + ib) * 4 + q, q>(t[0], y[ib], noop); });
          });
          z[ob] = z[ob] + t[0];
          pin_v(z[ob]);
        });
        skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
        wp += SL.T_DOWN * 64;
        return;
      }
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
        t[ob & 1] = zero16();
        static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            constexpr int TI = (ob * NHB + ib) * 4 + q;
            if constexpr (ib == 0 && ob > 0) {
              fragmm.template operator()<TI, q>(t[ob & 1], y[ib], [&]() QINCO_LAMBDA {
                static_for<4>([&]<int e>() QINCO_LAMBDA { z[ob - 1][4 * q + e] += t[(ob - 1) & 1][4 * q + e]; });
              });
              if constexpr (q == 3) pin_v(z[ob - 1]);
            } else {
              fragmm.template operator()<TI, q>(t[ob & 1], y[ib], noop);
            }
          });
        });
      });
      z[NEB - 1] = z[NEB - 1] + t[(NEB - 1) & 1];  // the only chain end per layer that drains the pipe
      pin_v(z[NEB - 1]);
      skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
      wp += SL.T_DOWN * 64;
    };
    // KHEAD: block 0's down-projection, K-outer: y block ib = relu(P[cid] + Q[g]) is finished just before its 4 NEB fragments, block
    // ib + LA is requested in its place; z = (T + U) + t at the end (the order of the ob-outer form)
    auto first_down = [&]() QINCO_LAMBDA {
      const float oh0 = (dg == half) ? 1.f : 0.f;
      f32x16 t4[NEB];
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { t4[ob] = zero16(); });
      static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise moves the group add behind the next ring wait and drains vmcnt there)
        // y block ib was requested LA blocks ago (ib < LA: in the head's burst, whose order is hipcc's: everything issued before
        // this block's... first ring wait has to land); younger than its loads: NEB ring DMAs per block and the blocks after it
        constexpr int YOUNGER = ib < LA ? NEB * ib + 4 * ib : NEB * LA + 4 * (LA - 1);
        f32x16 yv = QINCO_MFMA(one_hot_operand(qa[ib]), oh0, wait_block<YOUNGER>(yb[ib % LA]));
        relu16(yv);
        if constexpr (ib + LA < NHB) asm_load_block<(ib + LA) * 128>(yb[ib % LA], pptr);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&]<int q>() QINCO_LAMBDA {
          static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
            constexpr int TI = (ib * 4 + q) * NEB + ob;
            fragmm.template operator()<TI, q, khead_hosted(TI / 4)>(t4[ob], yv, noop);
          });
        });
      });
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
        z[ob] = QINCO_MFMA(one_hot_operand(ua[ob]), oh0, z[ob]);
        z[ob] = z[ob] + t4[ob];
        pin_v(z[ob]);
      });
      skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
      wp += SL.T_DOWN * 64;
    };
    if constexpr (KHEAD) {
      first_down();
      stamp(6);
    } else if constexpr (FOLD2) down_phase();   // block 0 (needs L >= 1: the host never picks FOLD2 for L == 0)
#pragma unroll 1
    for (int l = FOLD2 ? 1 : 0; l < a.L; ++l) {
      up_phase();
      down_phase();
    }
  } else {
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
        f32x16 acc = zero16();
        static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            fragmm.template operator()<(ob * NEB + ib) * 4 + q, q>(acc, z[ib], noop);
          });
        });
        relu16(acc);
        y[ob] = acc;
      });
      skip_pad.template operator()<NHB * NEB * 4, SL.T_UP>();
      wp += SL.T_UP * 64;
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
        f32x16 acc = zero16();
        static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            fragmm.template operator()<(ob * NHB + ib) * 4 + q, q>(acc, y[ib], noop);
          });
        });
        z[ob] = z[ob] + acc;
      });
      skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
      wp += SL.T_DOWN * 64;
    }
  }

  stamp(3);   // FFN blocks done
  if constexpr (SELEP) {
    if (a.sel_T > 0) {   // (a kernel argument: uniform over the launch)
      // ---- E': candidates in place (z <- (z + coeff*c) + xhat), distances, per-vector top-T, winners only ----------------------
      const long nv = g / a.F;
      const float* xptr = a.x + nv * D + half * 4;
      // every row fetches its parent beam's code history now, with the epilogue operands (a winner then stores it without another
      // round trip at the end of the workgroup's life); histories longer than 8 codes go the cooperative way below
      constexpr int HP = 8;
      int hv[HP];
      const bool hist_inline = a.sel_m <= HP;
      if (hist_inline && half == 0) {
#pragma unroll
        for (int jj = 0; jj < HP; ++jj) hv[jj] = jj < a.sel_m ? a.sel_hist_in[g * a.sel_M + jj] : 0;
      }
      float s2 = 0.f, sx = 0.f, xn = 0.f;
      {
        static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
          f32x16 o = z[ob];
          if (a.add_c) o = o + load_block(cptr + ob * 32);
          o = o + load_block(xhptr + ob * 32);
          const f32x16 xb = load_block(xptr + ob * 32);
#pragma unroll
          for (int i = 0; i < 16; ++i) {   // (the order of the generic epilogue below: the same distances to the bit)
            s2 = fmaf(o[i], o[i], s2);
            sx = fmaf(o[i], xb[i], sx);
            xn = fmaf(xb[i], xb[i], xn);
          }
          z[ob] = o;
        });
      }
      s2 += __shfl_xor(s2, 32);
      sx += __shfl_xor(sx, 32);
      xn += __shfl_xor(xn, 32);
      const float dist = (xn + s2) - 2.f * sx;
      // Selection by counting: a candidate's rank among the vector's C = (distance, index) keys -- beam_select_kernel's order
      // (select.hpp: sel_key, ties -> lower index) -- is the number of keys below its own: C / 2 LDS reads per lane, all rows at once,
      // no serial chain at the end of the workgroup's life (a one-wave wave_top_t here cost 6 % of a qinco2-S encode).  Ranks are a
      // permutation of 0 .. C-1: rank < T = winner, and the rank is its place in the next beam.
#if defined(QINCO_EXPERIMENT) && defined(QINCO_EXP_SELEP_SECOND_LDS)
      // experiment builds (scripts/exp_isa_lint_red.py): the rounds-3/4 layout -- the keys in LDS objects of their own -- which makes
      // hipcc drain the weight ring in front of its reads; kept to show that tests/test_isa.py turns red on it
      __shared__ unsigned long long sel_keys_own[128];
      __shared__ int sel_idx_own[128];
      unsigned long long* const sel_keys = sel_keys_own;
      int* const sel_idx = sel_idx_own;
#else
      unsigned long long* const sel_keys = reinterpret_cast<unsigned long long*>(lds_ring + RING4);   // [128]
      int* const sel_idx = reinterpret_cast<int*>(lds_ring + RING4 + 64);                              // [128]
#endif
      auto lds_barrier = [&]() QINCO_LAMBDA {   // (raw: __syncthreads would also wait for the ring's tail DMAs)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      const int C = a.A * a.F, T = a.sel_T;
      const int rl = wave * 32 + j;          // row within the workgroup
      const int v = rl / C, bi = rl - v * C;   // its vector within the workgroup, its flat candidate index
      const long n0 = ((long)blockIdx.x * 128) / C, nvec = a.R / C;
      const unsigned long long mine = ((unsigned long long)sel_key(dist) << 32) | (unsigned)bi;
      if (half == 0) sel_keys[rl] = mine;
      lds_barrier();
      stamp(7);   // distances computed, keys published
      int rk = 0;
      {   // (the two lanes of a row split the keys; eight independent LDS reads in flight, not one round trip per key)
        const unsigned long long* kv = sel_keys + v * C + half * (C / 2);
        int i = 0;
        for (; i + 8 <= C / 2; i += 8) {
          unsigned long long k8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) k8[u] = kv[i + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) rk += k8[u] < mine ? 1 : 0;
        }
        for (; i < C / 2; ++i) rk += kv[i] < mine ? 1 : 0;
        if (C & 1) rk += (half == 0 && sel_keys[v * C + C - 1] < mine) ? 1 : 0;
      }
      rk += __shfl_xor(rk, 32);
      const bool winner = valid && rk < T;
      if (winner) {
        const long orow = nv * T + rk;
        float* outp = a.sel_xhat_out + orow * D + half * 4;
        static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t = {z[ob][4 * q], z[ob][4 * q + 1], z[ob][4 * q + 2], z[ob][4 * q + 3]};
            *reinterpret_cast<f32x4*>(outp + ob * 32 + 8 * q) = t;
          }
        });
        if (half == 0) {
          if (hist_inline) {
            int* hout = a.sel_hist_out + orow * a.sel_M;
#pragma unroll
            for (int jj = 0; jj < HP; ++jj)
              if (jj < a.sel_m) hout[jj] = hv[jj];
            hout[a.sel_m] = cid;
          } else {
            sel_idx[v * C + rk] = bi;
          }
        }
      }
      if (hist_inline) {
        stamp(4);   // epilogue issued
        if constexpr (LDSR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);   // stores / tail DMAs retired
        return;
      }
      lds_barrier();
      // the winners' code histories -- the parent beam's codes, then the winner's own (qinco_inference.py:203-210) -- one element
      // per thread (a loop per winner would chain sel_m dependent round trips at the very end of the workgroup's life)
      const int per = T * (a.sel_m + 1);
      for (int e = threadIdx.x; e < (128 / C) * per; e += 256) {
        const int vv = e / per, r = e - vv * per, t = r / (a.sel_m + 1), jj = r - t * (a.sel_m + 1);
        const long nn = n0 + vv;
        if (nn >= nvec) break;
        const int wi = sel_idx[vv * C + t], f = wi / a.A;
        const int val = jj < a.sel_m ? a.sel_hist_in[(nn * a.F + f) * a.sel_M + jj]
                                     : (a.cand_ids ? a.cand_ids[nn * C + wi] : wi - f * a.A);
        a.sel_hist_out[(nn * T + t) * a.sel_M + jj] = val;
      }
      stamp(4);   // epilogue issued
      if constexpr (LDSR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(5);   // stores / tail DMAs retired
      return;
    }
  }
  if constexpr (KHEAD && !PROJ) {
    // ---- E (KHEAD, identity projections): the same candidates and distances with every operand requested in ONE burst.  The
    // generic epilogue below fetches a block's operands in front of its out_proj chain; without projections there is no chain and its
    // four blocks were four exposed round trips at the end of the workgroup's life (28 k cycles, profiles/r03_timeline.jsonl).
    // xhat (per group) and x (per vector) are the same bytes for every row of a group / vector: they arrive as one coalesced dword per
    // lane and block and reach the rows through a one-hot MFMA, like the head's U and Q (o + 1 * xhat rounds like o + xhat; 1 * x + 0 is x).
    const long nv = g / a.F;
    const long nbase = ((long)__builtin_amdgcn_readfirstlane((int)(nv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)nv);
    const int dn = (int)(nv - nbase), dn_last = __builtin_amdgcn_readlane(dn, 31);
    float xha[NDB], xa[NDB];
    {
      const int kg = half < dg_last ? half : dg_last, kn = half < dn_last ? half : dn_last;
      const float* hp = a.xhat + (gbase + kg) * D + j;
      static_for<NDB>([&]<int ob>() QINCO_LAMBDA { xha[ob] = hp[ob * 32]; });
      if (a.x) {
        const float* xp = a.x + (nbase + kn) * D + j;
        static_for<NDB>([&]<int ob>() QINCO_LAMBDA { xa[ob] = xp[ob * 32]; });
      }
    }
    f32x16 cb4[NDB];
    if (a.add_c) static_for<NDB>([&]<int ob>() QINCO_LAMBDA { cb4[ob] = load_block(cptr + ob * 32); });
    const float ohg = (dg == half) ? 1.f : 0.f, ohn = (dn == half) ? 1.f : 0.f;
#if defined(QINCO_EXPERIMENT) && defined(QINCO_EXP_FEW_LINES)
    float* outp = a.cand_out + ((QINCO_EXP_FEW_LINES & 2) ? (row & 1) : row) * D + half * 4;
#else
    float* outp = a.cand_out + row * D + half * 4;
#endif
    float s2 = 0.f, sx = 0.f, xn = 0.f;
    static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
      f32x16 o = z[ob];
      if (a.add_c) o = o + cb4[ob];
      o = QINCO_MFMA(one_hot_operand(xha[ob]), ohg, o);
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
          *reinterpret_cast<f32x4*>(outp + ob * 32 + 8 * q) = t;
        }
      }
      if (a.x) {
        f32x16 xb = QINCO_MFMA(one_hot_operand(xa[ob]), ohn, zero16());
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // (the order of the generic epilogue: the same distances to the bit)
          s2 = fmaf(o[i], o[i], s2);
          sx = fmaf(o[i], xb[i], sx);
          xn = fmaf(xb[i], xb[i], xn);
        }
      }
    });
    if (a.dist_out) {
      s2 += __shfl_xor(s2, 32);
      sx += __shfl_xor(sx, 32);
      xn += __shfl_xor(xn, 32);
      if (valid && half == 0) a.dist_out[row] = (xn + s2) - 2.f * sx;
    }
    stamp(4);
    if constexpr (LDSR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
    return;
  }
  // ---- E: out_proj + epilogue: cand = (out + coeff*c) + xhat ; dist = |x|^2 + |cand|^2 - 2 x.cand
  const long n = g / a.F;
  const float* xptr = a.x ? a.x + n * D + half * 4 : nullptr;
  float* outp = a.cand_out + row * D + half * 4;
  float s2 = 0.f, sx = 0.f, xn = 0.f;
  static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
    // operands of this block's epilogue are fetched before its GEMM chain so their latency hides under it
    f32x16 cblk, xhb, xb;
    if (a.add_c) cblk = load_block(cptr + ob * 32);
    xhb = load_block(xhptr + ob * 32);
    if (xptr) xb = load_block(xptr + ob * 32);
    f32x16 o;
    if constexpr (PROJ) {
      o = zero16();
      static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
        static_for<4>([&]<int q>() QINCO_LAMBDA {
          fragmm.template operator()<(ob * NEB + ib) * 4 + q, q>(o, z[ib], noop);
        });
      });
    } else {
      o = z[ob];
    }
    if (a.add_c) o = o + cblk;
    o = o + xhb;
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 t = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        *reinterpret_cast<f32x4*>(outp + ob * 32 + 8 * q) = t;
      }
    }
    if (xptr) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s2 = fmaf(o[i], o[i], s2);
        sx = fmaf(o[i], xb[i], sx);
        xn = fmaf(xb[i], xb[i], xn);
      }
    }
  });
  if (a.dist_out) {
    s2 += __shfl_xor(s2, 32);
    sx += __shfl_xor(sx, 32);
    xn += __shfl_xor(xn, 32);
    if (valid && half == 0) a.dist_out[row] = (xn + s2) - 2.f * sx;
  }
  stamp(4);   // epilogue issued
  // no LDS-DMA may be in flight when the wave ends (its LDS could be handed to the next workgroup)
  if constexpr (LDSR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(5);   // stores / tail DMAs retired
}

// U[g] = W_cat[:, De:] . xhat_g for every (vector, beam) group (FOLD), and Q[g] = W_up[0] . U[g] (FOLD2).  Same transposed-MFMA
// form: a wave keeps its 32 groups' xhat as B operands and consumes the weight fragments (wx: De x D, then wq: Dh x De, in
// (ob, ib, q) order) as A operands; U's C layout is Q's B layout, so Q chains on the U blocks still in registers.
// Round 2: the fragments come through LDS, shared by the 4 waves of the workgroup -- chunks of 16 fragments (16 KiB),
// double-buffered, each wave LDS-DMAs a quarter of the next chunk while the current one is consumed, one barrier per chunk.
// Round 1 let every wave load every fragment straight from L2 with no look-ahead: latency-bound at 35-45 % of the fp32-MFMA
// peak, and 4096 waves x 192 KiB of L2 -> L1 traffic per call at the qinco2-S shape.  Same MFMA order, same results.
template <int D, int DE, int DH>
__global__ void __launch_bounds__(256) xproj_kernel(XprojArgs a) {
  constexpr int NDB = D / 32, NEB = DE / 32, NHB = DH / 32;
  constexpr int CH = 16;                              // fragments per chunk
  constexpr int NFU = NEB * NDB * 4, NFQ = NHB * NEB * 4;
  __shared__ f32x4 lds_w[2 * CH * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int j = lane & 31, half = lane >> 5;
  const long g0 = ((long)blockIdx.x * 4 + wave) * 32;   // every wave runs (barriers); groups past G are clamped, never stored
  long g = g0 + j;
  const bool valid = g < a.G;
  if (!valid) g = a.G - 1;
  const float* xp = a.xhat + g * D + half * 4;
  f32x16 xt[NDB];
#pragma unroll
  for (int ib = 0; ib < NDB; ++ib) xt[ib] = load_block(xp + ib * 32);

  // chunk c of a stream -> buffer c & 1; wave w fetches fragments w, w + 4, ... of the chunk
  auto dma_chunk = [&]<int C, int NF>(const f32x4* stream) QINCO_LAMBDA {
    constexpr int n = (NF - C * CH) < CH ? (NF - C * CH) : CH;
    static_for<(n + 3) / 4>([&]<int i>() QINCO_LAMBDA {
      const int f = wave_u + 4 * i;
      if (f < n)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + (C * CH + f) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(lds_w + ((C & 1) * CH + f) * 64), 16, 0, 0);
    });
  };
  // boundary before chunk C: this wave's DMAs of chunk C have landed, every wave is done with the other buffer, chunk C+1 is
  // requested.  The wait is vmcnt(0): round 2 counted the block stores issued since the DMAs (4 per finished block) and waited
  // for exactly the DMAs -- correct only as long as hipcc emits exactly one global_store_dwordx4 per source-level store, which
  // is a property of code generation, not of the source (advisor, round 2); the full wait costs < 2 % of this kernel, which is
  // itself 1-3 % of a step.
  auto boundary = [&]<int C, int NF, int FPO>(const f32x4* stream) QINCO_LAMBDA {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr ((C + 1) * CH < NF) dma_chunk.template operator()<C + 1, NF>(stream);
  };
  // (the last fragment of a chunk is pinned: its LDS read -- and, reads being issued and returned in order, every earlier one --
  // has completed before this wave reaches the boundary that lets the other waves refill the buffer; see mlp_kernel fragmm)
  auto frag = [&]<int F>() QINCO_LAMBDA -> f32x4 {
    f32x4 w = lds_w[(((F / CH) & 1) * CH + F % CH) * 64 + lane];
    asm volatile("" ::: "memory");
    if constexpr (F % CH == CH - 1) pin4_v(w);
    return w;
  };
  // (lanes past G hold copies of group G-1: they store the same values to the same place)
  auto store_block = [&](float* p, const f32x16& v) QINCO_LAMBDA {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      *reinterpret_cast<f32x4*>(p + 8 * q) = t;
    }
  };

  float* up = a.uproj + g * DE + half * 4;
  f32x16 u[NEB];
  dma_chunk.template operator()<0, NFU>(a.wx);
  static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
    f32x16 acc = zero16();
    static_for<NDB * 4>([&]<int t>() QINCO_LAMBDA {
      constexpr int F = ob * NDB * 4 + t;
      if constexpr (F % CH == 0) boundary.template operator()<F / CH, NFU, NDB * 4>(a.wx);
      const f32x4 w = frag.template operator()<F>();
      static_for<4>([&]<int e>() QINCO_LAMBDA { acc = QINCO_MFMA(w[e], xt[t / 4][4 * (t % 4) + e], acc); });
    });
    u[ob] = acc;
    store_block(up + ob * 32, u[ob]);
  });
  if (a.wq) {   // FOLD2 (wave-uniform): Q = W_up[0] . U
    float* qp = a.qproj + g * DH + half * 4;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave has left the U stream's buffers
    dma_chunk.template operator()<0, NFQ>(a.wq);
    static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
      f32x16 acc = zero16();
      static_for<NEB * 4>([&]<int t>() QINCO_LAMBDA {
        constexpr int F = ob * NEB * 4 + t;
        if constexpr (F % CH == 0) boundary.template operator()<F / CH, NFQ, NEB * 4>(a.wq);
        const f32x4 w = frag.template operator()<F>();
        static_for<4>([&]<int e>() QINCO_LAMBDA { acc = QINCO_MFMA(w[e], u[t / 4][4 * (t % 4) + e], acc); });
      });
      store_block(qp + ob * 32, acc);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the wave ends
}

}  // namespace qinco
