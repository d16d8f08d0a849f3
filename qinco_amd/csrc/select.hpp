// In-wave sorting / top-T selection primitives (wave64, gfx950): the lexicographic arg-min across a wave, the bitonic sort of
// the 64 lane values, and the threshold-and-compact selection of the T smallest of a row that the pre-selection table kernels
// and the beam selection are built on.  Device functions and templates only: shared by the C-ABI translation unit
// (aux_kernels.hpp, ivf_kernel.hpp) and the per-shape kernel instances (presel_kernel.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "mlp_args.hpp"

namespace qinco {

#define QINCO_DEV __device__ __forceinline__

// Lexicographic (value, index) minimum across the 64 lanes of a wave, result in every lane (ties -> lower index,
// which is what argmin returns on the reference CPU path; topk's order among exact ties is unspecified).
// All on the VALU: DPP lane permutes inside the 16-lane rows, gfx950's v_permlane16_swap / v_permlane32_swap across
// rows.  (__shfl_xor is ds_bpermute_b32: an LDS-crossbar round trip per level; a float (value, index) compare-select
// per level and a ballot / v_readlane fast path were both slower than the integer form below: SGPR round trips.)
// Branch-free, SGPR-free: reduce the order-preserving integer image of the value with v_min_u32 (fused with the DPP
// permute: one instruction per level), then reduce the index among the lanes that hold the minimum the same way.
// Exactly the lexicographic (value, index) minimum (-0.0 is folded into +0.0 first, as a float compare treats it).
// Needs all 64 lanes active.
template <int CTRL>
QINCO_DEV unsigned dpp_u(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
QINCO_DEV unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
QINCO_DEV unsigned wave_umin(unsigned k) {
  k = umin(k, dpp_u<0xB1>(k));
  k = umin(k, dpp_u<0x4E>(k));
  k = umin(k, dpp_u<0x141>(k));
  k = umin(k, dpp_u<0x140>(k));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(k, k, false, false);
    k = umin((unsigned)r[0], (unsigned)r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(k, k, false, false);
    k = umin((unsigned)r[0], (unsigned)r[1]);
  }
  return k;
}
QINCO_DEV unsigned ordered_bits(float d) {
  const unsigned u = __builtin_bit_cast(unsigned, d + 0.f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
QINCO_DEV float from_ordered_bits(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, u);
}
template <int N>
QINCO_DEV void wave_argmin_u(float (&v)[N], int (&i)[N]) {
  unsigned key[N], m[N], c[N];
#pragma unroll
  for (int u = 0; u < N; ++u) key[u] = ordered_bits(v[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) m[u] = wave_umin(key[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) c[u] = key[u] == m[u] ? (unsigned)i[u] : 0x7fffffffu;
#pragma unroll
  for (int u = 0; u < N; ++u) c[u] = wave_umin(c[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) {
    v[u] = from_ordered_bits(m[u]);
    i[u] = (int)c[u];
  }
}

QINCO_DEV void wave_argmin(float& v, int& i) {
  float a[1] = {v};
  int b[1] = {i};
  wave_argmin_u<1>(a, b);
  v = a[0];
  i = b[0];
}

// ---------------------------------------------------------------------------------------------
// Top-T selection of a wave: threshold-and-compact instead of T dependent arg-min rounds.
//   1. every lane reduces its elements (index = lane, lane + 64, ...) to a lane minimum;
//   2. one bitonic sort of the 64 lane minima across the wave (21 compare-exchange steps on DPP row permutes and
//      gfx950's v_permlane16/32_swap, 3-5 VALU ops each); the value tau now in lane T-1 bounds the T-th smallest
//      element from above, because T lanes hold an element <= tau;
//   3. the elements <= tau (T <= S; S ~ 1.1 T .. 1.4 T for 2-8 elements per lane) are compacted into LDS as 64-bit
//      (ordered distance bits, index) keys with ballot / mbcnt;
//   4. every survivor counts the survivors with a smaller key: its rank in the exact lexicographic (distance, index)
//      order = its output slot (ascending, ties -> lower index: argmin / stable argsort, qinco_inference.py:173,200).
// ~170 VALU ops per selection against T x ~40 dependent ones.  Needs T <= 64 and S <= 64 (else the caller falls back to the
// rounds: massive exact ties only).  NaN distances sort last.  All 64 lanes must be active.
// ---------------------------------------------------------------------------------------------
QINCO_DEV unsigned sel_key(float d) { return d != d ? 0xffffffffu : ordered_bits(d); }

QINCO_DEV unsigned sel_pick(unsigned lo, unsigned hi, unsigned long long keepmin) {
  unsigned r;   // r = keepmin[lane] ? lo : hi, the lane mask being a compile-time constant in an SGPR pair
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "s"(keepmin));
  return r;
}
template <int K, int J>
constexpr unsigned long long sel_keepmin_mask() {
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) {
    const bool up = K >= 64 || (i & K) == 0;   // ascending block
    const bool low = (i & J) == 0;             // lower lane of its pair
    if (up == low) m |= 1ull << i;
  }
  return m;
}
QINCO_DEV unsigned sel_umax(unsigned a, unsigned b) { return a > b ? a : b; }
// compare-exchange with the lane J away inside a bitonic block of size K
template <int K, int J>
QINCO_DEV unsigned sel_cmpx(unsigned x) {
  unsigned lo, hi;
  if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // {even-row copy, odd-row copy} of each row pair
    lo = umin((unsigned)r[0], (unsigned)r[1]);
    hi = sel_umax((unsigned)r[0], (unsigned)r[1]);
  } else if constexpr (J == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    lo = umin((unsigned)r[0], (unsigned)r[1]);
    hi = sel_umax((unsigned)r[0], (unsigned)r[1]);
  } else {
    unsigned p;
    if constexpr (J == 1) p = dpp_u<0xB1>(x);         // quad_perm [1,0,3,2]
    else if constexpr (J == 2) p = dpp_u<0x4E>(x);    // quad_perm [2,3,0,1]
    else if constexpr (J == 8) p = dpp_u<0x128>(x);   // row_ror:8
    else {                                            // J == 4: banks 0,2 read lane+4 (row_shl:4), banks 1,3 lane-4
      static_assert(J == 4, "");
      p = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);
      p = (unsigned)__builtin_amdgcn_update_dpp((int)p, (int)x, 0x114, 0xF, 0xA, false);
    }
    lo = umin(x, p);
    hi = sel_umax(x, p);
  }
  return sel_pick(lo, hi, sel_keepmin_mask<K, J>());
}
template <int K, int J>
QINCO_DEV unsigned sel_merge(unsigned x) {
  x = sel_cmpx<K, J>(x);
  if constexpr (J > 1) x = sel_merge<K, J / 2>(x);
  return x;
}
// ascending across the 64 lanes
QINCO_DEV unsigned wave_sort64(unsigned x) {
  x = sel_merge<2, 1>(x);
  x = sel_merge<4, 2>(x);
  x = sel_merge<8, 4>(x);
  x = sel_merge<16, 8>(x);
  x = sel_merge<32, 16>(x);
  x = sel_merge<64, 32>(x);
  return x;
}

constexpr int SEL_SURV = 72;   // LDS entries (64-bit) a wave needs for wave_select_smallest: 64 survivors + padding to 8

// dv[0..C) distances in LDS (wave-private).  On success every survivor lane p < S holds one selected candidate:
// rank < T -> (rank, index) is an output pair; other lanes get rank = -1.  Returns false if the caller must fall back.
QINCO_DEV bool wave_select_smallest(const float* dv, int C, int T, unsigned long long* surv, int lane, int& rank, int& index) {
  rank = -1;
  index = 0;
  if (T > 64) return false;
  unsigned lm = 0xffffffffu;
  for (int k = lane; k < C; k += 64) lm = umin(lm, sel_key(dv[k]));
  const unsigned sorted = wave_sort64(lm);
  const unsigned tau = (unsigned)__builtin_amdgcn_readlane((int)sorted, T - 1);
  int S = 0;
  for (int k0 = 0; k0 < C; k0 += 64) {
    const int k = k0 + lane;
    const unsigned key = k < C ? sel_key(dv[k]) : 0xffffffffu;
    const bool m = k < C && key <= tau;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(m);
    const int pos = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (m && pos < 64) surv[pos] = ((unsigned long long)key << 32) | (unsigned)k;
    S += __builtin_popcountll(mask);
  }
  if (S > 64) return false;
  if (lane < 8) surv[S + lane] = ~0ull;   // pad to a multiple of 8 for the unrolled count
  __builtin_amdgcn_wave_barrier();
  const unsigned long long mine = lane < S ? surv[lane] : ~0ull;
  int r = 0;
  for (int j0 = 0; j0 < S; j0 += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) r += surv[j0 + u] < mine ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < S && r < T) {
    rank = r;
    index = (int)(unsigned)mine;
  }
  return true;
}

// The T smallest of dv[0..C) (LDS, wave-private; consumed), ascending, ties -> lower index: their flat indices into sel[0..T) (LDS).
// The selection of QINCoInferenceStepEncoder.forward's topk(F_out) (qinco_inference.py:200) as beam_select_kernel and the fused
// epilogue of mlp_kernel (SELEP) both run it: threshold-and-compact, or -- T == 1 (one arg-min round is already minimal), more
// than 64 survivors, massive ties -- T rounds of arg-min.
QINCO_DEV void wave_top_t(float* dv, int C, int T, unsigned long long* surv, int* sel, int lane) {
  int rank, index;
  if (T > 1 && wave_select_smallest(dv, C, T, surv, lane, rank, index)) {
    if (rank >= 0) sel[rank] = index;
  } else {
    for (int t = 0; t < T; ++t) {
      float bv = __builtin_inff();
      int bi = 0x7fffffff;
      for (int k = lane; k < C; k += 64) {  // k ascends: strict < keeps the lowest index
        const float v = dv[k];
        const bool take = v < bv;
        bv = take ? v : bv;
        bi = take ? k : bi;
      }
      wave_argmin(bv, bi);
      if (bi == 0x7fffffff) bi = 0;           // only NaN / +inf left: degenerate, keep in range
      if ((bi & 63) == lane) dv[bi] = __builtin_inff();
      if (lane == 0) sel[t] = bi;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// Lane-pair selection (round 5): the T smallest of a group's K = 256 distances while they still sit in the MFMA's C layout.
//
// A 32 x 256 table tile of a wave leaves the matrix pipe with lane (j, half) holding 128 of group j's distances (codeword
// k = 32 cb + 8 gq + 4 half + e in register 16 cb + 4 gq + e).  The wave-wide selection above takes ONE group at a time across the
// 64 lanes (sort of the lane minima, ballot / mbcnt compaction, rank by counting: ~240 VALU instructions per group, 7.6 k per tile --
// as long as the tile's 512 MFMAs; counters in profiles/r04_S_khead_pmc_*).  Here every lane pair (l, l ^ 32) works on ITS group and
// all 32 groups advance with every instruction, no cross-lane step except the exchange with the partner:
//   1. threshold: per lane 32 quad minima, its 16 smallest sorted in registers (two 63-comparator networks + a merge); with the
//      partner's the T-th smallest of the group's 64 quad minima tau bounds its T-th smallest distance from above (T quads hold an
//      element <= tau); T <= 16;
//   2. survivors: one pass over the 128 registers, "always write, advance when it survives": (distance bits, codeword) go to the
//      lane's list in LDS ([entry][lane]: conflict-free), ~18 survivors per group for T = 16;
//   3. order: the group's survivors -- lane 0's list, then lane 1's -- as 32 keys, 16 per lane, sorted by the same network and one
//      bitonic merge across the pair.  A key is a DOUBLE in [1, 2) whose mantissa is (ordered distance bits << 8 | codeword): doubles of
//      one binade order like their mantissas, so a compare-exchange is v_min_f64 + v_max_f64 instead of a 64-bit compare and four
//      selects.  Exactly the lexicographic (distance, index) order (ties -> lower index: argmin / stable argsort,
//      qinco_inference.py:173,200; -0.0 == +0.0).
// ~2.1 k VALU instructions per 32 groups.  Exactness does not depend on step 1: the survivors are {d <= tau}, so whenever a group has
// S >= T of them they contain its T smallest, ties included; a group with S < T (NaN / infinite distances in the minima) or with more
// survivors than the lists hold (massive exact ties) is redone -- with T == 1 and T > 32 -- by rounds of exact arg-min over the
// registers (pair_rounds: slow, always right, NaN last).
// ---------------------------------------------------------------------------------------------
constexpr int PAIR_LIST = 24;                   // survivors a lane can hold (one more entry absorbs the writes past the end)
constexpr int PAIR_GROUP_MAX = 32;              // survivors of a group the sort takes
constexpr int PAIR_T_MAX = 16;                  // largest T of the fast path: for T = 17 .. 32 a threshold from 32 bucket minima leaves ~100 survivors
constexpr unsigned PAIR_SENTINEL = 0x7fffffffu; // distance bits of an empty list entry (a positive NaN: never a survivor's)
// unsigned of LDS per wave: [entry][lane][distance bits, codeword]
template <int NL = 64>
constexpr int pair_lds_words() {
  static_assert(NL == 64, "one list per lane");
  return (PAIR_LIST + 1) * NL * 2;
}

constexpr int kSort16[63][2] = {
    {0, 1},   {2, 3},   {4, 5},   {6, 7},   {8, 9},   {10, 11}, {12, 13}, {14, 15}, {0, 2},   {1, 3},   {4, 6},   {5, 7},   {8, 10},
    {9, 11},  {12, 14}, {13, 15}, {1, 2},   {5, 6},   {9, 10},  {13, 14}, {0, 4},   {1, 5},   {2, 6},   {3, 7},   {8, 12},  {9, 13},
    {10, 14}, {11, 15}, {2, 4},   {3, 5},   {10, 12}, {11, 13}, {1, 2},   {3, 4},   {5, 6},   {9, 10},  {11, 12}, {13, 14}, {0, 8},
    {1, 9},   {2, 10},  {3, 11},  {4, 12},  {5, 13},  {6, 14},  {7, 15},  {4, 8},   {5, 9},   {6, 10},  {7, 11},  {2, 4},   {3, 5},
    {6, 8},   {7, 9},   {10, 12}, {11, 13}, {1, 2},   {3, 4},   {5, 6},   {7, 8},   {9, 10},  {11, 12}, {13, 14}};   // Batcher, 0-1 checked
constexpr int kMerge16[32][2] = {{0, 8},  {1, 9},  {2, 10}, {3, 11},  {4, 12},  {5, 13},  {6, 14},  {7, 15},  {0, 4},   {1, 5},   {2, 6},
                                 {3, 7},  {8, 12}, {9, 13}, {10, 14}, {11, 15}, {0, 2},   {1, 3},   {4, 6},   {5, 7},   {8, 10},  {9, 11},
                                 {12, 14}, {13, 15}, {0, 1}, {2, 3},  {4, 5},   {6, 7},   {8, 9},   {10, 11}, {12, 13}, {14, 15}};   // bitonic -> ascending

template <class F, int... Is>
QINCO_DEV void sel_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f.template operator()<Is>(), ...);
}
template <int N, class F>
QINCO_DEV void sel_static_for(F&& f) {
  sel_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// v_min / v_max as the instructions themselves: __builtin_fminf is llvm.minnum, in front of which hipcc canonicalises every operand it
// cannot prove quiet (a v_max_f32 x, x per MFMA result: 128 + 80 extra instructions here).  A signalling NaN cannot reach these: the
// distances are results of VALU arithmetic, the keys are constructed.  (Plain asm, not volatile: free to schedule.)
QINCO_DEV float pair_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
QINCO_DEV float pair_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
QINCO_DEV float pair_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
QINCO_DEV double pair_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
QINCO_DEV double pair_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// compare-exchange: v_min + v_max (f32 or f64)
template <class V>
QINCO_DEV void pair_cex(V& a, V& b) {
  const V lo = pair_min(a, b), hi = pair_max(a, b);
  a = lo;
  b = hi;
}
template <class V>
QINCO_DEV void pair_sort16(V (&v)[16]) {
  sel_static_for<63>([&]<int i>() __attribute__((always_inline)) { pair_cex(v[kSort16[i][0]], v[kSort16[i][1]]); });
}
template <class V>
QINCO_DEV void pair_merge16(V (&v)[16]) {
  sel_static_for<32>([&]<int i>() __attribute__((always_inline)) { pair_cex(v[kMerge16[i][0]], v[kMerge16[i][1]]); });
}
// the value lane l ^ 32 holds (v_permlane32_swap: the upper half of vdst trades places with the lower half of src)
QINCO_DEV unsigned pair_partner_u(unsigned v, int half) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return half ? (unsigned)r[0] : (unsigned)r[1];
}
QINCO_DEV float pair_partner(float v, int half) { return __builtin_bit_cast(float, pair_partner_u(__builtin_bit_cast(unsigned, v), half)); }
QINCO_DEV double pair_partner(double v, int half) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = pair_partner_u((unsigned)u, half), hi = pair_partner_u((unsigned)(u >> 32), half);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// codeword of register slot r of a lane (MFMA 32x32 C layout, see above)
constexpr unsigned pair_slot_codeword(int r) { return (unsigned)(((r >> 4) << 5) | (((r >> 2) & 3) << 3) | (r & 3)); }
QINCO_DEV double pair_key(unsigned dbits, unsigned k) {
  const unsigned key = dbits == PAIR_SENTINEL ? 0xffffffffu : ordered_bits(__builtin_bit_cast(float, dbits));
  return __builtin_bit_cast(double, ((unsigned long long)(0x3ff00000u | (key >> 24)) << 32) | (unsigned long long)((key << 8) | k));
}

// Exact rounds: rank t = the smallest (distance key, codeword) above rank t - 1, found over the registers of both lanes.
template <int NKB>
QINCO_DEV void pair_rounds(const f32x16 (&acc)[NKB], int half, int T, bool mine, int* __restrict__ out) {
  const unsigned kh = (unsigned)half << 2;
  unsigned long long prev = 0;
  for (int t = 0; t < T; ++t) {
    unsigned long long best = ~0ull;
#pragma unroll
    for (int r = 0; r < NKB * 16; ++r) {
      const unsigned long long key = ((unsigned long long)sel_key(acc[r >> 4][r & 15]) << 32) | (pair_slot_codeword(r) | kh);
      const bool take = (t == 0 || key > prev) && key < best;
      best = take ? key : best;
    }
    const unsigned plo = pair_partner_u((unsigned)best, half), phi = pair_partner_u((unsigned)(best >> 32), half);
    const unsigned long long pb = ((unsigned long long)phi << 32) | plo;
    best = pb < best ? pb : best;
    if (mine && half == 0) out[t] = best == ~0ull ? 0 : (int)(unsigned)best;   // (only NaNs left and T > their number: degenerate, keep in range)
    prev = best;
  }
}

// acc: the lane's 16 NKB distances; lists: pair_lds_words<64>() unsigned of LDS private to this wave; out: ids_out + group * T
// (written when store).  All 64 lanes must be active; lanes whose group is a copy (past the end of the launch) pass store = false.
template <int NKB>
QINCO_DEV void pair_top_t(const f32x16 (&acc)[NKB], int lane, int T, unsigned* lists, int* __restrict__ out, bool store,
                          int* __restrict__ went_to_rounds = nullptr) {
  static_assert(NKB == 8, "32 quads per lane: K = 256");
  constexpr int NL = 64, ES = NL * 2;                                              // lists per wave, words per entry row
  const int half = lane >> 5;
  const int li = lane;                                                             // this lane's list
  bool redo = true;   // this lane's group still has to go through the rounds
  int dbg_S = 0;
  if (T >= 2 && T <= PAIR_T_MAX) {   // (wave-uniform)
    // ---- 1. tau: the T-th smallest of the group's 64 quad minima (a quad = the four registers e = 0 .. 3 of a (cb, gq)).  Only a
    // lane's 16 smallest minima can matter: two sorted halves of 16, the lower 16 of their union (bitonic), sorted.
    // (Round 5's first version used 32 buckets of 8: ~21 survivors for T = 16 and one group in 2500 above the 32 the sort takes --
    // and ONE such group in a launch sends its wave through 16 exact rounds, ~50 us during which the rest of the chip idles:
    // 102 -> 170 us per launch.  With 64 buckets of 4: 17.6 survivors on average, 28 at most in 4 x 10^5 random groups.)
    float m[16];
    {
      float qa[16], qb[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const f32x16& lo = acc[b >> 2];
        const f32x16& hi = acc[4 + (b >> 2)];
        const int o = (b & 3) * 4;
        qa[b] = pair_min(pair_min3(lo[o], lo[o + 1], lo[o + 2]), lo[o + 3]);
        qb[b] = pair_min(pair_min3(hi[o], hi[o + 1], hi[o + 2]), hi[o + 3]);
      }
      pair_sort16(qa);
      pair_sort16(qb);
#pragma unroll
      for (int i = 0; i < 16; ++i) m[i] = pair_min(qa[i], qb[15 - i]);
    }
    pair_merge16(m);
    float z[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = pair_min(m[i], pair_partner(m[15 - i], half));   // the 16 smaller of the two sorted lists: bitonic
    pair_merge16(z);
    float tau = z[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tau = ((T - 1) & 15) == i ? z[i] : tau;
    // ---- 2. survivors {d <= tau} -> this lane's list, in register order
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* const mylist = (lds_u32*)lists + li * 2;   // entry e at mylist[e * ES + {0, 1}]  (32-bit LDS pointers: no address arithmetic per store)
#pragma unroll
    for (int e = 0; e <= PAIR_LIST; ++e) mylist[e * ES] = PAIR_SENTINEL;
    const unsigned kh = (unsigned)half << 2;
    lds_u32* cur = mylist;
    lds_u32* const lim = mylist + PAIR_LIST * ES;   // entry PAIR_LIST absorbs an over-long list
#pragma unroll
    for (int r = 0; r < NKB * 16; ++r) {
      const float d = acc[r >> 4][r & 15];
#if defined(QINCO_EXPERIMENT) && QINCO_PAIR_EXP == 1   // timing only (wrong results): every write goes to entry 0
      mylist[0] = __builtin_bit_cast(unsigned, d);
      mylist[1] = pair_slot_codeword(r) | kh;
#else
      cur[0] = __builtin_bit_cast(unsigned, d);
      cur[1] = pair_slot_codeword(r) | kh;
#endif
      lds_u32* const nxt = cur + (d <= tau ? ES : 0);
      cur = nxt < lim ? nxt : lim;
    }
    cur[0] = PAIR_SENTINEL;                      // the entry behind the last survivor holds the last register's distance
    const int cnt = (int)((cur - mylist) / ES);
    const int cnt_p = (int)pair_partner_u((unsigned)cnt, half);
    const int cnt0 = half ? cnt_p : cnt, S = cnt + cnt_p;
    const bool ok = cnt < PAIR_LIST && cnt_p < PAIR_LIST && S >= T && S <= PAIR_GROUP_MAX;
    __builtin_amdgcn_wave_barrier();
    // ---- 3. the group's survivors (lane 0's, then lane 1's; sentinels behind them), 16 per lane, sorted
    double key[16];
    const unsigned* const list0 = lists + (li & (NL / 2 - 1)) * 2;
    const unsigned* const list1 = lists + (li | (NL / 2)) * 2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = 16 * half + i;                                  // position in the group's list
      const bool second = p >= cnt0;
      int e = second ? p - cnt0 : p;
      e = e < PAIR_LIST ? e : PAIR_LIST;                            // (behind both lists: a sentinel entry)
      const unsigned* src = (second ? list1 : list0) + e * ES;
      const unsigned long long both = *reinterpret_cast<const unsigned long long*>(src);
      // (lane 0's entries behind its last survivor are sentinels too, but the group's list continues with lane 1's: p >= cnt0 reads there)
      key[i] = pair_key((unsigned)both, (unsigned)(both >> 32));
    }
    __builtin_amdgcn_wave_barrier();
    pair_sort16(key);
    double w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = pair_partner(key[15 - i], half);   // (the exchange needs both lanes of the pair: not under the branch)
    if (half == 0) {   // lane 0 keeps the 16 smaller of the 32, lane 1 the 16 larger: bitonic sequences
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = pair_min(key[i], w[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = pair_max(key[i], w[i]);
    }
    pair_merge16(w);
    if (ok && store) {   // lane `half` holds ranks 16 half .. 16 half + 15
      int* o = out + 16 * half;
      if ((T & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (16 * half + 4 * q < T) {
            int4 v;
            v.x = (int)((unsigned)__builtin_bit_cast(unsigned long long, w[4 * q]) & 0xffu);
            v.y = (int)((unsigned)__builtin_bit_cast(unsigned long long, w[4 * q + 1]) & 0xffu);
            v.z = (int)((unsigned)__builtin_bit_cast(unsigned long long, w[4 * q + 2]) & 0xffu);
            v.w = (int)((unsigned)__builtin_bit_cast(unsigned long long, w[4 * q + 3]) & 0xffu);
            *reinterpret_cast<int4*>(o + 4 * q) = v;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (16 * half + i < T) o[i] = (int)((unsigned)__builtin_bit_cast(unsigned long long, w[i]) & 0xffu);
      }
    }
    redo = !ok;
    dbg_S = S;
  }
  // (the self-test's view, qinco_selftest / scripts/exp_pair_select.py: bit `half` = this lane sends the group to the rounds, byte
  // 1 + half = the survivors it counted)
  if (went_to_rounds && store) atomicOr(went_to_rounds, (redo ? 1 << half : 0) | ((dbg_S & 255) << (8 + 8 * half)));
  // the pair's lanes agree on `redo` (both computed S); a wave without such a group skips the rounds
#if defined(QINCO_EXPERIMENT) && QINCO_PAIR_EXP == 2   // timing only: no rounds behind the fast path
  if (T >= 2 && T <= PAIR_T_MAX) return;
#endif
  if (__builtin_amdgcn_ballot_w64(redo) != 0) pair_rounds<NKB>(acc, half, T, redo && store, out);
}

// wave_sort64 of GP independent values, one compare-exchange step at a time across all of them (a dependent DPP chain needs
// wait states after every step; GP chains interleaved need none)
template <int K, int J, int GP>
QINCO_DEV void sel_merge_multi(unsigned (&x)[GP]) {
#pragma unroll
  for (int u = 0; u < GP; ++u) x[u] = sel_cmpx<K, J>(x[u]);
  if constexpr (J > 1) sel_merge_multi<K, J / 2, GP>(x);
}
template <int GP>
QINCO_DEV void wave_sort64_multi(unsigned (&x)[GP]) {
  sel_merge_multi<2, 1, GP>(x);
  sel_merge_multi<4, 2, GP>(x);
  sel_merge_multi<8, 4, GP>(x);
  sel_merge_multi<16, 8, GP>(x);
  sel_merge_multi<32, 16, GP>(x);
  sel_merge_multi<64, 32, GP>(x);
}

// The same selection for GP groups of 64 * NV distances side by side (rows of an LDS table, `ld` floats apart): one group's
// selection is a chain of dependent cross-lane steps and VALU -> SALU -> VALU round trips (~3700 cycles alone on a SIMD that
// holds a single wave), GP independent chains interleave in the instruction stream.  Returns a bit per group: 1 = selected
// (lanes with rank[u] >= 0 hold an output pair), 0 = fall back to the rounds for that group.
template <int GP, int NV>
QINCO_DEV unsigned wave_select_smallest_multi(const float* tab, int ld, int T, unsigned long long* surv, int lane, int (&rank)[GP],
                                              int (&index)[GP]) {
  unsigned key[GP][NV], lm[GP], tau[GP];
  int S[GP];
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    lm[u] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      key[u][i] = sel_key(tab[u * ld + lane + 64 * i]);
      lm[u] = umin(lm[u], key[u][i]);
    }
  }
  wave_sort64_multi<GP>(lm);
#pragma unroll
  for (int u = 0; u < GP; ++u) tau[u] = (unsigned)__builtin_amdgcn_readlane((int)lm[u], T - 1);
  unsigned long long mask[GP][NV];
#pragma unroll
  for (int u = 0; u < GP; ++u)
#pragma unroll
    for (int i = 0; i < NV; ++i) mask[u][i] = __builtin_amdgcn_ballot_w64(key[u][i] <= tau[u]);
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    int base = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask[u][i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[u][i], 0u));
      if (key[u][i] <= tau[u] && pos < 64) surv[u * SEL_SURV + pos] = ((unsigned long long)key[u][i] << 32) | (unsigned)(lane + 64 * i);
      base += __builtin_popcountll(mask[u][i]);
    }
    S[u] = base;
  }
  unsigned ok = 0;
  int smax = 0;
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    const int s = S[u] < 64 ? S[u] : 64;
    if (s + lane < SEL_SURV) surv[u * SEL_SURV + s + lane] = ~0ull;   // everything behind the survivors compares as "not smaller"
    if (S[u] <= 64) ok |= 1u << u;
    smax = s > smax ? s : smax;
  }
  __builtin_amdgcn_wave_barrier();
  unsigned long long mine[GP];
  int r[GP];
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    mine[u] = lane < S[u] && lane < 64 ? surv[u * SEL_SURV + lane] : ~0ull;
    r[u] = 0;
  }
  for (int j0 = 0; j0 < smax; j0 += 8) {
#pragma unroll
    for (int u = 0; u < GP; ++u)
#pragma unroll
      for (int v = 0; v < 8; ++v) r[u] += surv[u * SEL_SURV + j0 + v] < mine[u] ? 1 : 0;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    const bool sel = lane < S[u] && r[u] < T && ((ok >> u) & 1);
    rank[u] = sel ? r[u] : -1;
    index[u] = (int)(unsigned)mine[u];
  }
  return ok;
}

// The four waves of a workgroup have filled a shared (32 x LDK) distance table (one row per group, K distances); wave `wave`
// selects the T smallest of its 8 rows, ascending, ties -> lower index, into ids_out[(g0 + row) * T + rank].  The tail of the
// cooperative small-launch table kernels (dist_topk_mfma_coop_kernel, presel_xproj_coop_kernel).
template <int K, int LDK, int SGP>
QINCO_DEV void coop_select_rows(float* table, unsigned long long* surv, int lane, int wave, long g0, long G, int T, int* __restrict__ ids_out) {
  const long gend = G - g0 < 32 ? G - g0 : 32;
  const int r_lo = wave * 8, r_hi = r_lo + 8 < gend ? r_lo + 8 : (int)gend;   // this wave's rows of the table
  auto rounds = [&](int r) {   // T rounds of wave arg-min on one row
    float* dg = table + r * LDK;
    for (int t = 0; t < T; ++t) {
      float bv = __builtin_inff();
      int bi = 0x7fffffff;
#pragma unroll
      for (int k = lane; k < K; k += 64) {
        const float v = dg[k];
        const bool take = v < bv;
        bv = take ? v : bv;
        bi = take ? k : bi;
      }
      wave_argmin(bv, bi);
      if (bi == 0x7fffffff) bi = 0;
      if (lane == 0) ids_out[(g0 + r) * T + t] = bi;
      if ((bi & 63) == lane) dg[bi] = __builtin_inff();
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (T > 1 && T <= 64) {
    for (int r0 = r_lo; r0 < r_hi; r0 += SGP) {
      int rank[SGP], index[SGP];
      const unsigned ok = wave_select_smallest_multi<SGP, K / 64>(table + r0 * LDK, LDK, T, surv, lane, rank, index);
#pragma unroll
      for (int u = 0; u < SGP; ++u) {
        const int r = r0 + u;
        if (r >= r_hi) break;
        if ((ok >> u) & 1) {
          if (rank[u] >= 0) ids_out[(g0 + r) * T + rank[u]] = index[u];
          continue;
        }
        rounds(r);   // massive exact ties
      }
    }
  } else {
    for (int r = r_lo; r < r_hi; ++r) rounds(r);   // T == 1 or T > 64 (see dist_topk_mfma_kernel)
  }
}

}  // namespace qinco
