// In-wave sorting / top-T selection primitives (wave64, gfx950): the lexicographic arg-min across a wave, the bitonic sort of
// the 64 lane values, and the threshold-and-compact selection of the T smallest of a row that the pre-selection table kernels
// and the beam selection are built on.  Device functions and templates only: shared by the C-ABI translation unit
// (aux_kernels.hpp, ivf_kernel.hpp) and the per-shape kernel instances (presel_kernel.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
