// The pre-selection table + top-T on the matrix cores (K = 256): dist_topk_mfma_kernel for large launches, its cooperative
// form for small ones.  Templates only: instantiated for the dataset dimensions of the reference in the C-ABI translation unit
// (qinco_hip.hip, D in {32, 96, 128, 256, 768}) and, for any other D, in a kernel-instance module built on demand
// (mlp_inst.hip with -DQINCO_INSTANCE_MODULE: TableArgs / qinco_table_launch).
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_args.hpp"
#include "select.hpp"

namespace qinco {

// ---------------------------------------------------------------------------------------------
// K1+K2 on the matrix cores: the residual -> codebook table of dist_topk_kernel (aux_kernels.hpp) for K = 32*NKB
// codewords, evaluated like the IVF table (a wave = 32 groups as B operands, codebook fragments as A operands), then the
// T smallest per group, ascending.  r = x - xhat is formed on load (blocks are streamed, so any D fits), |r|^2 on the fly.
// <= 256 registers and 50 KiB of LDS, so that TWO workgroups share a CU: the table phase (matrix pipe) and the selection phase (VALU)
// of a wave do not overlap with themselves, only with another wave's.
// Selection (round 5): in the registers the MFMA left the distances in -- every lane pair selects its own group, all 32 groups of the
// tile at once (select.hpp pair_top_t: bucket-minimum threshold, one compaction pass, a 32-key sort on v_min_f64 / v_max_f64); T = 1,
// T > 32 and degenerate groups take exact arg-min rounds over the registers.
// (history: VALU table 621 us per 65 536 groups -> MFMA table + T arg-min rounds 235 us -> table through LDS + wave-wide
// threshold-and-compact per group 142 us per 131 072 (rounds 2-4, ~240 VALU instructions per group) -> this form)
// ---------------------------------------------------------------------------------------------
template <int D, int NKB>
__global__ void __launch_bounds__(256, 2)
dist_topk_mfma_kernel(const float* __restrict__ x, const float* __restrict__ xhat, int F,
                      const f32x4* __restrict__ cstream, const float* __restrict__ cnorm, long G, int T,
                      int* __restrict__ ids_out, int gpw) {
  // gpw = groups per wave (32, or 8 for small launches, which could not fill the chip with 32-group waves; lanes past gpw
  // repeat the last group in the MFMA tile and are not selected)
  constexpr int NDB = D / 32;
  __shared__ __attribute__((aligned(16))) unsigned pair_lists[4 * pair_lds_words<64>()];   // 12.5 KiB per wave: the survivors' lists
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long g0 = ((long)blockIdx.x * 4 + wave) * gpw;
  if (g0 >= G) return;  // wave-uniform; only wave-level ordering below
  long g = g0 + (j < gpw ? j : gpw - 1);
  if (g >= G) g = G - 1;
  const float* xp = x + (g / F) * D + half * 4;
  const float* hp = xhat ? xhat + g * D + half * 4 : nullptr;
  const f32x4* wp = cstream + lane;
  f32x16 acc[NKB];
#pragma unroll
  for (int cb = 0; cb < NKB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
  // Codebook fragments in the order they are used (feature block, q, codeword block) through a fenced register ring,
  // the raw x / xhat rows of the next feature block fetched before the current block's MFMAs.
  constexpr int NFR = NDB * 4 * NKB, P = 8;
  auto frag_ofs = [](int i) { return (((i % NKB) * NDB + i / (4 * NKB)) * 4 + (i / NKB) % 4) * 64; };
  f32x4 ring[P];
#pragma unroll
  for (int i = 0; i < P; ++i) ring[i] = wp[frag_ofs(i)];
  f32x4 xr[4], hr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xr[q] = *reinterpret_cast<const f32x4*>(xp + 8 * q);
    if (hp) hr[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
  }
  float rn = 0.f;
  static_assert((4 * NKB) % P == 0, "ring slots must not depend on the feature block");
  constexpr int IB_UNROLL = NDB > 2 ? 1 : NDB;  // feature-block loop rolled: 255 registers, no spills (2 waves per SIMD); unrolled it spills and is slower
#pragma unroll IB_UNROLL
  for (int ib = 0; ib < NDB; ++ib) {
    f32x16 rb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = xr[q];
      if (hp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = __fsub_rn(t[e], hr[q][e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rb[4 * q + e] = t[e];
        rn = fmaf(t[e], t[e], rn);
      }
    }
    if (ib + 1 < NDB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xr[q] = *reinterpret_cast<const f32x4*>(xp + (ib + 1) * 32 + 8 * q);
        if (hp) hr[q] = *reinterpret_cast<const f32x4*>(hp + (ib + 1) * 32 + 8 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) {
        const int i = (ib * 4 + q) * NKB + cb;
        const f32x4 w = ring[(q * NKB + cb) % P];
        ring[(q * NKB + cb) % P] = wp[frag_ofs(i + P)];  // past the end: inside the 16-fragment padding of the stream
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], rb[4 * q + e], acc[cb], 0, 0, 0);
      }
  }
  static_assert(NFR > 0, "");
  rn += __shfl_xor(rn, 32);
  // distances in place of the dot products: (|r|^2 + |c|^2) - 2 r.c  (the reference's association, utils.py:336-346)
#pragma unroll
  for (int cb = 0; cb < NKB; ++cb)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 cn = *reinterpret_cast<const f32x4*>(cnorm + cb * 32 + 8 * gq + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][4 * gq + e] = __fsub_rn(__fadd_rn(rn, cn[e]), __fmul_rn(2.f, acc[cb][4 * gq + e]));
    }
  // selection in the layout the matrix pipe left the distances in: the lane pair (j, j + 32) owns group j (select.hpp pair_top_t)
  const int gend = (int)((G - g0) < gpw ? (G - g0) : gpw);
  pair_top_t<NKB>(acc, lane, T, pair_lists + wave * pair_lds_words<64>(), ids_out + (g0 + j) * (long)T, j < gend);
}

// ---------------------------------------------------------------------------------------------
// The same table + top-T for SMALL launches (G <= 16 384 groups, e.g. the reference's default batch of 1024 vectors with 8
// beams): dist_topk_mfma_kernel fills the chip only with 8 groups per wave, and each of those waves still pays a full 32-column
// table (32.8 k MFMA cycles at D = 128) before it selects anything: 27 us per step at 8192 groups, 18 % of a split-form
// qinco2-S step.  Here the four waves of a workgroup share 32 groups: every wave computes a quarter of the codewords for all 32
// (8.2 k MFMA cycles), the distances meet in one LDS table, and after a barrier each wave selects 8 of the groups.
// ---------------------------------------------------------------------------------------------
template <int D, int NKB>
__global__ void __launch_bounds__(256)
dist_topk_mfma_coop_kernel(const float* __restrict__ x, const float* __restrict__ xhat, int F,
                           const f32x4* __restrict__ cstream, const float* __restrict__ cnorm, long G, int T,
                           int* __restrict__ ids_out) {
  constexpr int NDB = D / 32, K = NKB * 32, LDK = K + 4, SGP = 4, CPW = NKB / 4;   // codeword blocks per wave
  static_assert(NKB % 4 == 0, "the codeword blocks are split over four waves");
  __shared__ __attribute__((aligned(16))) float table[32 * LDK];
  __shared__ unsigned long long surv_all[4 * SGP * SEL_SURV];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  const long g0 = (long)blockIdx.x * 32;
  long g = g0 + j;
  if (g >= G) g = G - 1;
  const float* xp = x + (g / F) * D + half * 4;
  const float* hp = xhat ? xhat + g * D + half * 4 : nullptr;
  // fragment (cb, ib, q) of the stream = 64 lanes x float4 at ((cb * NDB + ib) * 4 + q) * 64
  const f32x4* wp = cstream + (long)(wave * CPW) * NDB * 4 * 64 + lane;
  f32x16 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float rn = 0.f;
  // fragments in consumption order (feature block, codeword block of this wave, q) through a register ring 16 deep, the groups' rows
  // of the next feature block requested before the current block's MFMAs (round 5: loaded in front of their MFMAs the fragments each
  // paid an L2 round trip -- see presel_kernel.hpp)
  constexpr int NF = NDB * CPW * 4, PF = NF < 16 ? NF : 16;
  auto fragment = [&](int i) -> f32x4 {
    const int ib = i / (CPW * 4), r = i % (CPW * 4);
    return wp[(((r / 4) * NDB + ib) * 4 + r % 4) * 64];
  };
  f32x4 ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) ring[i] = fragment(i);
  f32x4 xr[4], hr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xr[q] = *reinterpret_cast<const f32x4*>(xp + 8 * q);
    if (hp) hr[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
  }
  // (wide data: the feature-block loop unrolled by two only -- a ring slot is (8 ib + r) % 16 --; unrolled 24 times the D = 768
  // instance needs 512 registers and spills inside the loop)
  constexpr int IBU = (NDB > 8 && NDB % 2 == 0 && CPW * 4 * 2 == PF) ? 2 : NDB;
#pragma unroll IBU
  for (int ib = 0; ib < NDB; ++ib) {
    f32x16 rb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = xr[q];
      if (hp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = __fsub_rn(t[e], hr[q][e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        rb[4 * q + e] = t[e];
        rn = fmaf(t[e], t[e], rn);
      }
    }
    if (ib + 1 < NDB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xr[q] = *reinterpret_cast<const f32x4*>(xp + (ib + 1) * 32 + 8 * q);
        if (hp) hr[q] = *reinterpret_cast<const f32x4*>(hp + (ib + 1) * 32 + 8 * q);
      }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (ib * CPW + c) * 4 + q;
        const f32x4 w = ring[i % PF];
        if (i + PF < NF) ring[i % PF] = fragment(i + PF);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], rb[4 * q + e], acc[c], 0, 0, 0);
      }
  }
  rn += __shfl_xor(rn, 32);
  // distances (|r|^2 + |c|^2) - 2 r.c (the reference's association, utils.py:336-346) into row j of the shared table
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int cb = wave * CPW + c;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 cn = *reinterpret_cast<const f32x4*>(cnorm + cb * 32 + 8 * gq + 4 * half);
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = __fsub_rn(__fadd_rn(rn, cn[e]), __fmul_rn(2.f, acc[c][4 * gq + e]));
      *reinterpret_cast<f32x4*>(table + j * LDK + cb * 32 + 8 * gq + 4 * half) = d;
    }
  }
  __syncthreads();
  // (one wave per SIMD here and 8 groups per wave: the WAVE-wide selection, four groups side by side -- 64 lanes per group, ~9 k cycles
  // for the 8 -- beats the lane-pair form of the large-launch kernel, which would run 16 of its 64 lanes for ~18 k: measured, round 5)
  coop_select_rows<K, LDK, SGP>(table, surv_all + wave * SGP * SEL_SURV, lane, wave, g0, G, T, ids_out);
}


template <int D>
inline hipError_t launch_table_kernels(const TableArgs& a, hipStream_t st) {
  if (a.coop) {   // small launches: the four waves of a workgroup share 32 groups
    hipLaunchKernelGGL((dist_topk_mfma_coop_kernel<D, 8>), dim3((unsigned)((a.G + 31) / 32)), dim3(256), 0, st, a.x, a.xhat, a.F, a.cstream,
                       a.cnorm, a.G, a.T, a.ids_out);
  } else {
    const int gpw = a.G <= 16384 ? 8 : 32;  // (without the cooperative kernel: small launches with 8 groups per wave)
    hipLaunchKernelGGL((dist_topk_mfma_kernel<D, 8>), dim3((unsigned)((a.G + 4 * gpw - 1) / (4 * gpw))), dim3(256), 0, st, a.x, a.xhat,
                       a.F, a.cstream, a.cnorm, a.G, a.T, a.ids_out, gpw);
  }
  return hipGetLastError();
}

}  // namespace qinco
