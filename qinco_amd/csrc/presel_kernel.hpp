// Small launches of an encode step (G <= 16 384 groups: the reference's default batch of 1024 vectors x 8 beams): ONE kernel
// for the two things that read the groups' (x, xhat) rows before the fused MLP runs --
//   the pre-selection table + top-T   (qinco_inference.py:171-173: r = x - xhat, approx_pairwise_distance to the substep
//                                      codebook, topk ascending)                     = dist_topk_mfma_coop_kernel, and
//   the per-group half of the folded MLP head  U = W_cat[:, De:] xhat,  Q = W_up[0] U  (mlp_kernel.hpp FOLD / FOLD2) = xproj_kernel.
// Round 2 ran them as two launches: at 8192 groups the xproj launch was 64 workgroups on 256 CUs (31 us for 5 us of MFMA work
// per wave), the table 16.6 us, plus a launch gap -- a fifth of a qinco2-S step at that batch.  Here the four waves of a
// workgroup share 32 groups: wave w computes codeword blocks [w K/128, (w+1) K/128) of the table AND output blocks
// [w De/128, ..) of U for all 32 groups from the SAME (x, xhat) registers; distances and U meet in LDS; after one barrier wave w
// computes its quarter of Q from the shared U and selects the T smallest of its 8 groups.
// Bit-identical to the two separate kernels: every accumulation chain runs in the same order (table: feature block, q, e; U: the
// same; Q: input block, q, e), the distance expression and the selection are the shared code of select.hpp.
#pragma once
#include "mlp_kernel.hpp"
#include "select.hpp"

namespace qinco {

template <int D, int DE, int DH, int NKB>
__global__ void __launch_bounds__(256) presel_xproj_coop_kernel(XprojArgs a) {
  constexpr int NDB = D / 32, NEB = DE / 32, NHB = DH / 32, K = NKB * 32, LDK = K + 4, SGP = 4;
  constexpr int CPW = NKB / 4, UPW = NEB / 4, QPW = NHB / 4;   // codeword / U / Q blocks per wave
  static_assert(NKB % 4 == 0 && NEB % 4 == 0 && NHB % 4 == 0, "blocks are split over four waves");
  __shared__ __attribute__((aligned(16))) float table[32 * LDK];
  __shared__ unsigned long long surv_all[4 * SGP * SEL_SURV];
  __shared__ f32x4 ubuf[NEB * 4 * 64];     // U blocks in the MFMA B layout (= the C/D layout they were produced in), lane-linear
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  const long g0 = (long)blockIdx.x * 32;
  long g = g0 + j;
  const bool valid = g < a.G;
  if (!valid) g = a.G - 1;
  const float* xp = a.x + (g / a.F) * D + half * 4;
  const float* hp = a.xhat + g * D + half * 4;
  const f32x4* wt = a.cstream + (long)(wave * CPW) * NDB * 4 * 64 + lane;   // fragment (cb, ib, q) at ((cb * NDB + ib) * 4 + q) * 64
  const f32x4* wu = a.wx + (long)(wave * UPW) * NDB * 4 * 64 + lane;        // fragment (ob, ib, q) likewise
  f32x16 acc[CPW], uacc[UPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) acc[c] = zero16();
#pragma unroll
  for (int o = 0; o < UPW; ++o) uacc[o] = zero16();
  float rn = 0.f;
#pragma unroll
  for (int ib = 0; ib < NDB; ++ib) {
    f32x16 rb, xt;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 xq = *reinterpret_cast<const f32x4*>(xp + ib * 32 + 8 * q);
      const f32x4 hq = *reinterpret_cast<const f32x4*>(hp + ib * 32 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = __fsub_rn(xq[e], hq[e]);
        rb[4 * q + e] = t;
        xt[4 * q + e] = hq[e];
        rn = fmaf(t, t, rn);
      }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = wt[((c * NDB + ib) * 4 + q) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c] = QINCO_MFMA(w[e], rb[4 * q + e], acc[c]);
      }
#pragma unroll
    for (int o = 0; o < UPW; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = wu[((o * NDB + ib) * 4 + q) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) uacc[o] = QINCO_MFMA(w[e], xt[4 * q + e], uacc[o]);
      }
  }
  rn += __shfl_xor(rn, 32);
  // distances (|r|^2 + |c|^2) - 2 r.c (the reference's association, utils.py:336-346) into row j of the shared table
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int cb = wave * CPW + c;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 cn = *reinterpret_cast<const f32x4*>(a.cnorm + cb * 32 + 8 * gq + 4 * half);
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = __fsub_rn(__fadd_rn(rn, cn[e]), __fmul_rn(2.f, acc[c][4 * gq + e]));
      *reinterpret_cast<f32x4*>(table + j * LDK + cb * 32 + 8 * gq + 4 * half) = d;
    }
  }
  // U: to the other waves through LDS, to the fused MLP through HBM
  float* up = a.uproj + g * DE + half * 4;
#pragma unroll
  for (int o = 0; o < UPW; ++o) {
    const int ob = wave * UPW + o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = {uacc[o][4 * q], uacc[o][4 * q + 1], uacc[o][4 * q + 2], uacc[o][4 * q + 3]};
      ubuf[(ob * 4 + q) * 64 + lane] = t;
      if (valid) *reinterpret_cast<f32x4*>(up + ob * 32 + 8 * q) = t;
    }
  }
  __syncthreads();
  if (a.wq) {   // FOLD2 (wave-uniform): this wave's output blocks of Q = W_up[0] . U
    float* qp = a.qproj + g * DH + half * 4;
    const f32x4* wq = a.wq + (long)(wave * QPW) * NEB * 4 * 64 + lane;
#pragma unroll
    for (int o = 0; o < QPW; ++o) {
      f32x16 qacc = zero16();
#pragma unroll
      for (int ib = 0; ib < NEB; ++ib) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ub = ubuf[(ib * 4 + q) * 64 + lane];
          const f32x4 w = wq[((o * NEB + ib) * 4 + q) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) qacc = QINCO_MFMA(w[e], ub[e], qacc);
        }
      }
      if (valid) {
        const int ob = wave * QPW + o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = {qacc[4 * q], qacc[4 * q + 1], qacc[4 * q + 2], qacc[4 * q + 3]};
          *reinterpret_cast<f32x4*>(qp + ob * 32 + 8 * q) = t;
        }
      }
    }
  }
  coop_select_rows<K, LDK, SGP>(table, surv_all + wave * SGP * SEL_SURV, lane, wave, g0, a.G, a.T, a.ids_out);
}

}  // namespace qinco
