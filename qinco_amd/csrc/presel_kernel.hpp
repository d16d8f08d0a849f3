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
  // Weight fragments through a register ring PF deep (round 5).  Rounds 3-4 loaded every fragment right in front of its four MFMAs:
  // hipcc kept one or two loads in flight (`s_waitcnt vmcnt(1)` in front of nearly every MFMA group), so each of a wave's 80
  // fragments paid most of an L2 round trip -- ~70 k cycles for 20 k cycles of matrix work, 29 us per launch of 8192 groups.  The
  // fragments of the whole kernel are ONE sequence in consumption order -- per feature block the wave's table fragments, then its U
  // fragments; behind the barrier its Q fragments (which depend on nothing computed here) -- and fragment i + PF is requested when
  // fragment i is consumed: 16 x 256 cycles of matrix work ahead of every load.  Same MFMAs in the same order: same bits.
  constexpr int FPB = (CPW + UPW) * 4;                 // fragments per feature block
  // (D > 256: 24 feature blocks unrolled with the ring is more than hipcc's allocator copes with -- 411 spills at (768, 128, 256) --
  // those shapes keep the load-in-front-of-use form)
  constexpr bool PIPE = NDB <= 8;
  constexpr int NF1 = NDB * FPB, NFQ = QPW * NEB * 4, PF = PIPE ? 16 : 1;
  const bool has_q = a.wq != nullptr;                  // (wave-uniform; without a Q stream the ring's tail re-reads fragment 0 of W_x, never used)
  const f32x4* const wqs = (has_q ? a.wq + (long)(wave * QPW) * NEB * 4 * 64 : a.wx) + lane;
  auto fragment = [&]<int i>() QINCO_LAMBDA -> f32x4 {
    if constexpr (i < NF1) {
      constexpr int ib = i / FPB, r = i % FPB;
      if constexpr (r < CPW * 4) return wt[(((r / 4) * NDB + ib) * 4 + r % 4) * 64];
      else return wu[((((r - CPW * 4) / 4) * NDB + ib) * 4 + r % 4) * 64];
    } else if constexpr (i < NF1 + NFQ) {
      constexpr int f = i - NF1;                       // (o, ib, q) order: ((o * NEB + ib) * 4 + q)
      return wqs[(has_q ? f : 0) * 64];
    } else {
      return f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 ring[PF];
  if constexpr (PIPE) static_for<PF>([&]<int i>() QINCO_LAMBDA { ring[i] = fragment.template operator()<i>(); });
  auto take = [&]<int i>() QINCO_LAMBDA -> f32x4 {
    if constexpr (!PIPE) return fragment.template operator()<i>();
    const f32x4 w = ring[i % PF];
    if constexpr (i + PF < NF1 + NFQ) ring[i % PF] = fragment.template operator()<i + PF>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return w;
  };
  // the groups' rows of the next feature block are requested before the current block's MFMAs
  f32x4 xr[4], hr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xr[q] = *reinterpret_cast<const f32x4*>(xp + 8 * q);
    hr[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
  }
  float rn = 0.f;
  static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
    f32x16 rb, xt;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = __fsub_rn(xr[q][e], hr[q][e]);
        rb[4 * q + e] = t;
        xt[4 * q + e] = hr[q][e];
        rn = fmaf(t, t, rn);
      }
    }
    if constexpr (ib + 1 < NDB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xr[q] = *reinterpret_cast<const f32x4*>(xp + (ib + 1) * 32 + 8 * q);
        hr[q] = *reinterpret_cast<const f32x4*>(hp + (ib + 1) * 32 + 8 * q);
      }
    }
    static_for<CPW>([&]<int c>() QINCO_LAMBDA {
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        const f32x4 w = take.template operator()<ib * FPB + c * 4 + q>();
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c] = QINCO_MFMA(w[e], rb[4 * q + e], acc[c]);
      });
    });
    static_for<UPW>([&]<int o>() QINCO_LAMBDA {
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        const f32x4 w = take.template operator()<ib * FPB + CPW * 4 + o * 4 + q>();
#pragma unroll
        for (int e = 0; e < 4; ++e) uacc[o] = QINCO_MFMA(w[e], xt[4 * q + e], uacc[o]);
      });
    });
  });
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
    static_for<QPW>([&]<int o>() QINCO_LAMBDA {
      f32x16 qacc = zero16();
      static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
        static_for<4>([&]<int q>() QINCO_LAMBDA {
          const f32x4 ub = ubuf[(ib * 4 + q) * 64 + lane];
          const f32x4 w = take.template operator()<NF1 + (o * NEB + ib) * 4 + q>();
#pragma unroll
          for (int e = 0; e < 4; ++e) qacc = QINCO_MFMA(w[e], ub[e], qacc);
        });
      });
      if (valid) {
        const int ob = wave * QPW + o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = {qacc[4 * q], qacc[4 * q + 1], qacc[4 * q + 2], qacc[4 * q + 3]};
          *reinterpret_cast<f32x4*>(qp + ob * 32 + 8 * q) = t;
        }
      }
    });
  }
  coop_select_rows<K, LDK, SGP>(table, surv_all + wave * SGP * SEL_SURV, lane, wave, g0, a.G, a.T, a.ids_out);
}

}  // namespace qinco
