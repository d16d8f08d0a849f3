// Fused QINCo codeword-MLP kernel, 16-row tile form, for geometries whose activations do not fit the 32-row kernel's
// register plan (mlp_kernel.hpp keeps z = De/2 and y = Dh/2 registers per lane: De, Dh <= 384).  QINCo1 on 768-d data
// has De = D = 768 (reference config/model_args/qinco1.yaml: de = null => De = D, dh = 256).
//
// Same computation and epilogue as mlp_kernel (QINCoInferenceStep.forward, reference qinco/model/qinco_inference.py:31-40
// = QINCoStep.forward qinco_base.py:262-280; candidate + distance of QINCoInferenceStepEncoder.forward :190-199), same
// transposed-GEMM idea on v_mfma_f32_16x16x4_f32 (exact fp32): a wave owns 16 rows and all features; in the 16x16 C/D
// layout lane l holds row (l & 15) and features 4 (l >> 4) + r (r = 0..3) of a 16-feature block, which is the B-operand
// layout of the next layer when the host packs W with the matching K order -- activations stay in registers
// (z: De/4, y: max(De, Dh)/4 registers per lane).
// Weight stream: 1 KiB fragments (64 lanes x float4 = the A operands of the 4 MFMAs of one 16 x 16 weight block),
// every GEMM in K-OUTER order (for input block: for output block), so consecutive fragments accumulate into
// different register blocks (no dependent MFMA chains to manage), the residual add of the down-projection is the
// MFMA accumulation itself, and the ReLU is one pass per block.  The stream comes through the same shared LDS-DMA
// ring as mlp_kernel's SHR variant (wave w fetches the fragments = w mod 4, one raw s_barrier per 4 fragments).
// A fragment feeds 4 MFMAs x 8 passes = 128 matrix-pipe cycles (the 32-row kernel: 256), so this form needs twice the
// L2 -> LDS weight bandwidth per FLOP; it is the fallback for wide models, not the production path.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_args.hpp"
#include "mlp_kernel.hpp"

namespace qinco {

#define QINCO_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// (The bisection switches that isolated the co-residency failure of this kernel in round 2 -- profiles/r02_mlp16_bisect.log,
// DESIGN.md 3.1b -- are gone from this header; the commit history has them.)

// GK = DMAs per wave per ring boundary: 1 = a barrier / refill every 4 fragments (round 2), 2 = every 8 (VAR bit 1024; see
// mlp_kernel.hpp G8 -- a fragment feeds only 128 matrix-pipe cycles here, so the boundary is twice as expensive per FLOP).
// FOLD (VAR bit 16, round 3): the row-independent head leaves the kernel like in mlp_kernel's FOLD form -- z starts as
// T[cid] + U[group] (T per codeword: z_k + W_cat[:, :De] z_k + b, built at create; U per (vector, beam) group: W_cat[:, De:] xhat, by
// this kernel's MODE 1), so in_proj / bias / concat are neither in the stream nor executed: 2 (De + D) De of the row's FLOPs (16 % of
// a QINCo1 row at D = 768).
// MODE 1 = the group projection alone: rows are groups, y = W_x . xhat through the same ring, stored to a.uproj (stream = wx).
template <int D, int DE, int DH, int P, int GK = 1, bool FOLD = false, int MODE = 0>
__global__ void __launch_bounds__(256, 1) mlp16_kernel(MlpArgs a) {
  constexpr int GM = 4 * GK - 1;   // fragment-index mask of a ring group
  static_assert(GK == 1 || GK == 2 || GK == 4, "ring groups of 4, 8 or 16 fragments (16: measured, no gain over 8)");
  static_assert(P % 12 == 0 && P % (4 * GK) == 0 && P / 4 >= 2 * GK + 3, "ring depth against the group size");
  constexpr StreamDims SL = stream_dims(D, DE, DH, P, FOLD, false, 16);
  constexpr int NDB = SL.NDB, NEB = SL.NEB, NHB = SL.NHB;  // 16-feature blocks
  constexpr bool PROJ = SL.PROJ;
  constexpr int NYB0 = NHB > NEB ? NHB : NEB;
  constexpr int NYB = (PROJ && NDB > NYB0) ? NDB : NYB0;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const long tile = (long)blockIdx.x * 4 + wave;
  long row = tile * 16 + j;  // every wave runs (barriers); rows past R are clamped and never stored
  const bool valid = row < a.R;
  if (!valid) row = a.R - 1;
  const long g = row / a.A;
  const int cid = a.cand_ids ? a.cand_ids[row] : (int)(row - g * a.A);
  const float* cptr = a.codebook + (long)cid * D + kg * 4;
  const float* xhptr = a.xhat + g * D + kg * 4;
  auto load_blk = [](const float* p) QINCO_LAMBDA -> f32x4 { return *reinterpret_cast<const f32x4*>(p); };

  // ---- weight stream: shared LDS-DMA ring (see mlp_kernel.hpp, SHR) ----------------------------------------
  const f32x4* wp = a.wstream + lane;
  f32x4 ring[3];
  __shared__ f32x4 lds_ring[P * 64];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  f32x4* myring = lds_ring;
  const int wofs = wave_u * 64;
  auto dma = [&]<int T>() QINCO_LAMBDA {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + wofs + T * 64),
                                     (__attribute__((address_space(3))) void*)(myring + wofs + (T % P) * 64), 16, 0, 0);
  };
  auto wait_vm = [&]<int N>() QINCO_LAMBDA {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
  };
  // (P/4 - GK DMAs per wave in the prologue; "<= P/4 - GK - 1 outstanding" = every wave's first one landed)
  static_for<P / 4 - GK>([&]<int i>() QINCO_LAMBDA { dma.template operator()<4 * i>(); });
  wait_vm.template operator()<P / 4 - GK - 1>();
  __builtin_amdgcn_s_barrier();
  ring[0] = myring[lane];
  ring[1] = myring[64 + lane];
  auto take = [&]<int T>() QINCO_LAMBDA -> f32x4 {
    if constexpr ((T & GM) == 0) {   // before fragment T = 4 GK g a wave has issued P/4 - GK + GK g DMAs and needs GK g + GK + 1 landed
      wait_vm.template operator()<P / 4 - 2 * GK - 1>();
      __builtin_amdgcn_s_barrier();
      static_for<GK>([&]<int k>() QINCO_LAMBDA { dma.template operator()<T + P - 4 * GK + 4 * k>(); });
    }
    ring[(T + 2) % 3] = myring[((T + 2) % P) * 64 + lane];
    asm volatile("" ::: "memory");   // the ring reads keep their program order (see fragmm)
    return ring[T % 3];
  };
  auto skip_pad = [&]<int FROM, int TO>() QINCO_LAMBDA {
    static_for<TO - FROM>([&]<int i>() QINCO_LAMBDA { (void)take.template operator()<FROM + i>(); });
  };
  // acc (16 output features x 16 rows) += W[ob, ib] . b   -- 4 MFMAs, one per register of the input block
  auto fragmm = [&]<int T>(f32x4& acc, const f32x4& b) QINCO_LAMBDA {
    f32x4 w = take.template operator()<T>();
    // The ring recycles the slots of fragments <= T - 1 at the next barrier, so this wave's LDS read of every such fragment
    // must have COMPLETED before it arrives there.  Program order alone does not give that: hipcc moves the MFMAs -- and with
    // them the s_waitcnt lgkmcnt that completes the ds_read -- across s_barrier, and with three workgroups per CU (LDS pipe
    // ~75 % busy with this kernel's 1 KiB per wave per 128 cycles) the refill from L2 can land before a queued read executes:
    // about one wave in a hundred computed with a refilled (wrong) fragment.  The pin makes the fragment a register value at
    // this point of the program; asm volatile does not cross the barrier's fences.  (round-2 bisection)
    // The ring reads keep their program order (memory fence in take), LDS returns a wave's reads in order, so pinning the LAST
    // fragment in front of each barrier covers the earlier ones; a section whose live fragments do not end on such a fragment
    // finishes with lgkmcnt(0) (section_done).  (Pinning every fragment, the first form of the fix, cost 3.6 %.)
    if constexpr ((T & GM) == GM) pin4_v(w);
    static_for<4>([&]<int e>() QINCO_LAMBDA { acc = QINCO_MFMA16(w[e], b[e], acc); });
  };
  auto section_done = [&]() QINCO_LAMBDA {   // every LDS read of this wave has completed (once per section)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    asm volatile("" ::: "memory");
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 z[NEB];
  f32x4 y[NYB];

  if constexpr (MODE == 1) {   // U[g] = W_cat[:, De:] xhat_g   (a.A == 1: row == group)
    constexpr int T_X = round_up(NEB * NDB, P);
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { y[ob] = zero4; });
    f32x4 xb = load_blk(xhptr);
    static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
      const f32x4 b = xb;
      if constexpr (ib + 1 < NDB) xb = load_blk(xhptr + (ib + 1) * 16);
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NEB + ob>(y[ob], b); });
    });
    section_done();
    skip_pad.template operator()<NEB * NDB, T_X>();
    float* up = const_cast<float*>(a.uproj) + row * DE + kg * 4;
    if (valid) static_for<NEB>([&]<int ob>() QINCO_LAMBDA { *reinterpret_cast<f32x4*>(up + ob * 16) = y[ob]; });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  if constexpr (FOLD) {
    const float* tptr = a.ttab + (long)cid * DE + kg * 4;
    const float* uptr = a.uproj + g * DE + kg * 4;
    static_for<NEB>([&]<int ib>() QINCO_LAMBDA { z[ib] = load_blk(tptr + ib * 16) + load_blk(uptr + ib * 16); });
  } else {
  // ---- A: z = in_proj(c) ----------------------------------------------------------------------------------
  if constexpr (PROJ) {
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = zero4; });
    f32x4 cb = load_blk(cptr);
    static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
      const f32x4 cur = cb;
      if constexpr (ib + 1 < NDB) cb = load_blk(cptr + (ib + 1) * 16);
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NEB + ob>(z[ob], cur); });
    });
    section_done();
    skip_pad.template operator()<NEB * NDB, SL.T_IN>();
    wp += SL.T_IN * 64;
  } else {
    static_for<NEB>([&]<int ib>() QINCO_LAMBDA { z[ib] = load_blk(cptr + ib * 16); });
  }

  // ---- B: y = bias of the concat Linear ---------------------------------------------------------------------
  static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
    y[ob] = take.template operator()<ob>();
    pin4_v(y[ob]);   // (see fragmm: the bias fragments land straight in y and would stay in flight for a whole section)
  });
  skip_pad.template operator()<NEB, SL.T_BIAS>();
  wp += SL.T_BIAS * 64;

  // ---- C: y += W_cat . [z ; xhat]   then z = z + y   (QConcat.forward, qinco_base.py:60-64) ------------------
  {
    f32x4 xb = load_blk(xhptr);
    static_for<NEB + NDB>([&]<int ib>() QINCO_LAMBDA {
      f32x4 b;
      if constexpr (ib < NEB) {
        b = z[ib];
      } else {
        b = xb;
        if constexpr (ib + 1 < NEB + NDB) xb = load_blk(xhptr + (ib + 1 - NEB) * 16);
      }
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NEB + ob>(y[ob], b); });
    });
    section_done();
    skip_pad.template operator()<NEB*(NEB + NDB), SL.T_CAT>();
    wp += SL.T_CAT * 64;
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = z[ob] + y[ob]; });
  }
  }   // !FOLD

  // ---- D: L residual FFN blocks: z = z + W_down . relu(W_up . z)   (QBlockFFN.forward :93-97) ---------------
#pragma unroll 1
  for (int l = 0; l < a.L; ++l) {
    static_for<NHB>([&]<int ob>() QINCO_LAMBDA { y[ob] = zero4; });
    static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NHB + ob>(y[ob], z[ib]); });
    });
    static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
      static_for<4>([&]<int e>() QINCO_LAMBDA { y[ob][e] = relu1(y[ob][e]); });
    });
    section_done();
    skip_pad.template operator()<NEB * NHB, SL.T_UP>();
    wp += SL.T_UP * 64;
    // the down-projection accumulates straight into z: the residual add is the MFMA's C operand
    static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NEB + ob>(z[ob], y[ib]); });
    });
    section_done();
    skip_pad.template operator()<NHB * NEB, SL.T_DOWN>();
    wp += SL.T_DOWN * 64;
  }

  // ---- E: out_proj + epilogue: cand = (out + coeff*c) + xhat ; dist = |x|^2 + |cand|^2 - 2 x.cand ------------
  if constexpr (PROJ) {
    static_for<NDB>([&]<int ob>() QINCO_LAMBDA { y[ob] = zero4; });
    static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
      static_for<NDB>([&]<int ob>() QINCO_LAMBDA { fragmm.template operator()<ib * NDB + ob>(y[ob], z[ib]); });
    });
  }
  const long n = g / a.F;
  const float* xptr = a.x ? a.x + n * D + kg * 4 : nullptr;
  float* outp = a.cand_out + row * D + kg * 4;
  float s2 = 0.f, sx = 0.f, xn = 0.f;
  static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
    f32x4 o;
    if constexpr (PROJ) o = y[ob];
    else o = z[ob];
    if (a.add_c) o = o + load_blk(cptr + ob * 16);
    o = o + load_blk(xhptr + ob * 16);
    if (valid) *reinterpret_cast<f32x4*>(outp + ob * 16) = o;
    if (xptr) {
      const f32x4 xb = load_blk(xptr + ob * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s2 = fmaf(o[i], o[i], s2);
        sx = fmaf(o[i], xb[i], sx);
        xn = fmaf(xb[i], xb[i], xn);
      }
    }
  });
  if (a.dist_out) {
    s2 += __shfl_xor(s2, 16);
    sx += __shfl_xor(sx, 16);
    xn += __shfl_xor(xn, 16);
    s2 += __shfl_xor(s2, 32);
    sx += __shfl_xor(sx, 32);
    xn += __shfl_xor(xn, 32);
    if (valid && kg == 0) a.dist_out[row] = (xn + s2) - 2.f * sx;
  }
  // no LDS-DMA may be in flight when the wave ends (its LDS could be handed to the next workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace qinco
