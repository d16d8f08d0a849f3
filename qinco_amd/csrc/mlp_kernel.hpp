// Fused QINCo codeword-MLP kernel for gfx950 (MI355X / CDNA4).
//
// Computes, for R rows (row = one candidate codeword of one beam of one vector),
//     f(c, xhat) = out_proj( FFN_L(...FFN_1( concat(in_proj(c), xhat) )) ) + coeff*c
// followed by   cand = f(c, xhat) + xhat   and   dist = |x|^2 + |cand|^2 - 2 x.cand
// i.e. the body of QINCoInferenceStep.forward (reference qinco/model/qinco_inference.py:31-40,
// = QINCoStep.forward qinco/model/qinco_base.py:262-280, QConcat :60-64, QBlockFFN :93-97)
// plus the candidate / distance epilogue of QINCoInferenceStepEncoder.forward (:190-199).
//
// MI355X design (not a translation of the reference's ATen op sequence):
//  * The GEMMs are evaluated TRANSPOSED: Out^T[feat x rows] = W[feat_out x feat_in] . In^T[feat_in x rows]
//    on v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain numerics).  A wave owns 32 rows and ALL features.
//    In the 32x32 C/D layout lane l holds row (l&31) and features {(r&3)+8(r>>2)+4(l>>5)}; that is
//    exactly a legal B-operand layout for the next layer once the K-order of the weights is permuted
//    the same way (host packs the weights).  Activations therefore never leave the register file:
//    no LDS round trip, no barriers, no HBM traffic between the 2L+3 GEMMs.
//  * Weights are a single sequential stream of 1 KiB "fragments" (64 lanes x float4 = the A operands
//    of 4 consecutive MFMAs) packed on the host in exactly the order the kernel consumes them; a P-deep
//    register ring prefetches them from L2 / Infinity Cache (20 MB per step for qinco2-L: cache resident).
//  * 1 wave per SIMD (z: De/2 + h: Dh/2 accumulator registers), 4 independent waves per workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "mlp_args.hpp"

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

// relu without the NaN-canonicalising v_max x,x,x that fmaxf() costs (relu(NaN) -> 0 either way on this path)
QINCO_INL float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
QINCO_INL void relu16(f32x16& v) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = relu1(v[i]);
}

// VAR bit 0 (LAZY): chain epilogues are taken off the MFMA critical path.
//   * up-projection chains accumulate straight into y[ob]; the ReLU of block ib is applied later, inside the
//     first down-projection chain, one block ahead of its first use (its input is long finished: no drain);
//   * the residual add z[ob] += acc of chain ob is issued after the first input block of chain ob+1 (two
//     alternating accumulators); only the last chain of a layer is added eagerly.
//   Without LAZY every chain end drains the matrix pipe and runs ~64 dependent VALU ops before the next MFMA.
template <int D, int DE, int DH, int P, int VAR>
__global__ void __launch_bounds__(256, 1) mlp_kernel(MlpArgs a) {
  constexpr StreamDims SL = stream_dims(D, DE, DH, P);
  constexpr int NDB = SL.NDB, NEB = SL.NEB, NHB = SL.NHB;
  constexpr bool PROJ = SL.PROJ;
  constexpr int NYB = NHB > NEB ? NHB : NEB;
  constexpr bool LAZY = (VAR & 1) != 0;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile * 32 >= a.R) return;  // wave-uniform; no barriers / LDS in this kernel
  long row = tile * 32 + j;
  const bool valid = row < a.R;
  if (!valid) row = a.R - 1;
  const long g = row / a.A;
  const int cid = a.cand_ids ? a.cand_ids[row] : (int)(row - g * a.A);
  const float* cptr = a.codebook + (long)cid * D + half * 4;
  const float* xhptr = a.xhat + g * D + half * 4;

  // ---- weight stream with a P-deep register ring -------------------------------------------
  const f32x4* wp = a.wstream + lane;
  f32x4 ring[P];
#pragma unroll
  for (int i = 0; i < P; ++i) ring[i] = wp[i * 64];
  // take<T>(): fragment T of the current section (sections start at multiples of P).
  auto take = [&]<int T>() QINCO_LAMBDA -> f32x4 {
    f32x4 w = ring[T % P];
    ring[T % P] = wp[(T + P) * 64];
    return w;
  };
  auto skip_pad = [&]<int FROM, int TO>() QINCO_LAMBDA {
    static_for<TO - FROM>([&]<int i>() QINCO_LAMBDA { (void)take.template operator()<FROM + i>(); });
  };
  // 4 MFMAs of one fragment: acc += W[ob, 8 features of block ib] . b
  auto mfma4 = [&]<int q>(f32x16& acc, const f32x4& w, const f32x16& b) QINCO_LAMBDA {
    static_for<4>([&]<int e>() QINCO_LAMBDA { acc = QINCO_MFMA(w[e], b[4 * q + e], acc); });
  };

  f32x16 z[NEB];
  f32x16 y[NYB];

  // ---- A: z = in_proj(c)  (K-outer; c blocks streamed from the codebook, next block prefetched) ---------
  if constexpr (PROJ) {
    static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = zero16(); });
    f32x16 cb = load_block(cptr);
    static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
      f32x16 cur = cb;
      if constexpr (ib + 1 < NDB) cb = load_block(cptr + (ib + 1) * 32);
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
          f32x4 w = take.template operator()<(ib * 4 + q) * NEB + ob>();
          mfma4.template operator()<q>(z[ob], w, cur);
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
          f32x4 w = take.template operator()<(ib * 4 + q) * NEB + ob>();
          mfma4.template operator()<q>(y[ob], w, b);
        });
      });
    });
  }
  skip_pad.template operator()<NEB*(NEB + NDB) * 4, SL.T_CAT>();
  wp += SL.T_CAT * 64;
  static_for<NEB>([&]<int ob>() QINCO_LAMBDA { z[ob] = z[ob] + y[ob]; });

  // ---- D: L residual FFN blocks: z = z + W_down . relu(W_up . z)   (QBlockFFN.forward) ---------
  if constexpr (!LAZY) {
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
        f32x16 acc = zero16();
        static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            f32x4 w = take.template operator()<(ob * NEB + ib) * 4 + q>();
            mfma4.template operator()<q>(acc, w, z[ib]);
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
            f32x4 w = take.template operator()<(ob * NHB + ib) * 4 + q>();
            mfma4.template operator()<q>(acc, w, y[ib]);
          });
        });
        z[ob] = z[ob] + acc;
      });
      skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
      wp += SL.T_DOWN * 64;
    }
  } else {
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
      // up-projection: chains accumulate in place in y[ob]; no epilogue here
      static_for<NHB>([&]<int ob>() QINCO_LAMBDA {
        y[ob] = zero16();
        static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            f32x4 w = take.template operator()<(ob * NEB + ib) * 4 + q>();
            mfma4.template operator()<q>(y[ob], w, z[ib]);
          });
        });
      });
      skip_pad.template operator()<NHB * NEB * 4, SL.T_UP>();
      wp += SL.T_UP * 64;
      // down-projection: lazy ReLU one block ahead, residual adds one chain behind
      f32x16 acc[2];
      relu16(y[0]);
      static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
        acc[ob & 1] = zero16();
        static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            f32x4 w = take.template operator()<(ob * NHB + ib) * 4 + q>();
            mfma4.template operator()<q>(acc[ob & 1], w, y[ib]);
            if constexpr (q == 0 && ob == 0 && ib + 1 < NHB) relu16(y[ib + 1]);
            if constexpr (q == 0 && ib == 0 && ob > 0) z[ob - 1] = z[ob - 1] + acc[(ob - 1) & 1];
          });
        });
      });
      z[NEB - 1] = z[NEB - 1] + acc[(NEB - 1) & 1];  // the only chain end per layer that drains the pipe
      skip_pad.template operator()<NEB * NHB * 4, SL.T_DOWN>();
      wp += SL.T_DOWN * 64;
    }
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
          f32x4 w = take.template operator()<(ob * NEB + ib) * 4 + q>();
          mfma4.template operator()<q>(o, w, z[ib]);
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
}

}  // namespace qinco
