// IVF step 0 with an fp16 matrix-core FILTER and an exact fp32 decision (IVFBook.quantize, reference
// qinco/model/qinco_base.py:146-163: argmin_k |x|^2 + |c_k|^2 - 2 x.c_k over ivf_K = 2^20 centroids).
//
// The exact fp32 table (ivf_assign_kernel) costs 2 D ivf_K FLOPs per vector on the fp32 MFMA (157 TFLOP/s).  The same
// table on v_mfma_f32_32x32x16_f16 runs at the 16x higher fp16 rate, and an argmin does not need the table exactly -- it
// needs the set of centroids that could be the argmin:
//   pass A  s~_k = |c_k|^2 / 2 - x~.c~_k   (x~, c~ rounded to fp16, products exact, fp32 accumulate);  m~ = min_k s~_k
//           over a SAMPLE of one eighth of the centroids (the head of the stream): m~ >= the minimum over all of them, so the
//           threshold below only gets looser -- the argmin stays a candidate, about 8 per vector instead of 1
//   pass B  the table over all centroids; every k with  s~_k <= m~ + W/2  is appended to a candidate list
//   pass C  exact fp32 distance of every candidate (sequential fmaf, the association of approx_pairwise_distance) and
//           the usual 64-bit (distance, id) atomicMin.
// W = 2 E + 2 delta bounds |d~ - d| twice plus the fp32 evaluation error twice, so the argmin of the exact pass is always
// a candidate:   E = 2 [(2u + u^2) + 4 D 2^-24] |x| cmax + 2^-22 sqrt(D) (|x| + cmax),  u = 2^-11 (RNE to fp16; the
// last term covers fp16 subnormals),  delta = 4 (D + 4) 2^-24 (|x| + cmax)^2  -- Cauchy-Schwarz, rigorous, loose by
// about sqrt(D).  Random 128-d data: about one candidate per vector from the bound, seven more from the sampled minimum.  If the list overflows, or an input is outside
// the fp16 range, a device flag makes the exact fp32 kernel (launched behind, early-exit otherwise) redo the batch.
//
// Kernel shape (passes A and B share it): a wave keeps VS x 32 vectors as fp16 B operands (VS = 2 for D <= 256), streams
// 1 KiB centroid fragments (32 centroids x 16 features of fp16) through a fenced register ring, and reduces each 32 x 32
// tile on the VALU: ONE instruction per (vector, centroid) pair (the accumulators start at -|c|^2/2), plus a rare slow
// path in pass B.
#pragma once
#include <hip/hip_runtime.h>

#include "ivf_kernel.hpp"

namespace qinco {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned f32_ordered(float d) {
  unsigned u = __builtin_bit_cast(unsigned, d);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, u);
}

struct IvfF16Args {
  const h16x8* cstream;     // fp16 centroid fragments: (block of 32, k-step of 16 features) -> 64 lanes x 8 halfs
  const float* cnorm_half;  // -|c_k|^2 / 2 (fp32, of the fp32 centroids): the accumulators' start value, loaded as is
  int nblocks, blocks_per_slice;
  const float* x;           // (N, D) normalised vectors, fp32
  long N;
  unsigned* approx_min;     // (N) ordered-uint min of s~ (pass A writes, pass B reads)
  float cmax;               // max_k |c_k| (slightly inflated)
  // pass B
  int* cand_count;
  int cand_cap;
  int* cand_vec;
  int* cand_id;
  int* overflow;            // set when the list is full or an input leaves the fp16 range
  const int* perm;          // stream position of a block -> block index (pass B: candidate ids)
};

template <int D, int MODE>
__global__ void __launch_bounds__(256)
ivf_f16_kernel(IvfF16Args a) {
  constexpr int NK = D / 16;                 // k-steps (fragments) per block of 32 centroids
#ifndef QINCO_IVF_VS
#define QINCO_IVF_VS(D) ((D) <= 128 ? 4 : ((D) <= 256 ? 2 : 1))
#endif
  // sets of 32 vectors per wave.  A 1 KiB fragment feeds VS MFMAs of 32 cycles: with 2 sets and 2 waves per SIMD the eight
  // waves of a CU ask the L1 for 128 B/clk, twice what it delivers; 4 sets (one wave per SIMD: 293 registers) ask for 32.
  // Round 2, ivf_K = 2^20, 16 384 vectors: pass A 4.37 -> 4.00 (VGPR accumulators) -> 3.25 ms (4 sets), pass B 4.35 -> 3.56 ms.
  constexpr int VS = QINCO_IVF_VS(D);
  constexpr int P = NK >= 8 ? 8 : NK;        // ring depth in fragments (divides NK for every supported D; 16 was slower, twice)
  static_assert(NK % P == 0, "ring depth must divide the fragments per block");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long v0 = ((long)blockIdx.x * 4 + wave) * (32 * VS);
  if (v0 >= a.N) return;

  h16x8 xt[VS][NK];
  float thr[VS], mn[VS];
  long vec[VS];
  bool valid[VS];
#pragma unroll
  for (int s = 0; s < VS; ++s) {
    vec[s] = v0 + 32 * s + j;
    valid[s] = vec[s] < a.N;
    if (!valid[s]) vec[s] = a.N - 1;
    const float* xp = a.x + vec[s] * D + half * 8;
    float xn = 0.f, amax = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(xp + k * 16);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(xp + k * 16 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xt[s][k][e] = (_Float16)lo[e];
        xt[s][k][4 + e] = (_Float16)hi[e];
        xn = fmaf(lo[e], lo[e], xn);
        xn = fmaf(hi[e], hi[e], xn);
        amax = fmaxf(amax, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
      }
    }
    xn += __shfl_xor(xn, 32);
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    mn[s] = 0.f;
    thr[s] = 0.f;
    if constexpr (MODE == 1) {
      const float xnorm = sqrtf(xn) * 1.000001f;
      const float u = 0.00048828125f;  // 2^-11
      const float E = 2.f * ((2.f * u + u * u) + 4.f * D * 5.9604645e-8f) * xnorm * a.cmax +
                      2.3841858e-7f * sqrtf((float)D) * (xnorm + a.cmax);
      const float dl = 4.f * (D + 4) * 5.9604645e-8f * (xnorm + a.cmax) * (xnorm + a.cmax);
      const float w_half = (E + dl) * 1.0001f;  // (2E + 2 delta) / 2, in units of s = d / 2
      thr[s] = f32_from_ordered(a.approx_min[vec[s]]) + w_half;
      if (!(amax < 60000.f) || !(thr[s] < 3.0e38f)) atomicOr(a.overflow, 1);  // outside fp16 / not finite: exact path
    }
  }

  const int cb0 = blockIdx.y * a.blocks_per_slice;
  int cb1 = cb0 + a.blocks_per_slice;
  if (cb1 > a.nblocks) cb1 = a.nblocks;
  const h16x8* wp = a.cstream + (long)cb0 * (NK * 64) + lane;
  h16x8 ring[P];
#pragma unroll
  for (int i = 0; i < P; ++i) ring[i] = wp[i * 64];

  // The accumulators start at -|c|^2/2, so the tile comes out as  -s~ = x~.c~ - |c|^2/2.  Round 2: the negated half norms
  // are LOADED straight into the accumulators -- two sets of them, the next block's start values arrive while the current
  // block is on the matrix pipe -- and the accumulators are pinned to VGPRs.  Round 1 kept them in AGPRs, where the VALU cannot
  // read: per tile of 8 MFMAs it spent 16 v_xor (negation) + 16 v_accvgpr_write + 32 v_accvgpr_read next to the 16 v_max3 that
  // do the work, all serial with the MFMAs inside the wave (40 % of the fp16 peak).
  float nthr[VS];
#pragma unroll
  for (int s = 0; s < VS; ++s) {
    nthr[s] = -thr[s];
    mn[s] = -__builtin_inff();  // running maximum of -s~
  }
  f32x16 acc[2][VS];
  // (one set is loaded, the others are register copies: loading every set separately made the norms twice the fragments' L1
  // traffic at 4 sets -- 24 vector-memory instructions per block, 63 % of the wave cycles waiting for an issue slot)
  auto load_start = [&](const int cb, f32x16 (&dst)[VS]) __attribute__((always_inline)) {
    const int cbc = cb < a.nblocks ? cb : a.nblocks - 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.cnorm_half + cbc * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[0][4 * g + e] = t[e];
    }
#pragma unroll
    for (int s = 1; s < VS; ++s) dst[s] = dst[0];
  };
  auto block = [&](const int cb, f32x16 (&cur)[VS], f32x16 (&nxt)[VS]) __attribute__((always_inline)) {
    load_start(cb + 1, nxt);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const h16x8 w = ring[k % P];
      ring[k % P] = wp[(k + P) * 64];
      asm volatile("" ::: "memory");  // keep the ring loads where they are (see ivf_assign_kernel)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < VS; ++s) cur[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, xt[s][k], cur[s], 0, 0, 0);
    }
    wp += NK * 64;
    // lane holds centroids cb*32 + 8g + 4*half + e (register 4g + e) of vector j of each set
    if constexpr (MODE == 0) {
#pragma unroll
      for (int s = 0; s < VS; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) mn[s] = fmaxf(mn[s], cur[s][i]);
    } else {
      bool any = false;
#pragma unroll
      for (int s = 0; s < VS; ++s) {
        float m = cur[s][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, cur[s][i]);   // v_max3: 8 instructions, then ONE compare per set
        any |= m >= nthr[s];
      }
      if (__builtin_expect(__any(any), 0)) {  // rare: about one hit per vector in ivf_K centroids
#pragma unroll
        for (int s = 0; s < VS; ++s)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (valid[s] && cur[s][4 * g + e] >= nthr[s]) {
                const int pos = atomicAdd(a.cand_count, 1);
                if (pos < a.cand_cap) {
                  a.cand_vec[pos] = (int)vec[s];
                  a.cand_id[pos] = a.perm[cb] * 32 + 8 * g + 4 * half + e;
                } else {
                  atomicOr(a.overflow, 1);
                }
              }
      }
    }
  };
  int cb = cb0;
  load_start(cb0, acc[0]);
  // two blocks per trip: the accumulator sets alternate with compile-time indices, and the waitcnts inside straight-line code
  // are counted exactly (see ivf_assign_kernel)
  for (; cb + 1 < cb1; cb += 2) {
    block(cb, acc[0], acc[1]);
    block(cb + 1, acc[1], acc[0]);
  }
  if (cb < cb1) block(cb, acc[0], acc[1]);

  if constexpr (MODE == 0) {
#pragma unroll
    for (int s = 0; s < VS; ++s) {
      const float m = -fmaxf(mn[s], __shfl_xor(mn[s], 32));  // min of s~
      if (valid[s] && half == 0) atomicMin(a.approx_min + vec[s], f32_ordered(m));
    }
  }
}

// pass C: exact fp32 distance of each candidate, (|x|^2 + |c|^2) - 2 x.c with sequential fmaf sums, merged into the
// (distance, id) keys of ivf_assign_kernel.  One thread per candidate (about N of them).
__global__ void __launch_bounds__(256)
ivf_exact_kernel(const int* __restrict__ cand_count, int cand_cap, const int* __restrict__ overflow,
                 const int* __restrict__ cand_vec, const int* __restrict__ cand_id, const float* __restrict__ x,
                 const float* __restrict__ centroids, const float* __restrict__ cnorm, int D,
                 unsigned long long* __restrict__ best) {
  // after an overflow the fp32 table kernel decides alone: keys of the two kernels must not be mixed (their
  // distances round differently, which would reorder exact ties)
  if (*overflow) return;
  int cnt = *cand_count;
  if (cnt > cand_cap) cnt = cand_cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const int v = cand_vec[i], k = cand_id[i];
    const float* xp = x + (long)v * D;
    const float* cp = centroids + (long)k * D;
    float xn = 0.f, dot = 0.f;
    for (int d4 = 0; d4 < D; d4 += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + d4);
      const f32x4 cv = *reinterpret_cast<const f32x4*>(cp + d4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xn = fmaf(xv[e], xv[e], xn);
        dot = fmaf(xv[e], cv[e], dot);
      }
    }
    const float dist = __fsub_rn(__fadd_rn(xn, cnorm[k]), __fmul_rn(2.f, dot));
    atomicMin(best + v, ivf_key(dist, k));
  }
}

}  // namespace qinco
