// IVF step 0 (IVFBook.quantize, reference qinco/model/qinco_base.py:146-163): nearest of ivf_K (up to 2^20)
// coarse centroids for every vector, = argmin_k |x|^2 + |c_k|^2 - 2 x.c_k  (approx_pairwise_distance, utils.py:336-346).
// 2 D ivf_K FLOPs per vector (268 MFLOP at D=128, ivf_K=2^20) -- a GEMM-shaped table, so it runs on the fp32 MFMA
// with the same transposed trick as the MLP kernel: a wave keeps its 32 vectors as B operands in registers and
// streams centroid fragments (host-packed, 1 KiB = 32 centroids x 8 features) as A operands; the 32x32 result tile
// leaves lane l with 16 centroid distances of vector (l&31), reduced on the fly to a running (distance, id) minimum.
// The centroid table does not fit any cache (512 MB), so the grid is (vector tiles) x (centroid slices) and slices
// merge through one 64-bit atomicMin per vector on an order-preserving (distance bits, id) key: ties -> lower id,
// like argmin.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_args.hpp"

namespace qinco {

__device__ __forceinline__ unsigned long long ivf_key(float d, int idx) {
  unsigned u = __builtin_bit_cast(unsigned, d);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> unsigned
  return ((unsigned long long)u << 32) | (unsigned)idx;
}

template <int D>
__global__ void __launch_bounds__(256)
ivf_assign_kernel(const f32x4* __restrict__ cstream, const float* __restrict__ cnorm, int nblocks, int blocks_per_slice,
                  const float* __restrict__ x, long N, unsigned long long* __restrict__ best) {
  constexpr int NDB = D / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long v0 = ((long)blockIdx.x * 4 + wave) * 32;
  if (v0 >= N) return;
  long vec = v0 + j;
  const bool valid = vec < N;
  if (!valid) vec = N - 1;
  const float* xp = x + vec * D + half * 4;
  f32x16 xt[NDB];
  float xn = 0.f;
#pragma unroll
  for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = *reinterpret_cast<const f32x4*>(xp + ib * 32 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xt[ib][4 * q + e] = t[e];
        xn = fmaf(t[e], t[e], xn);
      }
    }
  }
  xn += __shfl_xor(xn, 32);

  const int cb0 = blockIdx.y * blocks_per_slice;
  int cb1 = cb0 + blocks_per_slice;
  if (cb1 > nblocks) cb1 = nblocks;
  const f32x4* wp = cstream + (long)cb0 * (NDB * 4 * 64) + lane;
  float bestd = __builtin_inff();
  int besti = 0x7fffffff;
  for (int cb = cb0; cb < cb1; ++cb) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = wp[(ib * 4 + q) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xt[ib][4 * q + e], acc, 0, 0, 0);
      }
    }
    wp += NDB * 4 * 64;
    // lane holds centroids cb*32 + 8g + 4*half + e  (g = r>>2, e = r&3) of its vector
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int base = cb * 32 + 8 * g + 4 * half;
      const f32x4 cn = *reinterpret_cast<const f32x4*>(cnorm + base);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = __fsub_rn(__fadd_rn(xn, cn[e]), __fmul_rn(2.f, acc[4 * g + e]));
        if (d < bestd) {  // ids grow within a lane: strict < keeps the first minimum
          bestd = d;
          besti = base + e;
        }
      }
    }
  }
  unsigned long long key = ivf_key(bestd, besti);
  const unsigned long long other = __shfl_xor(key, 32);
  if (other < key) key = other;
  if (valid && half == 0 && besti != 0x7fffffff) atomicMin(best + vec, key);
}

// codes0[n] = id part of the merged key (and the matching normalised centroid becomes xhat0 via gather_rows)
__global__ void ivf_finish_kernel(const unsigned long long* __restrict__ best, long N, int* __restrict__ ids) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) ids[i] = (int)(best[i] & 0xffffffffu);
}

}  // namespace qinco
