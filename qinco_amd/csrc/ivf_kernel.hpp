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

#include "aux_kernels.hpp"
#include "mlp_args.hpp"
#include "table_kernel.hpp"

namespace qinco {

__device__ __forceinline__ unsigned long long ivf_key(float d, int idx) {
  unsigned u = __builtin_bit_cast(unsigned, d);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> unsigned
  return ((unsigned long long)u << 32) | (unsigned)idx;
}

// largest divisor of nf that is <= cap (nf = 4 x feature blocks: at least 4)
constexpr int ivf_ring_depth(int nf, int cap) {
  int p = cap;
  while (nf % p) --p;
  return p;
}

template <int D>
__global__ void __launch_bounds__(256)
ivf_assign_kernel(const f32x4* __restrict__ cstream, const float* __restrict__ cnorm, int nblocks, int blocks_per_slice,
                  const float* __restrict__ x, long N, unsigned long long* __restrict__ best,
                  const int* __restrict__ only_if) {
  // only_if != nullptr: this launch is the exact fall-back behind the fp16-filter passes (ivf_f16_kernel.hpp) and
  // does nothing unless they raised their overflow flag
  if (only_if && *only_if == 0) return;
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
  // The slice's fragments are one sequential stream; a P-deep register ring keeps P fragments in flight across the
  // block boundary (without it every wave stalled on its 16 loads at the top of each block: 36 % of wave time in
  // s_waitcnt, matrix pipe 62 % busy).  The host pads the stream by P fragments.
  constexpr int NF = NDB * 4;                      // fragments per block of 32 centroids
  constexpr int P = ivf_ring_depth(NF, NDB > 8 ? 8 : 16);      // ring depth (divides NF: compile-time ring slots)
  const f32x4* wp = cstream + (long)cb0 * (NF * 64) + lane;
  f32x4 ring[P];
#pragma unroll
  for (int i = 0; i < P; ++i) ring[i] = wp[i * 64];
  float bestd = __builtin_inff();
  int besti = 0x7fffffff;
  auto block = [&](const int cb) __attribute__((always_inline)) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // the block's centroid norms are fetched before its MFMA chain, not in the epilogue (an exposed L2 round trip
    // per block otherwise)
    f32x4 cn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cn[g] = *reinterpret_cast<const f32x4*>(cnorm + cb * 32 + 8 * g + 4 * half);
#pragma unroll
    for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib * 4 + q;
        const f32x4 w = ring[i % P];
        ring[i % P] = wp[(i + P) * 64];
        // fences: without them hipcc sinks each ring load down to its use one block later (a vmcnt(0) every 4 MFMAs),
        // or hoists the block's 64 MFMAs above all of its loads
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xt[ib][4 * q + e], acc, 0, 0, 0);
      }
    }
    wp += NF * 64;
    // lane holds centroids cb*32 + 8g + 4*half + e  (g = r>>2, e = r&3) of its vector
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int base = cb * 32 + 8 * g + 4 * half;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = __fsub_rn(__fadd_rn(xn, cn[g][e]), __fmul_rn(2.f, acc[4 * g + e]));
        if (d < bestd) {  // ids grow within a lane: strict < keeps the first minimum
          bestd = d;
          besti = base + e;
        }
      }
    }
  };
  // two blocks per trip: hipcc waits for every outstanding load at a loop head (vmcnt of the back edge is merged
  // conservatively); inside straight-line code the waits are counted exactly
  int cb = cb0;
  if constexpr (NDB <= 8) {
    for (; cb + 1 < cb1; cb += 2) {
      block(cb);
      block(cb + 1);
    }
  } else {  // D = 768: the vectors alone take 384 registers
    for (; cb < cb1; ++cb) block(cb);
  }
  if (cb < cb1) block(cb);
  unsigned long long key = ivf_key(bestd, besti);
  const unsigned long long other = __shfl_xor(key, 32);
  if (other < key) key = other;
  if (valid && half == 0 && besti != 0x7fffffff) atomicMin(best + vec, key);
}

template <int D>
inline hipError_t launch_ivf_assign_kernel(const IvfArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(ivf_assign_kernel<D>, dim3((unsigned)((a.N + 127) / 128), (unsigned)a.slices), dim3(256), 0, st, a.cstream, a.cnorm,
                     a.nblocks, a.blocks_per_slice, a.x, a.N, a.best, a.only_if);
  return hipGetLastError();
}

// codes0[n] = id part of the merged key (and the matching normalised centroid becomes xhat0 via gather_rows)
__global__ void ivf_finish_kernel(const unsigned long long* __restrict__ best, long N, int* __restrict__ ids) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {   // a key nobody lowered (NaN input: no distance compares below +inf) keeps the reference's argmin of NaNs = 0
    const unsigned long long k = best[i];
    ids[i] = k == ~0ull ? 0 : (int)(k & 0xffffffffu);
  }
}

}  // namespace qinco
