// Split-fp16 evaluation of an fp32 GEMM on v_mfma_f32_32x32x16_f16 (feasibility study for a gated "split" mode of the
// residual blocks): x = xhi + xlo, W = Whi + Wlo (fp16 each), W.x ~= Whi.xhi + Whi.xlo + Wlo.xhi, fp32 accumulation.
//   part 1 (numerics): one wave, D[32x32] = W[32xK] . X[Kx32], K = 384, against a float64 host result, beside the fp32
//           MFMA (v_mfma_f32_32x32x2_f32) on the same data; inputs scaled down to push the lo parts / the hi parts into the
//           fp16 subnormal range (does the matrix pipe flush them?).
//   part 2 (rate): cycles per triplet (3 MFMAs = 96 cycles ideal) with 12 accumulators (K-outer order), adding the
//           LDS fragment reads, the workgroup-shared LDS-DMA ring with a barrier every G fragments, and the VALU split.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// 8 fp32 values -> hi (8 fp16, RNE) and lo = fp16(v - hi)
__device__ __forceinline__ void split8(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

// ---- part 1 -------------------------------------------------------------------------------------------
// W row-major [32][K], X row-major [K][32]; out16 / out32: [32][32] (feature m, row n).
__global__ void numerics(const float* W, const float* X, int K, float* out16, float* out16x4, float* out32) {
  const int lane = threadIdx.x, j = lane & 31, half = lane >> 5;
  f32x16 a16, a164, a32;
  for (int i = 0; i < 16; ++i) { a16[i] = 0.f; a164[i] = 0.f; a32[i] = 0.f; }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float wv[8], xv[8];
    for (int e = 0; e < 8; ++e) { wv[e] = W[j * K + k0 + 8 * half + e]; xv[e] = X[(k0 + 8 * half + e) * 32 + j]; }
    f16x8 whi, wlo, xhi, xlo;
    split8(wv, whi, wlo);
    split8(xv, xhi, xlo);
    a16 = MF16(whi, xhi, a16); a16 = MF16(whi, xlo, a16); a16 = MF16(wlo, xhi, a16);
    a164 = MF16(wlo, xlo, a164); a164 = MF16(wlo, xhi, a164); a164 = MF16(whi, xlo, a164); a164 = MF16(whi, xhi, a164);
  }
  for (int k0 = 0; k0 < K; k0 += 2) a32 = MF32(W[j * K + k0 + half], X[(k0 + half) * 32 + j], a32);
  for (int i = 0; i < 16; ++i) {
    int m = (i & 3) + 8 * (i >> 2) + 4 * half;
    out16[m * 32 + j] = a16[i]; out16x4[m * 32 + j] = a164[i]; out32[m * 32 + j] = a32[i];
  }
}

// ---- part 2 -------------------------------------------------------------------------------------------
constexpr int NB = 12;     // 32-feature blocks in and out (384 x 384 layer)
constexpr int P = 96;      // ring depth in 1 KiB fragments
template <int N> __device__ __forceinline__ void waitvm() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
// MODE 0: MFMAs only (operands in registers)   1: + two ds_read_b128 per triplet (static LDS), one pair of outputs ahead
//      2: + shared LDS-DMA ring, barrier every G fragments   3: 2 + the VALU split of z (AGPR master -> fp16 hi/lo per 32-block)
//      4: 1 + the DMAs alone (no waits, no barriers)          5: 1 + the barriers alone
template <int MODE, int G>
__global__ void __launch_bounds__(256) rate(const f32x4* __restrict__ w, float* out, long long* cyc, int iters) {
  __shared__ f32x4 lds[P * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4* wp = w + lane;
  f32x16 t[NB];
  for (int o = 0; o < NB; ++o) for (int i = 0; i < 16; ++i) t[o][i] = 0.f;
  f32x16 zm;                       // stands for the fp32 master of the current input block
  for (int i = 0; i < 16; ++i) zm[i] = 1.0f + lane * 1e-3f + i * 0.01f;
  for (int i = lane; i < P * 64; i += 64) if (wave == 0) lds[i] = wp[(i / 64) * 64];
  constexpr int PER = G / 4;       // DMAs per wave per group
  constexpr bool DMA = MODE == 2 || MODE == 3 || MODE == 4, BAR = MODE == 2 || MODE == 3 || MODE == 5;
  const f32x4* gsrc = wp + wave * PER * 64;          // this wave's share of a group
  f32x4* ldst = lds + wave * PER * 64;
  auto dma = [&](int slot0, long fragofs) {          // slot0: compile-time after unrolling
#pragma unroll
    for (int q = 0; q < PER; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (fragofs + q) * 64),
                                       (__attribute__((address_space(3))) void*)(ldst + ((slot0 + q) % P) * 64), 16, 0, 0);
  };
  __syncthreads();
  if constexpr (DMA) {
#pragma unroll
    for (int gq = 0; gq < P / G - 1; ++gq) dma(gq * G, gq * G);
  }
  f16x8 bh[2], bl[2];
  {
    float v[16]; for (int i = 0; i < 16; ++i) v[i] = zm[i];
    split8(v, bh[0], bl[0]); split8(v + 8, bh[1], bl[1]);
  }
  auto rd = [&](int slot) { return __builtin_bit_cast(f16x8, lds[(slot % P) * 64 + lane]); };
  long long t0 = __builtin_readcyclecounter();
  f16x8 cur[4], nxt[4];
  if constexpr (MODE != 0) for (int q = 0; q < 4; ++q) cur[q] = rd(q);
  for (int it = 0; it < iters; it += 2) {       // one trip = two input blocks = one ring revolution (96 fragments)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      if constexpr (MODE == 3) {
        asm volatile("" : "+a"(zm));
        float v[16]; for (int i = 0; i < 16; ++i) v[i] = zm[i] + (float)it;
        split8(v, bh[0], bl[0]); split8(v + 8, bh[1], bl[1]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int o = 0; o < NB; o += 2) {
          const int f = blk * 48 + (c * NB + o) * 2;   // ring slot of this pair's first fragment (compile time)
          if constexpr (MODE == 0) {
            cur[0] = bh[0]; cur[1] = bl[0]; cur[2] = bh[1]; cur[3] = bl[1];
          } else {
            // the next pair's fragments; when they open a new group, that group must have landed and the one before the
            // current one is free for the refill
            const int fn = (f + 4) % P;
            if ((fn % G) == 0) {
              if constexpr (BAR) {
                if constexpr (DMA) waitvm<(P / G - 3) * PER>();
                __builtin_amdgcn_s_barrier();
              }
              if constexpr (DMA) dma((fn + P - G) % P, (long)it * 48 + (f + 4) + P - G);
            }
            for (int q = 0; q < 4; ++q) nxt[q] = rd(fn + q);
            asm volatile("" ::: "memory");
            if constexpr (BAR) { if ((fn % G) == G - 4) asm volatile("" : "+v"(nxt[3])); }
          }
          t[o] = MF16(cur[0], bh[c], t[o]);     t[o + 1] = MF16(cur[2], bh[c], t[o + 1]);
          t[o] = MF16(cur[0], bl[c], t[o]);     t[o + 1] = MF16(cur[2], bl[c], t[o + 1]);
          t[o] = MF16(cur[1], bh[c], t[o]);     t[o + 1] = MF16(cur[3], bh[c], t[o + 1]);
          if constexpr (MODE != 0) for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int o = 0; o < NB; ++o) for (int i = 0; i < 16; ++i) s += t[o][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int G>
void run_rate(const char* name, const f32x4* w, float* out, long long* cyc, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate<MODE, G>), dim3(256), dim3(256), 0, 0, w, out, cyc, 8);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate<MODE, G>), dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(256); CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : h) mean += v; mean /= 256;
  double trip = (double)iters * 24;   // triplets per wave
  double tf = 256.0 * 4 * trip * 3 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("{\"rate\": \"%s\", \"cycles_per_triplet\": %.1f, \"ms\": %.3f, \"f16_TFLOPs\": %.0f, \"fp32_equiv_TFLOPs\": %.0f}\n", name, mean / trip,
         ms, tf, tf / 3);
}

int main() {
  // ---- numerics
  const int K = 384;
  std::vector<float> W(32 * K), X(K * 32);
  srand(1);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2 - 1); };
  float *dW, *dX, *d16, *d164, *d32;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4));
  CK(hipMalloc(&d16, 4096)); CK(hipMalloc(&d164, 4096)); CK(hipMalloc(&d32, 4096));
  for (float ws : {0.1f, 0.1f / 4096.f}) for (float xs : {1.0f, 1.0f / 1024.f, 1.0f / 65536.f}) {
    for (auto& v : W) v = rnd() * ws;
    for (auto& v : X) v = rnd() * xs;
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(numerics, dim3(1), dim3(64), 0, 0, dW, dX, K, d16, d164, d32);
    std::vector<float> o16(1024), o164(1024), o32(1024);
    CK(hipMemcpy(o16.data(), d16, 4096, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o164.data(), d164, 4096, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o32.data(), d32, 4096, hipMemcpyDeviceToHost));
    double e16 = 0, e164 = 0, e32 = 0, ref2 = 0, sab = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      double r = 0, a = 0;
      for (int k = 0; k < K; ++k) { r += (double)W[m * K + k] * X[k * 32 + n]; a += fabs((double)W[m * K + k] * X[k * 32 + n]); }
      e16 += pow(o16[m * 32 + n] - r, 2); e164 += pow(o164[m * 32 + n] - r, 2); e32 += pow(o32[m * 32 + n] - r, 2);
      ref2 += r * r; sab += a;
    }
    printf("{\"numerics\": {\"w_scale\": %g, \"x_scale\": %g}, \"rms_ref\": %.3e, \"rms_err_f16x3\": %.3e, \"rms_err_f16x4\": %.3e, \"rms_err_f32_mfma\": %.3e, "
           "\"err_over_sum_abs_f16x3\": %.3e, \"err_over_sum_abs_f32\": %.3e}\n", ws, xs, sqrt(ref2 / 1024), sqrt(e16 / 1024), sqrt(e164 / 1024),
           sqrt(e32 / 1024), sqrt(e16 / 1024) / (sab / 1024), sqrt(e32 / 1024) / (sab / 1024));
  }
  // ---- rate
  const int iters = 2000;
  const long nfrag = (long)(iters + 16) * 48 + 4 * P;
  f32x4* w; CK(hipMalloc(&w, nfrag * 1024)); CK(hipMemset(w, 0, nfrag * 1024));
  if (getenv("SPLIT16_RANDOM")) {   // random fp16 weights (|w| < 2): the matrix pipe's power draw depends on the operand bits
    std::vector<_Float16> hw((size_t)nfrag * 512);
    for (auto& v : hw) v = (_Float16)(rnd() * 2.0f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  }
  float* out; CK(hipMalloc(&out, 256 * 256 * 4));
  long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
  run_rate<0, 4>("mfma_only", w, out, cyc, iters);
  run_rate<1, 4>("lds_reads", w, out, cyc, iters);
  run_rate<2, 4>("ring_G4", w, out, cyc, iters);
  run_rate<2, 8>("ring_G8", w, out, cyc, iters);
  run_rate<2, 16>("ring_G16", w, out, cyc, iters);
  run_rate<2, 32>("ring_G32", w, out, cyc, iters);
  run_rate<3, 16>("ring_G16_split", w, out, cyc, iters);
  run_rate<4, 4>("dma_only_G4", w, out, cyc, iters);
  run_rate<4, 16>("dma_only_G16", w, out, cyc, iters);
  run_rate<5, 4>("barrier_only_G4", w, out, cyc, iters);
  run_rate<5, 16>("barrier_only_G16", w, out, cyc, iters);
  return 0;
}
