// What a VALU instruction costs next to f32 MFMAs on one SIMD (round 6; for knn_table_kernel / dist_topk_mfma_kernel / the
// short-MLP epilogues).  v_mfma_f32_32x32x2_f32 = 16 passes = 64 cycles of the matrix pipe.
//   (A) ONE wave per SIMD: a dependent MFMA chain with K independent VALU instructions (v_fma_f32 on private registers) behind every
//       MFMA: cycles per MFMA as K grows -- does the wave's own VALU work hide under its MFMAs?
//   (B) TWO waves per SIMD (a 512-thread workgroup, roles by HW_ID as in knn_roles_kernel.hpp): one wave runs the bare chain, the
//       other nothing but VALU instructions: cycles per MFMA of the first, cycles per VALU instruction of the second.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 scripts/ubench/mfma_valu.hip -o scripts/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int K, int KIND>
__global__ void __launch_bounds__(256) one_wave(float* out, long long* cyc, int iters) {
  __shared__ float pad[36 * 1024];   // 144 KiB: one workgroup per CU
  pad[threadIdx.x] = 0.f;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float a = 1.0f + threadIdx.x * 1e-4f, b = 0.5f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc = MF(a, b, acc);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(m * K + k) & 15]) : "v"(b));
        if (KIND == 1) asm volatile("v_cmp_le_f32 vcc, %0, %1" ::"v"(v[(m * K + k) & 15]), "v"(b) : "vcc");
        if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[(m * K + k) & 15]) : "v"(b));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = acc[0] + acc[7] + pad[threadIdx.x];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NOPS = 0, bool YOUNG = false>
__global__ void __launch_bounds__(512) two_roles(float* out, long long* cyc, int iters, int valu_per_iter, int prio) {
  __shared__ int maxw[4];
  __shared__ float pad[36 * 1024];
  __shared__ int claim[4];
  __shared__ int stop;
  pad[threadIdx.x] = 0.f;
  if (threadIdx.x < 4) { claim[threadIdx.x] = 0; maxw[threadIdx.x] = -1; }
  if (threadIdx.x == 4) stop = 0;
  __syncthreads();
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const int simd = (hw >> 4) & 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int order = 0;
  if (lane == 0) { order = atomicAdd(&claim[simd], 1); atomicMax(&maxw[simd], wave); }
  order = __builtin_amdgcn_readfirstlane(order);
  __syncthreads();
  const bool spread = claim[0] == 2 && claim[1] == 2 && claim[2] == 2 && claim[3] == 2;
  // YOUNG: the MFMA role goes to the YOUNGER wave of each SIMD (the higher wave index), the VALU role to the older one
  const bool mfma_role = __builtin_amdgcn_readfirstlane((int)(spread ? (YOUNG ? wave == maxw[simd] : order == 0) : wave < 4)) != 0;
  float s = 0.f;
  long long t0, t1, n = 0;
  if (mfma_role) {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-4f, b = 0.5f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc = MF(a, b, acc);
        // (B') the MFMA wave steps back from the issue stage while its MFMA runs: s_nop 15 = 16 idle cycles, NOPS of them
#pragma unroll
        for (int z = 0; z < NOPS; ++z) asm volatile("s_nop 15");
      }
    }
    t1 = __builtin_readcyclecounter();
    n = (long long)iters * 16;
    s = acc[0] + acc[5];
    if (lane == 0) atomicAdd(&stop, 1);
  } else {
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float b = 0.5f;
    t0 = __builtin_readcyclecounter();
    // VALU-only until every MFMA wave of the workgroup has finished (checked every 64 instructions through LDS)
    while (valu_per_iter > 0) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 15]) : "v"(b));
        if (KIND == 1) asm volatile("v_cmp_le_f32 vcc, %0, %1" ::"v"(v[k & 15]), "v"(b) : "vcc");
        if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k & 15]) : "v"(b));
      }
      n += 64;
      if (*(volatile int*)&stop >= 4) break;
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s + pad[threadIdx.x];
  if (lane == 0) {
    long long* o = cyc + (blockIdx.x * 8 + wave) * 4;
    o[0] = mfma_role; o[1] = t1 - t0; o[2] = n; o[3] = spread;
  }
}

template <int K, int KIND>
void runA(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((one_wave<K, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c[1024]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 1024; ++i) avg += c[i]; avg /= 1024;
  printf("(A) one wave per SIMD, %2d VALU (%s) behind every MFMA: %6.1f cycles per MFMA (64 = the pipe's rate)  -> %.1f extra per VALU instruction\n", K,
         KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_cmp_le_f32" : "v_xor_b32", avg / (iters * 16.0), K ? (avg / (iters * 16.0) - 64.0) / K : 0.0);
}
template <int KIND, int NOPS = 0, bool YOUNG = false>
void runB(float* out, long long* cyc, int valu, int prio) {
  const int iters = 2000;
  if (NOPS) printf("[MFMA wave: %d x s_nop 15 behind every MFMA] ", NOPS);
  if (YOUNG) printf("[MFMA role to the younger wave of each SIMD] ");
  hipLaunchKernelGGL((two_roles<KIND, NOPS, YOUNG>), dim3(256), dim3(512), 0, 0, out, cyc, iters, valu, prio);
  hipDeviceSynchronize();
  static long long c[256 * 8 * 4]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double mc = 0, mn = 0, vc = 0, vn = 0; int spread = 0;
  for (int w = 0; w < 2048; ++w) {
    if (c[w * 4]) { mc += c[w * 4 + 1]; mn += c[w * 4 + 2]; } else { vc += c[w * 4 + 1]; vn += c[w * 4 + 2]; }
    spread += (w % 8 == 0) && c[w * 4 + 3];
  }
  printf("(B) two waves per SIMD (%d of 256 workgroups 2+2+2+2), VALU wave %s (%s), prio %s: %6.1f cycles per MFMA", spread, valu ? "busy" : "idle",
         KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_cmp_le_f32" : "v_xor_b32", prio == 0 ? "none" : prio == 1 ? "MFMA wave" : "VALU wave", mc / mn);
  if (valu) printf(", %5.1f cycles per VALU instruction = %.2f VALU per MFMA", vc / vn, (mc / mn) / (vc / vn));
  printf("\n");
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 4 * 8);
  runA<0, 0>(out, cyc); runA<1, 0>(out, cyc); runA<2, 0>(out, cyc); runA<4, 0>(out, cyc); runA<8, 0>(out, cyc); runA<12, 0>(out, cyc); runA<16, 0>(out, cyc);
  runA<4, 1>(out, cyc); runA<8, 1>(out, cyc); runA<4, 2>(out, cyc); runA<8, 2>(out, cyc); runA<16, 2>(out, cyc);
  runB<0>(out, cyc, 0, 0);
  for (int prio = 0; prio < 3; ++prio) { runB<0>(out, cyc, 1, prio); runB<1>(out, cyc, 1, prio); runB<2>(out, cyc, 1, prio); }
  runB<0, 1>(out, cyc, 1, 0); runB<0, 2>(out, cyc, 1, 0);
  for (int rep = 0; rep < 2; ++rep)
    for (int prio = 0; prio < 3; ++prio) { runB<0, 0, true>(out, cyc, 1, prio); runB<1, 0, true>(out, cyc, 1, prio); runB<2, 0, true>(out, cyc, 1, prio); }
  return 0;
}
