// Micro-benchmark (round 5, for knn_table_kernel / dist_topk_mfma_kernel): cycles per 4-MFMA fragment (ideal 256 per wave, x waves
// per SIMD) when every wave feeds its chain from a REGISTER ring of plain global_load_dwordx4 over an L2-resident stream
// (1 MiB, every workgroup walks the same one) -- the loop of the table kernels -- against MFMAs only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 ring_hot.hip -o ring_hot && ./ring_hot
// Result (profiles/r05g_ubench_ring_hot.log; LDS padding pins the workgroups per CU, HW_ID confirms 1 or 2 waves on EVERY SIMD):
//   one wave per SIMD:  MFMAs only 152 TFLOP/s (0.97), with the ring 142-145 (0.90-0.92): the plain-load ring costs a lone wave 7-9 %
//   two waves per SIMD: MFMAs only 78-104 TFLOP/s (0.50-0.66, varies run to run; the measured wave of every workgroup still sees
//                       256 cycles per fragment: the other wave of its SIMD starves, then runs), with the ring 126-129 (0.80-0.82)
//   (cycles per fragment x fragments / launch time = 2.33 GHz with one wave, ~1.9 GHz with two + ring: part of the loss is the clock;
//    s_setprio 3 / 0 by wave-slot parity: no change)
// i.e. in delivered TFLOP/s two waves per SIMD that both always have a dependent f32 MFMA ready lose ~20 % against one; every
// two-workgroups-per-CU kernel of this library sits at or under that 0.80 (knn_table 0.77-0.80 in its loop, dist_topk_mfma 0.60,
// the short-MLP instances 0.84), the one-wave-per-SIMD kernels at 0.92.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int NF = 16;         // fragments per block (D = 128)
constexpr int BLOCKS = 64;     // blocks per pass over the stream (1 MiB)

template <int DEPTH, bool LOADS, int WPS, bool PRIO = false>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ w, float* out, long long* cyc, int passes, long long* wall, unsigned long long* where) {
  const long long w0 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {   // XCC id << 32 | HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SE/SH [15:12])
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    where[blockIdx.x * 4 + (threadIdx.x >> 6)] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  if (PRIO) {   // the two waves of a SIMD sit in different wave slots: the odd slot gets the higher priority
    unsigned hw2;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw2));
    if (hw2 & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
  }
  __shared__ float pad[WPS == 2 ? 18 * 1024 : 36 * 1024];   // 72 / 144 KiB of LDS: exactly WPS workgroups fit a CU (without it the
  pad[threadIdx.x] = 0.f;                                     // dispatcher put up to three of these small workgroups on one CU)
  const int lane = threadIdx.x & 63;
  const f32x4* wp = w + lane;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x4 r[DEPTH];
  for (int i = 0; i < DEPTH; ++i) r[i] = wp[i * 64];
  float b = 1.0f + lane * 1e-4f;
  long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    const f32x4* q = wp;
    for (int blk = 0; blk < BLOCKS; ++blk) {
#pragma unroll
      for (int t = 0; t < NF; ++t) {
        const f32x4 x = r[t % DEPTH];
        if (LOADS) r[t % DEPTH] = q[(t + DEPTH) * 64];
        asm volatile("" ::: "memory");
        SB;
        acc = MF(x[0], b, acc); acc = MF(x[1], b, acc); acc = MF(x[2], b, acc); acc = MF(x[3], b, acc);
      }
      q += NF * 64;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = acc[0] + acc[9] + pad[threadIdx.x];
  for (int i = 0; i < DEPTH; ++i) s += r[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[2 * blockIdx.x] = w0; wall[2 * blockIdx.x + 1] = wall_clock64(); }
}

template <int DEPTH, bool LOADS, int WPS, bool PRIO = false>
void run(const f32x4* w, int passes) {
  float* out; long long* cyc; long long* wall;
  hipMalloc(&wall, 1024 * 16);
  unsigned long long* where; hipMalloc(&where, 4096 * 8);
  const int grid = 256 * WPS;   // WPS workgroups of 4 waves per CU (a 512-thread workgroup does NOT spread its 8 waves 2 per SIMD:
                                // MFMAs only took 3x the one-wave time, i.e. three waves on one SIMD)
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<DEPTH, LOADS, WPS, PRIO>), dim3(grid), dim3(256), 0, 0, w, out, cyc, passes, wall, where);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double tflops = (double)grid * 4 * passes * BLOCKS * NF * 4 * 4096.0 / (ms * 1e-3) / 1e12;
  long long c[1024]; hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += c[i]; avg /= grid;
  printf("%sring depth %2d  loads %d  waves/SIMD %d: %.1f cycles per fragment per wave (ideal %d)  launch %.3f ms = %.1f TFLOP/s\n", PRIO ? "[prio by wave slot] " : "", DEPTH,
         (int)LOADS, WPS, avg / ((double)passes * BLOCKS * NF), 256 * WPS, ms, tflops);
  {  // when did the workgroups run?  (wall_clock64: 100 MHz)
    static long long wl[2048]; hipMemcpy(wl, wall, grid * 16, hipMemcpyDeviceToHost);
    long long lo = wl[0]; for (int i = 0; i < grid; ++i) lo = wl[2 * i] < lo ? wl[2 * i] : lo;
    int late = 0; double life = 0, last = 0;
    for (int i = 0; i < grid; ++i) {
      const double st = (wl[2 * i] - lo) * 1e-5, en = (wl[2 * i + 1] - lo) * 1e-5;   // ms
      late += st > 0.1; life += en - st; last = en > last ? en : last;
    }
    static unsigned long long wh[4096]; hipMemcpy(wh, where, grid * 4 * 8, hipMemcpyDeviceToHost);
    int hist[9] = {0};   // SIMDs by the number of this launch's waves they hold
    for (int i = 0; i < grid * 4; ++i) {
      if (wh[i] == ~0ull) continue;
      const unsigned long long key = ((wh[i] >> 32) << 16) | ((wh[i] >> 4) & 0xfff);
      int n = 0;
      for (int k = i; k < grid * 4; ++k)
        if (wh[k] != ~0ull && ((((wh[k] >> 32) << 16) | ((wh[k] >> 4) & 0xfff)) == key)) { ++n; if (k != i) wh[k] = ~0ull; }
      hist[n > 8 ? 8 : n]++;
    }
    printf("    SIMDs holding 1 / 2 / 3 / 4 / 5+ of the launch's waves: %d / %d / %d / %d / %d\n", hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7] + hist[8]);
    printf("    workgroups: %d of %d started more than 0.1 ms after the first; mean life %.3f ms; last end %.3f ms\n", late, grid, life / grid, last);
  }
  hipFree(out); hipFree(cyc); hipFree(wall);
}
int main() {
  const int passes = 32;
  f32x4* w; size_t bytes = (size_t)(BLOCKS * NF + 64) * 1024; hipMalloc(&w, bytes); hipMemset(w, 0, bytes);
  run<8, false, 1>(w, passes); run<8, true, 1>(w, passes); run<16, true, 1>(w, passes);
  run<8, false, 2>(w, passes); run<8, true, 2>(w, passes); run<16, true, 2>(w, passes);
  run<8, false, 2, true>(w, passes); run<8, true, 2, true>(w, passes); run<16, true, 2, true>(w, passes);
  return 0;
}
