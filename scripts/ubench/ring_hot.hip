// Micro-benchmark (round 5, for knn_table_kernel / dist_topk_mfma_kernel): cycles per 4-MFMA fragment (ideal 256 per wave, x waves
// per SIMD) when every wave feeds its chain from a REGISTER ring of plain global_load_dwordx4 over an L2-resident stream
// (1 MiB, every workgroup walks the same one) -- the loop of the table kernels -- against MFMAs only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 ring_hot.hip -o ring_hot && ./ring_hot
// Result (profiles/r05g_ubench_ring_hot.log): ONE wave per SIMD 153 TFLOP/s with MFMAs only, 141-144 with the ring (depth 8 / 16):
// the plain-load ring costs a single wave 7-9 %, hot.  The two-workgroups-per-CU rows are NOT understood and not used as evidence:
// MFMAs only take exactly 3x the one-workgroup time for 2x the work (102.8 TFLOP/s) while every wave's own cycle counter shows the
// full rate (256 cycles per fragment) -- as if a third of the workgroups ran alone after the others; with the ring 126-128 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int NF = 16;         // fragments per block (D = 128)
constexpr int BLOCKS = 64;     // blocks per pass over the stream (1 MiB)

template <int DEPTH, bool LOADS, int WPS>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ w, float* out, long long* cyc, int passes) {
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
  float s = acc[0] + acc[9];
  for (int i = 0; i < DEPTH; ++i) s += r[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH, bool LOADS, int WPS>
void run(const f32x4* w, int passes) {
  float* out; long long* cyc;
  const int grid = 256 * WPS;   // WPS workgroups of 4 waves per CU (a 512-thread workgroup does NOT spread its 8 waves 2 per SIMD:
                                // MFMAs only took 3x the one-wave time, i.e. three waves on one SIMD)
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<DEPTH, LOADS, WPS>), dim3(grid), dim3(256), 0, 0, w, out, cyc, passes);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double tflops = (double)grid * 4 * passes * BLOCKS * NF * 4 * 4096.0 / (ms * 1e-3) / 1e12;
  long long c[1024]; hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += c[i]; avg /= grid;
  printf("ring depth %2d  loads %d  waves/SIMD %d: %.1f cycles per fragment per wave (ideal %d)  launch %.3f ms = %.1f TFLOP/s\n", DEPTH,
         (int)LOADS, WPS, avg / ((double)passes * BLOCKS * NF), 256 * WPS, ms, tflops);
  hipFree(out); hipFree(cyc);
}
int main() {
  const int passes = 32;
  f32x4* w; size_t bytes = (size_t)(BLOCKS * NF + 64) * 1024; hipMalloc(&w, bytes); hipMemset(w, 0, bytes);
  run<8, false, 1>(w, passes); run<8, true, 1>(w, passes); run<16, true, 1>(w, passes);
  run<8, false, 2>(w, passes); run<8, true, 2>(w, passes); run<16, true, 2>(w, passes);
  return 0;
}
