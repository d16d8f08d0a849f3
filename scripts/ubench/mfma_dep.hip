// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for 1 / 2 / 4 independent accumulator chains,
// one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep && ./mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
  f32x16 acc[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH>
void run(int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * NCH;
  double tf = n * 4096.0 * blocks * 4 / (ms * 1e-3) / 1e12;
  printf("chains=%d blocks=%d: %.2f s_memtime-cycles/MFMA (100 MHz ticks x?), %.3f ms, %.1f TFLOP/s\n", NCH, blocks, c / n, ms, tf);
}
int main() {
  run<1>(256); run<2>(256); run<4>(256); run<1>(1024); run<2>(1024);
  return 0;
}
