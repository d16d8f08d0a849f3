// Micro-benchmark: cycles per v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles) for 1 / 2 / 4 independent accumulator chains, one
// wave per SIMD -- does a dependent chain (acc = mfma(a, b, acc) back to back, what mlp16_kernel's fragmm issues four at a time)
// run at the full rate like the 32x32x2 form does (mfma_dep.hip: 64.0 cycles in every case)?
//   hipcc --offload-arch=gfx950 -O3 mfma_dep16.hip -o mfma_dep16 && ./mfma_dep16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH, int RUN>   // RUN = consecutive MFMAs on one chain before moving to the next (mlp16: 4)
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x4 acc[NCH];
  for (int c = 0; c < NCH; ++c)
    for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / RUN; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < RUN; ++r) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH, int RUN>
void run(int blocks) {
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  const int iters = 4000;
  hipLaunchKernelGGL((k<NCH, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((k<NCH, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 20;
  const double n = (double)iters * 16 * NCH;            // MFMAs per wave
  const double tf = n * 2048.0 * blocks * 4 / (ms * 1e-3) / 1e12;
  printf("chains=%d run=%d blocks=%d: %.3f ms, %.1f TFLOP/s (%.3f of 157.3)\n", NCH, RUN, blocks, ms, tf, tf / 157.3);
  hipFree(out);
}
int main() {
  run<1, 16>(256);
  run<2, 1>(256);
  run<2, 4>(256);
  run<4, 1>(256);
  run<4, 4>(256);
  run<8, 1>(256);
  run<1, 16>(512);
  run<2, 1>(512);
  return 0;
}
