// Where does a decode step of the small-launch kernel (csrc/mlp_small_kernel.hpp) spend its cycles?  Launches the decode
// instantiation of one shape on synthetic operands with -DQINCO_TIMELINE stamps (wave 0 of every workgroup) and prints, per stamp
// interval, the median over the workgroups.  Build + run on the GPU box:
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -DQINCO_EXPERIMENT -DQINCO_TIMELINE -DQD=128 -DQDE=128 -DQDH=256 -DQF2=1 -DQNT=3 \
//         -Iqinco_amd/csrc scripts/ubench/small_timeline.hip -o /tmp/small_timeline && /tmp/small_timeline 12288 7 2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlp_small_kernel.hpp"

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

using namespace qinco;

// Calibration: what does __builtin_readcyclecounter() count, and at what clock does the matrix pipe run?  One wave per SIMD issues N
// independent v_mfma_f32_16x16x4_f32 (32 pipe cycles each) between two stamps; HIP events give the wall time of the same launch.
__global__ void __launch_bounds__(256, 1) calib_kernel(int n, unsigned long long* ticks, float* sink) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float av = 1.0f + threadIdx.x * 1e-9f, bv = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k & 3] = QINCO_MFMA16(av, bv, acc[k & 3]);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = 1.f;
}

static void calibrate() {
  const int n = 1 << 18, grid = 256;
  unsigned long long* dt;
  float* ds;
  CHECK(hipMalloc(&dt, grid * 4 * 8));
  CHECK(hipMalloc(&ds, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_kernel, dim3(grid), dim3(256), 0, 0, n, dt, ds);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(grid * 4);
    CHECK(hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    const double med = (double)t[t.size() / 2];
    printf("calibration %d: %d MFMAs per wave (%.0f pipe cycles) in %.1f us: %.0f ticks (%.3f ticks per pipe cycle), %.3f ticks/ns, "
           "pipe clock if one MFMA per 32 cycles: %.3f GHz\n", rep, n, n * 32.0, ms * 1e3, med, med / (n * 32.0), med / (ms * 1e6),
           n * 32.0 / (ms * 1e6));
  }
}

int main(int argc, char** argv) {
  calibrate();
  const long R = argc > 1 ? atol(argv[1]) : 12288;
  const int steps = argc > 2 ? atoi(argv[2]) : 7;
  const int L = argc > 3 ? atoi(argv[3]) : 2;
  constexpr SmallDims S = small_dims(QD, QDE, QDH, QF2 != 0);
  constexpr SmallPlan PL = small_plan(QD, QDE, QDH, QNT, QF2 != 0, true);
  static_assert(PL.ok, "no plan");
  const int M = steps + 1, K = 256;
  const size_t stream_f4 = ((size_t)steps * S.step(L) + 64) * kSmallWaves * 64;
  std::vector<float> hw(stream_f4 * 4);
  unsigned long long seed = 12345;
  auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (float)((seed >> 40) & 0xffff) / 65536.f - 0.5f; };
  for (float& v : hw) v = 0.05f * rnd();
  float *dw, *dtt, *dpt, *dcb, *dout;
  CHECK(hipMalloc(&dw, hw.size() * 4));
  CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> tab((size_t)M * K * (QDE + QDH + QD));
  for (float& v : tab) v = rnd();
  CHECK(hipMalloc(&dtt, tab.size() * 4));
  CHECK(hipMemcpy(dtt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  dpt = dtt + (size_t)M * K * QDE;
  dcb = dpt + (size_t)M * K * QDH;
  std::vector<SmallStep> hs(M);
  for (int m = 0; m < M; ++m) hs[m] = SmallStep{dtt + (size_t)m * K * QDE, dpt + (size_t)m * K * QDH, dcb + (size_t)m * K * QD};
  SmallStep* dsteps;
  CHECK(hipMalloc(&dsteps, M * sizeof(SmallStep)));
  CHECK(hipMemcpy(dsteps, hs.data(), M * sizeof(SmallStep), hipMemcpyHostToDevice));
  std::vector<int> codes((size_t)M * R);
  for (int& c : codes) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; c = (int)((seed >> 33) % K); }
  int* dcodes;
  CHECK(hipMalloc(&dcodes, codes.size() * 4));
  CHECK(hipMemcpy(dcodes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&dout, (size_t)R * QD * 4));
  const unsigned grid = (unsigned)((R + 16 * QNT - 1) / (16 * QNT));
  unsigned long long* dtl;
  CHECK(hipMalloc(&dtl, (size_t)grid * kSmallWaves * 64 * 8));
  CHECK(hipMemset(dtl, 0, (size_t)grid * kSmallWaves * 64 * 8));
  SmallArgs a{};
  a.wstream = reinterpret_cast<const f32x4*>(dw);
  a.steps = dsteps;
  a.m_first = 1;
  a.m_count = steps;
  a.L = L;
  a.add_c = 1;
  a.R = R;
  a.codes_t = dcodes;
  a.codebook0 = dcb;
  a.out = dout;
  a.mean = nullptr;
  a.std_ = 1.f;
  a.Duser = QD;
  a.timeline = dtl;
  auto kern = mlp_small_kernel<QD, QDE, QDH, QNT, QF2 != 0, true>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL.lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kSmallWaves), PL.lds_bytes, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)R * steps * (2.0 * QD * QDE + (QF2 ? 2.0 * QDE * QDH : 0.0) + 4.0 * L * QDE * QDH - (QF2 ? 2.0 * QDE * QDH : 0.0) +
                                              (QD != QDE ? 2.0 * QD * QDE : 0.0));
    printf("launch %d: %.1f us, %u workgroups, PW=%d DB=%d lds=%u, executed %.1f TFLOP/s (%.3f of 157.3)\n", rep, ms * 1e3, grid, PL.PW, (int)PL.DB,
           PL.lds_bytes, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12);
  }
  std::vector<unsigned long long> tl((size_t)grid * kSmallWaves * 64);
  CHECK(hipMemcpy(tl.data(), dtl, tl.size() * 8, hipMemcpyDeviceToHost));
  // stamps per wave: 0 = ring prologue, then per step: FOLD2 ? 6 : 2 head stamps, 4 per remaining block, 1 step end
  const int per_step = (QF2 ? 6 : 2) + 4 * (L - (QF2 ? 1 : 0)) + 1;
  const int nst = std::min(64, 1 + per_step * steps);
  printf("stamp intervals (cycles), median over %u workgroups, per wave 0..%d; %d stamps per step\n", grid, kSmallWaves - 1, per_step);
  for (int i = 1; i < nst; ++i) {
    printf("  step %d stamp %2d:", (i - 1) / per_step, (i - 1) % per_step);
    for (int w = 0; w < kSmallWaves; ++w) {
      std::vector<long> d;
      for (unsigned g = 0; g < grid; ++g) {
        const unsigned long long a0 = tl[((size_t)g * kSmallWaves + w) * 64 + i - 1], a1 = tl[((size_t)g * kSmallWaves + w) * 64 + i];
        if (a0 && a1) d.push_back((long)(a1 - a0));
      }
      if (d.empty()) continue;
      std::sort(d.begin(), d.end());
      printf(" %7ld", d[d.size() / 2]);
    }
    printf("\n");
  }
  std::vector<long> tot;
  for (unsigned g = 0; g < grid; ++g) tot.push_back((long)(tl[(size_t)g * kSmallWaves * 64 + nst - 1] - tl[(size_t)g * kSmallWaves * 64]));
  std::sort(tot.begin(), tot.end());
  printf("first..last stamp (wave 0): median %ld, max %ld cycles\n", tot[tot.size() / 2], tot.back());
  return 0;
}
