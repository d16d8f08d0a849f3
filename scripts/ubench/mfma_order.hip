// Does a scalar chain of fmaf reproduce the bits of a v_mfma_f32_32x32x2_f32 chain?  (round 6: an exact pass that recomputes a few
// (query, row) pairs must hand the table kernel's bits back.)  knn_table_kernel<128, false> writes |q|^2 + |x|^2 - 2 q.x for 64 queries x
// 1024 rows; the host recomputes every pair with sequential fmaf in the order the kernel's MFMAs contract the features
// (block ib, quarter q, e = 0..3; per MFMA k = 0 then k = 1: features ib*32 + 8q + e, then ib*32 + 8q + 4 + e) and in two other orders.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I qinco_amd/csrc scripts/ubench/mfma_order.hip -o scripts/ubench/mfma_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "knn_kernel.hpp"
using namespace qinco;
int main() {
  const int D = 128, N = 1024, Q = 64;
  std::vector<float> db(N * D), q(Q * D);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 4.f; };
  for (auto& v : db) v = rnd();
  for (auto& v : q) v = rnd() * (1.f + 3.f * (rnd() > 1.5f));
  float *ddb, *dq, *dqn, *dt; f32x4* qs;
  hipMalloc(&ddb, db.size() * 4); hipMalloc(&dq, q.size() * 4); hipMalloc(&dqn, Q * 4); hipMalloc(&dt, (size_t)Q * N * 4);
  hipMalloc(&qs, ((size_t)Q * D + 16 * 256) * 4); hipMemset(qs, 0, ((size_t)Q * D + 16 * 256) * 4);
  hipMemcpy(ddb, db.data(), db.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
  knn_pack_rows_kernel<<<64, 256>>>(dq, Q, D, qs, dqn, Q / 32);
  KnnFilt f{};
  hipLaunchKernelGGL((knn_table_kernel<128, false>), dim3(N / 128), dim3(256), 0, 0, qs, dqn, Q / 32, ddb, (long)N, 1L, dt, (long)N, f);
  std::vector<float> t((size_t)Q * N), qn(Q);
  hipMemcpy(t.data(), dt, t.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(qn.data(), dqn, Q * 4, hipMemcpyDeviceToHost);
  auto norm_kernel_order = [&](const float* x) {   // knn_table_kernel: lane half h sums its features (ib, q, e) in order, then the halves are added
    float h[2] = {0.f, 0.f};
    for (int hh = 0; hh < 2; ++hh)
      for (int ib = 0; ib < 4; ++ib) for (int qq = 0; qq < 4; ++qq) for (int e = 0; e < 4; ++e) { const float v = x[ib * 32 + 8 * qq + 4 * hh + e]; h[hh] = fmaf(v, v, h[hh]); }
    return h[0] + h[1];
  };
  long bad[3] = {0, 0, 0};
  for (int qi = 0; qi < Q; ++qi)
    for (int n = 0; n < N; ++n) {
      const float* a = &q[qi * D]; const float* x = &db[n * D];
      float acc[3] = {0.f, 0.f, 0.f};
      for (int ib = 0; ib < 4; ++ib) for (int qq = 0; qq < 4; ++qq) for (int e = 0; e < 4; ++e) {
        const int f0 = ib * 32 + 8 * qq + e, f1 = f0 + 4;
        acc[0] = fmaf(a[f1], x[f1], fmaf(a[f0], x[f0], acc[0]));                  // k = 0 then k = 1, each its own rounding
        acc[1] = fmaf(a[f0], x[f0], fmaf(a[f1], x[f1], acc[1]));                  // k = 1 then k = 0
        acc[2] = (float)((double)acc[2] + (double)a[f0] * x[f0] + (double)a[f1] * x[f1]);   // both products exact, one rounding
      }
      const float xn = norm_kernel_order(x);
      for (int o = 0; o < 3; ++o) {
        float d = (qn[qi] + xn) - 2.f * acc[o];
        unsigned u, v; std::memcpy(&u, &d, 4); std::memcpy(&v, &t[(size_t)qi * N + n], 4);
        bad[o] += u != v;
      }
    }
  printf("pairs %d; differing bits: fmaf k0,k1 sequential %ld | fmaf k1,k0 %ld | 2-term exact sum, one rounding %ld\n", Q * N, bad[0], bad[1], bad[2]);
  return 0;
}
