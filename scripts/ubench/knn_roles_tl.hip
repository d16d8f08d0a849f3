// Timeline of knn_table_roles_kernel<128> (round 6) against knn_table_kernel<128, true> on the search's shape (10^6 rows, one chunk
// of 2048 queries, ~0.2 % survivors): launch time of both, and per wave of eight workgroups of the two-role kernel: role, SIMD,
// cycles alive, cycles inside the per-block barrier, cycles in list flushes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -DQINCO_KNN_TIMELINE -I qinco_amd/csrc scripts/ubench/knn_roles_tl.hip -o scripts/ubench/knn_roles_tl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "knn_roles_kernel.hpp"
using namespace qinco;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    unsigned h2 = h * 2654435761u + 12345u; h2 ^= h2 >> 15; h2 *= 2246822519u; h2 ^= h2 >> 13;
    const float u1 = ((h >> 8) + 1) * (1.f / 16777217.f), u2 = (h2 >> 8) * (1.f / 16777216.f);
    p[i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
}
int main() {
  const int D = 128; const long N = 1000000; const int Q = 2048, nqb = Q / 32;
  float *db, *q, *qnorm; f32x4* qs; unsigned *tau, *cnt; unsigned long long* cand; int* rs; long long* tl;
  CK(hipMalloc(&db, N * D * 4)); CK(hipMalloc(&q, (long)Q * D * 4)); CK(hipMalloc(&qs, ((long)Q * D + 16 * 256) * 4)); CK(hipMalloc(&qnorm, Q * 4));
  CK(hipMalloc(&tau, Q * 8)); cnt = tau + Q; CK(hipMalloc(&cand, (long)Q * kKnnCap * 8)); CK(hipMalloc(&rs, 8)); CK(hipMalloc(&tl, 8 * 8 * 8 * 8));
  CK(hipMemset(qs, 0, ((long)Q * D + 16 * 256) * 4)); CK(hipMemset(rs, 0, 8)); CK(hipMemset(tl, 0, 4096));
  fill<<<4096, 256>>>(db, N * D, 1u); fill<<<256, 256>>>(q, (long)Q * D, 77u);
  knn_pack_rows_kernel<<<1024, 256>>>(q, Q, D, qs, qnorm, nqb);
  std::vector<unsigned> h(Q, 0u);
  const float thr = 165.f; unsigned key; { unsigned u; std::memcpy(&u, &thr, 4); key = u | 0x80000000u; }
  for (auto& v : h) v = key;
  CK(hipMemcpy(tau, h.data(), Q * 4, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_tl), &tl, sizeof(tl)));
  KnnFilt f{}; f.tau = tau; f.cnt = cnt; f.cand = cand; f.nq_valid = Q;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid((unsigned)((N + 127) / 128));
  for (int form = 0; form < 6; ++form)
    for (int rep = 0; rep < 3; ++rep) {
      if (form != 1 && form != 5) { const int pm = form == 0 ? 0 : form - 1; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_prio), &pm, sizeof(pm))); }
      CK(hipMemset(cnt, 0, Q * 4));
      hipEventRecord(e0, 0);
      if (form == 5) hipLaunchKernelGGL((knn_table_kernel<128, true, 1>), grid, dim3(256), 0, 0, qs, qnorm, nqb, db, N, 1L, (float*)nullptr, 0L, f);
      else if (form != 1) hipLaunchKernelGGL((knn_table_roles_kernel<128>), grid, dim3(512), 0, 0, qs, qnorm, nqb, db, N, f, rs);
      else hipLaunchKernelGGL((knn_table_kernel<128, true>), grid, dim3(256), 0, 0, qs, qnorm, nqb, db, N, 1L, (float*)nullptr, 0L, f);
      hipEventRecord(e1, 0); CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned> c(Q); CK(hipMemcpy(c.data(), cnt, Q * 4, hipMemcpyDeviceToHost));
      double tot = 0; for (auto v : c) tot += v;
      printf("%s: %.3f ms = %.1f TFLOP/s  (survivors %.3f %% of the pairs)\n", form == 0 ? "two roles, MFMA prio 3" : form == 1 ? "one role " : form == 5 ? "one role, ONE wave per SIMD" : form == 2 ? "two roles, no prio" : form == 3 ? "two roles, filter prio 3" : "two roles, filter prio 1", ms, 2.0 * D * N * Q / ms / 1e9, 100.0 * tot / ((double)N * Q));
    }
  std::vector<long long> t(512); CK(hipMemcpy(t.data(), tl, 4096, hipMemcpyDeviceToHost));
  for (int w = 0; w < 64; ++w) {
    const long long* o = &t[w * 8];
    printf("wg %5lld wave %d: %s simd %lld pair %lld  alive %8lld cycles  in barriers %8lld (%.0f per block)  flushes %7lld\n", o[6], w & 7, o[0] ? "MFMA  " : "filter", o[1], o[5], o[2], o[3],
           (double)o[3] / (nqb + 1), o[4]);
  }
  int hrs[2]; CK(hipMemcpy(hrs, rs, 8, hipMemcpyDeviceToHost)); printf("workgroups spread / by index: %d / %d\n", hrs[0], hrs[1]);
  return 0;
}
