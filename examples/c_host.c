/* A host with no Python in the process: the C ABI of include/qinco_hip.h end to end.
 *
 *   gcc -O2 -I include examples/c_host.c -o /tmp/c_host -L qinco_amd -lqinco_hip -Wl,-rpath,$PWD/qinco_amd -lm
 *   /tmp/c_host            (needs an MI355X; prints one line and exits 0)
 *
 * Builds a small QINCo2-shaped model (D = 128, de = 128, dh = 256, L = 2, M = 4, A = 16, B = 8: the qinco2-S geometry) from a
 * seeded generator in the reference's state_dict layout (qinco/model/qinco_base.py:229-260), then does what
 * qinco/search/search_tasks.py:85-137 does with the reference's model object: encode a "database" from host memory
 * (qinco_encode_host), gather the shard's codes (qinco_gather_codes, world = 1: no communicator), decode them
 * (qinco_decode_host) and report the reconstruction error next to the error of RANDOM code rows decoded by the same model -- the
 * beam search must beat them by a wide margin (the weights are random, so this says nothing about quantisation quality, only
 * that encode searches and decode inverts it) -- and the self-consistency of encode's tracked reconstruction with decode.
 * The device-pointer entry points need a HIP allocation; they are resolved from libamdhip64 with dlsym so that this file needs
 * no HIP headers. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qinco_hip.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform(void) {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}
static float gauss(void) {
  double a = 0.0;
  for (int i = 0; i < 12; ++i) a += uniform();
  return (float)(a - 6.0);
}
static float* randn(size_t n, float scale) {
  float* p = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = scale * gauss();
  return p;
}
#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != 0) {                                                              \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, qinco_last_error());        \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(void) {
  enum { D = 128, DE = 128, DH = 256, L = 2, M = 4, K = 256, A = 16, B = 8, N = 3000 };
  qinco_desc d = {D, DE, DH, L, M, K, A, B, 0, 0, 1024};
  const float* cb[M];
  const float* sub[M];
  const float* cw[M];
  const float* cbias[M];
  const float* up[M * L];
  const float* down[M * L];
  memset(sub, 0, sizeof(sub));
  memset(cw, 0, sizeof(cw));
  memset(cbias, 0, sizeof(cbias));
  memset(up, 0, sizeof(up));
  memset(down, 0, sizeof(down));
  float scale = 1.0f;
  for (int m = 0; m < M; ++m, scale *= 0.6f) {
    float* c = randn((size_t)K * D, scale);
    cb[m] = c;
    if (m == 0) continue;
    float* s = (float*)malloc((size_t)K * D * sizeof(float));      /* pre-selection codebook = main codebook + 10 % noise */
    for (size_t i = 0; i < (size_t)K * D; ++i) s[i] = c[i] + 0.1f * scale * gauss();
    sub[m] = s;
    float* w = randn((size_t)DE * (DE + D), 0.6f / sqrtf((float)(DE + D)));
    for (int i = 0; i < DE; ++i)
      for (int j = DE; j < DE + D; ++j) w[(size_t)i * (DE + D) + j] *= 0.15f;  /* f(c, xhat) dominated by c */
    cw[m] = w;
    cbias[m] = randn(DE, 0.05f);
    for (int l = 0; l < L; ++l) {
      up[m * L + l] = randn((size_t)DH * DE, 0.6f / sqrtf((float)DE));
      down[m * L + l] = randn((size_t)DE * DH, 0.6f / sqrtf((float)DH));
    }
  }
  float* mean = randn(D, 0.5f);
  qinco_weights w = {mean, 2.0f, cb, sub, NULL, NULL, cw, cbias, up, down};   /* De == D: no in_proj / out_proj */

  qinco_handle h = NULL;
  CHECK(qinco_create(&d, &w, &h));
  char desc[256];
  qinco_describe(h, desc, (int32_t)sizeof(desc));

  float* x = (float*)malloc((size_t)N * D * sizeof(float));
  for (size_t i = 0; i < (size_t)N * D; ++i) x[i] = mean[i % D] + 2.0f * gauss();
  uint8_t* codes = (uint8_t*)malloc((size_t)N * M);
  float* xhat_n = (float*)malloc((size_t)N * D * sizeof(float));
  float* dec = (float*)malloc((size_t)N * D * sizeof(float));
  CHECK(qinco_encode_host(h, x, QINCO_X_F32, 0, N, codes, QINCO_CODE_U8, xhat_n, 0));     /* 3 passes of max_batch rows */
  CHECK(qinco_decode_host(h, codes, QINCO_CODE_U8, N, dec, 0));
  uint8_t* rcodes = (uint8_t*)malloc((size_t)N * M);
  float* rdec = (float*)malloc((size_t)N * D * sizeof(float));
  for (size_t i = 0; i < (size_t)N * M; ++i) rcodes[i] = (uint8_t)(uniform() * K);
  CHECK(qinco_decode_host(h, rcodes, QINCO_CODE_U8, N, rdec, 0));

  /* the end-of-job collective on device buffers (one rank: a device copy) */
  void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!hip) hip = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
  int (*hipMalloc_)(void**, size_t) = hip ? (int (*)(void**, size_t))dlsym(hip, "hipMalloc") : NULL;
  int (*hipMemcpy_)(void*, const void*, size_t, int) = hip ? (int (*)(void*, const void*, size_t, int))dlsym(hip, "hipMemcpy") : NULL;
  int (*hipDeviceSynchronize_)(void) = hip ? (int (*)(void))dlsym(hip, "hipDeviceSynchronize") : NULL;
  int gathered_ok = -1;
  if (hipMalloc_ && hipMemcpy_ && hipDeviceSynchronize_) {
    void *dsrc = NULL, *ddst = NULL;
    uint8_t* back = (uint8_t*)malloc((size_t)N * M);
    const int64_t counts[1] = {N};
    if (hipMalloc_(&dsrc, (size_t)N * M) || hipMalloc_(&ddst, (size_t)N * M) || hipMemcpy_(dsrc, codes, (size_t)N * M, 1 /* H2D */)) return 2;
    CHECK(qinco_gather_codes(dsrc, N, M, QINCO_CODE_U8, ddst, counts, 1, 0, 0, NULL, NULL));
    if (hipDeviceSynchronize_() || hipMemcpy_(back, ddst, (size_t)N * M, 2 /* D2H */)) return 2;
    gathered_ok = memcmp(back, codes, (size_t)N * M) == 0;
  }

  double err = 0.0, err0 = 0.0, self = 0.0, scale_x = 0.0;
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < D; ++j) {
      const double xi = x[(size_t)i * D + j];
      const double e = xi - dec[(size_t)i * D + j], e0 = xi - rdec[(size_t)i * D + j];
      const double s = (double)xhat_n[(size_t)i * D + j] * 2.0 + mean[j] - dec[(size_t)i * D + j];
      err += e * e;
      err0 += e0 * e0;
      self = fmax(self, fabs(s));
      scale_x = fmax(scale_x, fabs(xi));
    }
  }
  err /= N;
  err0 /= N;
  printf("c_host: %s | %d vectors, MSE %.3f (random code rows: %.3f), encode-vs-decode reconstruction differs by %.2e relative, gather %s\n", desc,
         N, err, err0, self / scale_x, gathered_ok == 1 ? "ok" : gathered_ok == 0 ? "MISMATCH" : "skipped");
  CHECK(qinco_destroy(h));
  return (err < 0.75 * err0 && self / scale_x < 1e-5 && gathered_ok != 0) ? 0 : 3;
}
