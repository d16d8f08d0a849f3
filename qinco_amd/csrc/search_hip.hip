// C ABI of the evaluation stages either side of the hot path (include/qinco_hip.h): brute-force top-k of the
// small-db search (run_search_full_direct_small_db, reference qinco/search/search_tasks.py:551-603) and the
// squared-error sum of compute_MSE / AnyVectMSE (qinco/qinco_tasks.py:87-148, qinco/metrics.py:29-58).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "abi_util.hpp"
#include "knn_kernel.hpp"
#include "knn_roles_kernel.hpp"
#include "rerank_kernel.hpp"

using namespace qinco;
#define fail qinco::abi_fail

struct qinco_knn_s {
  int device = 0;
  int D = 0;
  f32x4* qstream = nullptr;  // packed query fragments of one chunk (+16 fragments of padding)
  float* qnorm = nullptr;
  size_t q_rows = 0;         // chunk capacity in query rows (multiple of 32)
  float* table = nullptr;
  size_t table_elems = 0;
  // the filtered form (knn_kernel.hpp): per-chunk candidate lists, thresholds, counters, and one fall-back flag per chunk
  unsigned long long* cand = nullptr;
  size_t cand_bytes = 0;
  unsigned* tau = nullptr;   // [chunk] thresholds, then [chunk] counters
  size_t tau_bytes = 0;
  int* ovf = nullptr;
  size_t ovf_bytes = 0;
  long last_chunks = 0, last_filtered = 0;   // of the last search: chunks, chunks that took the filtered form
  int filter_mode = 1;       // 0: never, 1: where it pays (n >= filter_min_n and a sampling stride >= 4)
  long filter_min_n = 65536;
  int roles_mode = 0;        // opt-in: the filtered table as knn_table_roles_kernel (MFMA waves + filter waves) where D <= 128
  int* roles_stat = nullptr; // {workgroups whose waves covered all four SIMDs, workgroups that fell back to index roles}
  long q_stream_bytes = 0;   // query fragments of one chunk: every wave walks them, they should stay in L2 (0: 1 MiB, 2 MiB for D >= 512)
  // staging for the host form
  void* s_db = nullptr;
  size_t s_db_bytes = 0;
  void* s_q = nullptr;
  size_t s_q_bytes = 0;
  void* s_ids = nullptr;
  size_t s_ids_bytes = 0;
  void* s_dist = nullptr;
  size_t s_dist_bytes = 0;
};

static int grow(void** p, size_t* cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  HIP_TRY(hipMalloc(p, need ? need : 16));
  *cap = need;
  return 0;
}

static bool knn_dim_ok(int D) { return D == 32 || D == 64 || D == 96 || D == 128 || D == 256 || D == 768; }

extern "C" int qinco_knn_create(int32_t D, qinco_knn* out) {
  if (!out) return fail(QINCO_ERR_INVALID, "qinco_knn_create: null argument");
  if (!knn_dim_ok(D)) return fail(QINCO_ERR_UNSUPPORTED, "qinco_knn_create: no table kernel instance for D=%d", D);
  qinco_knn_s* s = new qinco_knn_s();
  s->D = D;
  if (hipGetDevice(&s->device) != hipSuccess) {
    delete s;
    return fail(QINCO_ERR_HIP, "hipGetDevice failed (no HIP device?)");
  }
  // knn_select_kernel / knn_cand_select_kernel hold 80 KiB of static LDS (histogram + the 8192-key buffer) and two workgroups of
  // knn_table_kernel<D, true> (57 KiB each) share a CU: a 160 KiB-LDS part (gfx950).  This library is built for gfx950 only; the
  // check turns a launch failure on anything else into a message.
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device) != hipSuccess || lds < kKnnLdsNeeded) {
    delete s;
    return fail(QINCO_ERR_UNSUPPORTED, "qinco_knn_create: the search kernels need %d KiB of LDS per workgroup, this device offers %d KiB",
                kKnnLdsNeeded >> 10, lds >> 10);
  }
  if (hipMalloc((void**)&s->roles_stat, 2 * sizeof(int)) != hipSuccess || hipMemset(s->roles_stat, 0, 2 * sizeof(int)) != hipSuccess) {
    delete s;
    return fail(QINCO_ERR_HIP, "qinco_knn_create: hipMalloc failed");
  }
  *out = s;
  return QINCO_OK;
}

extern "C" int qinco_knn_roles_stats(qinco_knn s, int64_t* out2) {
  if (!s || !out2) return fail(QINCO_ERR_INVALID, "qinco_knn_roles_stats: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  int h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, s->roles_stat, sizeof(h), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(s->roles_stat, 0, sizeof(h)));
  out2[0] = h[0];
  out2[1] = h[1];
  return QINCO_OK;
}

extern "C" int qinco_knn_destroy(qinco_knn s) {
  if (!s) return QINCO_OK;
  (void)hipDeviceSynchronize();
  for (void* p : {(void*)s->qstream, (void*)s->qnorm, (void*)s->table, (void*)s->cand, (void*)s->tau, (void*)s->ovf, (void*)s->roles_stat, s->s_db, s->s_q,
                  s->s_ids, s->s_dist})
    if (p) (void)hipFree(p);
  delete s;
  return QINCO_OK;
}

template <int D>
static void launch_table_d(qinco_knn_s* s, int nqblocks, const float* db, long n, long stride, long ldt, bool filt, const KnnFilt& f,
                           hipStream_t st) {
  const dim3 grid((unsigned)((n + 127) / 128));
  if constexpr (D <= 128) {
    if (filt && s->roles_mode && stride == 1) {
      hipLaunchKernelGGL((knn_table_roles_kernel<D>), grid, dim3(512), 0, st, s->qstream, s->qnorm, nqblocks, db, n, f, s->roles_stat);
      return;
    }
  }
  if (filt)
    hipLaunchKernelGGL((knn_table_kernel<D, true>), grid, dim3(256), 0, st, s->qstream, s->qnorm, nqblocks, db, n, stride, s->table, ldt, f);
  else
    hipLaunchKernelGGL((knn_table_kernel<D, false>), grid, dim3(256), 0, st, s->qstream, s->qnorm, nqblocks, db, n, stride, s->table, ldt, f);
}

// columns 0 .. n-1 of the table = database rows 0, stride, 2 stride, ...
static int launch_table(qinco_knn_s* s, int nqblocks, const float* db, long n, long stride, long ldt, bool filt, const KnnFilt& f,
                        hipStream_t st) {
  switch (s->D) {
    case 32: launch_table_d<32>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    case 64: launch_table_d<64>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    case 96: launch_table_d<96>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    case 128: launch_table_d<128>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    case 256: launch_table_d<256>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    case 768: launch_table_d<768>(s, nqblocks, db, n, stride, ldt, filt, f, st); break;
    default: return fail(QINCO_ERR_UNSUPPORTED, "no table kernel instance for D=%d", s->D);
  }
  HIP_TRY(hipGetLastError());
  return QINCO_OK;
}

extern "C" int qinco_knn_set_option(qinco_knn s, int32_t option, int64_t value) {
  if (!s) return fail(QINCO_ERR_INVALID, "qinco_knn_set_option: null handle");
  switch (option) {
    case QINCO_KNN_OPT_FILTER: s->filter_mode = value != 0; return QINCO_OK;
    case QINCO_KNN_OPT_FILTER_MIN_N:
      if (value < 1) return fail(QINCO_ERR_INVALID, "qinco_knn_set_option: filter_min_n must be >= 1");
      s->filter_min_n = (long)value;
      return QINCO_OK;
    case QINCO_KNN_OPT_ROLES: s->roles_mode = value != 0; return QINCO_OK;
    case QINCO_KNN_OPT_QUERY_BYTES:
      if (value < 4096) return fail(QINCO_ERR_INVALID, "qinco_knn_set_option: query_bytes must be >= 4096");
      s->q_stream_bytes = (long)value;
      return QINCO_OK;
    default: return fail(QINCO_ERR_INVALID, "qinco_knn_set_option: unknown option %d", (int)option);
  }
}

extern "C" int qinco_knn_last_stats(qinco_knn s, int64_t* out3) {
  if (!s || !out3) return fail(QINCO_ERR_INVALID, "qinco_knn_last_stats: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  out3[0] = s->last_chunks;
  out3[1] = s->last_filtered;
  out3[2] = 0;
  if (s->last_filtered > 0) {
    std::vector<int> h((size_t)s->last_chunks);
    HIP_TRY(hipMemcpy(h.data(), s->ovf, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int v : h) out3[2] += v != 0;
  }
  return QINCO_OK;
}

extern "C" int qinco_knn_search(qinco_knn s, const float* db, int64_t n, const float* queries, int64_t nq, int32_t k,
                                int64_t* ids_out, float* dist_out, void* stream) {
  if (!s) return fail(QINCO_ERR_INVALID, "qinco_knn_search: null handle");
  if (n < 1 || n > (int64_t)1 << 31 || nq < 0 || k < 1 || k > kKnnMaxK || k > n)
    return fail(QINCO_ERR_INVALID, "qinco_knn_search: need 1 <= k <= min(n, %d), n <= 2^31 (n=%lld, k=%d)", kKnnMaxK, (long long)n, k);
  if (nq == 0) return QINCO_OK;
  if (!db || !queries || !ids_out) return fail(QINCO_ERR_INVALID, "qinco_knn_search: null buffer");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int D = s->D;
  const long ldt = (n + 31) / 32 * 32;
  // chunk of queries: table of at most 8 GiB (the card has 288 GB; the database is re-read once per chunk)
  long chunk = (long)(((size_t)8 << 30) / ((size_t)ldt * 4)) / 32 * 32;
  if (chunk < 32) chunk = 32;
  if (chunk > kKnnMaxChunk) chunk = kKnnMaxChunk;
  // every wave loops over the whole query stream: keep it L2-resident (4 MiB per XCD), about 1 MiB
  // (measured at N = 10^6, round 5: D = 128 is indifferent between 0.5 and 2 MiB; D = 768 gains 7 % from 2 MiB -- 4 passes over the
  // database instead of 7 and waves that live twice as long)
  const long q_bytes = s->q_stream_bytes ? s->q_stream_bytes : (D >= 512 ? (long)2 << 20 : (long)1 << 20);
  const long l2_rows = q_bytes / (D * 4) / 32 * 32;
  if (chunk > l2_rows) chunk = l2_rows;
  if (chunk < 32) chunk = 32;   // (a query_bytes below one 32-row block of this D -- 4096 B at D >= 64 -- must not give 0 rows)
  const long nq_pad = (nq + 31) / 32 * 32;
  if (chunk > nq_pad) chunk = nq_pad;
  int rc;
  if (s->q_rows < (size_t)chunk) {
    size_t cap = 0;
    if (s->qstream) (void)hipFree(s->qstream);
    if (s->qnorm) (void)hipFree(s->qnorm);
    s->qstream = nullptr;
    s->qnorm = nullptr;
    s->q_rows = 0;
    if ((rc = grow((void**)&s->qstream, &cap, ((size_t)chunk * D + 16 * 256) * sizeof(float)))) return rc;
    HIP_TRY(hipMemsetAsync(s->qstream, 0, ((size_t)chunk * D + 16 * 256) * sizeof(float), st));
    cap = 0;
    if ((rc = grow((void**)&s->qnorm, &cap, (size_t)chunk * sizeof(float)))) return rc;
    s->q_rows = (size_t)chunk;
  }
  {
    size_t bytes = s->table_elems * sizeof(float);
    if ((rc = grow((void**)&s->table, &bytes, (size_t)chunk * ldt * sizeof(float)))) return rc;
    s->table_elems = bytes / sizeof(float);
  }
  // The filtered form (knn_kernel.hpp): sampling stride so that a query expects about kKnnCap / 4 candidates (k * stride of them
  // on exchangeable data); below a stride of 4 the sample pass costs more than the table it saves.
  long stride = (long)kKnnCap / (4 * (long)k);
  if (stride > 32) stride = 32;
  const bool filtered = s->filter_mode != 0 && n >= s->filter_min_n && stride >= 4 && (n + stride - 1) / stride >= k;
  const long ns = filtered ? (long)((n + stride - 1) / stride) : 0;   // sample rows 0, stride, 2 stride, ...
  const long lds_ = (ns + 31) / 32 * 32;
  const long nchunks = (long)((nq + chunk - 1) / chunk);
  if (filtered) {
    if ((rc = grow((void**)&s->cand, &s->cand_bytes, (size_t)chunk * kKnnCap * sizeof(unsigned long long)))) return rc;
    if ((rc = grow((void**)&s->tau, &s->tau_bytes, (size_t)chunk * 2 * sizeof(unsigned)))) return rc;
    if ((rc = grow((void**)&s->ovf, &s->ovf_bytes, (size_t)nchunks * sizeof(int)))) return rc;
    HIP_TRY(hipMemsetAsync(s->ovf, 0, (size_t)nchunks * sizeof(int), st));
  }
  s->last_chunks = nchunks;
  s->last_filtered = filtered ? nchunks : 0;
  long ci = 0;
  for (int64_t q0 = 0; q0 < nq; q0 += chunk, ++ci) {
    const long cq = (nq - q0 < chunk) ? (long)(nq - q0) : chunk;
    const long nqb = (cq + 31) / 32;
    long g = (nqb * 32 * (D / 4) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(knn_pack_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, queries + q0 * D, cq, D, s->qstream, s->qnorm, nqb);
    HIP_TRY(hipGetLastError());
    long long* ids_q = reinterpret_cast<long long*>(ids_out) + q0 * k;
    float* dist_q = dist_out ? dist_out + q0 * k : nullptr;
    KnnFilt f{};
    if (filtered) {
      unsigned* cnt = s->tau + chunk;
      // (1) thresholds from the sample  (2) the whole table, survivors appended  (3) each query's candidates sorted
      if ((rc = launch_table(s, (int)nqb, db, ns, stride, lds_, false, f, st))) return rc;
      hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)cq), dim3(kKnnThreads), 0, st, s->table, lds_, ns, (int)k, nullptr,
                         nullptr, s->tau, cnt, nullptr);
      HIP_TRY(hipGetLastError());
      f.tau = s->tau;
      f.cnt = cnt;
      f.cand = s->cand;
      f.nq_valid = (int)cq;
      if ((rc = launch_table(s, (int)nqb, db, (long)n, 1, ldt, true, f, st))) return rc;
      hipLaunchKernelGGL(knn_cand_select_kernel, dim3((unsigned)cq), dim3(kKnnThreads), 0, st, s->cand, cnt, (int)k, ids_q, dist_q,
                         s->ovf + ci);
      HIP_TRY(hipGetLastError());
      f = KnnFilt{};
      f.pred = s->ovf + ci;   // the unfiltered kernels below run only if a list overflowed (or came up short)
    }
    if ((rc = launch_table(s, (int)nqb, db, (long)n, 1, ldt, false, f, st))) return rc;
    hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)cq), dim3(kKnnThreads), 0, st, s->table, ldt, (long)n, (int)k, ids_q, dist_q,
                       nullptr, nullptr, f.pred);
    HIP_TRY(hipGetLastError());
  }
  return QINCO_OK;
}

extern "C" int qinco_knn_search_host(qinco_knn s, const float* db, int64_t n, const float* queries, int64_t nq, int32_t k,
                                     int64_t* ids_out, float* dist_out) {
  if (!s) return fail(QINCO_ERR_INVALID, "qinco_knn_search_host: null handle");
  if (nq == 0 && n >= 1 && k >= 1 && k <= n) return QINCO_OK;
  if (!db || !queries || !ids_out || n < 1 || nq < 0 || k < 1) return fail(QINCO_ERR_INVALID, "qinco_knn_search_host: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  const size_t dbb = (size_t)n * s->D * 4, qb = (size_t)nq * s->D * 4, ib = (size_t)nq * k * 8, fb = (size_t)nq * k * 4;
  if ((rc = grow(&s->s_db, &s->s_db_bytes, dbb))) return rc;
  if ((rc = grow(&s->s_q, &s->s_q_bytes, qb))) return rc;
  if ((rc = grow(&s->s_ids, &s->s_ids_bytes, ib))) return rc;
  if (dist_out && (rc = grow(&s->s_dist, &s->s_dist_bytes, fb))) return rc;
  HIP_TRY(hipMemcpy(s->s_db, db, dbb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->s_q, queries, qb, hipMemcpyHostToDevice));
  if ((rc = qinco_knn_search(s, (const float*)s->s_db, n, (const float*)s->s_q, nq, k, (int64_t*)s->s_ids,
                             dist_out ? (float*)s->s_dist : nullptr, nullptr)))
    return rc;
  HIP_TRY(hipMemcpy(ids_out, s->s_ids, ib, hipMemcpyDeviceToHost));
  if (dist_out) HIP_TRY(hipMemcpy(dist_out, s->s_dist, fb, hipMemcpyDeviceToHost));
  return QINCO_OK;
}

extern "C" int qinco_sqerr_sum(const float* a, const float* b, int64_t count, double* sum_out, void* stream) {
  if (!sum_out || count < 0) return fail(QINCO_ERR_INVALID, "qinco_sqerr_sum: bad argument");
  *sum_out = 0.0;
  if (count == 0) return QINCO_OK;
  if (!a || !b) return fail(QINCO_ERR_INVALID, "qinco_sqerr_sum: null buffer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, sizeof(double)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(double), st);
  if (e == hipSuccess) {
    long g = (count + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(sqerr_sum_kernel, dim3((unsigned)g), dim3(256), 0, st, a, b, (long)count, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(sum_out, d, sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(QINCO_ERR_HIP, "qinco_sqerr_sum failed: %s", hipGetErrorString(e));
  return QINCO_OK;
}

// ---------------------------------------------------------------------------------------------
// re-rank of per-query shortlists (run_search_ivf's re-rank stages, search_tasks.py:447-472, 497-507)
// ---------------------------------------------------------------------------------------------
extern "C" int qinco_rerank(const float* xq, const float* cand, int64_t nq, int32_t ns, int32_t D, int32_t k, const int64_t* ids_in,
                            const int32_t* codes_in, int32_t Mc, int64_t* pos_out, float* dist_out, int64_t* ids_out, int32_t* codes_out,
                            void* stream) {
  if (nq < 0 || ns < 1 || D < 1 || k < 1 || k > ns) return fail(QINCO_ERR_INVALID, "qinco_rerank: need nq >= 0, 1 <= k <= ns, D >= 1");
  if (nq == 0) return QINCO_OK;
  if (!xq || !cand) return fail(QINCO_ERR_INVALID, "qinco_rerank: null buffer");
  if ((ids_out && !ids_in) || (codes_out && (!codes_in || Mc < 1))) return fail(QINCO_ERR_INVALID, "qinco_rerank: an output without its input");
  int P = 1;
  while (P < ns) P <<= 1;
  const size_t lds = (size_t)P * 8 + (size_t)D * 4;
  if (lds > 160 * 1024) return fail(QINCO_ERR_UNSUPPORTED, "qinco_rerank: a shortlist of %d rows of %d features does not fit the 160 KiB of LDS", ns, D);
  // (the attribute applies to the CURRENT device's copy of the function: the high-water mark is kept per device -- a process that
  // drives a second GPU has to raise the limit there too)
  static size_t raised[64] = {};
  int devid = 0;
  HIP_TRY(hipGetDevice(&devid));
  size_t& mark = raised[devid & 63];
  if (lds > 64 * 1024 && lds > mark) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rerank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mark = lds;
  }
  RerankArgs a{xq, cand, (long)nq, ns, D, k, P, reinterpret_cast<const long long*>(ids_in), codes_in, Mc,
               reinterpret_cast<long long*>(pos_out), dist_out, reinterpret_cast<long long*>(ids_out), codes_out};
  hipLaunchKernelGGL(rerank_kernel, dim3((unsigned)nq), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return QINCO_OK;
}
