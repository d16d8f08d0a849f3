// Re-rank of a per-query shortlist: the two re-rank stages of run_search_ivf (reference qinco/search/search_tasks.py:447-472 and
// :497-507) -- compute_batch_distances(xq[:, None], shortlist, approx=True) (utils.py:349-383: |a|^2 + |b|^2 - 2 a.b), argsort per
// query, take_along_dim of the ids (and of the code rows) of the first k -- as ONE kernel: a workgroup per query.
//   * a wave per candidate row: lanes stride the D features (coalesced 256 B segments from HBM -- this is an HBM-bound stage:
//     nq x ns x D x 4 B read once), dot product and norm reduced across the wave;
//   * the ns (distance, position) keys of the query meet in LDS and are sorted there (bitonic, 64-bit keys: ascending distance,
//     ties -> the earlier shortlist position; torch.argsort leaves the order among exact ties unspecified);
//   * the first k positions, their distances, their database ids and their code rows are written out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "select.hpp"

namespace qinco {

struct RerankArgs {
  const float* xq;        // (nq, D)
  const float* cand;      // (nq, ns, D) decoded shortlist
  long nq;
  int ns, D, k, P;        // P = ns rounded up to a power of two (sort size)
  const long long* ids_in;   // (nq, ns) database ids of the shortlist, or nullptr
  const int* codes_in;    // (nq, ns, Mc) code rows of the shortlist, or nullptr
  int Mc;
  long long* pos_out;     // (nq, k) shortlist positions, or nullptr
  float* dist_out;        // (nq, k) or nullptr
  long long* ids_out;     // (nq, k) or nullptr
  int* codes_out;         // (nq, k, Mc) or nullptr
};

__global__ void __launch_bounds__(256) rerank_kernel(RerankArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long rr_keys[];   // P keys, then D floats of the query
  float* xs = reinterpret_cast<float*>(rr_keys + a.P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long q = blockIdx.x;
  const float* xq = a.xq + q * a.D;
  for (int j = tid; j < a.D; j += 256) xs[j] = xq[j];
  for (int i = a.ns + tid; i < a.P; i += 256) rr_keys[i] = ~0ull;   // padding sorts last
  __syncthreads();
  float an = 0.f;   // |x|^2, the same value in every thread (summed in index order)
  for (int j = 0; j < a.D; ++j) an = fmaf(xs[j], xs[j], an);
  const float* cq = a.cand + q * (long)a.ns * a.D;
  for (int c = wave; c < a.ns; c += 4) {
    const float* row = cq + (long)c * a.D;
    float ab = 0.f, bn = 0.f;
    for (int j = lane; j < a.D; j += 64) {
      const float v = row[j];
      ab = fmaf(xs[j], v, ab);
      bn = fmaf(v, v, bn);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      ab += __shfl_xor(ab, s);
      bn += __shfl_xor(bn, s);
    }
    if (lane == 0) {
      const float d = (an + bn) - 2.f * ab;   // approx_compute_batch_distances: anorms + bnorms - 2 bmm
      rr_keys[c] = ((unsigned long long)sel_key(d) << 32) | (unsigned)c;
    }
  }
  __syncthreads();
  // bitonic sort of the P keys, ascending
  for (int kk = 2; kk <= a.P; kk <<= 1) {
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      for (int i = tid; i < a.P; i += 256) {
        const int l = i ^ jj;
        if (l > i) {
          const unsigned long long x = rr_keys[i], y = rr_keys[l];
          const bool up = (i & kk) == 0;
          if ((x > y) == up) {
            rr_keys[i] = y;
            rr_keys[l] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < a.k; t += 256) {
    const unsigned long long key = rr_keys[t];
    const int c = (int)(unsigned)key;
    if (a.pos_out) a.pos_out[q * a.k + t] = c;
    if (a.dist_out) {
      unsigned u = (unsigned)(key >> 32);
      a.dist_out[q * a.k + t] = u == 0xffffffffu ? __builtin_nanf("") : from_ordered_bits(u);
    }
    if (a.ids_out) a.ids_out[q * a.k + t] = a.ids_in[q * a.ns + c];
  }
  if (a.codes_out) {
    for (int e = tid; e < a.k * a.Mc; e += 256) {
      const int t = e / a.Mc, m = e - t * a.Mc;
      const int c = (int)(unsigned)rr_keys[t];
      a.codes_out[(q * a.k + t) * a.Mc + m] = a.codes_in[(q * (long)a.ns + c) * a.Mc + m];
    }
  }
}

}  // namespace qinco
