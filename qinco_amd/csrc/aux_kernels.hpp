// Auxiliary gfx950 kernels of the QINCo encode/decode path: (de)normalisation, the residual->codebook
// L2 distance table fused with top-T selection, the per-vector beam top-B select/prune, gathers.
// All of these are <0.1 % of the FLOPs; they are HBM/LDS/latency work:
// coalesced row reads, LDS-staged codebook tiles, wave64 shuffle reductions for the selections.
#pragma once
#include "select.hpp"

namespace qinco {

// self-test kernels (qinco_selftest): one wave per problem
__global__ void __launch_bounds__(64) selftest_sort_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out) {
  const int lane = threadIdx.x;
  out[(long)blockIdx.x * 64 + lane] = wave_sort64(in[(long)blockIdx.x * 64 + lane]);
}
// pair_top_t (select.hpp) on tiles of 32 rows x 256 distances given in memory: lane (j, half) loads row j in the MFMA C layout
// (codeword 32 cb + 8 gq + 4 half + e -> register 16 cb + 4 gq + e).  coop != 0: 8 rows per wave, the other lanes repeat them and
// do not store (a wave with most of its pairs idle).
__global__ void __launch_bounds__(64) selftest_pair_kernel(const float* __restrict__ d, int T, int* __restrict__ ids, int coop,
                                                           int* __restrict__ rounds) {
  __shared__ __attribute__((aligned(16))) unsigned lists[pair_lds_words<64>()];
  const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
  const int rows = coop ? 8 : 32;
  const long row = (long)blockIdx.x * rows + (coop ? (j & 7) : j);
  f32x16 acc[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(d + row * 256 + cb * 32 + 8 * gq + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cb][4 * gq + e] = t[e];
    }
  if (T == 0) {   // (self-test of the exchange itself: every lane reports what pair_partner hands it for its own lane number / a double of it)
    ids[blockIdx.x * 128 + lane] = (int)pair_partner_u((unsigned)lane + 1000u, half);
    ids[blockIdx.x * 128 + 64 + lane] = (int)pair_partner((double)lane * 0.5 + 7.0, half);
    return;
  }
  pair_top_t<8>(acc, lane, T, lists, ids + row * T, coop ? j < 8 : true, rounds + row);
}

__global__ void __launch_bounds__(64) selftest_select_kernel(const float* __restrict__ d, int C, int T, int* __restrict__ ids,
                                                             int* __restrict__ fell_back) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  float* dv = lds;
  unsigned long long* surv = reinterpret_cast<unsigned long long*>(lds + ((C + 3) & ~3));
  for (int k = lane; k < C; k += 64) dv[k] = d[(long)blockIdx.x * C + k];
  __builtin_amdgcn_wave_barrier();
  int rank, index;
  const bool ok = wave_select_smallest(dv, C, T, surv, lane, rank, index);
  if (ok && rank >= 0) ids[(long)blockIdx.x * T + rank] = index;
  if (lane == 0) fell_back[blockIdx.x] = ok ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// K8: x_n = (x - mean) / std   (qinco_inference.py:277).  x is fp32 or uint8 rows with a byte stride
// (bvecs rows are d+4 bytes apart: search_tasks.py:109-110 converts the raw slice with .to(float32)).
// ---------------------------------------------------------------------------------------------
__global__ void normalize_kernel(const void* __restrict__ x, int x_dtype, long row_stride_bytes,
                                 const float* __restrict__ mean, float std_, float* __restrict__ out,
                                 long n, int D, int Dp) {
  // D = the model's data dimension (columns of x), Dp >= D = the kernels' padded dimension (columns of out; the padding
  // features are exact zeros in the input, the codebooks and the weights, so they contribute exact zeros everywhere)
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = n * Dp;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx / Dp;
    int d = (int)(idx - r * Dp);
    if (d >= D) { out[idx] = 0.f; continue; }
    const char* rowp = reinterpret_cast<const char*>(x) + r * row_stride_bytes;
    float v = x_dtype == 0 ? reinterpret_cast<const float*>(rowp)[d]
                           : (float)reinterpret_cast<const unsigned char*>(rowp)[d];
    out[idx] = mean ? __fdiv_rn(__fsub_rn(v, mean[d]), std_) : v;   // mean == nullptr: x is already normalised
  }
}

// x = xhat * std + mean  (qinco_inference.py:281): two roundings (no fma contraction), like ATen mul + add.
// xhat rows are Dp floats apart, out rows D.
__global__ void denormalize_kernel(const float* __restrict__ xhat, const float* __restrict__ mean,
                                   float std_, float* __restrict__ out, long n, int D, int Dp) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = n * D;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx / D;
    int d = (int)(idx - r * D);
    const float v = xhat[r * Dp + d];
    out[idx] = mean ? __fadd_rn(__fmul_rn(v, std_), mean[d]) : v;
  }
}

// ---------------------------------------------------------------------------------------------
// K1+K2: table[g][k] = |r_g|^2 + |c_k|^2 - 2 r_g.c_k with r_g = x[g/F] - xhat[g] (xhat may be null),
// then the T smallest per group, ascending  (approx_pairwise_distance utils.py:336-346 + topk/argmin,
// qinco_inference.py:171-173 and :241-245).
//
// One workgroup = TG groups x all K codewords.  Thread t owns codewords t, t+256, ...; the codebook is
// staged through LDS in [256 codewords][DC features] tiles (coalesced 128 B row segments from HBM/L2,
// padded rows -> conflict-free per-thread row reads), the TG residual rows sit in LDS and are read as
// wave-uniform broadcasts.  Selection: T rounds of wave64 shuffle arg-min per group.
// ---------------------------------------------------------------------------------------------
constexpr int DT_TG = 16;   // groups per workgroup
constexpr int DT_DC = 32;   // feature chunk staged per pass
constexpr int DT_CP = 36;   // padded LDS row (floats): 16 B aligned, breaks the 32-float stride

__global__ void __launch_bounds__(256)
dist_topk_kernel(const float* __restrict__ x, const float* __restrict__ xhat, int F,
                 const float* __restrict__ codebook, const float* __restrict__ cnorm, int K, int D,
                 long G, int T, int* __restrict__ ids_out /* (G,T) */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r_lds = lds;                          // [TG][D]
  float* c_lds = r_lds + DT_TG * D;            // [256][DT_CP]
  float* d_lds = c_lds + 256 * DT_CP;          // [TG][K]
  float* rn_lds = d_lds + DT_TG * K;           // [TG]

  const int tid = threadIdx.x;
  const long g0 = (long)blockIdx.x * DT_TG;

  // residual rows -> LDS
  for (int i = tid; i < DT_TG * D; i += 256) {
    int gl = i / D, d = i - gl * D;
    long g = g0 + gl;
    if (g >= G) g = G - 1;
    float v = x[(g / F) * D + d];
    if (xhat) v = __fsub_rn(v, xhat[g * D + d]);
    r_lds[i] = v;
  }
  __syncthreads();
  if (tid < DT_TG) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(r_lds[tid * D + d], r_lds[tid * D + d], s);
    rn_lds[tid] = s;
  }

  for (int kc = 0; kc < K; kc += 256) {
    float acc[DT_TG];
#pragma unroll
    for (int g = 0; g < DT_TG; ++g) acc[g] = 0.f;
    for (int dc = 0; dc < D; dc += DT_DC) {
      __syncthreads();
      // stage codebook[kc..kc+255][dc..dc+31]: thread -> (row = i/8, float4 column = i%8)
      for (int i = tid; i < 256 * (DT_DC / 4); i += 256) {
        int rrow = i >> 3, c4 = i & 7;
        int k = kc + rrow;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) v = *reinterpret_cast<const float4*>(codebook + (long)k * D + dc + c4 * 4);
        *reinterpret_cast<float4*>(c_lds + rrow * DT_CP + c4 * 4) = v;
      }
      __syncthreads();
#pragma unroll
      for (int d4 = 0; d4 < DT_DC; d4 += 4) {
        float4 cv = *reinterpret_cast<const float4*>(c_lds + tid * DT_CP + d4);
#pragma unroll
        for (int g = 0; g < DT_TG; ++g) {
          float4 rv = *reinterpret_cast<const float4*>(r_lds + g * D + dc + d4);
          acc[g] = fmaf(rv.x, cv.x, acc[g]);
          acc[g] = fmaf(rv.y, cv.y, acc[g]);
          acc[g] = fmaf(rv.z, cv.z, acc[g]);
          acc[g] = fmaf(rv.w, cv.w, acc[g]);
        }
      }
    }
    int k = kc + tid;
    if (k < K) {
      float cn = cnorm[k];
#pragma unroll
      for (int g = 0; g < DT_TG; ++g)
        d_lds[g * K + k] = __fsub_rn(__fadd_rn(rn_lds[g], cn), __fmul_rn(2.f, acc[g]));
    }
  }
  __syncthreads();

  // selection: wave w handles groups w, w+4, ...
  const int lane = tid & 63, wave = tid >> 6;
  for (int gl = wave; gl < DT_TG; gl += 4) {
    long g = g0 + gl;
    if (g >= G) break;
    float* dg = d_lds + gl * K;
    for (int t = 0; t < T; ++t) {
      float bv = __builtin_inff();
      int bi = 0x7fffffff;
      for (int k = lane; k < K; k += 64) {  // k ascends: strict < keeps the lowest index
        const float v = dg[k];
        const bool take = v < bv;
        bv = take ? v : bv;
        bi = take ? k : bi;
      }
      wave_argmin(bv, bi);
      if (bi == 0x7fffffff) bi = 0;           // all remaining +inf/NaN: degenerate, keep in range
      if (lane == 0) ids_out[g * T + t] = bi;
      if ((bi & 63) == lane) dg[bi] = __builtin_inff();   // owner lane retires the winner
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Step 0 / decode init: xhat[g] = codebook[ids[g]] ; optionally hist[g][0] = ids[g]
// (qinco_inference.py:246-249 and :70-72).
// ---------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ codebook, const int* __restrict__ ids,
                                   long G, int D, float* __restrict__ xhat, int* __restrict__ hist, int M) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = G * (D / 4);
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long g = idx / (D / 4);
    int c4 = (int)(idx - g * (D / 4));
    int id = ids[g];
    reinterpret_cast<float4*>(xhat)[idx] = reinterpret_cast<const float4*>(codebook + (long)id * D)[c4];
    if (hist && c4 == 0) hist[g * M] = id;
  }
}

// ---------------------------------------------------------------------------------------------
// K5 (selection part) + K6: per vector, the T smallest of its C = F*A candidate distances (ascending),
// then  parent = idx / A ; code = cand_ids[...] (or idx % A) ; re-thread the code history and gather the
// next xhat  (qinco_inference.py:200-222; base model qinco_base.py:346-372).
// One wave per vector; distances staged in LDS; T rounds of shuffle arg-min.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
beam_select_kernel(const float* __restrict__ dist, const float* __restrict__ cand,
                   const int* __restrict__ cand_ids, long N, int F, int A, int D, int T, int m, int M,
                   const int* __restrict__ hist_in /* (N,F,M) */, int* __restrict__ hist_out /* (N,T,M) */,
                   float* __restrict__ xhat_out /* (N,T,D) */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 4 + wave;
  if (n >= N) return;   // wave-uniform; only wave-level sync below
  const int C = F * A;
  // per wave: C distances, the T selected flat indices, the survivor list of wave_select_smallest
  const int per_wave = ((C + T + 3) & ~3) + 2 * SEL_SURV;
  float* dv = lds + (long)wave * per_wave;
  int* sel = reinterpret_cast<int*>(dv + C);
  unsigned long long* surv = reinterpret_cast<unsigned long long*>(dv + ((C + T + 3) & ~3));
  const float* dsrc = dist + n * C;
  for (int k = lane; k < C; k += 64) dv[k] = dsrc[k];
  __builtin_amdgcn_wave_barrier();
  wave_top_t(dv, C, T, surv, sel, lane);
  // parents, codes, history re-threading and the xhat gather of all T survivors, spread over the 64 lanes
  for (int e = lane; e < T * (m + 1); e += 64) {
    const int t = e / (m + 1), j = e - t * (m + 1);
    const int bi = sel[t];
    const int f = bi / A;
    int v;
    if (j < m) v = hist_in[(n * F + f) * M + j];
    else v = cand_ids ? cand_ids[n * C + bi] : (bi - f * A);
    hist_out[(n * T + t) * M + j] = v;
  }
  const int d4 = D / 4;
  for (int e = lane; e < T * d4; e += 64) {
    const int t = e / d4, j = e - t * d4;
    const float4* src = reinterpret_cast<const float4*>(cand + (n * C + sel[t]) * D);
    reinterpret_cast<float4*>(xhat_out + (n * T + t) * D)[j] = src[j];
  }
}

// codes_out[n][m] = hist[n][0][m]   (beam 0 is the best: topk output is ascending, :253)
__global__ void emit_codes_kernel(const int* __restrict__ hist, long N, int T, int M, void* __restrict__ out,
                                  int code_dtype) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = N * M;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long n = idx / M;
    int mm = (int)(idx - n * M);
    int v = hist[(n * T) * M + mm];
    if (code_dtype == 0) reinterpret_cast<long long*>(out)[idx] = v;
    else if (code_dtype == 1) reinterpret_cast<int*>(out)[idx] = v;
    else reinterpret_cast<unsigned char*>(out)[idx] = (unsigned char)v;
  }
}

// Decode input: codes (n, M) of any integer dtype -> (M, n) int32 with range check folded into a flag.
__global__ void import_codes_kernel(const void* __restrict__ codes, int code_dtype, long N, int M,
                                    const int* __restrict__ Kvals, int* __restrict__ out /* (M,N) */,
                                    int* __restrict__ err_flag) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = N * M;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long n = idx / M;
    int mm = (int)(idx - n * M);
    long long v;
    if (code_dtype == 0) v = reinterpret_cast<const long long*>(codes)[idx];
    else if (code_dtype == 1) v = reinterpret_cast<const int*>(codes)[idx];
    else v = reinterpret_cast<const unsigned char*>(codes)[idx];
    if (v < 0 || v >= Kvals[mm]) { atomicOr(err_flag, 1); v = 0; }
    out[(long)mm * N + n] = (int)v;
  }
}

}  // namespace qinco
