// Auxiliary gfx950 kernels of the QINCo encode/decode path: (de)normalisation, the residual->codebook
// L2 distance table fused with top-T selection, the per-vector beam top-B select/prune, gathers.
// All of these are <0.1 % of the FLOPs; they are HBM/LDS/latency work:
// coalesced row reads, LDS-staged codebook tiles, wave64 shuffle reductions for the selections.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qinco {

#define QINCO_DEV __device__ __forceinline__

// Lexicographic (value, index) minimum across the 64 lanes of a wave, result in every lane (ties -> lower index,
// which is what argmin returns on the reference CPU path; topk's order among exact ties is unspecified).
// All on the VALU: DPP lane permutes inside the 16-lane rows, gfx950's v_permlane16_swap / v_permlane32_swap across
// rows.  (__shfl_xor is ds_bpermute_b32: an LDS-crossbar round trip per level; a float (value, index) compare-select
// per level and a ballot / v_readlane fast path were both slower than the integer form below: SGPR round trips.)
// Branch-free, SGPR-free: reduce the order-preserving integer image of the value with v_min_u32 (fused with the DPP
// permute: one instruction per level), then reduce the index among the lanes that hold the minimum the same way.
// Exactly the lexicographic (value, index) minimum (-0.0 is folded into +0.0 first, as a float compare treats it).
// Needs all 64 lanes active.
template <int CTRL>
QINCO_DEV unsigned dpp_u(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
QINCO_DEV unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
QINCO_DEV unsigned wave_umin(unsigned k) {
  k = umin(k, dpp_u<0xB1>(k));
  k = umin(k, dpp_u<0x4E>(k));
  k = umin(k, dpp_u<0x141>(k));
  k = umin(k, dpp_u<0x140>(k));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(k, k, false, false);
    k = umin((unsigned)r[0], (unsigned)r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(k, k, false, false);
    k = umin((unsigned)r[0], (unsigned)r[1]);
  }
  return k;
}
QINCO_DEV unsigned ordered_bits(float d) {
  const unsigned u = __builtin_bit_cast(unsigned, d + 0.f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
QINCO_DEV float from_ordered_bits(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, u);
}
template <int N>
QINCO_DEV void wave_argmin_u(float (&v)[N], int (&i)[N]) {
  unsigned key[N], m[N], c[N];
#pragma unroll
  for (int u = 0; u < N; ++u) key[u] = ordered_bits(v[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) m[u] = wave_umin(key[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) c[u] = key[u] == m[u] ? (unsigned)i[u] : 0x7fffffffu;
#pragma unroll
  for (int u = 0; u < N; ++u) c[u] = wave_umin(c[u]);
#pragma unroll
  for (int u = 0; u < N; ++u) {
    v[u] = from_ordered_bits(m[u]);
    i[u] = (int)c[u];
  }
}

QINCO_DEV void wave_argmin(float& v, int& i) {
  float a[1] = {v};
  int b[1] = {i};
  wave_argmin_u<1>(a, b);
  v = a[0];
  i = b[0];
}

// ---------------------------------------------------------------------------------------------
// K8: x_n = (x - mean) / std   (qinco_inference.py:277).  x is fp32 or uint8 rows with a byte stride
// (bvecs rows are d+4 bytes apart: search_tasks.py:109-110 converts the raw slice with .to(float32)).
// ---------------------------------------------------------------------------------------------
__global__ void normalize_kernel(const void* __restrict__ x, int x_dtype, long row_stride_bytes,
                                 const float* __restrict__ mean, float std_, float* __restrict__ out,
                                 long n, int D) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = n * D;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx / D;
    int d = (int)(idx - r * D);
    const char* rowp = reinterpret_cast<const char*>(x) + r * row_stride_bytes;
    float v = x_dtype == 0 ? reinterpret_cast<const float*>(rowp)[d]
                           : (float)reinterpret_cast<const unsigned char*>(rowp)[d];
    out[idx] = mean ? __fdiv_rn(__fsub_rn(v, mean[d]), std_) : v;   // mean == nullptr: x is already normalised
  }
}

// x = xhat * std + mean  (qinco_inference.py:281): two roundings (no fma contraction), like ATen mul + add.
__global__ void denormalize_kernel(const float* __restrict__ xhat, const float* __restrict__ mean,
                                   float std_, float* __restrict__ out, long n, int D) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = n * D;
  for (; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int d = (int)(idx % D);
    out[idx] = mean ? __fadd_rn(__fmul_rn(xhat[idx], std_), mean[d]) : xhat[idx];
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
  float* dv = lds + (long)wave * C;
  const float* dsrc = dist + n * C;
  for (int k = lane; k < C; k += 64) dv[k] = dsrc[k];
  __builtin_amdgcn_wave_barrier();
  for (int t = 0; t < T; ++t) {
    float bv = __builtin_inff();
    int bi = 0x7fffffff;
    for (int k = lane; k < C; k += 64) {  // k ascends: strict < keeps the lowest index
      const float v = dv[k];
      const bool take = v < bv;
      bv = take ? v : bv;
      bi = take ? k : bi;
    }
    wave_argmin(bv, bi);
    if (bi == 0x7fffffff) bi = 0;
    if ((bi & 63) == lane) dv[bi] = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    const int f = bi / A;
    const long rowi = n * C + bi;
    const int code = cand_ids ? cand_ids[rowi] : (bi - f * A);
    int* ho = hist_out + (n * T + t) * M;
    const int* hi = hist_in + (n * F + f) * M;
    for (int j = lane; j < m; j += 64) ho[j] = hi[j];
    if (lane == 0) ho[m] = code;
    const float4* src = reinterpret_cast<const float4*>(cand + rowi * D);
    float4* dst = reinterpret_cast<float4*>(xhat_out + (n * T + t) * D);
    for (int j = lane; j < D / 4; j += 64) dst[j] = src[j];
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
