// Look-up decoders that consume the codes after the hot path:
//   reconstruct_from_fixed_codebooks  (reference qinco/search/search_utils.py:105-115)   xhat = sum_m cb[m][codes[:, m]]
//   PairwiseDecoderIVF.forward        (reference qinco/search/pairwise_decoder.py:88-93, map_codes :126-130)
//                                     xhat = sum_j cb[j][ codes[a_j] * K_base + codes[b_j] ]
// One kernel: out[n] = sum_j table_j[ codes[n][a_j] * mul + (b_j >= 0 ? codes[n][b_j] : 0) ], accumulated in j order in
// fp32 exactly like the reference's `xhat = cb[0][c0]; xhat += cb[j][c_j]` loop (bit-reproducible).
// HBM-bound gather-add: per vector  Mc code bytes + J x 4D gathered table bytes in, 4D out.  A half-wave owns one
// vector (float4 per lane for D = 128), all J row reads are issued before the adds so they overlap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qinco {

constexpr int kLutMaxJ = 64;

struct LutArgs {
  const float* tables;   // (J, Kt, D)
  int J, D;
  long Kt;
  int a[kLutMaxJ], b[kLutMaxJ];
  long mul;
  const void* codes;     // (n, Mc)
  int code_dtype, Mc;
  long n;
  float* out;            // (n, D)
  int* err_flag;
};

__global__ void __launch_bounds__(256) lut_decode_kernel(LutArgs a) {
  const int D4 = a.D / 4;
  const long total = a.n * D4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long n = idx / D4;
    const int c4 = (int)(idx - n * D4);
    auto code = [&](int col) -> long {
      const long o = n * a.Mc + col;
      if (a.code_dtype == 0) return (long)reinterpret_cast<const long long*>(a.codes)[o];
      if (a.code_dtype == 1) return (long)reinterpret_cast<const int*>(a.codes)[o];
      return (long)reinterpret_cast<const unsigned char*>(a.codes)[o];
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < a.J; j0 += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u;
        if (j < a.J) {
          long k = code(a.a[j]) * a.mul + (a.b[j] >= 0 ? code(a.b[j]) : 0);
          if (k < 0 || k >= a.Kt) { atomicOr(a.err_flag, 1); k = 0; }
          v[u] = reinterpret_cast<const float4*>(a.tables + ((long)j * a.Kt + k) * a.D)[c4];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u;
        if (j < a.J) {
          if (j == 0) acc = v[u];
          else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
      }
    }
    reinterpret_cast<float4*>(a.out)[idx] = acc;
  }
}

}  // namespace qinco
