// Micro-benchmark of the fused-MLP inner loop: cycles per "fragment" (4 dependent fp32 MFMAs = 256 cycles ideal)
// when the ring traffic of the real kernel is added piece by piece.  One wave per SIMD (256 blocks x 256 threads).
//   MODE 0: MFMAs only                         MODE 1: + ds_read_b128 per fragment
//   MODE 2: + LDS-DMA (global_load_lds) + counted vmcnt     MODE 3: 1 + 2 (the production loop), compiler placement
//   MODE 4: 3 with the pieces pinned between the MFMAs      MODE 5: plain global_load_dwordx4 register ring (depth 8)
//   MODE 6: 3 but alternating two accumulators per fragment MODE 7: 6 + pinned pieces between independent MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int P = 36;

__device__ __forceinline__ void wait_vm(int) {}
template <int N> __device__ __forceinline__ void waitvm() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ w, float* out, long long* cyc, int iters) {
  __shared__ f32x4 lds[4 * P * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4* myring = lds + wave * P * 64;
  const f32x4* wp = w + lane;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  f32x4 r[8];
  for (int i = 0; i < 8; ++i) r[i] = wp[i * 64];
  float b = 1.0f + lane * 1e-4f;
  auto dma = [&](int slot, const f32x4* src) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(myring + slot * 64), 16, 0, 0);
  };
  auto dma_sh = [&](int slot, const f32x4* src) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + slot * 64), 16, 0, 0);
  };
  if (MODE == 15) {
    for (int i = 0; i < P - 8; ++i) if ((i & 3) == wave) dma_sh(i, wp + i * 64);
  } else if (MODE == 13 || MODE == 14 || MODE == 16 || MODE == 17) {
    for (int i = 0; i < P - 4; ++i) if ((i & 3) == wave) dma_sh(i, wp + i * 64);
  } else if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7 || MODE >= 8)
    for (int i = 0; i < P - 1; ++i) dma(i, wp + i * 64);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < P; ++t) {   // one trip = P fragments
      f32x4 x = r[t & 7];
      if constexpr (MODE == 0) {
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
      } else if constexpr (MODE == 1) {
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        r[(t + 2) & 7] = myring[((t + 2) % P) * 64 + lane];
      } else if constexpr (MODE == 2) {
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        waitvm<P - 4>();
        dma((t + P - 1) % P, wp + (t + P - 1) * 64);
      } else if constexpr (MODE == 3 || MODE == 6) {
        f32x16& a = (MODE == 6 && (t & 1)) ? acc1 : acc0;
        waitvm<P - 4>();
        r[(t + 2) & 7] = myring[((t + 2) % P) * 64 + lane];
        dma((t + P - 1) % P, wp + (t + P - 1) * 64);
        a = MF(x[0], b, a); a = MF(x[1], b, a); a = MF(x[2], b, a); a = MF(x[3], b, a);
      } else if constexpr (MODE == 4 || MODE == 7) {
        f32x16& a = (MODE == 7 && (t & 1)) ? acc1 : acc0;
        f32x16& o = (MODE == 7 && (t & 1)) ? acc0 : acc1;   // the other chain (MODE 7 only touches it via ordering)
        (void)o;
        a = MF(x[0], b, a); SB;
        waitvm<P - 4>(); r[(t + 2) & 7] = myring[((t + 2) % P) * 64 + lane]; SB;
        a = MF(x[1], b, a); SB;
        dma((t + P - 1) % P, wp + (t + P - 1) * 64); SB;
        a = MF(x[2], b, a); a = MF(x[3], b, a);
      } else if constexpr (MODE == 8) {           // DMA issue only, never waited for inside the loop
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        dma((t + P - 1) % P, wp + (t + P - 1) * 64);
      } else if constexpr (MODE == 9) {           // DMA + wait, but always the same 36 KiB of source (cache hits)
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        waitvm<P - 4>();
        dma((t + P - 1) % P, w + lane + t * 64);
      } else if constexpr (MODE == 10) {          // DMA every other fragment (half the bytes)
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        if (t & 1) { waitvm<P / 2 - 4>(); dma((t + P - 1) % P, wp + (t + P - 1) * 64); }
      } else if constexpr (MODE == 11) {          // DMA alone in the shadow of MFMA 1 (address prepared earlier)
        const f32x4* src = wp + (t + P - 1) * 64;
        asm volatile("" : "+v"(src));
        acc0 = MF(x[0], b, acc0); SB;
        dma((t + P - 1) % P, src); SB;
        acc0 = MF(x[1], b, acc0); SB;
        waitvm<P - 3>(); SB;
        acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
      } else if constexpr (MODE == 12) {          // 8 B/lane DMA pieces: two global_load_lds_dwordx2... not available: use 2x dword
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        waitvm<P - 4>();
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const float*)(w) + lane + t * 64),
                                         (__attribute__((address_space(3))) void*)(myring + ((t + P - 1) % P) * 64), 4, 0, 0);
      } else if constexpr (MODE == 13 || MODE == 14) {
        // ONE ring shared by the 4 waves of the workgroup: wave w DMAs the fragments t = w (mod 4), everybody reads all
        // of them; a raw s_barrier every 4 fragments (after the issuer's counted vmcnt) publishes the landed group.
        f32x4* shring = lds;   // P fragments shared (P*1 KiB)
        if ((t & 3) == 0) {
          waitvm<(P / 4) - 2>();               // my DMA belonging to the NEXT group of 4 has landed
          __builtin_amdgcn_s_barrier();        // ... and so have the other three waves'
        }
        r[(t + 2) & 7] = shring[((t + 2) % P) * 64 + lane];
        if ((t & 3) == wave) dma_sh((t + P - 4) % P, wp + (t + P - 4) * 64);
        if constexpr (MODE == 14) {
          f32x16& a = (t & 1) ? acc1 : acc0;
          a = MF(x[0], b, a); a = MF(x[1], b, a); a = MF(x[2], b, a); a = MF(x[3], b, a);
        } else {
          acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        }
      } else if constexpr (MODE == 15) {
        // shared ring, barrier every 8 fragments, each wave DMAs 2 of the 8
        f32x4* shring = lds;
        if ((t & 7) == 0) {
          waitvm<(P / 4) - 4>();
          __builtin_amdgcn_s_barrier();
          dma_sh((t + P - 8) % P + wave, wp + (t + P - 8 + wave) * 64);
          dma_sh((t + P - 4) % P + wave, wp + (t + P - 4 + wave) * 64);
        }
        r[(t + 2) & 7] = shring[((t + 2) % P) * 64 + lane];
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
      } else if constexpr (MODE == 16) {
        // shared ring as MODE 13 but branch-free issue (wave offset folded into the pointers), like the real kernel
        f32x4* shring = lds;
        if ((t & 3) == 0) {
          waitvm<(P / 4) - 3>();
          __builtin_amdgcn_s_barrier();
          dma_sh((t + P - 4) % P + wave, wp + (t + P - 4 + wave) * 64);
        }
        r[(t + 2) & 7] = shring[((t + 2) % P) * 64 + lane];
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
      } else if constexpr (MODE == 17) {
        // MODE 16 with a 2-deep register ring read ONE fragment ahead
        f32x4* shring = lds;
        if ((t & 3) == 0) {
          waitvm<(P / 4) - 3>();
          __builtin_amdgcn_s_barrier();
          dma_sh((t + P - 4) % P + wave, wp + (t + P - 4 + wave) * 64);
        }
        r[(t + 1) & 7] = shring[((t + 1) % P) * 64 + lane];
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
      } else if constexpr (MODE == 5) {
        acc0 = MF(x[0], b, acc0); acc0 = MF(x[1], b, acc0); acc0 = MF(x[2], b, acc0); acc0 = MF(x[3], b, acc0);
        r[t & 7] = wp[(t + 8) * 64];
      }
    }
    wp += P * 64;
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = acc0[0] + acc0[9] + acc1[3];
  for (int i = 0; i < 8; ++i) s += r[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const f32x4* w, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
  hipDeviceSynchronize();
  long long c[256]; hipMemcpy(c, cyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
  printf("MODE %d: %.1f cycles per fragment (ideal 256)\n", MODE, avg / ((double)iters * P));
  hipFree(out); hipFree(cyc);
}
int main() {
  const int iters = 544;   // 544 x 36 = 19584 fragments = one C2 step's stream (20 MB)
  f32x4* w; hipMalloc(&w, (size_t)(iters + 2) * P * 1024); hipMemset(w, 0, (size_t)(iters + 2) * P * 1024);
  run<0>(w, iters); run<1>(w, iters); run<2>(w, iters); run<3>(w, iters); run<4>(w, iters); run<5>(w, iters);
  run<6>(w, iters); run<7>(w, iters); run<8>(w, iters); run<9>(w, iters); run<10>(w, iters); run<11>(w, iters); run<12>(w, iters); run<13>(w, iters); run<14>(w, iters); run<15>(w, iters); run<16>(w, iters); run<17>(w, iters);
  return 0;
}
