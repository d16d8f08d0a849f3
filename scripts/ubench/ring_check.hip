// Ring checker: the workgroup-shared LDS-DMA weight ring of mlp16_kernel / mlp_kernel (SHR), with the arithmetic replaced by a
// CHECK of every fragment the ring delivers.  Fragment f of the stream holds, in lane l, {f*64+l, ~(f*64+l), f, l}; a wave that
// receives anything else records (section, fragment, what it got).  This separates "the ring delivered wrong bytes" from
// everything else the fused kernel does, under 1 / 2 / 3 workgroups per CU (dynamic-LDS padding), with or without MFMA work
// between fragments, with or without the distance epilogue's loads / shuffles / stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 ring_check.hip -o ring_check && ./ring_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int P = 48;

#define INL __device__ __forceinline__
#define LAMBDA __attribute__((always_inline))
template <class F, int... Is>
INL void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f.template operator()<Is>(), ...); }
template <int N, class F>
INL void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct Args {
  const u32x4* stream;   // nsec * P fragments (+ P of padding)
  int nsec;
  unsigned* rec;         // per wave: [n_bad, first_sec, first_T, got.x, got.y, got.z, got.w, smid]
  const float* x;        // epilogue: rows to load
  float* out;            // epilogue: stores
  int epilogue;
  int mfma;              // MFMAs per fragment (0 or 4)
  const unsigned* probe; // PROBE: cold rows loaded into VGPRs by ordinary loads while LDS-DMAs are in flight (value = its own index)
  unsigned* probe_bad;   // PROBE: [count of wrong probe values]
  f32x4* accout;         // per lane: the MFMA accumulator at the end (every wave computes the same chain -> all waves must agree)
};

// LIVE = fragments of each 48-fragment section that are consumed (the rest are "padding": their barriers / DMAs happen, the
// LDS reads are dead code, exactly like skip_pad in the real kernels)
// PROBE: an ordinary global load is issued BEFORE the prologue DMAs (like cand_ids[row] in the real kernels) and one per section in
// the middle of the stream (like the codeword / xhat block prefetches); hipcc waits for them with a COUNTED vmcnt that assumes
// LDS-DMA loads and VGPR loads complete in issue order.
template <int LIVE, int CONSERVATIVE, int PROBE = 0>
__global__ void __launch_bounds__(256, 1) ring_check(Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const u32x4* wp = a.stream + lane;
  u32x4 ring[3];
  __shared__ u32x4 lds_ring[P * 64];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  u32x4* myring = lds_ring;
  const int wofs = wave_u * 64;
  auto dma = [&]<int T>() LAMBDA {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + wofs + T * 64),
                                     (__attribute__((address_space(3))) void*)(myring + wofs + (T % P) * 64), 16, 0, 0);
  };
  auto wait_vm = [&]<int N>() LAMBDA {
    asm volatile("" ::: "memory");
    if constexpr (CONSERVATIVE) __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
  };
  const long gw = (long)blockIdx.x * 4 + wave;
  unsigned probe0 = 0, pbad = 0;
  if constexpr (PROBE) probe0 = a.probe[gw * 64 * 64 + lane * 64];   // one cold 256-byte-strided line per lane
  static_for<P / 4 - 1>([&]<int i>() LAMBDA { dma.template operator()<4 * i>(); });
  wait_vm.template operator()<P / 4 - 2>();
  __builtin_amdgcn_s_barrier();
  ring[0] = myring[lane];
  ring[1] = myring[64 + lane];
  if constexpr (PROBE) pbad += probe0 != (unsigned)(gw * 64 * 64 + lane * 64);
  auto take = [&]<int T>() LAMBDA -> u32x4 {
    if constexpr ((T & 3) == 0) {
      wait_vm.template operator()<P / 4 - 3>();
      __builtin_amdgcn_s_barrier();
      dma.template operator()<T + P - 4>();
    }
    ring[(T + 2) % 3] = myring[((T + 2) % P) * 64 + lane];
    return ring[T % 3];
  };
  unsigned nbad = 0, fsec = 0, fT = 0;
  u32x4 fgot = {0, 0, 0, 0};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float bval = 1.0f + lane * 1e-3f;
#pragma unroll 1
  for (int sec = 0; sec < a.nsec; ++sec) {
    unsigned probe1 = 0;
    static_for<P>([&]<int T>() LAMBDA {
      if constexpr (PROBE && T == 2) probe1 = a.probe[gw * 64 * 64 + lane * 64 + 1 + sec];   // issued mid-stream ...
      if constexpr (PROBE && T == 7) pbad += probe1 != (unsigned)(gw * 64 * 64 + lane * 64 + 1 + sec);   // ... used 5 fragments later
      const u32x4 w = take.template operator()<T>();
      if constexpr (T < LIVE) {
        const unsigned f = (unsigned)(sec * P + T);
        const unsigned e0 = f * 64u + (unsigned)lane;
        const bool bad = (w[0] != e0) | (w[1] != ~e0) | (w[2] != f) | (w[3] != (unsigned)lane);
        if (bad) {
          if (nbad == 0) {
            fsec = sec;
            fT = T;
            fgot = w;
          }
          nbad++;
        }
        if (a.mfma)
          static_for<4>([&]<int e>() LAMBDA {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)(w[e] & 0xffffu) * 0.001f + 1.0f, bval, acc, 0, 0, 0);
          });
      }
    });
    wp += P * 64;
  }
  if (a.epilogue) {   // the distance epilogue of mlp16_kernel: row loads, cross-lane sums, a row store and a 4-byte store
    const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const f32x4 xb = *reinterpret_cast<const f32x4*>(a.x + row * 32 + (lane >> 4) * 4);
    float s = acc[0] * xb[0] + acc[1] * xb[1] + acc[2] * xb[2] + acc[3] * xb[3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    *reinterpret_cast<f32x4*>(a.out + row * 32 + (lane >> 4) * 4) = acc + xb;
    if ((lane >> 4) == 0) a.out[(long)gridDim.x * 64 * 32 + row] = s;
  } else if (a.mfma && acc[0] == 123.456f) {
    a.out[0] = acc[1];
  }
  if (a.accout) a.accout[((long)blockIdx.x * 4 + wave) * 64 + lane] = acc;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (PROBE) {
    if (pbad) atomicAdd(a.probe_bad, pbad);
  }
  // one record per lane that saw something wrong (lane 0 otherwise)
  unsigned smid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(smid));
  const unsigned long long anybad = __builtin_amdgcn_ballot_w64(nbad != 0);
  const int rep = anybad ? __builtin_ctzll(anybad) : 0;
  if (lane == rep) {
    unsigned* r = a.rec + ((long)blockIdx.x * 4 + wave) * 8;
    r[0] = (unsigned)__builtin_popcountll(anybad) | (nbad << 8);
    r[1] = fsec;
    r[2] = fT | ((unsigned)lane << 16);
    r[3] = fgot[0];
    r[4] = fgot[1];
    r[5] = fgot[2];
    r[6] = fgot[3];
    r[7] = smid;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LIVE, int CONS, int PROBE = 0>
static void run(const char* name, Args a, int grid, int pad_kib, unsigned* d_rec, int reps) {
  const size_t nw = (size_t)grid * 4;
  std::vector<unsigned> rec(nw * 8);
  for (int r = 0; r < reps; ++r) {
    CK(hipMemset(d_rec, 0, nw * 8 * sizeof(unsigned)));
    if (PROBE) CK(hipMemset(a.probe_bad, 0, 4));
    hipLaunchKernelGGL((ring_check<LIVE, CONS, PROBE>), dim3(grid), dim3(256), (size_t)pad_kib * 1024, 0, a);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(rec.data(), d_rec, nw * 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    long bad_waves = 0, bad_frags = 0;
    int shown = 0;
    for (size_t w = 0; w < nw; ++w) {
      const unsigned* q = &rec[w * 8];
      if (q[0] == 0) continue;
      bad_waves++;
      bad_frags += q[0] >> 8;
      if (shown < 6 && r == 0) {
        const unsigned expf = q[1] * P + (q[2] & 0xffff);
        printf("    wg %zu wave %zu: %u lanes bad, lane %u saw %u bad fragments; first at sec %u T %u (fragment %u, slot %u): got "
               "{%u, ~%u, frag %u, lane %u}  -> that is fragment %u (delta %d) lane %u\n",
               w / 4, w % 4, q[0] & 0xff, q[2] >> 16, q[0] >> 8, q[1], q[2] & 0xffff, expf, (q[2] & 0xffff) % P, q[3], ~q[4], q[5], q[6],
               q[5], (int)q[5] - (int)expf, q[6]);
        shown++;
      }
    }
    unsigned pb = 0;
    if (PROBE) CK(hipMemcpy(&pb, a.probe_bad, 4, hipMemcpyDeviceToHost));
    long acc_bad = -1;
    if (a.accout && a.mfma) {   // the accumulators of all waves must be bit-equal (same fragments, same B operand per lane)
      std::vector<float> ha(nw * 64 * 4);
      CK(hipMemcpy(ha.data(), a.accout, ha.size() * 4, hipMemcpyDeviceToHost));
      acc_bad = 0;
      for (size_t w = 1; w < nw; ++w)
        if (memcmp(&ha[w * 256], &ha[0], 1024) != 0) acc_bad++;
    }
    printf("%-44s grid %5d pad %2d KiB live %2d mfma %d epi %d  rep %d: %ld bad waves of %zu, %ld bad fragments, %u bad probe values, %ld waves with a different MFMA result\n", name,
           grid, pad_kib, LIVE, a.mfma, a.epilogue, r, bad_waves, nw, bad_frags, pb, acc_bad);
  }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 3000;
  const int nsec = argc > 2 ? atoi(argv[2]) : 12;
  const size_t nf = (size_t)(nsec + 1) * P + P;
  std::vector<unsigned> h(nf * 64 * 4);
  for (size_t f = 0; f < nf; ++f)
    for (unsigned l = 0; l < 64; ++l) {
      unsigned* q = &h[(f * 64 + l) * 4];
      q[0] = (unsigned)(f * 64 + l);
      q[1] = ~q[0];
      q[2] = (unsigned)f;
      q[3] = l;
    }
  unsigned *d_stream, *d_rec;
  float *d_x, *d_out;
  CK(hipMalloc(&d_stream, h.size() * 4));
  CK(hipMemcpy(d_stream, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_rec, (size_t)grid * 4 * 8 * 4));
  CK(hipMalloc(&d_x, (size_t)grid * 64 * 32 * 4));
  CK(hipMemset(d_x, 0, (size_t)grid * 64 * 32 * 4));
  CK(hipMalloc(&d_out, (size_t)grid * 64 * 33 * 4));
  f32x4* d_acc;
  CK(hipMalloc(&d_acc, (size_t)grid * 4 * 64 * 16));
  Args a{};
  a.stream = reinterpret_cast<const u32x4*>(d_stream);
  a.nsec = nsec;
  a.rec = d_rec;
  a.x = d_x;
  a.out = d_out;
  a.accout = d_acc;
  {   // probe rows: value = own index; 64 * 64 words per wave (16 KiB), so every load is a cold line
    const size_t np = (size_t)grid * 4 * 64 * 64;
    std::vector<unsigned> hp(np);
    for (size_t i = 0; i < np; ++i) hp[i] = (unsigned)i;
    unsigned *d_probe, *d_pb;
    CK(hipMalloc(&d_probe, np * 4));
    CK(hipMemcpy(d_probe, hp.data(), np * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pb, 4));
    a.probe = d_probe;
    a.probe_bad = d_pb;
  }
  for (int pad : {36, 24, 0}) {
    a.mfma = 4;
    a.epilogue = 1;
    run<48, 0, 1>("PROBE: VGPR loads among the DMAs, all live", a, grid, pad, d_rec, 3);
    run<8, 0, 1>("PROBE: VGPR loads among the DMAs, 8 live", a, grid, pad, d_rec, 3);
    a.mfma = 0;
    run<8, 0, 1>("PROBE: VGPR loads among the DMAs, 8 live", a, grid, pad, d_rec, 3);
    for (int mfma : {4, 0})
      for (int epi : {1, 0}) {
        a.mfma = mfma;
        a.epilogue = epi;
        run<48, 0>("all fragments live", a, grid, pad, d_rec, 2);
        run<8, 0>("8 live of 48 (padding-dominated)", a, grid, pad, d_rec, 2);
      }
    a.mfma = 4;
    a.epilogue = 1;
    run<8, 1>("8 live, every wait = s_waitcnt 0", a, grid, pad, d_rec, 2);
    run<48, 1>("all live, every wait = s_waitcnt 0", a, grid, pad, d_rec, 2);
  }
  return 0;
}
