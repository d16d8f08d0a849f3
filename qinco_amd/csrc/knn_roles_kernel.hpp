// knn_table_roles_kernel<D> -- the filtered table of the small-db search (knn_kernel.hpp, form (2)) with the wave's two jobs
// given to two waves (round 6).  Same pairs, same instruction sequence per pair, same keys: bit-identical candidate sets.
//
// Why: knn_table_kernel<D, true> runs two waves per SIMD that BOTH compute and filter.  Two waves that each always have a
// dependent f32 MFMA ready deliver 0.80-0.82 of the matrix pipe between them (scripts/ubench/ring_hot.hip; one wave alone 0.91
// with its ring of loads), and a lone wave that also filters needs ~430 cycles per 256-cycle step (its LDS reads, the ballot's
// scalar branch and the list writes sit between its own dependent MFMAs, in order).  MI355X_MICROARCH.md ("Wave scheduling"): the
// matrix pipe and the VALU are separate -- an MFMA-only wave and a VALU-only wave on one SIMD run concurrently, both near their
// own maximum.  So, per workgroup of eight waves on a CU of its own:
//   * four MFMA waves, ONE PER SIMD: stream = the register ring's global loads, the chain of 4 NF MFMAs per block of 32 queries,
//     four `ds_write_b128` of the finished 32 x 32 tile and one raw `s_barrier` per block.  The tile of block qb is written
//     behind the first MFMAs of block qb + 1 (second accumulator) and the barrier comes a few MFMAs later, when the writes have
//     landed: the pipe never waits for either.  Raised priority (`s_setprio`): its few non-MFMA instructions win arbitration.
//   * four VALU waves: wait at the barrier, read "their" MFMA wave's tile (four `ds_read_b128`, the MFMA layout as is), run the 16
//     (query, 32 rows) slices of knn_table_kernel's filter -- threshold compare, ballot, survivors to the wave's LDS list by
//     prefix count -- and flush the list to the per-query candidate lists in HBM (one returning atomic per entry) when it could
//     overflow and at the end.  Tiles are double-buffered: the barrier a filter wave passes for tile qb + 1 is the one the MFMA
//     wave must pass before it overwrites tile qb's buffer with tile qb + 2.
// Which waves are which: the dispatcher does NOT always put two of a 512-thread workgroup's waves on every SIMD (ring_hot.hip saw
// three on one), so the roles are taken at run time: every wave reads its SIMD from HW_ID, the first wave to claim a SIMD is
// that SIMD's MFMA wave, the others filter.  If the eight waves do not cover all four SIMDs the roles fall back to the wave
// index (correct, just not one MFMA wave per SIMD); the kernel counts both cases (qinco_knn_roles_stats).
// LDS: thresholds 33 KiB + tiles 32 KiB + lists 24 KiB = 89 KiB -> one workgroup per CU by construction.
#pragma once
#include "knn_kernel.hpp"

namespace qinco {

#ifdef QINCO_KNN_TIMELINE
// experiment builds only (scripts/ubench/knn_roles_tl.hip): per wave of the first workgroups {role, SIMD, cycles alive, cycles
// inside lds_barrier, cycles in the filter's flushes}
__device__ long long* g_knn_tl = nullptr;
__device__ int g_knn_prio = 0;   // experiment: 0 = MFMA waves at priority 3, 1 = nobody, 2 = filter waves at priority 3, 3 = filter 1
#define KNN_TL(x) x
#else
#define KNN_TL(x)
#endif

template <int D>
__global__ void __launch_bounds__(512)
knn_table_roles_kernel(const f32x4* __restrict__ qstream, const float* __restrict__ qnorm, int nqblocks,
                       const float* __restrict__ db, long N, KnnFilt f, int* __restrict__ roles_stat) {
  constexpr int NDB = D / 32;
  constexpr int NF = NDB * 4;   // fragments (of 4 MFMAs) per block of 32 queries
  static_assert(NDB >= 1 && NDB <= 4, "the two-role form keeps the whole block in a ring of NF <= 16 fragments");
  // where the tile write and the barrier sit inside the NEXT block's chain: behind fragment H, behind fragment H2
  constexpr int H = NF >= 16 ? 2 : 1;
  constexpr int H2 = NF >= 16 ? 5 : (NF >= 8 ? 3 : 2);
  static_assert(H < H2 && H2 <= NF, "pipeline points");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  __shared__ uint2 s_qt[kKnnMaxChunk + 32];   // per query row of the chunk: (|q|^2, threshold as a float: knn_tau_float) bits
  __shared__ f32x4 s_tile[4][2][4][64];       // [pair][buffer][quarter of the accumulator][lane]
  __shared__ unsigned long long s_key[4][kKnnWaveList];
  __shared__ unsigned s_row[4][kKnnWaveList];
  __shared__ float s_xn[4][32];
  __shared__ int s_claim[4];
  __shared__ int s_nfilt;
  for (int i = threadIdx.x; i < nqblocks * 32; i += 512)
    s_qt[i] = make_uint2(__builtin_bit_cast(unsigned, qnorm[i]), __builtin_bit_cast(unsigned, knn_tau_float(i < f.nq_valid ? f.tau[i] : 0u)));
  if (threadIdx.x < 4) s_claim[threadIdx.x] = 0;
  if (threadIdx.x == 4) s_nfilt = 0;
  __syncthreads();
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));   // SIMD in bits [5:4]
  const int simd = (int)((hw >> 4) & 3);
  int order = 0;
  if (lane == 0) order = atomicAdd(&s_claim[simd], 1);
  order = __builtin_amdgcn_readfirstlane(order);
  __syncthreads();
  const bool spread = s_claim[0] > 0 && s_claim[1] > 0 && s_claim[2] > 0 && s_claim[3] > 0;
  bool mfma_role;
  int pair;   // MFMA wave `pair` and filter wave `pair` work on database rows (blockIdx.x * 4 + pair) * 32 ...
  if (spread) {
    mfma_role = order == 0;
    if (mfma_role) {
      pair = simd;
    } else {
      int fi = 0;
      if (lane == 0) fi = atomicAdd(&s_nfilt, 1);
      pair = __builtin_amdgcn_readfirstlane(fi);
    }
  } else {
    mfma_role = wave < 4;
    pair = wave & 3;
  }
  mfma_role = __builtin_amdgcn_readfirstlane((int)mfma_role) != 0;
  pair = __builtin_amdgcn_readfirstlane(pair);
  if (roles_stat && threadIdx.x == 0) atomicAdd(roles_stat + (spread ? 0 : 1), 1);

  const long n0 = ((long)blockIdx.x * 4 + pair) * 32;
  long row = n0 + j;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  // (raw barrier: __syncthreads would also drain vmcnt, i.e. the MFMA wave's ring of loads)
  KNN_TL(long long tl_bar = 0; long long tl_flush = 0; const long long tl_t0 = __builtin_readcyclecounter();)
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    KNN_TL(const long long b0 = __builtin_readcyclecounter();)
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    KNN_TL(tl_bar += __builtin_readcyclecounter() - b0;)
    asm volatile("" ::: "memory");
  };

  if (mfma_role) {
#ifdef QINCO_KNN_TIMELINE
    if (g_knn_prio == 0) __builtin_amdgcn_s_setprio(3);
#else
    __builtin_amdgcn_s_setprio(3);
#endif
    const float* xp = db + row * D + half * 4;
    f32x16 xt[NDB];
    float xn = 0.f;
#pragma unroll
    for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 t = *reinterpret_cast<const f32x4*>(xp + ib * 32 + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xt[ib][4 * q + e] = t[e];
          xn = fmaf(t[e], t[e], xn);   // (the order of knn_table_kernel: same |x|^2 bits)
        }
      }
    }
    xn += __shfl_xor(xn, 32);
    if (half == 0) s_xn[pair][j] = xn;
    const f32x4* wp = qstream + lane;
    f32x4 ring[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) ring[i] = wp[i * 64];
    lds_barrier();   // #0: |x|^2 published
    auto chain = [&]<int I0, int I1>(f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        const f32x4 w = ring[i];
        ring[i] = wp[(i + NF) * 64];
        // fences: see knn_table_kernel (hipcc sinks the ring loads to their use, or hoists the MFMAs above them)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xt[i >> 2][4 * (i & 3) + e], acc, 0, 0, 0);
      }
    };
    auto write_tile = [&](const f32x16& t, const int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
        s_tile[pair][buf][q][lane] = v;
      }
    };
    auto zero = [&](f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    };
    // block qb's chain with block qb - 1's tile leaving under it
    auto step = [&](const int qb, f32x16& acc, const f32x16& prev) __attribute__((always_inline)) {
      zero(acc);
      chain.template operator()<0, H>(acc);
      write_tile(prev, (qb - 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      chain.template operator()<H, H2>(acc);
      lds_barrier();   // B(qb - 1): tile qb - 1 is in LDS; the filter waves have left the buffer tile qb will take
      __builtin_amdgcn_sched_barrier(0);
      chain.template operator()<H2, NF>(acc);
      wp += NF * 64;
    };
    f32x16 acc0, acc1;
    zero(acc0);
    chain.template operator()<0, NF>(acc0);
    wp += NF * 64;
    int qb = 1;
    for (; qb + 1 < nqblocks; qb += 2) {
      step(qb, acc1, acc0);
      step(qb + 1, acc0, acc1);
    }
    if (qb < nqblocks) {
      step(qb, acc1, acc0);
      write_tile(acc1, qb & 1);
    } else {
      write_tile(acc0, (qb - 1) & 1);
    }
    lds_barrier();   // B(nqblocks - 1)
  } else {
    int lcount = 0;  // entries of this wave's survivor list (uniform)
    auto flush = [&]() __attribute__((always_inline)) {
      KNN_TL(const long long f0 = __builtin_readcyclecounter();)
      for (int i = lane; i < lcount; i += 64) {  // this wave's own DS writes, in order: no barrier
        const unsigned long long key = s_key[pair][i];
        const unsigned qrow = s_row[pair][i];
        const unsigned pos = atomicAdd(f.cnt + qrow, 1u);
        if (pos < (unsigned)kKnnCap) f.cand[(long)qrow * kKnnCap + pos] = key;
      }
      lcount = 0;
      KNN_TL(tl_flush += __builtin_readcyclecounter() - f0;)
    };
    KNN_TL(if (g_knn_prio == 2) __builtin_amdgcn_s_setprio(3); if (g_knn_prio == 3) __builtin_amdgcn_s_setprio(1);)
    lds_barrier();   // #0
    const float xn = s_xn[pair][j];
    for (int qb = 0; qb < nqblocks; ++qb) {
      lds_barrier();   // B(qb)
      // the tile and the 16 (|q|^2, threshold) pairs of this lane's query rows: 4 + 8 wide LDS reads, all in flight together
      f32x16 t;
      uint2 qt[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = s_tile[pair][qb & 1][q][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * q + e] = v[e];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4* qp = reinterpret_cast<const uint4*>(s_qt + qb * 32 + 8 * g + 4 * half);
        const uint4 a = qp[0], b = qp[1];
        qt[4 * g + 0] = make_uint2(a.x, a.y);
        qt[4 * g + 1] = make_uint2(a.z, a.w);
        qt[4 * g + 2] = make_uint2(b.x, b.y);
        qt[4 * g + 3] = make_uint2(b.z, b.w);
      }
      // 16 independent slices (query row qb * 32 + 8 (v >> 2) + 4 half + (v & 3) against this lane's database row): keys and votes
      // first -- branch-free, 16 chains the VALU can interleave (the one-slice-at-a-time form of knn_table_kernel is a serial chain
      // of ~40 dependent instructions per slice: as a wave's only work it took longer than the block's MFMAs) --
      float d[16];
      unsigned long long m[16];
      unsigned long long any = 0;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        d[v] = knn_dist_raw(__builtin_bit_cast(float, qt[v].x), xn, t[v]);
        m[v] = __builtin_amdgcn_ballot_w64(valid && d[v] <= __builtin_bit_cast(float, qt[v].y));   // (float threshold: knn_tau_float)
        any |= m[v];
      }
      // ... then the survivors (none in 7 slices of 8) take their slots in the wave's list, slice by slice: the order of the list
      // does not matter (the candidate sort orders keys), its content is what knn_table_kernel<D, true> appends
      if (any) {  // uniform
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if ((v & 3) == 0 && lcount + 256 > kKnnWaveList) flush();   // (room for 4 x 64 entries, checked once per 4 slices)
          if (m[v]) {  // uniform
            if ((m[v] >> lane) & 1) {
              const int p = lcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[v] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[v], 0u));
              s_key[pair][p] = ((unsigned long long)knn_key_hi(d[v]) << 32) | (unsigned)(n0 + j);   // (d is not a NaN here)
              s_row[pair][p] = (unsigned)(qb * 32 + 8 * (v >> 2) + 4 * half + (v & 3));
            }
            lcount += __popcll(m[v]);
          }
        }
      }
    }
    flush();
  }
  KNN_TL(if (g_knn_tl && blockIdx.x % 256 == 0 && blockIdx.x / 256 < 8 && lane == 0) {
    long long* o = g_knn_tl + ((blockIdx.x / 256) * 8 + wave) * 8;
    o[0] = mfma_role; o[1] = simd; o[2] = __builtin_readcyclecounter() - tl_t0; o[3] = tl_bar; o[4] = tl_flush; o[5] = pair; o[6] = blockIdx.x;
  })
}

}  // namespace qinco
