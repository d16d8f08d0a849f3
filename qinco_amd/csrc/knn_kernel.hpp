// Brute-force L2 top-k of the small-db search that follows the hot path:
//   run_search_full_direct_small_db (reference qinco/search/search_tasks.py:551-603):
//     dists = approx_pairwise_distance(queries, xhat_db)  (utils.py:336-346:  |q|^2 + |x|^2 - 2 q.x, fp32)
//     shortlist = dists.argsort(-1)[:, :100]
// Two kernels per chunk of queries:
//   knn_table_kernel<D>  the (queries x db) distance table on the fp32 MFMA with the transposed trick of the IVF
//                        kernel, roles swapped: a wave keeps 32 DATABASE rows as B operands in registers and
//                        streams the chunk's query fragments (packed on the device, L2-resident) as A operands, so
//                        the 32x32 result tile leaves lanes 0..31 with 32 consecutive database columns of one query
//                        row: 128-byte contiguous stores into table[query][db].  The database is read once per chunk.
//                        2 D FLOP per pair; 4 B written per pair.  MFMA-bound (D=128: 4 KiB stored per 4096 pipe cycles
//                        per SIMD = 2.4 TB/s of HBM writes chip-wide).
//   knn_select_kernel    one workgroup per query row: radix select of the k smallest 64-bit keys
//                        (order-preserving distance bits, db index) -- 12-bit LDS histograms refine a key prefix
//                        until the k-th key's bucket and everything below it fit an LDS buffer, one compaction pass
//                        collects them, a bitonic sort orders them.  Ties resolve to the lower index (stable argsort).
//                        HBM-bound: the row is read 2 (rarely 3+) times: 8-12 B per pair.
// Large databases (round 5) take the FILTERED form instead, which never writes the (queries x db) table: 4 B written and
// 8-12 B read per pair made the search HBM-bound at half the matrix pipe's rate (40 GB of table per 10^4 queries x 10^6 rows).
//   (1) the two kernels above on a strided 1/s SAMPLE of the database (rows 0, s, 2s, ...) give tau[q] = the k-th smallest
//       key of the sample: an upper bound of the k-th smallest key of the whole database (the sample is a subset);
//   (2) knn_table_kernel<D, FILT = true> computes the whole table on the matrix pipe -- the same instruction sequence per
//       pair, hence the same bits -- and appends only the pairs with key <= tau[q] to the query's candidate list
//       (about k s of them; one L2 atomic per survivor, 0.2 % of the pairs);
//   (3) knn_cand_select_kernel sorts each query's candidates as 64-bit (distance, index) keys in LDS: the first k.
// The candidate set contains the k smallest keys whatever the data (every key <= the k-th smallest is <= tau), so the
// result is the unfiltered form's bit for bit.  A list that overflows its kKnnCap slots (adversarial data: thousands of
// rows at exactly tau) or holds fewer than k keys (NaN distances) raises the chunk's flag, and the unfiltered kernels --
// enqueued behind it, predicated on that flag -- redo the chunk: no host round trip, the call stays asynchronous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mlp_args.hpp"

namespace qinco {

// rows (n, D) row-major -> fragment stream of ceil(n/32) blocks (same layout as put_frag in qinco_hip.hip:
// fragment (block, ib, q): lane l, component e = rows[block*32 + (l&31)][ib*32 + 8q + 4(l>>5) + e]); rows >= n are 0.
// norms[r] = sum_d rows[r][d]^2 (sequential fp32), 0 for padding rows.
__global__ void __launch_bounds__(256)
knn_pack_rows_kernel(const float* __restrict__ rows, long n, int D, f32x4* __restrict__ stream, float* __restrict__ norms,
                     long nblocks) {
  const int NDB = D / 32;
  const long total = nblocks * NDB * 4 * 64;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int l = (int)(t & 63);
    const long frag = t >> 6;
    const int q = (int)(frag & 3);
    const long bi = frag >> 2;
    const int ib = (int)(bi % NDB);
    const long blk = bi / NDB;
    const long r = blk * 32 + (l & 31);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < n) v = *reinterpret_cast<const f32x4*>(rows + r * D + ib * 32 + 8 * q + 4 * (l >> 5));
    stream[t] = v;
  }
  const long nrows = nblocks * 32;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    if (r < n)
      for (int d = 0; d < D; ++d) s = fmaf(rows[r * D + d], rows[r * D + d], s);
    norms[r] = s;
  }
}

constexpr int kKnnCap = 8192;      // LDS candidate buffer (64 KiB of 64-bit keys) = slots of a query's candidate list
constexpr int kKnnThreads = 1024;
constexpr int kKnnMaxK = 2048;

__device__ __forceinline__ unsigned knn_key_hi(float d) {
  const unsigned u = __builtin_bit_cast(unsigned, d);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> unsigned (NaNs sort last)
}

// |q|^2 + |x|^2 - 2 q.x with the reference's roundings (utils.py:336-346); a NaN comes out as THE positive quiet NaN: which sign and
// payload an fsub / fma hands on depends on the instruction form hipcc picks (a negated operand flips a NaN's sign), the two forms of
// the search compile differently, and a negative NaN's key sorts FIRST -- the forms disagreed on corrupt rows one run in four.
__device__ __forceinline__ float knn_dist(float qn, float xn, float dot) {
  const float d = __fsub_rn(__fadd_rn(qn, xn), __fmul_rn(2.f, dot));
  return d != d ? __builtin_bit_cast(float, 0x7fc00000u) : d;
}

// The same distance without the NaN rule, for the filter's threshold test only (a NaN fails `d <= t` whatever its sign).
__device__ __forceinline__ float knn_dist_raw(float qn, float xn, float dot) {
  return __fsub_rn(__fadd_rn(qn, xn), __fmul_rn(2.f, dot));
}
// The threshold the filter compares with, as a FLOAT: the distance whose key's high word is tau.  `d <= t` as floats keeps every
// pair whose key is <= tau (the key order is the float order with -0 below +0: the float test can only keep MORE -- a +0 against a
// threshold of -0 -- and the candidate sort orders by the full keys anyway).  A NaN threshold (fewer than k finite distances in the
// sample) and the 0 of a padding row (-> a negative NaN) keep nothing: the list comes up short and the chunk is redone unfiltered,
// as it is when the key test floods the list.  One v_cmp per pair instead of the five VALU instructions of the key: a wave is held
// for the 16 passes of each of its f32 MFMAs and nothing of its own stream hides under them, so every filter instruction is issue
// time its chain does not get (round 6: scripts/ubench/mfma_valu.hip).
__device__ __forceinline__ float knn_tau_float(unsigned tau) {
  return __builtin_bit_cast(float, (tau & 0x80000000u) ? (tau & 0x7fffffffu) : ~tau);
}

// The filtered form's arguments (FILT = true); `pred` also predicates the unfiltered form when it runs as the fall-back.
struct KnnFilt {
  const unsigned* tau;        // per query row of the chunk: high word of the k-th smallest key of the sample
  unsigned* cnt;              // per query row: candidates appended (may exceed kKnnCap: the list overflowed)
  unsigned long long* cand;   // (chunk, kKnnCap) keys
  int nq_valid;               // query rows of the chunk (rows beyond are padding of the last block)
  const int* pred;            // nullptr: always run; else run only if *pred != 0
};

// database row of tile column c = c * row_stride (row_stride = 1: the database itself; s: its strided sample, N = sample rows)
//
// FILT: nothing of the table is stored.  The thresholds of the chunk's query rows sit in LDS; behind a block's MFMAs every
// (query, 32 database rows) slice of the tile is compared with its threshold and the wave votes: `__ballot` of the survivors
// (none in 7 slices of 8 at the usual 0.2 %) -> slots in the WAVE's own LDS list by prefix count, the list's length a scalar
// -- no atomic, no memory round trip inside the MFMA loop (a first version appended straight to the per-query lists: 34
// `s_waitcnt vmcnt(0)` in the loop, each draining the fragment ring, and 268 registers = one wave per SIMD).  The list
// (about 130 entries over a wave's life at k = 100) is flushed to the per-query lists in HBM when it could overflow and at
// the end: one returning atomic per entry, 64 entries per wait.
constexpr int kKnnLdsNeeded = 4096 * 4 + kKnnCap * 8 + 64;   // knn_select_kernel: hist + buf (+ scalars): the largest static LDS here
static_assert(kKnnLdsNeeded <= 160 * 1024, "the selection kernels' LDS must fit a gfx950 CU");
constexpr int kKnnMaxChunk = 4096;   // query rows per chunk (qinco_knn_search): thresholds in LDS
constexpr int kKnnWaveList = 512;    // entries of a wave's survivor list (room for 4 x 64 checked once per 8 query rows).
static_assert(2 * ((kKnnMaxChunk + 32) * 8 + 4 * kKnnWaveList * 12) <= 160 * 1024,
              "knn_table_kernel<D, true> is launched for two workgroups per CU (waves_per_eu(2)): both residents' LDS must fit");
// Two workgroups per CU = two waves per SIMD, measured: one wave per SIMD (lists of 2048 entries = 128 KiB of LDS) 27.4 -> 33.3 ms.
// Per-wave cycle stamps (scripts/exp_knn_timeline.py): a wave needs ~545 cycles per 4-MFMA step of this loop (256 of matrix pipe)
// when it has its SIMD nearly to itself in the launch's tail and ~640 when it shares it -- the pipe is 0.80 busy inside the loop.

template <int D, bool FILT, int WPE = (D <= 128 ? 2 : 1)>   // WPE: waves per SIMD (1: the LDS lists are sized so that one workgroup fills a CU)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE)))
knn_table_kernel(const f32x4* __restrict__ qstream, const float* __restrict__ qnorm, int nqblocks,
                 const float* __restrict__ db, long N, long row_stride, float* __restrict__ table, long ldt, KnnFilt f) {
  constexpr int NDB = D / 32;
  if (!FILT && f.pred && *f.pred == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  __shared__ uint2 s_qt[FILT ? kKnnMaxChunk + 32 : 1];   // per query row of the chunk: (|q|^2, threshold as a float: knn_tau_float) bits
  constexpr int LIST = (FILT && WPE == 1 && D <= 128) ? 2 * kKnnWaveList : kKnnWaveList;   // (1024 entries: 81 KiB with the thresholds -> one workgroup per CU)
  __shared__ unsigned long long s_key[FILT ? 4 : 1][FILT ? LIST : 1];
  __shared__ unsigned s_row[FILT ? 4 : 1][FILT ? LIST : 1];
  if constexpr (FILT) {
    // a padding row's fragments and norm are 0: its distances are |x|^2 >= 0, keys >= 0x80000000 -- above a threshold of 0
    for (int i = threadIdx.x; i < nqblocks * 32; i += 256)
      s_qt[i] = make_uint2(__builtin_bit_cast(unsigned, qnorm[i]), __builtin_bit_cast(unsigned, knn_tau_float(i < f.nq_valid ? f.tau[i] : 0u)));
    __syncthreads();
  }
  const long n0 = ((long)blockIdx.x * 4 + wave) * 32;
  if (n0 >= N) return;
  long row = n0 + j;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  const float* xp = db + row * row_stride * D + half * 4;
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
        xn = fmaf(t[e], t[e], xn);
      }
    }
  }
  xn += __shfl_xor(xn, 32);

  constexpr int NF = NDB * 4;
  constexpr int P = (NF % 16 != 0) ? NF : (NDB > 8 ? 8 : 16);  // register ring, as in ivf_assign_kernel
  const f32x4* wp = qstream + lane;
  f32x4 ring[P];
#pragma unroll
  for (int i = 0; i < P; ++i) ring[i] = wp[i * 64];
  // the ring's loads are BUFFER loads (round 6, see the filtered form below): lane offset in a VGPR, fragment offset in an SGPR
  const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(qstream), 0, 0x7fffffff, 0x00020000);
  const int voff = lane * 16;
  int soff = 0;   // byte offset of the current block's first fragment in the stream (uniform)
  auto ring_load = [&](const int frag) __attribute__((always_inline)) -> f32x4 {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (four fragments share one scalar offset: the 1 KiB steps inside a 4 KiB group ride in the instruction's 12-bit immediate)
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, voff + (frag & 3) * 1024, soff + (frag >> 2) * 4096, 0);
    return __builtin_bit_cast(f32x4, r);
  };
  float* tp = table + n0 + j;
  int lcount = 0;  // entries of this wave's survivor list (uniform)
  auto flush = [&]() __attribute__((always_inline)) {
    if constexpr (FILT) {
      for (int i = lane; i < lcount; i += 64) {  // this wave's own DS writes, in order: no barrier
        const unsigned long long key = s_key[wave][i];
        const unsigned qrow = s_row[wave][i];
        const unsigned pos = atomicAdd(f.cnt + qrow, 1u);
        if (pos < (unsigned)kKnnCap) f.cand[(long)qrow * kKnnCap + pos] = key;
      }
      lcount = 0;
    }
  };
  auto block = [&](const int qb) __attribute__((always_inline)) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 qn[4];  // fetched ahead of the MFMA chain (see ivf_assign_kernel)
#pragma unroll
    for (int g = 0; g < 4; ++g) qn[g] = *reinterpret_cast<const f32x4*>(qnorm + qb * 32 + 8 * g + 4 * half);
#pragma unroll
    for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = ib * 4 + q;
        const f32x4 w = ring[i % P];
        ring[i % P] = ring_load(i + P);
        // fences: without them hipcc sinks each ring load down to its use one block later (a vmcnt(0) every 4 MFMAs),
        // or hoists the block's 64 MFMAs above all of its loads
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xt[ib][4 * q + e], acc, 0, 0, 0);
      }
    }
    soff += NF * 1024;
    // lane holds queries qb*32 + 8g + 4*half + e of database row n0 + j
    if (valid) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int base = qb * 32 + 8 * g + 4 * half;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          tp[(long)(base + e) * ldt] = knn_dist(qn[g][e], xn, acc[4 * g + e]);
      }
    }
  };
  if constexpr (FILT) {
    // Software pipeline inside the wave: block qb - 1's tile waits in the other accumulator and is filtered during block qb's chain.
    // (History: the filter behind its own block's chain -- both waves of a SIMD in their MFMA phase, then both in their
    // ~700-instruction filter phase with the matrix pipe idle: 0.68 of the pipe at D = 128; round 5: one (query, 32 rows) slice behind
    // every 64 / NF-th of the chain with the next slice's LDS read a slice ahead: 0.70.)
    // Round 6 -- every instruction of a wave that is not an MFMA takes the matrix pipe away from that wave's chain for its issue
    // time (scripts/ubench/mfma_valu.hip: ~5 cycles per VALU instruction behind a dependent v_mfma_f32_32x32x2_f32, fully additive;
    // only the OTHER wave of the SIMD can fill the gap, and only with an MFMA of its own).  Rounds 5's per-slice filter spent ~16
    // such instructions per 4 MFMAs (ballot through v_cndmask + v_cmp_ne, a 64-bit VALU address per ring load, a branch and an LDS
    // read per slice): 0.70-0.75 of the pipe.  Now: (a) the ring loads are BUFFER loads -- lane offset in one VGPR, the fragment's
    // offset in an SGPR: one SALU add per load instead of three VALU instructions; (b) the filter of block qb - 1 runs as ONE burst
    // behind the first fragment of block qb: 8 wide LDS reads, the 16 distances with packed adds / fmas, 16 compares straight into
    // SGPR pairs (the ballot IS the compare's result), one branch.
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid);
    auto burst = [&](const f32x16& prev, const int pq) __attribute__((always_inline)) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {   // two halves of 8 slices (registers): slice v = 8 hb + 4 gg + e -> query row pq * 32 + 8 (2 hb + gg) + 4 half + e
        uint2 qt[8];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const uint4* qp = reinterpret_cast<const uint4*>(s_qt + pq * 32 + 8 * (2 * hb + gg) + 4 * half);
          const uint4 a = qp[0], b = qp[1];
          qt[4 * gg + 0] = make_uint2(a.x, a.y);
          qt[4 * gg + 1] = make_uint2(a.z, a.w);
          qt[4 * gg + 2] = make_uint2(b.x, b.y);
          qt[4 * gg + 3] = make_uint2(b.z, b.w);
        }
        float d[8];
        unsigned long long m[8];
        unsigned long long any = 0;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          d[v] = knn_dist_raw(__builtin_bit_cast(float, qt[v].x), xn, prev[8 * hb + v]);
          m[v] = __builtin_amdgcn_ballot_w64(d[v] <= __builtin_bit_cast(float, qt[v].y)) & vmask;   // (float threshold: knn_tau_float)
          any |= m[v];
        }
        if (any) {  // uniform; none in 7 slices of 8
#pragma unroll
          for (int v = 0; v < 8; ++v) {
            if ((v & 3) == 0 && lcount + 256 > LIST) flush();   // (room for 4 x 64 entries, checked once per 4 slices)
            if (m[v]) {  // uniform
              if ((m[v] >> lane) & 1) {   // (d is not a NaN here: its key needs no NaN rule)
                const int p = lcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[v] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[v], 0u));
                s_key[wave][p] = ((unsigned long long)knn_key_hi(d[v]) << 32) | (unsigned)(n0 + j);
                s_row[wave][p] = (unsigned)(pq * 32 + 8 * (2 * hb + (v >> 2)) + 4 * half + (v & 3));
              }
              lcount += __popcll(m[v]);
            }
          }
        }
      }
    };
    auto blockp = [&](const int qb, f32x16& acc, const f32x16& prev, auto have_prev) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ib = 0; ib < NDB; ++ib) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = ib * 4 + q;
          const f32x4 w = ring[i % P];
          ring[i % P] = ring_load(i + P);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], xt[ib][4 * q + e], acc, 0, 0, 0);
          if constexpr (decltype(have_prev)::value) {
            if (i == 0) {
              burst(prev, qb - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      soff += NF * 1024;
    };
    f32x16 acc0, acc1 = {};
    if constexpr (NDB <= 8) {
      blockp(0, acc0, acc1, std::false_type{});
      int qb = 1;
      for (; qb + 1 < nqblocks; qb += 2) {
        blockp(qb, acc1, acc0, std::true_type{});
        blockp(qb + 1, acc0, acc1, std::true_type{});
      }
      if (qb < nqblocks) {
        blockp(qb, acc1, acc0, std::true_type{});
        burst(acc1, qb);
      } else {
        burst(acc0, qb - 1);
      }
    } else {  // D = 768: 384 registers of database rows leave no room for a second accumulator (12 spilled): filter behind the chain
      for (int qb = 0; qb < nqblocks; ++qb) {
        blockp(qb, acc0, acc1, std::false_type{});
        burst(acc0, qb);
      }
    }
    flush();
  } else {
    int qb = 0;  // two blocks per trip (where the registers allow it): see ivf_assign_kernel
    if constexpr (NDB <= 8) {
      for (; qb + 1 < nqblocks; qb += 2) {
        block(qb);
        block(qb + 1);
      }
    } else {
      for (; qb < nqblocks; ++qb) block(qb);
    }
    if (qb < nqblocks) block(qb);
  }
}

// ---------------------------------------------------------------------------------------------
// selection
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long knn_key(float d, unsigned idx) {
  return ((unsigned long long)knn_key_hi(d) << 32) | idx;
}
__device__ __forceinline__ float knn_key_dist(unsigned long long key) {
  unsigned u = (unsigned)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, u);
}

// buf[0 .. count) unsorted keys in LDS (count <= kKnnCap, a barrier behind the last write): bitonic sort, the first k emitted
__device__ __forceinline__ void knn_sort_emit(unsigned long long* buf, int count, int k, long q, long long* __restrict__ ids_out,
                                              float* __restrict__ dist_out, unsigned* __restrict__ tau_out) {
  const int tid = threadIdx.x;
  int n2 = 64;
  while (n2 < count) n2 <<= 1;
  for (int i = count + tid; i < n2; i += kKnnThreads) buf[i] = ~0ull;
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < n2 / 2; t += kKnnThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = buf[lo], b = buf[hi];
        if ((a > b) == up) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  if (tau_out) {  // the sample pass of the filtered form: only the k-th key's high word is wanted
    if (tid == 0) tau_out[q] = (unsigned)(buf[k - 1] >> 32);
    return;
  }
  for (int i = tid; i < k; i += kKnnThreads) {
    const unsigned long long key = buf[i];
    ids_out[q * k + i] = (long long)(key & 0xffffffffull);
    if (dist_out) dist_out[q * k + i] = knn_key_dist(key);
  }
}

// table: (nq, ldt) fp32, row q holds N distances.  ids_out (nq, k) int64, dist_out (nq, k) fp32 or nullptr -- or, with tau_out
// (the sample pass of the filtered form), tau_out[q] = high word of the k-th smallest key and cnt_out[q] = 0, nothing else.
// pred: nullptr or the chunk's fall-back flag (run only if set).
__global__ void __launch_bounds__(kKnnThreads)
knn_select_kernel(const float* __restrict__ table, long ldt, long N, int k, long long* __restrict__ ids_out,
                  float* __restrict__ dist_out, unsigned* __restrict__ tau_out, unsigned* __restrict__ cnt_out,
                  const int* __restrict__ pred) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned long long buf[kKnnCap];
  __shared__ unsigned s_ub, s_bucket, s_before, s_cnt, s_count;
  const int tid = threadIdx.x;
  if (pred && *pred == 0) return;
  if (cnt_out && tid == 0) cnt_out[blockIdx.x] = 0;
  const float* row = table + (long)blockIdx.x * ldt;

  unsigned long long prefix = 0;  // the k-th key starts with these `pbits` bits
  int pbits = 0;
  unsigned long long below = 0;   // keys strictly below the prefix range
  // histogram of the next digit over keys matching the prefix; bins above a running bound are not needed
  // (the bound only shrinks), which removes almost all LDS atomics after the first segment
  while (true) {
    const int w = (64 - pbits) < 12 ? (64 - pbits) : 12;
    const int shift = 64 - pbits - w;
    for (int i = tid; i < 4096; i += kKnnThreads) hist[i] = 0;
    if (tid == 0) s_ub = 4095;
    __syncthreads();
    const unsigned need = (unsigned)(k - below);  // rank of the k-th key inside the prefix range (>= 1)
    constexpr long SEG = (long)kKnnThreads * 16;  // multiple of 4 * kKnnThreads
    int seg_i = 0;
    for (long s0 = 0; s0 < N; s0 += SEG, ++seg_i) {
      const unsigned ub = s_ub;
      const long s1 = (s0 + SEG < N) ? s0 + SEG : N;
      for (long i = s0 + 4 * tid; i < s1; i += 4 * kKnnThreads) {  // rows are 128-byte aligned (ldt % 32 == 0)
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long key = knn_key(v[e], (unsigned)(i + e));
          if (i + e < N && (pbits == 0 || (key >> (64 - pbits)) == prefix)) {
            const unsigned dgt = (unsigned)(key >> shift) & ((1u << w) - 1u);
            if (dgt <= ub) atomicAdd(&hist[dgt], 1u);
          }
        }
      }
      __syncthreads();
      const bool last = s1 >= N;
      if (last || (seg_i & (seg_i + 1)) == 0) {  // after segments 1, 2, 4, 8, ... and at the end
        if (tid < 64) {
          // wave 0: lane l owns bins [64 l, 64 l + 64); find the first bin where the running count reaches `need`
          unsigned sum = 0;
          for (int b = 0; b < 64; ++b) sum += hist[tid * 64 + b];
          unsigned incl = sum;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off);
            if (tid >= off) incl += o;
          }
          const unsigned excl = incl - sum;
          const bool mine = excl < need && incl >= need;
          if (mine) {
            unsigned run = excl;
            for (int b = 0; b < 64; ++b) {
              const unsigned c = hist[tid * 64 + b];
              if (run + c >= need) {
                s_bucket = tid * 64 + b;
                s_before = run;
                s_cnt = c;
                s_ub = tid * 64 + b;
                break;
              }
              run += c;
            }
          }
        }
        __syncthreads();
      }
    }
    const unsigned bucket = s_bucket, before = s_before, cnt = s_cnt;
    __syncthreads();
    below += before;
    prefix = (prefix << w) | bucket;
    pbits += w;
    if (below + cnt <= (unsigned long long)kKnnCap || pbits >= 64) break;
  }

  // collect every key whose leading pbits are <= prefix (>= k of them, <= kKnnCap)
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (long i = 4 * tid; i < N; i += 4 * kKnnThreads) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned long long key = knn_key(v[e], (unsigned)(i + e));
      if (i + e < N && (key >> (64 - pbits)) <= prefix) {
        const unsigned pos = atomicAdd(&s_count, 1u);
        if (pos < (unsigned)kKnnCap) buf[pos] = key;
      }
    }
  }
  __syncthreads();
  knn_sort_emit(buf, (int)s_count, k, (long)blockIdx.x, ids_out, dist_out, tau_out);
}

// The filtered form's selection: query row q's candidate list (cnt[q] keys, every key <= tau[q], the k smallest among them)
// -> sorted in LDS, the first k emitted.  An overflowed list (cnt > kKnnCap) or one with fewer than k keys raises *ovf:
// the predicated unfiltered kernels behind this launch redo the whole chunk.
__global__ void __launch_bounds__(kKnnThreads)
knn_cand_select_kernel(const unsigned long long* __restrict__ cand, const unsigned* __restrict__ cnt, int k,
                       long long* __restrict__ ids_out, float* __restrict__ dist_out, int* __restrict__ ovf) {
  __shared__ unsigned long long buf[kKnnCap];
  const int tid = threadIdx.x;
  const unsigned count = cnt[blockIdx.x];
  if (count > (unsigned)kKnnCap || count < (unsigned)k) {
    if (tid == 0) atomicOr(ovf, 1);
    return;
  }
  const unsigned long long* src = cand + (long)blockIdx.x * kKnnCap;
  for (unsigned i = tid; i < count; i += kKnnThreads) buf[i] = src[i];
  __syncthreads();
  knn_sort_emit(buf, (int)count, k, (long)blockIdx.x, ids_out, dist_out, nullptr);
}

// sum over count elements of (a - b)^2, fp64 accumulation (AnyVectMSE.update, reference qinco/metrics.py:43-50)
__global__ void __launch_bounds__(256)
sqerr_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, long count, double* __restrict__ out) {
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += (double)d * (double)d;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace qinco
