// Small-launch form of the fused QINCo codeword MLP for gfx950 (MI355X / CDNA4).
//
// mlp_kernel.hpp gives a wave 32 rows and ALL features: 128 rows per workgroup, so a launch below 256 CUs x 128 rows leaves
// CUs idle -- and that is every decode call the reference makes (compute_MSE decodes cfg.batch = 1024 rows per call,
// qinco_tasks.py:112-125; the search re-rank decodes cfg.search.batch_size = 12 288, search_tasks.py:475-486) and every greedy
// encode step at the reference's batch (1024 vectors x A = 16 candidates).  This form turns the split around:
//
//  * a workgroup owns 16 * NT rows (NT = 1..4 tiles of 16) and its eight waves (two per SIMD: while one issues its ring and
//    address bookkeeping the other's MFMAs keep the matrix pipe busy) split the OUTPUT features of every GEMM
//    (wave w owns the 16-feature output blocks 8 j + w); the host picks NT so that the launch has ~one workgroup per CU;
//  * a layer's input must then be seen by all the waves: activations live in LDS between the GEMMs, in the B-operand layout of
//    v_mfma_f32_16x16x4_f32 (1 KiB per block and row tile, lane-linear ds_read_b128 / ds_write_b128), one workgroup barrier per
//    GEMM (two when the two activation buffers do not fit next to the weight rings);
//  * every wave streams ITS quarter of the weights, as A operands in consumption order, through a private LDS-DMA ring
//    (global_load_lds_dwordx4, counted vmcnt by hand like mlp_kernel's rings); one 1 KiB fragment feeds 4 NT MFMAs;
//  * decode runs EVERY step of a row tile in one launch (xhat never leaves the workgroup; the reference's loop
//    qinco_inference.py:66-75): 2 launches per decode call instead of 2 (M - 1) + 3.
//
// Numerics: same products, same order as mlp_kernel.  Within a 16-feature block the features sit in the order in which the
// 32-row kernel's fragments contract them (mlp_args.hpp small_feat), every chain starts from zero and ends in the same
// element-wise operations (T + U, relu(P + Q), z + chain, (o + c) + xhat), and the candidate distances are accumulated per
// (row, half) in mlp_kernel's feature order from an LDS copy of the candidate tile.  Reference semantics: QINCoInferenceStep.forward
// (qinco/model/qinco_inference.py:31-40 = QINCoStep.forward qinco_base.py:262-280), the epilogue of
// QINCoInferenceStepEncoder.forward (:190-199) and QINCoInferenceDecoder.forward (:66-75).
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_args.hpp"
#include "mlp_kernel.hpp"

namespace qinco {

#ifndef QINCO_MFMA16
#define QINCO_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// LDS plan of an instantiation (host and device agree through this one function).
struct SmallPlan {
  bool ok;
  bool DB;          // two activation buffers: one barrier per GEMM
  int PW;           // weight-ring depth per wave, fragments
  int ACTB;         // 16-feature blocks per activation buffer
  int LAG;          // a ring slot is refilled LAG reads after it was read
  unsigned lds_bytes;
};
// Decode's head GEMM U = W_x xhat reads the workgroup's xhat tile from LDS: NDB 16-feature blocks x NT row tiles of 1 KiB.  At
// D = 768 that is 48 KiB per row tile, more than fits next to the rings for NT > 1: the tile is then published in chunks of
// kSmallHeadChunk blocks and the GEMM's accumulation chains run on across the chunks (same products, same order: same bits).
constexpr int kSmallHeadChunk = 16;
constexpr int small_head_chunk(int NDB) { return (NDB > 24 && NDB % kSmallHeadChunk == 0) ? kSmallHeadChunk : NDB; }
constexpr SmallPlan small_plan(int D, int DE, int DH, int NT, bool fold2, bool dec) {
  const SmallDims S = small_dims(D, DE, DH, fold2);
  SmallPlan p{};
  int actb = S.NEB > S.NHB ? S.NEB : S.NHB;
  if (dec && small_head_chunk(S.NDB) > actb) actb = small_head_chunk(S.NDB);
  // (registers: a decode wave carries xhat, c and the next step's c for its NDW blocks of every row tile -- 18 x 4 NT at D = 768)
  if (dec && S.NDW * NT > 12) return p;   // (NT = 3 at D = 768 compiles to 74-103 spilled registers with scratch traffic inside the GEMMs)
  int nobmax = S.NEW > S.NHW ? S.NEW : S.NHW;
  if (S.PROJ && S.NDW > nobmax) nobmax = S.NDW;
  p.ACTB = actb;
  p.LAG = nobmax + 1;
  const long slot = (long)actb * NT * 1024;
  const long ct = dec ? 0 : (long)16 * NT * (D + 4) * 4;   // candidate tile of the encode epilogue (aliases the activations)
  const long avail = 160 * 1024;
  const int pw_min = p.LAG + 4, pw_max = 16;
#if defined(QINCO_EXPERIMENT) && defined(QINCO_SMALL_FORCE_SINGLE)   // A/B: one activation buffer, the LDS goes to the weight rings
  const int db_first = 0;
#else
  const int db_first = 1;
#endif
  for (int db = db_first; db >= 0; --db) {
    long act = (db ? 2 : 1) * slot;
    if (ct > act) act = ct;
    long pw = (avail - act) / (1024 * kSmallWaves);
    if (pw > pw_max) pw = pw_max;
    if (pw - p.LAG - 1 > 63) pw = 64 + p.LAG;
    if (pw >= pw_min) {
      p.ok = true;
      p.DB = db != 0;
      p.PW = (int)pw;
      p.lds_bytes = (unsigned)(act + pw * 1024 * kSmallWaves);
      return p;
    }
  }
  return p;
}

// Loads hosted in a GEMM (mlp_small_kernel: gemm_hosting): GPF of the G loads are issued right behind ring read q = NOW + k (k = 0, 1, ...).
constexpr int hosted_after(int q, int NOW, int G, int GPF) {
  const int k = q - NOW;
  if (k < 0 || k * GPF >= G) return 0;
  return (k + 1) * GPF <= G ? GPF : G - k * GPF;
}
// ... and how many of them are younger than the DMA of the fragment that ring read r waits for (issued at read r - WIN)
constexpr int hosted_extra(int r, int NOW, int G, int GPF, int WIN) {
  int n = 0;
  for (int q = r - WIN; q <= r - 1; ++q)
    if (q >= 0) n += hosted_after(q, NOW, G, GPF);
  return n;
}

template <int D, int DE, int DH, int NT, bool FOLD2, bool DEC>
__global__ void __launch_bounds__(64 * kSmallWaves, 1) mlp_small_kernel(SmallArgs a) {
  constexpr int NW = kSmallWaves;
  constexpr SmallDims S = small_dims(D, DE, DH, FOLD2);
  constexpr SmallPlan PL = small_plan(D, DE, DH, NT, FOLD2, DEC);
  static_assert(PL.ok, "no LDS plan for this instantiation");
  constexpr int NDB = S.NDB, NEB = S.NEB, NHB = S.NHB, NDW = S.NDW, NEW = S.NEW, NHW = S.NHW;
  constexpr bool PROJ = S.PROJ;
  constexpr int PW = PL.PW, LAG = PL.LAG;
  constexpr bool DB = PL.DB;
  constexpr int SLOT4 = PL.ACTB * NT * 64;   // f32x4 per activation buffer
  static_assert(PROJ || NDW == NEW, "identity projections: De == D");
  constexpr bool LATE = DEC && ((NEW + NHW) * NT > 16 || NDW * NT > 8);   // table gathers after their GEMM (register budget: 256 per wave)

  extern __shared__ __attribute__((aligned(16))) f32x4 lds_small[];
  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  const int foff = (kg >> 1) + 4 * (kg & 1);   // small_feat(r, kg) = foff + {0, 2, 8, 10}[r]
  f32x4* const ring = lds_small + wave_u * PW * 64;
  f32x4* const act = lds_small + NW * PW * 64;

#ifdef QINCO_TIMELINE
  int stamp_i = 0;
  auto stamp = [&]() QINCO_LAMBDA {
    if (a.timeline && lane == 0 && stamp_i < 64) a.timeline[((long)blockIdx.x * NW + wave_u) * 64 + stamp_i] = __builtin_readcyclecounter();
    ++stamp_i;
  };
#else
  auto stamp = []() QINCO_LAMBDA {};
#endif
  const long base = (long)blockIdx.x * (16 * NT);
  long row[NT];
  bool valid[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    row[t] = base + 16 * t + n16;
    valid[t] = row[t] < a.R;
    if (!valid[t]) row[t] = a.R - 1;   // clamped rows compute a copy of the last row and are never stored
  }

  // ---- weight stream: private LDS-DMA ring ------------------------------------------------------------------------------
  // Fragment f of this wave sits at wstream + (NW f + wave) KiB and goes to ring slot f % PW.  A read of fragment g waits until at most
  // PW - LAG - 1 of the wave's vector-memory operations are outstanding -- the DMAs of g + 1 .. g + PW - LAG - 1, so g has landed;
  // any other load issued in between only makes the wait stricter -- reads the slot into registers and issues the DMA of
  // fragment g + PW - LAG into the slot of g - LAG, whose register copy has been consumed (pinned) by then.
  const f32x4* wsrc = a.wstream + wave_u * 64 + lane;
  int rd_slot = 0, wr_slot = PW - LAG;
  auto wait_vm = [&]<int N>() QINCO_LAMBDA {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
  };
  auto dma = [&](int slot) QINCO_LAMBDA {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                     (__attribute__((address_space(3))) void*)(ring + __builtin_amdgcn_readfirstlane(slot) * 64), 16, 0, 0);
    wsrc += NW * 64;
  };
#pragma unroll
  for (int i = 0; i < PW - LAG; ++i) dma(i);
  stamp();   // 0: ring prologue issued
  // EXTRA: loads other than ring DMAs issued since the DMA of the fragment being read (gathers hosted in a GEMM, below): they
  // are younger than it, so the count that lets it land grows by exactly that many.
  auto ring_read = [&]<int EXTRA>() QINCO_LAMBDA -> f32x4 {
#if defined(QINCO_EXPERIMENT) && defined(QINCO_SMALL_NO_DMA)   // timing A/B only (wrong results): the ring is never refilled
    const f32x4 v0 = ring[rd_slot * 64 + lane];
    rd_slot = rd_slot + 1 == PW ? 0 : rd_slot + 1;
    return v0;
#endif
    static_assert(PW - LAG - 1 + EXTRA <= 63, "vmcnt field");
    wait_vm.template operator()<PW - LAG - 1 + EXTRA>();
    const f32x4 v = ring[rd_slot * 64 + lane];
    asm volatile("" ::: "memory");
    dma(wr_slot);
    asm volatile("" ::: "memory");   // loads hosted behind this read stay behind its DMA (hosted_extra counts them as younger)
    rd_slot = rd_slot + 1 == PW ? 0 : rd_slot + 1;
    wr_slot = wr_slot + 1 == PW ? 0 : wr_slot + 1;
    return v;
  };
  // workgroup barrier with every LDS access of this wave completed first (a raw s_barrier: __syncthreads would also drain vmcnt,
  // i.e. the weight ring).  hipcc moves MFMAs -- and the lgkmcnt wait of the ds_read that feeds them -- across s_barrier
  // (mlp16_kernel.hpp), so the wait is explicit.
  auto barrier = [&]() QINCO_LAMBDA {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // acc[j][t] = sum over the NIB input blocks (read from `src`) of W[NW j + wave][ib] . in[ib][t]; chains start from zero.
  // The GEMM can HOST G independent loads (table rows of the next decode step), GPF of them behind each of its first ring reads:
  // vmcnt counts in order, so a load issued between two ring DMAs must be counted by every ring wait that looks across it
  // (hosted_extra) -- issued in one batch in front of a GEMM the same loads would stop its first ring read for their whole latency.
  auto gemm_hosting = [&]<int NIB, int NOW, int G, int GPF, bool ZERO = true>(f32x4 (&acc)[NOW][NT], const f32x4* src, auto&& gf) QINCO_LAMBDA {
    constexpr int WIN = PW - LAG;
    static_assert(G == 0 || NOW + (G + GPF - 1) / GPF - 1 + WIN <= NIB * NOW - 1, "hosted loads must age out inside the GEMM");
    f32x4 wf[NOW];
    f32x4 bn[NT];
    static_for<NOW>([&]<int j>() QINCO_LAMBDA { wf[j] = ring_read.template operator()<0>(); });
#pragma unroll
    for (int t = 0; t < NT; ++t) bn[t] = src[t * 64 + lane];
    if constexpr (ZERO) {   // (a chunk of a longer contraction continues the chains it is handed)
#pragma unroll
      for (int j = 0; j < NOW; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[j][t] = zero4;
    }
    auto body = [&]<bool LAST, int ib>() QINCO_LAMBDA {
      f32x4 bc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bc[t] = bn[t];
      static_for<NOW>([&]<int j>() QINCO_LAMBDA {
        f32x4 wv = wf[j];
        pin4_v(wv);   // this fragment's LDS read -- and, LDS returning a wave's reads in order, every earlier one -- has completed
        // Waves w and w + 4 share a SIMD; left alone the older one wins every arbitration, runs ahead and then idles at the barrier
        // while its partner finishes alone at a single wave's issue rate.  Taking turns at the higher priority, two fragments at a
        // time and in opposite phases, keeps both in step: -4 % (qinco2-S) ... -7 % (qinco2-L) on a decode (scripts/gpu_small_prio_sweep.sh,
        // profiles/r04_small_prio.log; one turn per fragment, four per turn, or a fixed priority for one of them: all slower).
        if ((((ib * NOW + j) >> 1) & 1) ^ (wave_u >> 2)) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
        if constexpr (!LAST) {
          constexpr int r = (ib + 1) * NOW + j;   // index of this read within the GEMM
          wf[j] = ring_read.template operator()<hosted_extra(r, NOW, G, GPF, WIN)>();
          static_for<hosted_after(r, NOW, G, GPF)>([&]<int i>() QINCO_LAMBDA { gf.template operator()<(r - NOW) * GPF + i>(); });
        }
        // The next block's B operands are requested HERE, behind the first pin: hipcc completes a pin with lgkmcnt(0), i.e. it waits
        // for every LDS read issued before it -- reads issued just in front of a pin expose a whole LDS round trip per block
        // (measured: 12-17 % of a GEMM), reads issued behind it have this fragment's 4 NT MFMAs to arrive.
        if constexpr (!LAST && j == 0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) bn[t] = src[((ib + 1) * NT + t) * 64 + lane];
        }
        static_for<4>([&]<int e>() QINCO_LAMBDA {
          static_for<NT>([&]<int t>() QINCO_LAMBDA { acc[j][t] = QINCO_MFMA16(wv[e], bc[t][e], acc[j][t]); });
        });
      });
    };
    static_for<NIB - 1>([&]<int ib>() QINCO_LAMBDA { body.template operator()<false, ib>(); });
    body.template operator()<true, NIB - 1>();
  };
  auto no_gather = []<int>() QINCO_LAMBDA {};
  auto gemm = [&]<int NIB, int NOW>(f32x4 (&acc)[NOW][NT], const f32x4* src) QINCO_LAMBDA {
    gemm_hosting.template operator()<NIB, NOW, 0, 1>(acc, src, no_gather);
  };
  auto gemm_more = [&]<int NIB, int NOW>(f32x4 (&acc)[NOW][NT], const f32x4* src) QINCO_LAMBDA {
    gemm_hosting.template operator()<NIB, NOW, 0, 1, false>(acc, src, no_gather);
  };

  // activation buffers: the GEMM after a publish reads what the publish wrote
  int cur = 0;
  auto publish = [&]<int NOW, int NB>(const f32x4 (&v)[NOW][NT]) QINCO_LAMBDA {
    if constexpr (DB) cur ^= 1;
    else barrier();   // one buffer: every wave has finished the GEMM that read it
    f32x4* dst = act + cur * SLOT4;
    static_for<NOW>([&]<int j>() QINCO_LAMBDA {
      if (NW * j + wave_u < NB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[((NW * j + wave_u) * NT + t) * 64 + lane] = v[j][t];
      }
    });
    barrier();
  };
  // blocks [B0, B0 + NB) of a tile whose blocks NW j + wave this wave holds, to positions 0 .. NB - 1 of the buffer
  auto publish_chunk = [&]<int NOW, int B0, int NB>(const f32x4 (&v)[NOW][NT]) QINCO_LAMBDA {
    if constexpr (DB) cur ^= 1;
    else barrier();
    f32x4* dst = act + cur * SLOT4;
    static_for<NOW>([&]<int j>() QINCO_LAMBDA {
      const int b = NW * j + wave_u;
      if (b >= B0 && b < B0 + NB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[((b - B0) * NT + t) * 64 + lane] = v[j][t];
      }
    });
    barrier();
  };
  auto src_buf = [&]() QINCO_LAMBDA -> const f32x4* { return act + cur * SLOT4; };

  // Row gathers, as GLOBAL loads (a pointer read from memory is generic to hipcc: flat loads, which also count in lgkmcnt and
  // would be waited for by every pin and barrier).  b past the end: the last block, never used.
  typedef const __attribute__((address_space(1))) float* gfp;
  typedef const __attribute__((address_space(1))) f32x4* gf4p;
  // four features of block b of a natural-layout row (U, Q, xhat, step-0 codebook), in block layout
  auto gather4 = [&]<int NB>(const float* rowp, int b) QINCO_LAMBDA -> f32x4 {
    gfp p = (gfp)(rowp + 16 * (b < NB ? b : NB - 1) + foff);
    return f32x4{p[0], p[2], p[8], p[10]};
  };
  // ... of a row of the block-layout copies of T, P and the codebooks (SmallStep): one 16-byte load
  auto gatherp = [&]<int NB>(const float* rowp, int b) QINCO_LAMBDA -> f32x4 {
    return *(gf4p)(rowp + 16 * (b < NB ? b : NB - 1) + 4 * kg);
  };

  f32x4 z[NEW][NT];    // this wave's blocks of z
  f32x4 xh[NDW][NT];   // decode: this wave's blocks of xhat, carried from step to step
  // this wave's blocks of the step's table rows T[code], P[code], c[code]
  f32x4 tg[NEW][NT];
  f32x4 pg[FOLD2 ? NHW : 1][NT];
  f32x4 cg[NDW][NT];
  int cid[NT];
  long grp[NT];
  auto gather_t = [&](const SmallStep& st, const int (&id)[NT]) QINCO_LAMBDA {
    static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) tg[j][t] = gatherp.template operator()<NEB>(st.ttab + (long)id[t] * DE, NW * j + wave_u);
    });
  };
  auto gather_p = [&](const SmallStep& st, const int (&id)[NT]) QINCO_LAMBDA {
    if constexpr (FOLD2) {
      static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) pg[j][t] = gatherp.template operator()<NHB>(st.ptab + (long)id[t] * DH, NW * j + wave_u);
      });
    }
  };
  auto gather_c = [&](const SmallStep& st, const int (&id)[NT], f32x4 (&dst)[NDW][NT]) QINCO_LAMBDA {
    if (a.add_c) {
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[j][t] = gatherp.template operator()<NDB>(st.codebook + (long)id[t] * D, NW * j + wave_u);
      });
    }
  };
  // Decode: where the rows of a step's tables are gathered.
  //   HOST  one step ahead, spread over the first ring reads of the step's down-projection 0 (FOLD2; no ring wait ever sees them);
  //   LATE  behind the GEMM whose result they meet (wide models with many row tiles: no registers to hold rows across a GEMM;
  //         three exposed L2 round trips per step, on steps of hundreds of microseconds);
  //   else  all of a step's rows in one batch at its start (one exposed round trip per step).
  constexpr int G_HOST = (NEW + NHW + NDW) * NT;
  constexpr int F_DOWN0 = NHB * NEW;                              // fragments of the down-projection per wave
  constexpr int HOST_ROOM = F_DOWN0 - 1 - (PW - LAG) - NEW + 1;   // ring reads of that GEMM that can host loads
  constexpr int GPF = HOST_ROOM > 0 ? (G_HOST + HOST_ROOM - 1) / HOST_ROOM : 99;
  // (wide data with several row tiles: a second set of c rows for the next step does not fit the 256 registers of a wave)
  constexpr bool HOST = DEC && FOLD2 && !LATE && GPF <= 4 && PW - LAG - 1 + GPF * (PW - LAG) <= 63 && NDW * NT <= 8;
  f32x4 cgn[HOST ? NDW : 1][NT];   // HOST: c rows of the next step (cg is still needed by this step's epilogue)

  // One step's MLP up to the last down-projection: z.  Decode: st / cid = this step, stn / cidn = the next one (HOST).
  auto run_step = [&](const SmallStep& st, const SmallStep& stn, const int (&cidn)[NT]) QINCO_LAMBDA {
    f32x4 acc_e[NEW][NT];
    f32x4 acc_h[NHW][NT];
    f32x4 y[NHW][NT];
    int l0 = 0;
    if constexpr (DEC) {
      // ---- head in the kernel: U = W_x xhat (chain from zero, xproj_kernel's order), z = T[code] + U ---------------------
      constexpr int HC = small_head_chunk(NDB);
      if constexpr (HC == NDB) {
        publish.template operator()<NDW, NDB>(xh);
        stamp();   // step + 0: xhat published
        gemm.template operator()<NDB, NEW>(acc_e, src_buf());
      } else {   // D = 768: the xhat tile in chunks of HC blocks, the chains run on across them
        static_for<NDB / HC>([&]<int c>() QINCO_LAMBDA {
          publish_chunk.template operator()<NDW, c * HC, HC>(xh);
          if constexpr (c == 0) {
            stamp();
            gemm.template operator()<HC, NEW>(acc_e, src_buf());
          } else {
            gemm_more.template operator()<HC, NEW>(acc_e, src_buf());
          }
        });
      }
      stamp();   // step + 1: U = W_x xhat
      if constexpr (FOLD2) publish.template operator()<NEW, NEB>(acc_e);   // U is the input of Q = W_up[0] U
      if constexpr (LATE) gather_t(st, cid);
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = tg[j][t] + acc_e[j][t];
      });
      if constexpr (FOLD2) {
        // Q = W_up[0] U (chain from zero), y = relu(P[code] + Q)
        stamp();   // step + 2: U published
        gemm.template operator()<NEB, NHW>(acc_h, src_buf());
        stamp();   // step + 3: Q = W_up[0] U
        if constexpr (LATE) gather_p(st, cid);
        static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            y[j][t] = pg[j][t] + acc_h[j][t];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j][t][e] = relu1(y[j][t][e]);
          }
        });
      }
    } else {
      // ---- head from the tables and the per-group projections: z = T[cid] + U[g], y = relu(P[cid] + Q[g]) ------------------
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = tg[j][t] + gather4.template operator()<NEB>(a.uproj + grp[t] * DE, NW * j + wave_u);
      });
      if constexpr (FOLD2) {
        static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            y[j][t] = pg[j][t] + gather4.template operator()<NHB>(a.qproj + grp[t] * DH, NW * j + wave_u);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j][t][e] = relu1(y[j][t][e]);
          }
        });
      }
    }
    if constexpr (FOLD2) {   // block 0: the down-projection of y = relu(P + Q)
      publish.template operator()<NHW, NHB>(y);
      stamp();   // step + 4: y published
      if constexpr (HOST) {
        // the next step's rows: T and P into the registers this step has just consumed, c into a second set
        auto host = [&]<int i>() QINCO_LAMBDA {
          if constexpr (i < NEW * NT) {
            tg[i / NT][i % NT] = gatherp.template operator()<NEB>(stn.ttab + (long)cidn[i % NT] * DE, NW * (i / NT) + wave_u);
          } else if constexpr (i < (NEW + NHW) * NT) {
            constexpr int k = i - NEW * NT;
            pg[k / NT][k % NT] = gatherp.template operator()<NHB>(stn.ptab + (long)cidn[k % NT] * DH, NW * (k / NT) + wave_u);
          } else {
            constexpr int k = i - (NEW + NHW) * NT;
            cgn[k / NT][k % NT] = gatherp.template operator()<NDB>(stn.codebook + (long)cidn[k % NT] * D, NW * (k / NT) + wave_u);
          }
        };
        gemm_hosting.template operator()<NHB, NEW, G_HOST, GPF>(acc_e, src_buf(), host);
      } else {
        gemm.template operator()<NHB, NEW>(acc_e, src_buf());
      }
      stamp();   // step + 5: down-projection 0
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = z[j][t] + acc_e[j][t];
      });
      l0 = 1;
    }
    // ---- residual FFN blocks: z = z + W_down relu(W_up z)   (QBlockFFN.forward, qinco_base.py:93-97) -----------------------
#pragma unroll 1
    for (int l = l0; l < a.L; ++l) {
      publish.template operator()<NEW, NEB>(z);
      stamp();   // block: z published
      gemm.template operator()<NEB, NHW>(acc_h, src_buf());
      stamp();   // block: up-projection
      static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_h[j][t][e] = relu1(acc_h[j][t][e]);
      });
      publish.template operator()<NHW, NHB>(acc_h);
      stamp();   // block: y published
      gemm.template operator()<NHB, NEW>(acc_e, src_buf());
      stamp();   // block: down-projection
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = z[j][t] + acc_e[j][t];
      });
    }
  };

  // ---- out_proj + (o + c) + xhat for this wave's blocks of D: o[j][t] --------------------------------------------------------
  auto out_blocks = [&](f32x4 (&o)[NDW][NT], const f32x4 (&xprev)[NDW][NT], auto&& after_gemm) QINCO_LAMBDA {
    if constexpr (PROJ) {
      publish.template operator()<NEW, NEB>(z);
      gemm.template operator()<NEB, NDW>(o, src_buf());
      after_gemm();
    } else {
      after_gemm();
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) o[j][t] = z[j][t];
      });
    }
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (a.add_c) o[j][t] = o[j][t] + cg[j][t];
        o[j][t] = o[j][t] + xprev[j][t];
      }
    });
  };

  if constexpr (DEC) {
    // ---- QINCoInferenceDecoder.forward: xhat = cw[0]; xhat += f_m(cw[m], xhat) ---------------------------------------------
    const int m_last = a.m_first + a.m_count - 1;
    typedef const __attribute__((address_space(1))) int* gip;
    auto codes_of = [&](int m, int (&dst)[NT]) QINCO_LAMBDA {   // (steps past the last: the last one again, never used)
      gip cp = (gip)(a.codes_t + (long)(m < m_last ? m : m_last) * a.R);
#pragma unroll
      for (int t = 0; t < NT; ++t) dst[t] = cp[row[t]];
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) cid[t] = ((gip)a.codes_t)[row[t]];
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) xh[j][t] = gather4.template operator()<NDB>(a.codebook0 + (long)cid[t] * D, NW * j + wave_u);
    });
    // codes are read two steps ahead of their step, table pointers one: the rows of step m + 1 are gathered during step m
    int cid_n1[NT], cid_n2[NT];
    codes_of(a.m_first, cid);
    codes_of(a.m_first + 1, cid_n1);
    SmallStep st = a.steps[a.m_first];
    if constexpr (!LATE) {   // the first step's rows: one batch
      gather_t(st, cid);
      gather_p(st, cid);
      gather_c(st, cid, cg);
    }
#pragma unroll 1
    for (int s = 0; s < a.m_count; ++s) {
      const int m = a.m_first + s;
      const SmallStep stn = a.steps[m < m_last ? m + 1 : m_last];
      codes_of(m + 2, cid_n2);
      run_step(st, stn, cid_n1);
      f32x4 o[NDW][NT];
      out_blocks(o, xh, [&]() QINCO_LAMBDA {
        if constexpr (LATE) gather_c(st, cid, cg);   // (behind the out_proj GEMM: the c rows do not sit in registers across it)
      });
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) xh[j][t] = o[j][t];
      });
      if constexpr (HOST) {
        static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) cg[j][t] = cgn[j][t];
        });
      } else if constexpr (!LATE) {
        if (s + 1 < a.m_count) {
          gather_t(stn, cid_n1);
          gather_p(stn, cid_n1);
          gather_c(stn, cid_n1, cg);
        }
      }
      st = stn;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        cid[t] = cid_n1[t];
        cid_n1[t] = cid_n2[t];
      }
      stamp();   // step end: xhat updated
    }
    // x = xhat * std + mean (two roundings, denormalize_kernel), the model's own D columns
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
      const int b = NW * j + wave_u;
      if (b < NDB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (!valid[t]) continue;
          float* op = a.out + row[t] * a.Duser;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * b + foff + (r & 1) * 2 + (r >> 1) * 8;
            if (f < a.Duser) {
              const float v = xh[j][t][r];
              op[f] = a.mean ? __fadd_rn(__fmul_rn(v, a.std_), a.mean[f]) : v;
            }
          }
        }
      }
    });
  } else {
    // ---- one encode step: candidates and their distances to x (QINCoInferenceStepEncoder.forward :178-199) ----------------
    const SmallStep st = a.steps[a.m_first];
    typedef const __attribute__((address_space(1))) int* gip;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      grp[t] = row[t] / a.A;
      cid[t] = a.cand_ids ? ((gip)a.cand_ids)[row[t]] : (int)(row[t] - grp[t] * a.A);
    }
    // every row this step gathers, in one batch in front of its first GEMM (one exposed round trip per launch)
    gather_t(st, cid);
    gather_p(st, cid);
    gather_c(st, cid, cg);
    f32x4 xprev[NDW][NT];
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) xprev[j][t] = gather4.template operator()<NDB>(a.xhat + grp[t] * D, NW * j + wave_u);
    });
    run_step(st, st, cid);
    f32x4 o[NDW][NT];
    out_blocks(o, xprev, []() QINCO_LAMBDA {});
    // the candidate tile in natural layout (rows D + 4 floats apart), over the activation buffers
    constexpr int CS = D + 4;
    float* ct = reinterpret_cast<float*>(act);
    barrier();   // every wave has finished its last GEMM's reads
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
      const int b = NW * j + wave_u;
      if (b < NDB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float* p = ct + (16 * t + n16) * CS + 16 * b + foff;
          p[0] = o[j][t][0];
          p[2] = o[j][t][1];
          p[8] = o[j][t][2];
          p[10] = o[j][t][3];
        }
      }
    });
    barrier();
    const int tid = threadIdx.x;
    // distances: thread (row, half) adds its 16 features of every 32-block in mlp_kernel's register order, the halves meet by shuffle
    if (a.dist_out && tid < 32 * NT) {   // (32 NT <= 128 threads: the first waves)
      const int rl = tid >> 1, half = tid & 1;
      long r = base + rl;
      const bool ok = r < a.R;
      if (!ok) r = a.R - 1;
      const float* xp = a.x + ((r / a.A) / a.F) * D + half * 4;
      const float* cp = ct + rl * CS + half * 4;
      float s2 = 0.f, sx = 0.f, xn = 0.f;
#pragma unroll 4
      for (int q = 0; q < D / 8; ++q) {
        const f32x4 ov = *reinterpret_cast<const f32x4*>(cp + 8 * q);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s2 = fmaf(ov[e], ov[e], s2);
          sx = fmaf(ov[e], xv[e], sx);
          xn = fmaf(xv[e], xv[e], xn);
        }
      }
      s2 += __shfl_xor(s2, 1);
      sx += __shfl_xor(sx, 1);
      xn += __shfl_xor(xn, 1);
      if (ok && half == 0) a.dist_out[r] = (xn + s2) - 2.f * sx;
    }
    // candidates: coalesced rows
    for (int i = tid; i < 16 * NT * (D / 4); i += 64 * NW) {
      const int rl = i / (D / 4), c4 = i - rl * (D / 4);
      if (base + rl < a.R)
        *reinterpret_cast<f32x4*>(a.cand_out + (base + rl) * D + 4 * c4) = *reinterpret_cast<const f32x4*>(ct + rl * CS + 4 * c4);
    }
  }
  // no LDS-DMA may be in flight when the wave ends (its LDS could be handed to the next workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace qinco
