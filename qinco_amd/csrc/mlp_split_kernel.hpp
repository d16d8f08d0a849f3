// Split-fp16 form of the fused codeword-MLP kernel (VAR bit 512; opt-in: qinco_create_ex flag QINCO_CREATE_SPLIT_F16) and of
// its per-group pre-GEMM (xproj_split_kernel, at the end of this file).
//
// Same function, tile (a wave owns 32 rows and all features), folded head (z = T[cid] + U[group], y = relu(P[cid] + Q[group]))
// and fp32 candidate / distance epilogue as mlp_kernel.hpp -- the GEMMs (the L residual FFN blocks, 94 % of the FLOPs at the
// qinco2-L shape, and out_proj when D has an even number of 32-blocks) are evaluated differently: fp32-in MFMA runs at the f32
// vector rate on gfx950 (1/16 of the fp16 rate), so each fp32 operand is split into two fp16 values, v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (22 significand bits), and
//     W.x  ~=  Whi.xhi + Whi.xlo + Wlo.xhi            (the dropped Wlo.xlo term is 2^-22 relative)
// on v_mfma_f32_32x32x16_f16 with fp32 accumulation: three MFMAs of 32 cycles for the K = 16 that costs eight fp32 MFMAs of
// 64 cycles.  Every product of two fp16 values is exact in fp32, the accumulation is fp32 as before; measured against
// float64 the result is in the same error class as the fp32 MFMA chain (scripts/ubench/split16.hip: 2.5e-7 vs 2.3e-7 rms at
// K = 384).  fp16 has 5 exponent bits: a lo part below 2^-14 is subnormal and loses bits, so the operands are kept at
// magnitudes where that does not matter by exact power-of-two scalings chosen on the host (qinco_hip.hip, qinco_create_ex):
//     z is held as z' = 2^c z, h as h' = 2^a h, W_up' = 2^d W_up, W_down' = 2^b W_down (d, b per layer) and the chain
//     accumulators are scaled back by one fp32 multiply in the epilogue they need anyway (smul[] below).
//
// Register plan (one wave per SIMD, up to 512 registers):
//   z'  fp32 master (VALU-updated by the residual add; split on the fly as the up-projection's B operand): NEB x 16 registers.
//       z' + y + the weight quads need 478 registers at De = Dh = 384 and hipcc cannot balance that over the VGPR / AGPR
//       halves (200+ spills), so the last NEB - 8 blocks of z' are parked in LDS (wave-private, 16 KiB per wave: with the
//       96 KiB ring exactly the 160 KiB of a CU) and fetched for the one split and the one residual add they see per layer.
//       (Halving the hidden layer instead is not an option: the second half's up-projection needs the block's INPUT z'.)
//   y   NHB x 16 AGPRs: the up-projection's chain accumulators (K-outer order: all NHB chains advance together, so no two
//       consecutive MFMAs depend on each other), converted IN PLACE (relu, scale, split) into the fp16 hi / lo B operands of
//       the down-projection (16 fp32 = 8 + 8 packed registers)
//   t   two chain accumulators for the down-projection (a pair of output blocks at a time, hi/lo products interleaved)
// Weight stream: fragments of 1 KiB = one MFMA A operand (32 features x 16 k x fp16) in consumption order -- per pair of
// output blocks (o, o+1) and K-chunk: hi(o), lo(o), hi(o+1), lo(o+1) -- through the workgroup-shared LDS-DMA ring, read one
// quad (4 fragments = 6 MFMAs = 192 cycles) ahead; the ring turns over 5.3x faster than in the fp32 kernel, so the barrier /
// refill cadence is a group of 16 fragments (each wave DMAs 4), measured best in scripts/ubench/split16.hip.
#pragma once
#include "mlp_kernel.hpp"

namespace qinco {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define QINCO_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// fp16 hi / lo parts of a 32-feature block of one row tile, as the two K = 16 chunks' B operands.  A lane's 16 values are
// registers r = 0..15 of the 32x32 C/D layout; chunk c takes r = 8c .. 8c+7 (the host packs the weights' K order to match).
struct SplitBlock {
  f16x8 h[2], l[2];
};

QINCO_INL SplitBlock split_block(const f32x16& v) {
  SplitBlock s;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const _Float16 hi = (_Float16)v[8 * c + i];   // round to nearest even
      s.h[c][i] = hi;
      s.l[c][i] = (_Float16)(v[8 * c + i] - (float)hi);   // exact difference, then rounded
    }
  return s;
}

// LDS accesses as inline asm.  hipcc orders every LDS instruction it can see behind ALL LDS-DMAs in flight (s_waitcnt vmcnt(0)
// in front of the first ds_read after each refill: SIInsertWaitcnts cannot tell which bytes a global_load_lds writes) --
// that would expose the full L2 latency once per group of 16 fragments.  The ring protocol below is the real ordering; the
// price of going behind the compiler's back is that the arrival of these reads is ours to wait for (lds_arrived*).
template <int OFF>
QINCO_INL void lds_read128(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
QINCO_INL void lds_write128(unsigned addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
QINCO_INL void lds_arrived4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// The workgroup-shared LDS-DMA weight ring of the split-form kernels (P fragments of 1 KiB, groups of G = 16, each of the four
// waves DMAs PER = 4 fragments of a group).  Protocol at the boundary in front of group g of the stream: every LDS read this
// wave has issued -- the previous group's -- has completed (lgkmcnt(0); hipcc would otherwise let that wait, and the MFMAs
// behind it, sink below the barrier that lets the other waves refill the group: mlp_kernel.hpp, fragmm); this wave's DMAs of
// group g have landed (it has issued NG-1 groups beyond the ones already consumed, so "at most (NG-2) x PER outstanding" means
// the oldest of them is complete; younger loads / stores of the wave only make the wait stricter); barrier; the previous group's
// slots are refilled with the group NG-1 ahead.  Sections of the stream are multiples of P, so ring slots are compile-time
// constants; `wsrc` is the origin of the current section.
template <int P>
struct SplitRing {
  static constexpr int G = 16, NG = P / G, PER = G / 4;
  const f32x4* wsrc;
  f32x4* wdst;
  unsigned addr, addr_hi;
  QINCO_INL void init(const f32x4* stream, f32x4* lds, int lane, int wave_u) {
    wsrc = stream + lane + wave_u * PER * 64;
    wdst = lds + wave_u * PER * 64;
    addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds + lane * 16;
    addr_hi = addr + 48 * 1024;
  }
  template <int T0>
  QINCO_INL void dma_group() {
    static_for<PER>([&]<int q>() QINCO_LAMBDA { dma<T0 + G - P, q>(); });
  }
  template <int T, int q>   // the q-th DMA of the refill that belongs to the boundary in front of fragment T
  QINCO_INL void dma() {
    if constexpr (q < PER)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (T + P - G + q) * 64),
                                       (__attribute__((address_space(3))) void*)(wdst + ((T + P - G + q) % P) * 64), 16, 0, 0);
  }
  QINCO_INL void prologue() {
    static_for<NG - 1>([&]<int i>() QINCO_LAMBDA { dma_group<i * G>(); });
  }
  template <int T>
  QINCO_INL void sync() {
    if constexpr (T % G == 0) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0070 | (((NG - 2) * PER) & 15) | ((((NG - 2) * PER) >> 4) << 14));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  template <int T, int q>
  QINCO_INL void read(f32x4& dst) {
    constexpr int slot = (T + q) % P;
    if constexpr (slot < 48) lds_read128<slot * 1024>(dst, addr); else lds_read128<(slot - 48) * 1024>(dst, addr_hi);
  }
  template <int T>
  QINCO_INL void ldquad(f32x4 (&dst)[4]) {
    sync<T>();
    if constexpr (T % G == 0) static_for<PER>([&]<int q>() QINCO_LAMBDA { dma<T, q>(); });
    static_for<4>([&]<int q>() QINCO_LAMBDA { read<T, q>(dst[q]); });
  }
  // one step: the six MFMAs of quad `cur` with the ring traffic of the next quad (fragments TN ..) spread between them
  template <int TN, class F>
  QINCO_INL void step(f32x4 (&cur)[4], f32x4 (&nxt)[4], f32x16& t0, f32x16& t1, const f16x8& bh, const f16x8& bl, F&& extra) {
#define QINCO_SB __builtin_amdgcn_sched_barrier(0)
#define QINCO_H(v) __builtin_bit_cast(f16x8, v)
    t0 = QINCO_MFMA_H(QINCO_H(cur[0]), bh, t0);  QINCO_SB;
    sync<TN>();                                  QINCO_SB;
    t1 = QINCO_MFMA_H(QINCO_H(cur[2]), bh, t1);  QINCO_SB;
    read<TN, 0>(nxt[0]);
    read<TN, 1>(nxt[1]);                         QINCO_SB;
    t0 = QINCO_MFMA_H(QINCO_H(cur[0]), bl, t0);  QINCO_SB;
    read<TN, 2>(nxt[2]);
    read<TN, 3>(nxt[3]);                         QINCO_SB;
    t1 = QINCO_MFMA_H(QINCO_H(cur[2]), bl, t1);  QINCO_SB;
    if constexpr (TN % G == 0) { dma<TN, 0>(); dma<TN, 1>(); }
    QINCO_SB;
    t0 = QINCO_MFMA_H(QINCO_H(cur[1]), bh, t0);  QINCO_SB;
    if constexpr (TN % G == 0) { dma<TN, 2>(); dma<TN, 3>(); }
    QINCO_SB;
    t1 = QINCO_MFMA_H(QINCO_H(cur[3]), bh, t1);
    extra();
    QINCO_SB;
#undef QINCO_H
#undef QINCO_SB
    lds_arrived4(nxt[0], nxt[1], nxt[2], nxt[3]);
    static_for<4>([&]<int q>() QINCO_LAMBDA { cur[q] = nxt[q]; });
  }
  template <int FROM, int TO>   // the padding of a section, then the next section's origin
  QINCO_INL void end_section(f32x4 (&cur)[4], f32x4 (&nxt)[4]) {
    static_for<(TO - FROM) / 4>([&]<int i>() QINCO_LAMBDA {
      ldquad<FROM + 4 * i + 4>(nxt);
      lds_arrived4(nxt[0], nxt[1], nxt[2], nxt[3]);
      static_for<4>([&]<int q>() QINCO_LAMBDA { cur[q] = nxt[q]; });
    });
    wsrc += TO * 64;
  }
};

// out_proj (D x De) takes the split form when it exists and has an even number of output blocks (host packer and kernel)
constexpr bool split_out_proj(int D, int DE) { return D != DE && (D / 32) % 2 == 0; }
// output blocks per out_proj pass: they accumulate in y's registers
constexpr int split_out_group(int D, int DH) { return D / 32 < DH / 32 ? D / 32 : DH / 32; }

template <int D, int DE, int DH, int P>
__global__ void __launch_bounds__(256, 1) mlp_split_kernel(MlpArgs a) {
  constexpr StreamDims SL = stream_dims(D, DE, DH, P, true, true);
  constexpr int NDB = SL.NDB, NEB = SL.NEB, NHB = SL.NHB;
  constexpr bool PROJ = SL.PROJ;
  constexpr int NZV = NEB > 8 ? 8 : NEB, NPARK = NEB - NZV;   // z blocks in registers / parked in LDS
  constexpr bool SPLIT_OUT = split_out_proj(D, DE);            // out_proj in the split form (the host packs accordingly)
  constexpr int OG = NDB < NHB ? NDB : NHB;                    // ... OG output blocks per pass
  static_assert(!SPLIT_OUT || (NDB % OG == 0 && OG % 2 == 0), "out_proj passes");
  constexpr int G = 16, NG = P / G, PER = G / 4;   // ring group (barrier / refill cadence), groups in the ring, DMAs per wave
  constexpr int T_UPS = round_up(NHB * NEB * 4, P), T_DOWNS = round_up(NEB * NHB * 4, P);
  static_assert(P % G == 0 && NG >= 4, "ring: at least 4 groups of 16 fragments");
  static_assert(NEB % 2 == 0 && NHB % 2 == 0 && NZV % 2 == 0, "output blocks are processed in pairs");
  static_assert(T_UPS == SL.T_UP && T_DOWNS == SL.T_DOWN, "same section sizes as the fp32 kernel's stream");

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const long tile = (long)blockIdx.x * 4 + wave;
  long row = tile * 32 + j;      // every wave runs (barriers); rows past R are clamped and never stored
  const bool valid = row < a.R;
  if (!valid) row = a.R - 1;
#ifdef QINCO_TIMELINE
  auto stamp = [&](int i) QINCO_LAMBDA {
    if (a.timeline && lane == 0) a.timeline[tile * 8 + i] = __builtin_readcyclecounter();
  };
#else
  auto stamp = [](int) QINCO_LAMBDA {};
#endif
  stamp(0);
  const long g = row / a.A;
  const int cid = a.cand_ids ? a.cand_ids[row] : (int)(row - g * a.A);
  const float* cptr = a.codebook + (long)cid * D + half * 4;
  const float* xhptr = a.xhat + g * D + half * 4;

  // ---- weight ring ------------------------------------------------------------------------------------
  __shared__ f32x4 lds_ring[P * 64];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  SplitRing<P> ring;
  ring.init(a.wstream, lds_ring, lane, wave_u);
  static_assert(PER <= 4, "a step carries at most four DMAs");

  const float zs = a.smul[0], zsi = a.smul[1];

  // ---- z': blocks < NZV in registers, the others in this wave's LDS park (lane-linear float4 quarters: conflict free) ----
  __shared__ f32x4 zpark[NPARK > 0 ? 4 * NPARK * 4 * 64 : 1];
  const unsigned zp_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)zpark + (wave_u * NPARK * 4 * 64 + lane) * 16;
  f32x16 z[NZV];
  auto zget = [&]<int ob>() QINCO_LAMBDA -> f32x16 {
    if constexpr (ob < NZV) {
      return z[ob];
    } else {
      f32x4 t[4];
      static_for<4>([&]<int q>() QINCO_LAMBDA { lds_read128<((ob - NZV) * 4 + q) * 1024>(t[q], zp_addr); });
      lds_arrived4(t[0], t[1], t[2], t[3]);
      f32x16 v;
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        static_for<4>([&]<int e>() QINCO_LAMBDA { v[4 * q + e] = t[q][e]; });
      });
      return v;
    }
  };
  auto zset = [&]<int ob>(const f32x16& v) QINCO_LAMBDA {
    if constexpr (ob < NZV) {
      z[ob] = v;
      pin_v(z[ob]);
    } else {
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        const f32x4 t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        lds_write128<((ob - NZV) * 4 + q) * 1024>(zp_addr, t);
      });
    }
  };

  // ---- head: z' = 2^c (T[cid] + U[group]);  FFN block 0's hidden layer is folded: h = relu(P[cid] + Q[group]),
  //      y = split(2^a h).
  // The table rows T[cid], P[cid] are 32 different rows per wave.  Gathered lane by lane in the C/D layout (16 B per lane, 32
  // rows per instruction) they cost the texture addresser ~64 cycles per instruction, shared by the CU's four waves: 96
  // instructions = 24 k cycles at the qinco2-S shape, 36 % of its tile, however many are in flight (measured: batching the
  // loads changed nothing).  So the rows are fetched COALESCED -- a slice of 2 blocks = 256 B of a row per 16 lanes, four rows
  // per instruction -- and transposed through LDS: this wave's quarter of the weight ring, which is idle until the head is
  // done (the ring prologue starts behind it).  The per-group rows U[g], Q[g] are fetched in the same lane mapping (their
  // duplicates within a load coalesce) and added before the transpose. ----
  f32x16 y[NHB];     // up-projection accumulators, then (bit pattern of) SplitBlock
  {
    constexpr int RSQ = 17;                          // staging row stride in float4 (272 B: rows shift by 4 banks)
    static_assert(32 * RSQ <= P * 16, "the staging buffer is a quarter of the ring");
    f32x4* stage = lds_ring + wave_u * (P * 16);
    const float m0 = a.smul[2];
    int rc[8], rg[8];                                // codeword / group of the row that this lane's 16-lane group fetches in load i
    static_for<8>([&]<int i>() QINCO_LAMBDA {
      rc[i] = __shfl(cid, 4 * i + (lane >> 4));
      rg[i] = __shfl((int)g, 4 * i + (lane >> 4));     // (groups = rows / A < 2^31)
    });
    constexpr int NSZ = NEB / 2, NS = NSZ + NHB / 2;   // slices of T, then of P
    struct Slice {
      f32x4 r[8], u[8];   // 8 + 8 coalesced loads: table / per-group rows 4i + (lane >> 4), float4 (lane & 15) of the slice
    };
    constexpr int AHEAD = 1;   // slices in flight beyond the current one (64 registers each)
    Slice sl[AHEAD + 1];
    auto fetch = [&]<int S>(Slice& d) QINCO_LAMBDA {
      constexpr bool Z = S < NSZ;
      constexpr int s = Z ? S : S - NSZ;
      const float* tab = Z ? a.ttab : a.ptab;
      const float* grp = Z ? a.uproj : a.qproj;
      constexpr int stride = Z ? DE : DH;
      static_for<8>([&]<int i>() QINCO_LAMBDA {
        d.r[i] = *reinterpret_cast<const f32x4*>(tab + (long)rc[i] * stride + s * 64 + (lane & 15) * 4);
      });
      static_for<8>([&]<int i>() QINCO_LAMBDA {
        d.u[i] = *reinterpret_cast<const f32x4*>(grp + (long)rg[i] * stride + s * 64 + (lane & 15) * 4);
      });
    };
    auto staged_block = [&]<int b>() QINCO_LAMBDA -> f32x16 {
      f32x16 v;
      static_for<4>([&]<int q>() QINCO_LAMBDA {
        const f32x4 t = stage[j * RSQ + b * 8 + 2 * q + half];
        static_for<4>([&]<int e>() QINCO_LAMBDA { v[4 * q + e] = t[e]; });
      });
      return v;
    };
    static_for<AHEAD>([&]<int S>() QINCO_LAMBDA {
      if constexpr (S < NS) fetch.template operator()<S>(sl[S % (AHEAD + 1)]);
    });
    static_for<NS>([&]<int S>() QINCO_LAMBDA {
      if constexpr (S + AHEAD < NS) fetch.template operator()<S + AHEAD>(sl[(S + AHEAD) % (AHEAD + 1)]);
      Slice& c = sl[S % (AHEAD + 1)];
      static_for<8>([&]<int i>() QINCO_LAMBDA { stage[(4 * i + (lane >> 4)) * RSQ + (lane & 15)] = c.r[i] + c.u[i]; });
      const f32x16 v0 = staged_block.template operator()<0>();
      const f32x16 v1 = staged_block.template operator()<1>();
      if constexpr (S < NSZ) {
        zset.template operator()<2 * S>(v0 * zs);
        zset.template operator()<2 * S + 1>(v1 * zs);
      } else {
        constexpr int ob = 2 * (S - NSZ);
        f32x16 w0 = v0, w1 = v1;
        relu16(w0);
        relu16(w1);
        y[ob] = __builtin_bit_cast(f32x16, split_block(w0 * m0));
        y[ob + 1] = __builtin_bit_cast(f32x16, split_block(w1 * m0));
        pin_a(y[ob]);
        pin_a(y[ob + 1]);
      }
    });
    // every wave is done with its staging buffer before anybody's DMA lands in the ring
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ring.prologue();
  }

  f32x4 cur[4], nxt[4];
  stamp(1);   // head operands assembled
  ring.template ldquad<0>(cur);
  lds_arrived4(cur[0], cur[1], cur[2], cur[3]);
  stamp(2);   // first quad of the stream in registers

  // One step = one quad = hi(o), lo(o), hi(o+1), lo(o+1) for one K-chunk: the six MFMAs of the current quad (the two chains
  // interleaved) with the next quad's ring traffic -- group boundary, one DMA and one LDS read per gap -- spread between them,
  // plus `extra`: VALU work of a neighbouring stage that the MFMAs do not depend on; then the next quad must have arrived.
  // Everything is pinned by scheduling barriers: left alone, hipcc hoists the MFMAs above the reads (the wait then exposes
  // the LDS latency in every step), and a wave that issues its four reads and four DMAs in one go leaves the pipe idle.
  auto step = [&]<int TN>(f32x16& t0, f32x16& t1, const f16x8& bh, const f16x8& bl, auto&& extra) QINCO_LAMBDA {
    ring.template step<TN>(cur, nxt, t0, t1, bh, bl, extra);
  };
  // the rest of a section (padding up to a multiple of P): read and drop, the ring protocol keeps running
  auto end_section = [&]<int FROM, int TO>() QINCO_LAMBDA { ring.template end_section<FROM, TO>(cur, nxt); };
  // accumulators of hidden block ob -> B operands of the down-projection, in place
  auto convert_y = [&]<int ob>(float mup) QINCO_LAMBDA {
    f32x16 v = y[ob];
    relu16(v);
    y[ob] = __builtin_bit_cast(f32x16, split_block(v * mup));
    pin_a(y[ob]);
  };

  // ---- down-projection: z' += 2^(c-a-b) W_down' . h'   (output-pair outer) -------------------------------------------
  // LAZY (after an up-projection): only y[0] is converted on entry; block ib+1's conversion rides on the first pair's step
  // (ib, 0).  The residual add of pair og-1 rides on pair og's first step (two alternating accumulator pairs), so that only the
  // last chain end of the phase drains the matrix pipe.
  auto down_phase = [&]<bool LAZY>(float mdn, float mup) QINCO_LAMBDA {
    f32x16 t[2][2];
    auto residual = [&]<int og>() QINCO_LAMBDA {
      zset.template operator()<2 * og>(zget.template operator()<2 * og>() + t[og & 1][0] * mdn);
      zset.template operator()<2 * og + 1>(zget.template operator()<2 * og + 1>() + t[og & 1][1] * mdn);
    };
    static_for<NEB / 2>([&]<int og>() QINCO_LAMBDA {
      t[og & 1][0] = zero16();
      t[og & 1][1] = zero16();
      static_for<NHB>([&]<int ib>() QINCO_LAMBDA {
        static_for<2>([&]<int c>() QINCO_LAMBDA {
          constexpr int T = ((og * NHB + ib) * 2 + c) * 4;
          const SplitBlock s = __builtin_bit_cast(SplitBlock, y[ib]);
          step.template operator()<T + 4>(t[og & 1][0], t[og & 1][1], s.h[c], s.l[c], [&]() QINCO_LAMBDA {
            constexpr int nb = ib + 1 < NHB ? ib + 1 : 0;
            if constexpr (LAZY && og == 0 && c == 0 && ib + 1 < NHB) convert_y.template operator()<nb>(mup);
            constexpr int pg = og > 0 ? og - 1 : 0;
            if constexpr (og > 0 && ib == 0 && c == 0) residual.template operator()<pg>();
          });
        });
      });
    });
    residual.template operator()<NEB / 2 - 1>();
    end_section.template operator()<NEB * NHB * 4, T_DOWNS>();
  };
  // ---- up-projection: y = W_up' . z'   (K-outer: every chain advances by one K-chunk per pass; the split of input block
  //      ib+1 rides on block ib's first steps).  The conversion of y belongs to the down-projection that follows. ----
  auto up_phase = [&](float mup) QINCO_LAMBDA {
    static_for<NHB>([&]<int ob>() QINCO_LAMBDA { y[ob] = zero16(); });
    SplitBlock sb[2];
    sb[0] = split_block(zget.template operator()<0>());
    static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
      static_for<2>([&]<int c>() QINCO_LAMBDA {
        static_for<NHB / 2>([&]<int op>() QINCO_LAMBDA {
          constexpr int T = ((ib * 2 + c) * (NHB / 2) + op) * 4;
          step.template operator()<T + 4>(y[2 * op], y[2 * op + 1], sb[ib & 1].h[c], sb[ib & 1].l[c], [&]() QINCO_LAMBDA {
            constexpr int nb = ib + 1 < NEB ? ib + 1 : 0;
            if constexpr (c == 0 && op == 0 && ib + 1 < NEB) sb[nb & 1] = split_block(zget.template operator()<nb>());
          });
        });
      });
    });
    convert_y.template operator()<0>(mup);
    end_section.template operator()<NHB * NEB * 4, T_UPS>();
  };

  down_phase.template operator()<false>(a.smul[3], 0.f);   // block 0 (its up-projection is folded into P, Q)
  stamp(6);
#pragma unroll 1
  for (int l = 1; l < a.L; ++l) {
    const float mup = a.smul[2 + 2 * l];
    up_phase(mup);
    down_phase.template operator()<true>(a.smul[3 + 2 * l], mup);
  }

  stamp(3);   // FFN blocks done
  // Underflow statistic (qinco_split_stats): how many of the activations' fp16 lo parts are subnormal, i.e. have lost bits.
  // fp16 overflow raises the error flag; underflow is silent by nature, so every 64th workgroup counts it on the final z'
  // (the register-resident blocks) -- a model whose activations sit far below the scalings' design range shows up here.
  if (a.stats && (blockIdx.x & 63) == 0) {
    unsigned nsub = 0, nall = 0;
    static_for<NZV>([&]<int ob>() QINCO_LAMBDA {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = z[ob][i];
        const _Float16 hi = (_Float16)v;
        const float rem = v - (float)hi;                       // what the lo part has to carry
        const float lo = (float)(_Float16)rem;
        nall += (v != 0.f);
        nsub += (rem != 0.f && __builtin_fabsf(lo) < 6.103515625e-05f);   // subnormal, or flushed to zero altogether
      }
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nsub += __shfl_xor(nsub, o);
      nall += __shfl_xor(nall, o);
    }
    if (lane == 0) {
      atomicAdd(a.stats, (unsigned long long)nall);
      atomicAdd(a.stats + 1, (unsigned long long)nsub);
    }
  }
  // ---- tail: out_proj + epilogue: cand = (out + coeff*c) + xhat ; dist = |x|^2 + |cand|^2 - 2 x.cand  (mlp_kernel.hpp E) ----
  const long n = g / a.F;
  const float* xptr = a.x ? a.x + n * D + half * 4 : nullptr;
  float* outp = a.cand_out + row * D + half * 4;
  float s2 = 0.f, sx = 0.f, xn = 0.f;
  struct Epi {
    f32x16 cblk, xhb, xb;
  };
  auto load_epi = [&]<int ob>(Epi& e) QINCO_LAMBDA {
    if (a.add_c) e.cblk = load_block(cptr + ob * 32);
    e.xhb = load_block(xhptr + ob * 32);
    if (xptr) e.xb = load_block(xptr + ob * 32);
  };
  constexpr int NPRE = NDB <= 4 ? NDB : (SPLIT_OUT && OG <= 4 ? OG : 0);   // blocks whose operands are fetched together
  auto epilogue = [&]<int ob>(f32x16 o, const Epi& e) QINCO_LAMBDA {
    if (a.add_c) o = o + e.cblk;
    o = o + e.xhb;
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 t = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
        *reinterpret_cast<f32x4*>(outp + ob * 32 + 8 * q) = t;
      }
    }
    if (xptr) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s2 = fmaf(o[i], o[i], s2);
        sx = fmaf(o[i], e.xb[i], sx);
        xn = fmaf(e.xb[i], e.xb[i], xn);
      }
    } else {   // decode: no distance, but the overflow check below still needs to see a non-finite output
#pragma unroll
      for (int i = 0; i < 16; ++i) s2 = fmaf(o[i], o[i], s2);
    }
  };
  if constexpr (SPLIT_OUT) {
    // out_proj in the split form too (it is 5 % of the MFMA cycles at D = 128 but 26 % at D = 768 once the blocks run on the fp16
    // pipe): K-outer like the up-projection, OG output blocks per pass in y's registers, epilogue operands one block ahead.
    const float mout = a.smul[2 + 2 * a.L];
    static_for<NDB / OG>([&]<int ps>() QINCO_LAMBDA {
      static_for<OG>([&]<int ob>() QINCO_LAMBDA { y[ob] = zero16(); });
      SplitBlock sb[2];
      sb[0] = split_block(zget.template operator()<0>());
      static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
        static_for<2>([&]<int c>() QINCO_LAMBDA {
          static_for<OG / 2>([&]<int op>() QINCO_LAMBDA {
            constexpr int T = ps * NEB * OG * 4 + ((ib * 2 + c) * (OG / 2) + op) * 4;
            step.template operator()<T + 4>(y[2 * op], y[2 * op + 1], sb[ib & 1].h[c], sb[ib & 1].l[c], [&]() QINCO_LAMBDA {
              constexpr int nb = ib + 1 < NEB ? ib + 1 : 0;
            if constexpr (c == 0 && op == 0 && ib + 1 < NEB) sb[nb & 1] = split_block(zget.template operator()<nb>());
            });
          });
        });
      });
      if constexpr (NPRE > 0) {
        Epi e[OG];
        static_for<OG>([&]<int ob>() QINCO_LAMBDA { load_epi.template operator()<ps * OG + ob>(e[ob]); });
        static_for<OG>([&]<int ob>() QINCO_LAMBDA { epilogue.template operator()<ps * OG + ob>(y[ob] * mout, e[ob]); });
      } else {
        Epi e[2];
        load_epi.template operator()<ps * OG>(e[0]);
        static_for<OG>([&]<int ob>() QINCO_LAMBDA {
          if constexpr (ob + 1 < OG) load_epi.template operator()<ps * OG + ob + 1>(e[(ob + 1) & 1]);
          epilogue.template operator()<ps * OG + ob>(y[ob] * mout, e[ob & 1]);
        });
      }
    });
  } else {
    Epi ep[NPRE > 0 ? NPRE : 1];
    if constexpr (NPRE > 0) static_for<NPRE>([&]<int ob>() QINCO_LAMBDA { load_epi.template operator()<ob>(ep[ob]); });
    static_for<NDB>([&]<int ob>() QINCO_LAMBDA {
      Epi e;
      if constexpr (NPRE > 0) e = ep[ob]; else load_epi.template operator()<ob>(e);
      f32x16 o;
      if constexpr (PROJ) {   // fp32 (odd number of output blocks: D = 96, the 32-d test models)
        o = zero16();
        static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
          constexpr int T = (ob * NEB + ib) * 4;     // fragments q = 0..3 of block pair (ob, ib): fp32 A operands, 4 MFMAs each
          const f32x16 zb = zget.template operator()<ib>();
          ring.template ldquad<T + 4>(nxt);
          __builtin_amdgcn_sched_barrier(0);
          static_for<4>([&]<int q>() QINCO_LAMBDA {
            static_for<4>([&]<int e4>() QINCO_LAMBDA { o = QINCO_MFMA(cur[q][e4], zb[4 * q + e4], o); });
          });
          __builtin_amdgcn_sched_barrier(0);
          lds_arrived4(nxt[0], nxt[1], nxt[2], nxt[3]);
          static_for<4>([&]<int q>() QINCO_LAMBDA { cur[q] = nxt[q]; });
        });
        o = o * zsi;
      } else {
        o = zget.template operator()<ob>() * zsi;
      }
      epilogue.template operator()<ob>(o, e);
    });
  }
  if (a.dist_out) {
    s2 += __shfl_xor(s2, 32);
    sx += __shfl_xor(sx, 32);
    xn += __shfl_xor(xn, 32);
    if (valid && half == 0) a.dist_out[row] = (xn + s2) - 2.f * sx;
    // an fp32 value beyond the fp16 range (|z'| > 65504) became inf in the split and NaN in the products: say so instead of
    // letting the selection sort a NaN (qinco_check / the host entry points report it; the fp32 path has no such limit)
    if (a.err && !(s2 < 3.0e38f)) *a.err = 2;
  } else if (a.err && !(s2 < 3.0e38f)) {   // decode: this lane's part of |out|^2 is inf / NaN
    *a.err = 2;
  }
  stamp(4);
  // no LDS-DMA may be in flight when the wave ends (its LDS could be handed to the next workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(5);
}

// ------------------------------------------------------------------------------------------------------------------------
// xproj in the split form: U[g] = W_cat[:, De:] . xhat_g and Q[g] = W_up[0] . U[g] for every (vector, beam) group, the
// per-group half of the folded head (mlp_kernel.hpp xproj_kernel is the fp32 form: 5-14 % of a split-form encode).  Same tile
// and ring protocol as mlp_split_kernel: a wave owns 32 groups, both GEMMs K-outer (all output chains advance together), xhat
// blocks fetched and split one block ahead, U kept in registers between the two GEMMs; U and Q are stored as fp32.
// Stream: [U section: for ib < D/32: for c: for o < De/32: hi, lo][Q section: for ib < De/32: for c: for o < Dh/32: hi, lo].
// smul = [2^cx, 1 / (2^cx s_u), 2^cu, 1 / (2^cu s_q)].
// ------------------------------------------------------------------------------------------------------------------------
template <int D, int DE, int DH, int P>
__global__ void __launch_bounds__(256, 1) xproj_split_kernel(XprojArgs a) {
  constexpr int NDB = D / 32, NEB = DE / 32, NHB = DH / 32;
  constexpr int T_U = round_up(4 * NDB * NEB, P);   // (the Q section, round_up(4 NEB NHB, P) fragments, ends the stream)
  static_assert(NEB % 2 == 0 && NHB % 2 == 0, "output blocks are processed in pairs");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int j = lane & 31, half = lane >> 5;
  long g = ((long)blockIdx.x * 4 + wave) * 32 + j;   // every wave runs (barriers); groups past G are clamped, never stored
  const bool valid = g < a.G;
  if (!valid) g = a.G - 1;
  const float* xp = a.xhat + g * D + half * 4;
  const float xs = a.smul[0], mu = a.smul[1], us = a.smul[2], mq = a.smul[3];

  __shared__ f32x4 lds_ring[P * 64];
  SplitRing<P> ring;
  ring.init(a.wx, lds_ring, lane, wave_u);
  ring.prologue();
  f32x4 cur[4], nxt[4];
  SplitBlock sb[2];
  sb[0] = split_block(load_block(xp) * xs);
  ring.template ldquad<0>(cur);
  lds_arrived4(cur[0], cur[1], cur[2], cur[3]);

  auto store_block = [&](float* p, const f32x16& v) QINCO_LAMBDA {
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4*>(p + 8 * q) = t;
      }
    }
  };

  // ---- U = W_cat[:, De:] . xhat ----
  f32x16 u[NEB];
  static_for<NEB>([&]<int ob>() QINCO_LAMBDA { u[ob] = zero16(); });
  static_for<NDB>([&]<int ib>() QINCO_LAMBDA {
    static_for<2>([&]<int c>() QINCO_LAMBDA {
      static_for<NEB / 2>([&]<int op>() QINCO_LAMBDA {
        constexpr int T = ((ib * 2 + c) * (NEB / 2) + op) * 4;
        constexpr int nb = ib + 1 < NDB ? ib + 1 : 0;
        ring.template step<T + 4>(cur, nxt, u[2 * op], u[2 * op + 1], sb[ib & 1].h[c], sb[ib & 1].l[c], [&]() QINCO_LAMBDA {
          if constexpr (c == 0 && op == 0 && ib + 1 < NDB) sb[nb & 1] = split_block(load_block(xp + nb * 32) * xs);
        });
      });
    });
  });
  ring.template end_section<4 * NDB * NEB, T_U>(cur, nxt);
  float* up = a.uproj + g * DE + half * 4;
  static_for<NEB>([&]<int ob>() QINCO_LAMBDA {
    u[ob] = u[ob] * mu;
    store_block(up + ob * 32, u[ob]);
  });

  // ---- Q = W_up[0] . U ----
  f32x16 qa[NHB];
  static_for<NHB>([&]<int ob>() QINCO_LAMBDA { qa[ob] = zero16(); });
  sb[0] = split_block(u[0] * us);
  static_for<NEB>([&]<int ib>() QINCO_LAMBDA {
    static_for<2>([&]<int c>() QINCO_LAMBDA {
      static_for<NHB / 2>([&]<int op>() QINCO_LAMBDA {
        constexpr int T = ((ib * 2 + c) * (NHB / 2) + op) * 4;
        constexpr int nb = ib + 1 < NEB ? ib + 1 : 0;
        ring.template step<T + 4>(cur, nxt, qa[2 * op], qa[2 * op + 1], sb[ib & 1].h[c], sb[ib & 1].l[c], [&]() QINCO_LAMBDA {
          if constexpr (c == 0 && op == 0 && ib + 1 < NEB) sb[nb & 1] = split_block(u[nb] * us);
        });
      });
    });
  });
  float* qp = a.qproj + g * DH + half * 4;
  static_for<NHB>([&]<int ob>() QINCO_LAMBDA { store_block(qp + ob * 32, qa[ob] * mq); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the wave ends
}

}  // namespace qinco
