// Small-launch form of the fused QINCo codeword MLP for gfx950 (MI355X / CDNA4).
//
// mlp_kernel.hpp gives a wave 32 rows and ALL features: 128 rows per workgroup, so a launch below 256 CUs x 128 rows leaves
// CUs idle -- and that is every decode call the reference makes (compute_MSE decodes cfg.batch = 1024 rows per call,
// qinco_tasks.py:112-125; the search re-rank decodes cfg.search.batch_size = 12 288, search_tasks.py:475-486) and every greedy
// encode step at the reference's batch (1024 vectors x A = 16 candidates).  This form turns the split around:
//
//  * a workgroup owns 16 * NT rows (NT = 1..4 tiles of 16) and its four waves split the OUTPUT features of every GEMM
//    (wave w owns the 16-feature output blocks 4 j + w); the host picks NT so that the launch has ~one workgroup per CU;
//  * a layer's input must then be seen by all four waves: activations live in LDS between the GEMMs, in the B-operand layout of
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
constexpr SmallPlan small_plan(int D, int DE, int DH, int NT, bool fold2, bool dec) {
  const SmallDims S = small_dims(D, DE, DH, fold2);
  SmallPlan p{};
  int actb = S.NEB > S.NHB ? S.NEB : S.NHB;
  if (dec && S.NDB > actb) actb = S.NDB;
  int nobmax = S.NEW > S.NHW ? S.NEW : S.NHW;
  if (S.PROJ && S.NDW > nobmax) nobmax = S.NDW;
  p.ACTB = actb;
  p.LAG = nobmax + 1;
  const long slot = (long)actb * NT * 1024;
  const long ct = dec ? 0 : (long)16 * NT * (D + 4) * 4;   // candidate tile of the encode epilogue (aliases the activations)
  const long avail = 160 * 1024;
  const int pw_min = p.LAG + 7, pw_max = 32;
  for (int db = 1; db >= 0; --db) {
    long act = (db ? 2 : 1) * slot;
    if (ct > act) act = ct;
    long pw = (avail - act) / 4096;
    if (pw > pw_max) pw = pw_max;
    if (pw - p.LAG - 1 > 63) pw = 64 + p.LAG;
    if (pw >= pw_min) {
      p.ok = true;
      p.DB = db != 0;
      p.PW = (int)pw;
      p.lds_bytes = (unsigned)(act + pw * 4096);
      return p;
    }
  }
  return p;
}

template <int D, int DE, int DH, int NT, bool FOLD2, bool DEC>
__global__ void __launch_bounds__(256, 1) mlp_small_kernel(SmallArgs a) {
  constexpr SmallDims S = small_dims(D, DE, DH, FOLD2);
  constexpr SmallPlan PL = small_plan(D, DE, DH, NT, FOLD2, DEC);
  static_assert(PL.ok, "no LDS plan for this instantiation");
  constexpr int NDB = S.NDB, NEB = S.NEB, NHB = S.NHB, NDW = S.NDW, NEW = S.NEW, NHW = S.NHW;
  constexpr bool PROJ = S.PROJ;
  constexpr int PW = PL.PW, LAG = PL.LAG;
  constexpr bool DB = PL.DB;
  constexpr int SLOT4 = PL.ACTB * NT * 64;   // f32x4 per activation buffer
  static_assert(PROJ || NDW == NEW, "identity projections: De == D");

  extern __shared__ __attribute__((aligned(16))) f32x4 lds_small[];
  const int lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  const int foff = (kg >> 1) + 4 * (kg & 1);   // small_feat(r, kg) = foff + {0, 2, 8, 10}[r]
  f32x4* const ring = lds_small + wave_u * PW * 64;
  f32x4* const act = lds_small + 4 * PW * 64;

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
  // Fragment f of this wave sits at wstream + (4 f + wave) KiB and goes to ring slot f % PW.  A read of fragment g waits until at most
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
    wsrc += 4 * 64;
  };
#pragma unroll
  for (int i = 0; i < PW - LAG; ++i) dma(i);
  auto ring_read = [&]() QINCO_LAMBDA -> f32x4 {
    wait_vm.template operator()<PW - LAG - 1>();
    const f32x4 v = ring[rd_slot * 64 + lane];
    asm volatile("" ::: "memory");
    dma(wr_slot);
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

  // acc[j][t] = sum over the NIB input blocks (read from `src`) of W[4 j + wave][ib] . in[ib][t]; chains start from zero
  auto gemm = [&]<int NIB, int NOW>(f32x4 (&acc)[NOW][NT], const f32x4* src) QINCO_LAMBDA {
    f32x4 wf[NOW];
    f32x4 bn[NT];
#pragma unroll
    for (int j = 0; j < NOW; ++j) wf[j] = ring_read();
#pragma unroll
    for (int t = 0; t < NT; ++t) bn[t] = src[t * 64 + lane];
#pragma unroll
    for (int j = 0; j < NOW; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[j][t] = zero4;
    auto body = [&]<bool LAST>(int ib) QINCO_LAMBDA {
      f32x4 bc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bc[t] = bn[t];
      if constexpr (!LAST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = src[((ib + 1) * NT + t) * 64 + lane];
      }
      static_for<NOW>([&]<int j>() QINCO_LAMBDA {
        f32x4 wv = wf[j];
        pin4_v(wv);   // this fragment's LDS read -- and, LDS returning a wave's reads in order, every earlier one -- has completed
        if constexpr (!LAST) wf[j] = ring_read();
        static_for<4>([&]<int e>() QINCO_LAMBDA {
          static_for<NT>([&]<int t>() QINCO_LAMBDA { acc[j][t] = QINCO_MFMA16(wv[e], bc[t][e], acc[j][t]); });
        });
      });
    };
#ifdef QINCO_SMALL_ROLLED
#pragma unroll 1
    for (int ib = 0; ib < NIB - 1; ++ib) body.template operator()<false>(ib);
#else
    static_for<NIB - 1>([&]<int ib>() QINCO_LAMBDA { body.template operator()<false>(ib); });
#endif
    body.template operator()<true>(NIB - 1);
  };

  // activation buffers: the GEMM after a publish reads what the publish wrote
  int cur = 0;
  auto publish = [&]<int NOW, int NB>(const f32x4 (&v)[NOW][NT]) QINCO_LAMBDA {
    if constexpr (DB) cur ^= 1;
    else barrier();   // one buffer: every wave has finished the GEMM that read it
    f32x4* dst = act + cur * SLOT4;
    static_for<NOW>([&]<int j>() QINCO_LAMBDA {
      if (4 * j + wave_u < NB) {
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[((4 * j + wave_u) * NT + t) * 64 + lane] = v[j][t];
      }
    });
    barrier();
  };
  auto src_buf = [&]() QINCO_LAMBDA -> const f32x4* { return act + cur * SLOT4; };

  // four features of block b of a table row, in block layout (b past the end: the last block, never used)
  auto gather4 = [&]<int NB>(const float* rowp, int b) QINCO_LAMBDA -> f32x4 {
    const float* p = rowp + 16 * (b < NB ? b : NB - 1) + foff;
    return f32x4{p[0], p[2], p[8], p[10]};
  };

  f32x4 z[NEW][NT];    // this wave's blocks of z
  f32x4 xh[NDW][NT];   // decode: this wave's blocks of xhat, carried from step to step
  int cid[NT];
  long grp[NT];

  auto run_step = [&](const SmallStep& st) QINCO_LAMBDA {
    f32x4 acc_e[NEW][NT];
    f32x4 acc_h[NHW][NT];
    f32x4 y[NHW][NT];
    int l0 = 0;
    if constexpr (DEC) {
      // ---- head in the kernel: U = W_x xhat (chain from zero, xproj_kernel's order), z = T[code] + U ---------------------
      publish.template operator()<NDW, NDB>(xh);
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = gather4.template operator()<NEB>(st.ttab + (long)cid[t] * DE, 4 * j + wave_u);
      });
      gemm.template operator()<NDB, NEW>(acc_e, src_buf());
      if constexpr (FOLD2) publish.template operator()<NEW, NEB>(acc_e);   // U is the input of Q = W_up[0] U
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = z[j][t] + acc_e[j][t];
      });
      if constexpr (FOLD2) {
        // Q = W_up[0] U (chain from zero), y = relu(P[code] + Q)
        static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) y[j][t] = gather4.template operator()<NHB>(st.ptab + (long)cid[t] * DH, 4 * j + wave_u);
        });
        gemm.template operator()<NEB, NHW>(acc_h, src_buf());
        static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            y[j][t] = y[j][t] + acc_h[j][t];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j][t][e] = relu1(y[j][t][e]);
          }
        });
      }
    } else {
      // ---- head from the tables and the per-group projections: z = T[cid] + U[g], y = relu(P[cid] + Q[g]) ------------------
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          z[j][t] = gather4.template operator()<NEB>(st.ttab + (long)cid[t] * DE, 4 * j + wave_u) +
                    gather4.template operator()<NEB>(a.uproj + grp[t] * DE, 4 * j + wave_u);
      });
      if constexpr (FOLD2) {
        static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            y[j][t] = gather4.template operator()<NHB>(st.ptab + (long)cid[t] * DH, 4 * j + wave_u) +
                      gather4.template operator()<NHB>(a.qproj + grp[t] * DH, 4 * j + wave_u);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j][t][e] = relu1(y[j][t][e]);
          }
        });
      }
    }
    if constexpr (FOLD2) {   // block 0: the down-projection of y = relu(P + Q)
      publish.template operator()<NHW, NHB>(y);
      gemm.template operator()<NHB, NEW>(acc_e, src_buf());
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
      gemm.template operator()<NEB, NHW>(acc_h, src_buf());
      static_for<NHW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_h[j][t][e] = relu1(acc_h[j][t][e]);
      });
      publish.template operator()<NHW, NHB>(acc_h);
      gemm.template operator()<NHB, NEW>(acc_e, src_buf());
      static_for<NEW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) z[j][t] = z[j][t] + acc_e[j][t];
      });
    }
  };

  // ---- out_proj + (o + c) + xhat for this wave's blocks of D: o[j][t] --------------------------------------------------------
  auto out_blocks = [&](const SmallStep& st, f32x4 (&o)[NDW][NT], const f32x4 (&xprev)[NDW][NT]) QINCO_LAMBDA {
    f32x4 cw[NDW][NT];
    if (a.add_c) {
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) cw[j][t] = gather4.template operator()<NDB>(st.codebook + (long)cid[t] * D, 4 * j + wave_u);
      });
    }
    if constexpr (PROJ) {
      publish.template operator()<NEW, NEB>(z);
      gemm.template operator()<NEB, NDW>(o, src_buf());
    } else {
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) o[j][t] = z[j][t];
      });
    }
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (a.add_c) o[j][t] = o[j][t] + cw[j][t];
        o[j][t] = o[j][t] + xprev[j][t];
      }
    });
  };

  if constexpr (DEC) {
    // ---- QINCoInferenceDecoder.forward: xhat = cw[0]; xhat += f_m(cw[m], xhat) ---------------------------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) cid[t] = a.codes_t[row[t]];
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) xh[j][t] = gather4.template operator()<NDB>(a.codebook0 + (long)cid[t] * D, 4 * j + wave_u);
    });
    int cid_next[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cid_next[t] = a.codes_t[(long)a.m_first * a.R + row[t]];
#pragma unroll 1
    for (int s = 0; s < a.m_count; ++s) {
      const int m = a.m_first + s;
      const SmallStep st = a.steps[m];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        cid[t] = cid_next[t];
        cid_next[t] = a.codes_t[(long)(s + 1 < a.m_count ? m + 1 : m) * a.R + row[t]];   // one step ahead of its use
      }
      run_step(st);
      f32x4 o[NDW][NT];
      out_blocks(st, o, xh);
      static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
        for (int t = 0; t < NT; ++t) xh[j][t] = o[j][t];
      });
    }
    // x = xhat * std + mean (two roundings, denormalize_kernel), the model's own D columns
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
      const int b = 4 * j + wave_u;
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
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      grp[t] = row[t] / a.A;
      cid[t] = a.cand_ids ? a.cand_ids[row[t]] : (int)(row[t] - grp[t] * a.A);
    }
    run_step(st);
    f32x4 xprev[NDW][NT];
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
#pragma unroll
      for (int t = 0; t < NT; ++t) xprev[j][t] = gather4.template operator()<NDB>(a.xhat + grp[t] * D, 4 * j + wave_u);
    });
    f32x4 o[NDW][NT];
    out_blocks(st, o, xprev);
    // the candidate tile in natural layout (rows D + 4 floats apart), over the activation buffers
    constexpr int CS = D + 4;
    float* ct = reinterpret_cast<float*>(act);
    barrier();   // every wave has finished its last GEMM's reads
    static_for<NDW>([&]<int j>() QINCO_LAMBDA {
      const int b = 4 * j + wave_u;
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
    if (a.dist_out && tid < 32 * NT) {
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
    for (int i = tid; i < 16 * NT * (D / 4); i += 256) {
      const int rl = i / (D / 4), c4 = i - rl * (D / 4);
      if (base + rl < a.R)
        *reinterpret_cast<f32x4*>(a.cand_out + (base + rl) * D + 4 * c4) = *reinterpret_cast<const f32x4*>(ct + rl * CS + 4 * c4);
    }
  }
  // no LDS-DMA may be in flight when the wave ends (its LDS could be handed to the next workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace qinco
