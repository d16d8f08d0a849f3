// Argument block and weight-stream geometry shared by the fused-MLP kernel (device) and the packer (host).
#pragma once
#include <hip/hip_runtime.h>

#if defined(QINCO_TIMELINE) && !defined(QINCO_EXPERIMENT)
#error "QINCO_TIMELINE (per-wave cycle stamps) belongs to experiment builds: scripts/build_exp_lib.py adds -DQINCO_EXPERIMENT"
#endif

namespace qinco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Bumped by hand whenever the MEANING of an argument block or of a launcher changes without changing its size (instance_abi).
constexpr int kInstanceAbiVersion = 4;
// Launchers a kernel-instance module hands over in qinco_instance_info's fns[] (mlp, xproj, table, IVF, small-launch form or null).
// Part of instance_abi(): a module exporting another number of launchers is refused at load.
constexpr int kInstanceNFns = 5;

// Fragment (1 KiB = 64 lanes x float4) counts of each section of the packed per-step weight stream.
// Every section is padded to a multiple of the ring depth P so ring slots are compile-time constants.
struct StreamDims {
  int NDB, NEB, NHB;
  bool PROJ;
  int T_IN, T_BIAS, T_CAT, T_UP, T_DOWN, T_OUT;
  bool FOLD2;   // the first FFN block's up-projection is not in the stream either (mlp_kernel.hpp FOLD2)
  constexpr long total(int L) const {
    return (long)T_IN + T_BIAS + T_CAT + (long)L * (T_UP + T_DOWN) - ((FOLD2 && L > 0) ? T_UP : 0) + T_OUT;
  }
};

// fold = true: in_proj / bias / concat are not in the stream (the kernel starts from z = T[cid] + U[group], see
// mlp_kernel.hpp FOLD).  tile = 32: mlp_kernel (32-feature blocks, 4 fragments per block pair); tile = 16:
// mlp16_kernel (16-feature blocks, 1 fragment per block pair, folded since round 3: FOLD only, no FOLD2).
constexpr StreamDims stream_dims(int D, int DE, int DH, int P, bool fold = false, bool fold2 = false, int tile = 32) {
  StreamDims s{};
  const int fpp = tile == 32 ? 4 : 1;  // fragments (256 weights) per pair of blocks
  s.FOLD2 = fold2;
  s.NDB = D / tile;
  s.NEB = DE / tile;
  s.NHB = DH / tile;
  s.PROJ = (D != DE);
  s.T_IN = (s.PROJ && !fold) ? round_up(s.NEB * s.NDB * fpp, P) : 0;
  s.T_BIAS = fold ? 0 : round_up(s.NEB * fpp, P);
  s.T_CAT = fold ? 0 : round_up(s.NEB * (s.NEB + s.NDB) * fpp, P);
  s.T_UP = round_up(s.NHB * s.NEB * fpp, P);
  s.T_DOWN = round_up(s.NEB * s.NHB * fpp, P);
  s.T_OUT = s.PROJ ? round_up(s.NDB * s.NEB * fpp, P) : 0;
  return s;
}

struct MlpArgs {
  const f32x4* wstream;   // packed weight stream of this step (+P fragments of tail padding)
  int L;                  // number of residual FFN blocks
  const float* codebook;  // (K, D) main codebook of this step
  const int* cand_ids;    // (R) codebook row per MLP row, or nullptr -> id = row % A
  int A;                  // candidates per (vector, beam) group
  int F;                  // beams per vector (groups per x row)
  const float* xhat;      // (R/A, D) current reconstruction per group
  const float* x;         // (R/A/F, D) normalised target, or nullptr (decode: no distances)
  long R;                 // rows
  float* cand_out;        // (R, D): f(c,xhat)+xhat
  float* dist_out;        // (R) or nullptr
  int add_c;              // 1: + c (QINCo2), 0: qinco1_mode (res_codeword_coeff = 0)
  const float* ttab;      // FOLD: (K, De) per-codeword table  T_k = z_k + W_cat[:, :De] z_k + b,  z_k = in_proj(c_k)
  const float* uproj;     // FOLD: (R/A, De) per-group  U_g = W_cat[:, De:] xhat_g   (xproj_kernel)
  const float* ptab;      // FOLD2: (K, Dh)  P_k = W_up[0] T_k
  const float* qproj;     // FOLD2: (R/A, Dh) Q_g = W_up[0] U_g
  int* err;               // split-fp16 form: sticky flag, set to 2 when a candidate distance is not finite (fp16 overflow)
  const float* smul;      // split-fp16 form (mlp_split_kernel.hpp): [2^c, 2^-c, then per FFN block l: m_up[l], m_down[l]]
  unsigned long long* stats;  // split-fp16 form: [elements sampled, elements whose fp16 lo part is subnormal] (every 64th workgroup)
  // SELEP instances (mlp_kernel.hpp): sel_T > 0 = the per-vector top-T of the step is taken in the kernel's epilogue (beam_select_kernel's
  // job: qinco_inference.py:200-222) -- needs the vector's F * A candidates inside one workgroup (128 % (F A) == 0); nothing is
  // written to cand_out / dist_out then, only the T winners: their rows to sel_xhat_out, their code histories to sel_hist_out
  int sel_T, sel_m, sel_M;    // beams kept, step index, codes per history row
  const int* sel_hist_in;     // (N, F, M)
  int* sel_hist_out;          // (N, T, M)
  float* sel_xhat_out;        // (N, T, D)
#ifdef QINCO_TIMELINE     // experiment builds only (scripts/exp_timeline.py): per-wave cycle stamps of the kernel's phases
  unsigned long long* timeline;   // (tiles, 8)
#endif
};

struct XprojArgs {
  const f32x4* wx;        // W_cat[:, De:] (De x D) as fragments in (ob, ib, q) order
  const float* xhat;      // (G, D)
  float* uproj;           // (G, De)
  long G;
  const f32x4* wq;        // FOLD2: W_up[0] (Dh x De) as fragments in (ob, ib, q) order, or nullptr
  float* qproj;           // FOLD2: (G, Dh)
  const float* smul;      // split-fp16 form (xproj_split_kernel): [2^cx, 1/(2^cx s_u), 2^cu, 1/(2^cu s_q)]; wx = the whole stream
  // small launches: the step's pre-selection rides in the same kernel (presel_kernel.hpp); cstream == nullptr: plain xproj
  const float* x;         // (G / F, D) normalised targets
  int F;                  // beams per vector
  const f32x4* cstream;   // pre-selection codebook (K = 256) as MFMA fragments (cb, ib, q)
  const float* cnorm;     // (K) |c_k|^2
  int T;                  // candidates to keep per group
  int* ids_out;           // (G, T)
};
// The pre-selection table + top-T alone (table_kernel.hpp): step 0, and encode steps whose launch is too large for the fused
// small-launch kernel.  Compiled-in for the reference's dataset dimensions; a module built on demand brings its own for its D.
struct TableArgs {
  const float* x;         // (G / F, D) normalised targets
  const float* xhat;      // (G, D) or nullptr (step 0: r = x)
  int F;
  const f32x4* cstream;   // codebook (K = 256) as MFMA fragments (cb, ib, q)
  const float* cnorm;     // (K) |c_k|^2
  long G;
  int T;
  int* ids_out;           // (G, T)
  int coop;               // 1: the cooperative small-launch kernel
};

// The exact fp32 coarse assignment of an IVF model (ivf_kernel.hpp).  Compiled-in for the reference's dataset dimensions (with the
// fp16 filter in front, ivf_f16_kernel.hpp); a module built on demand brings the fp32 kernel for its D.
struct IvfArgs {
  const f32x4* cstream;   // centroids as MFMA fragments (block of 32, feature block, q)
  const float* cnorm;     // (ivf_K) |c_k|^2  (rows added to reach a multiple of 32 carry 1e30: never the minimum)
  int nblocks, blocks_per_slice, slices;
  const float* x;         // (N, D) normalised vectors
  long N;
  unsigned long long* best;   // (N) merged (distance, id) keys
  const int* only_if;     // nullptr, or: do nothing unless *only_if != 0
};

// ---- small-launch form of the fused MLP (mlp_small_kernel.hpp): a workgroup owns 16 * NT rows and its kSmallWaves waves split the
// OUTPUT features of every GEMM; activations meet in LDS between the GEMMs.  16-feature blocks; within a block the features sit in
// the order the 32-row kernel contracts them, so both kernels add the same products in the same order.
// Lane group kg = lane >> 4, register r of a block <-> feature offset small_feat(r, kg) of the block.
constexpr int kSmallWaves = 8;   // two waves per SIMD: one wave's ring / address bookkeeping issues under the other's MFMAs
constexpr int small_feat(int r, int kg) { return 8 * (r >> 1) + 2 * (r & 1) + (kg >> 1) + 4 * (kg & 1); }
// position p (0..15) of the contraction order <-> feature offset (p = 4 r + kg)
constexpr int small_feat_at(int p) { return small_feat(p >> 2, p & 3); }

// Fragments (1 KiB) per wave of each section of a step's small-form stream.  Wave w owns the output blocks NW j + w (NW =
// kSmallWaves); a section is [input block][j] per wave, the waves' fragments interleaved ((f * NW + w) KiB), zero fragments where
// NW j + w is past the end.
struct SmallDims {
  int NDB, NEB, NHB;   // 16-feature blocks
  int NDW, NEW, NHW;   // output blocks per wave (ceil(N / kSmallWaves))
  bool PROJ, FOLD2;
  int F_HX, F_HQ, F_UP, F_DOWN, F_OUT;
  constexpr int head() const { return F_HX + (FOLD2 ? F_HQ : 0); }                       // the in-kernel head (decode)
  constexpr long step(int L) const { return (long)head() + (long)L * (F_UP + F_DOWN) - (FOLD2 ? F_UP : 0) + F_OUT; }
};
constexpr SmallDims small_dims(int D, int DE, int DH, bool fold2) {
  SmallDims s{};
  s.NDB = D / 16;
  s.NEB = DE / 16;
  s.NHB = DH / 16;
  s.NDW = (s.NDB + kSmallWaves - 1) / kSmallWaves;
  s.NEW = (s.NEB + kSmallWaves - 1) / kSmallWaves;
  s.NHW = (s.NHB + kSmallWaves - 1) / kSmallWaves;
  s.PROJ = D != DE;
  s.FOLD2 = fold2;
  s.F_HX = s.NDB * s.NEW;
  s.F_HQ = s.NEB * s.NHW;
  s.F_UP = s.NEB * s.NHW;
  s.F_DOWN = s.NHB * s.NEW;
  s.F_OUT = s.PROJ ? s.NEB * s.NDW : 0;
  return s;
}

struct SmallStep {          // per QINCo step, in device memory; every table in BLOCK LAYOUT: entry 4 kg + r of a 16-block is feature
                            // small_feat(r, kg), so the four registers of a lane (group kg) are one 16-byte load
  const float* ttab;        // (K, De)  T_k = z_k + W_cat[:, :De] z_k + b
  const float* ptab;        // (K, Dh)  P_k = W_up[0] T_k   (FOLD2)
  const float* codebook;    // (K, D)
};

struct SmallArgs {
  const f32x4* wstream;     // small-form stream, positioned at the first fragment this launch consumes
  const SmallStep* steps;   // indexed by step m
  int m_first, m_count;     // steps this launch runs: decode 1 .. M-1, an encode step: one
  int L;
  int add_c;
  long R;                   // rows
  // ---- one encode step: head from the tables and the per-group projections (xproj / presel_xproj kernels)
  const int* cand_ids;      // (R) or nullptr -> row % A
  int A, F;
  const float* xhat;        // (R / A, D)
  const float* x;           // (R / A / F, D)
  const float* uproj;       // (R / A, De)
  const float* qproj;       // (R / A, Dh)
  float* cand_out;          // (R, D)
  float* dist_out;          // (R)
  // ---- decode, every step in one launch: the head is computed in the kernel (same association: T[code] + W_x xhat)
  const int* codes_t;       // (M, R) step-major codes
  const float* codebook0;   // step 0: xhat = codebook0[code]
  float* out;               // (R, Duser) de-normalised reconstruction
  const float* mean;        // nullptr: leave normalised
  float std_;
  int Duser;
#ifdef QINCO_TIMELINE       // experiment builds only (scripts/ubench/small_timeline.hip): cycle stamps of every wave
  unsigned long long* timeline;   // (workgroups, kSmallWaves, 64)
#endif
};

// Source-version check between the library and a module built on demand: the sizes of the argument blocks they exchange.
constexpr int instance_abi() {
  return (int)((sizeof(MlpArgs) << 20) | (sizeof(XprojArgs) << 8) | sizeof(IvfArgs)) ^ (int)((sizeof(TableArgs) << 12) | (sizeof(SmallArgs) << 4)) ^
         (int)(sizeof(SmallStep) << 17) ^ (kInstanceNFns << 23) ^ (kInstanceAbiVersion << 26);
}
static_assert(sizeof(MlpArgs) < 2048 && sizeof(XprojArgs) < 4096 && sizeof(IvfArgs) < 256, "instance_abi packing");

// 1 if the instance's xproj launcher serves XprojArgs::cstream (the fused small-launch kernel exists for the shape)
constexpr bool presel_coop_ok(int DE, int DH, int var) {
  return (DE / 32) % 4 == 0 && (DH / 32) % 4 == 0 && DE <= 384 && (var & 16) && !(var & 128) && !(var & 512);
}

}  // namespace qinco
