// libqinco_hip.so -- C ABI (include/qinco_hip.h) over the gfx950 kernels.
// Host side: weight packing into the kernel's stream order, scratch management, the per-step launch
// sequence of encode (qinco_inference.py:239-254 + :156-224 / :78-140) and decode (:66-75).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "abi_util.hpp"
#include "aux_kernels.hpp"
#include "ivf_f16_kernel.hpp"
#include "ivf_kernel.hpp"
#include "lut_kernel.hpp"
#include "mlp_args.hpp"
#include "mlp_launch.hpp"

using namespace qinco;

// ---------------------------------------------------------------------------------------------
// per-shape launchers (shapes.def)
// ---------------------------------------------------------------------------------------------
#define QINCO_SHAPE(D, DE, DH, P, VAR)                                                                        \
  extern "C" hipError_t qinco_mlp_launch_##D##_##DE##_##DH##_##P##_##VAR(const qinco::MlpArgs*, hipStream_t); \
  extern "C" hipError_t qinco_xproj_launch_##D##_##DE##_##DH##_##P##_##VAR(const qinco::XprojArgs*, hipStream_t);
#include "shapes.def"
#undef QINCO_SHAPE

#define QINCO_SMALL_SHAPE(D, DE, DH, F2) \
  extern "C" hipError_t qinco_small_launch_##D##_##DE##_##DH##_##F2(const qinco::SmallArgs*, int, int, hipStream_t);
#include "small_shapes.def"
#undef QINCO_SMALL_SHAPE

namespace qinco {
static const MlpInstance g_instances[] = {
#define QINCO_SHAPE(d, de, dh, p, var) \
  {d, de, dh, p, var, &qinco_mlp_launch_##d##_##de##_##dh##_##p##_##var, &qinco_xproj_launch_##d##_##de##_##dh##_##p##_##var, nullptr, nullptr, nullptr},
#include "shapes.def"
#undef QINCO_SHAPE
};

// the small-launch form (mlp_small_kernel.hpp) of the compiled-in shapes
struct SmallInstance {
  int D, De, Dh, fold2;
  small_launch_fn fn;
};
static const SmallInstance g_small[] = {
#define QINCO_SMALL_SHAPE(d, de, dh, f2) {d, de, dh, f2, &qinco_small_launch_##d##_##de##_##dh##_##f2},
#include "small_shapes.def"
#undef QINCO_SMALL_SHAPE
};
// ... of an fp32 instance with the folded head (the small form starts from z = T[codeword] + U[group] like it)
static small_launch_fn find_small_launcher(const MlpInstance* i) {
  if (!i || !(i->var & 16) || (i->var & (512 | 128))) return nullptr;   // (not the split form, not the 16-row tile form: other orders)
  if (i->small) return i->small;
  for (const SmallInstance& s : g_small)
    if (s.D == i->D && s.De == i->De && s.Dh == i->Dh && s.fold2 == ((i->var & 32) ? 1 : 0)) return s.fn;
  return nullptr;
}

// Instances built on demand for geometries shapes.def does not list (qinco_load_instance): each lives in a shared object of
// its own -- one translation unit of csrc/mlp_inst.hip with its shape as -D flags -- and is never unloaded.  A deque: element
// addresses handed out to handles stay valid when more instances are loaded.
static std::deque<MlpInstance> g_loaded;

const MlpInstance* find_mlp_instance(int D, int De, int Dh, int want_P, int want_var) {
  const MlpInstance* first = nullptr;
  auto visit = [&](const MlpInstance& i) -> const MlpInstance* {
    if (i.D != D || i.De != De || i.Dh != Dh) return nullptr;
    if (!first) first = &i;
    return (i.P == want_P && i.var == want_var) ? &i : nullptr;
  };
  for (const MlpInstance& i : g_instances)
    if (const MlpInstance* hit = visit(i)) return hit;
  for (const MlpInstance& i : g_loaded)
    if (const MlpInstance* hit = visit(i)) return hit;
  return first;
}

static bool ivf_compiled_in(int D) { return D == 32 || D == 96 || D == 128 || D == 256 || D == 768; }
// the exact coarse-assignment kernel a loaded module brings for its D (any module of that D will do)
static ivf_launch_fn find_ivf_launcher(int D) {
  for (const MlpInstance& i : g_loaded)
    if (i.D == D && i.ivf) return i.ivf;
  return nullptr;
}

// The kernels work on 32-feature blocks.  Any other geometry is zero-padded up to the next multiple of 32 when the weights are
// packed: padding features are exact zeros in the inputs, codebooks and weights, so they add exact zeros to every sum.  A model
// WITH in/out projections (De != D) must keep them after padding (the kernels read "De == D" as "identity projections").
static void padded_geometry(int D, int De, int Dh, int* Dp, int* Dep, int* Dhp) {
  *Dp = round_up(D, 32);
  *Dep = round_up(De, 32);
  *Dhp = round_up(Dh, 32);
  if (De != D && *Dep == *Dp) *Dep += 32;
}
}  // namespace qinco

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int qinco::abi_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
#define fail qinco::abi_fail

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct qinco_handle_s {
  qinco_desc d{};      // the geometry the kernels run (D, De, Dh padded to multiples of 32; L >= 1)
  qinco_desc user{};   // the model's own hyper-parameters: I/O row widths and FLOP accounting
  int device = 0;
  int num_cu = 256;
  int A = 0, B = 1;  // active search widths
  const MlpInstance* inst = nullptr;
  StreamDims sd{};

  float* mean = nullptr;
  float std_ = 1.f;
  std::vector<float*> codebook, sub_codebook, cnorm, sub_cnorm;
  std::vector<f32x4*> cb_stream, sub_stream;   // the same codebooks as MFMA A-operand fragments (table kernels)
  // FOLD: per-codeword head table T (K, De) and the xhat half of the concat weight as fragments, per step
  bool fold = false, fold2 = false;
  std::vector<float*> ttab, ptab;            // T (K, De);  FOLD2: P = W_up[0] T (K, Dh)
  std::vector<float*> ttab_s, ptab_s, cb_s;  // block-layout copies of T, P and the codebooks for the small-launch form (upload_block_layout)
  std::vector<f32x4*> wx_stream, wq_stream;  // W_cat[:, De:] and (FOLD2) W_up[0] as fragments for xproj_kernel
  float* uproj = nullptr;      // (max_batch * B, De [+ Dh]) scratch: U_g = W_cat[:, De:] xhat_g  [then Q_g = W_up[0] U_g]
  float* duproj = nullptr;     // decode counterpart (dec_cap, De [+ Dh])
  std::vector<f32x4*> wstream;
  bool split16 = false;             // split-fp16 FFN blocks (QINCO_FLAG_SPLIT_F16; mlp_split_kernel.hpp)
  std::vector<float*> smul;         // per step: [2^c, 2^-c, m_up[0], m_down[0], m_up[1], ...]
  std::vector<float*> xsmul;        // per step, xproj_split_kernel: [2^cx, 1/(2^cx s_u), 2^cu, 1/(2^cu s_q)]
  // decode runs one row per group: nothing to share, so the folded head only adds the xproj launch and the U / Q round trip
  // through HBM.  When the shape has an un-folded instance, decode uses it with its own (complete) weight stream.
  const MlpInstance* dec_inst = nullptr;
  StreamDims dec_sd{};
  // the encode instance with the per-vector top-T in its epilogue (VAR bit 2048): the KHEAD + SELEP form by default where the shape has it,
  // the twin's on request (QINCO_CREATE_EPILOGUE_SELECT)
  const MlpInstance* sel_inst = nullptr;
  // production instance with KHEAD (VAR bit 4096): its twin without, for launches whose groups the KHEAD kernel does not take (a wave's
  // 32 rows must span at most two groups: A >= 32 or A == 16) and for the epilogue-selection instance; alt_wstream = the same weights
  // with block 0's down-projection ob-outer
  const MlpInstance* alt_inst = nullptr;
  std::vector<f32x4*> alt_wstream;
  std::vector<f32x4*> dec_wstream;
  // small-launch form (mlp_small_kernel.hpp): its weight stream (every step, contiguous), per-step table pointers, largest NT
  small_launch_fn small = nullptr;
  bool want_small = false;   // decided before the per-step tables are built (they get block-layout copies)
  SmallDims ssd{};
  f32x4* small_stream = nullptr;
  SmallStep* small_steps = nullptr;
  int small_max_nt[2] = {0, 0};   // [encode step, decode]
  int* kvals = nullptr;
  int* err_flag = nullptr;
  unsigned long long* split_stats = nullptr;   // split form: [activations sampled, fp16 lo parts subnormal] (mlp_split_kernel.hpp)
  qinco_split_report calib{};                  // split form: result of the create-time calibration against the fp32 instance
  bool ever_overflowed = false;
  // IVF step 0
  int K0 = 0;                       // rows of codebook[0] the model has (ivf_K or K): the range of step-0 codes
  f32x4* ivf_stream = nullptr;      // centroids packed as MFMA A-operand fragments
  qinco::ivf_launch_fn ivf_module = nullptr;   // D outside the compiled-in set: the module's exact fp32 kernel (no fp16 filter)
  unsigned long long* ivf_best = nullptr;  // (max_batch) merged (distance, id) keys
  // fp16-filter passes of the IVF assignment (ivf_f16_kernel.hpp)
  bool table_valu = false;          // QINCO_CREATE_TABLE_VALU: VALU pre-selection table kernel (A/B)
  bool table_coop = true;           // small launches: the cooperative table kernel (QINCO_CREATE_TABLE_NO_COOP clears it)
  bool no_presel_fusion = false;    // QINCO_CREATE_NO_PRESEL_FUSION: pre-selection and xproj as two launches at every size (A/B)
  long table_coop_max = 16384;      // ... up to this many groups
  bool ivf_f16 = false;
  void* ivf_h16 = nullptr;           // centroids as fp16 MFMA fragments
  float* ivf_cnorm_half = nullptr;   // -|c|^2 / 2
  float ivf_cmax = 0.f;
  unsigned* ivf_amin = nullptr;      // (max_batch) approximate minima
  int* ivf_perm = nullptr;           // stream position of a block of 32 centroids -> its index (the sample comes first)
  int ivf_sample_blocks = 0;
  int* ivf_cand = nullptr;           // [count, overflow, pad, pad][vec (cap)][id (cap)]
  int ivf_cand_cap = 0;

  // scratch (sized for d.max_batch, A, B)
  int64_t cap_n = 0;
  int cap_A = -1, cap_B = -1;
  size_t beam_lds_max = 0;   // dynamic LDS already granted to beam_select_kernel
  size_t table_lds_max = 0;  // ... and to dist_topk_kernel
  float* xn = nullptr;
  float* xhat[2] = {nullptr, nullptr};
  int* hist[2] = {nullptr, nullptr};
  int* top_ids = nullptr;
  int* codes_t = nullptr;
  float* cand = nullptr;
  float* dist = nullptr;
  // decode scratch: its own (larger) chunk, a decode row costs only 2 D floats + M ints
  int64_t dec_cap = 0;
  float* dxhat[2] = {nullptr, nullptr};

  // host-path staging: a two-deep pipeline of pinned host buffers and device buffers (HostPipe below)
  struct HostPipe* pipe = nullptr;

  // Every call on a handle works in the SAME scratch (xhat, hist, cand, dist, codes_t, dxhat, uproj, err_flag): calls are ordered
  // by the stream they are given, and a call on ANOTHER stream than the previous one (torch's current stream, then the host
  // pipeline's private non-blocking stream; two user streams) first waits for that one's work (scratch_enter / scratch_leave).
  hipEvent_t scratch_done = nullptr;
  hipStream_t scratch_stream = nullptr;
  bool scratch_used = false;

  // profiling
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double prof_flops = 0.0;        // algorithmic (rows x R_mlp)
  double prof_flops_exec = 0.0;   // what the matrix pipe executes for them (the folded head is tables + one GEMM per group)

  std::vector<void*> owned;  // every device allocation, for destroy
#ifdef QINCO_TIMELINE
  unsigned long long* tl = nullptr;
  size_t tl_cap = 0, tl_tiles = 0;
#endif
};

// the MFMA table kernel is instantiated here for K = 256 and the D of the compiled-in MLP / IVF instances; a kernel-instance module
// built on demand brings one for its own D (MlpInstance::table)
static bool builtin_table_dim(int D) { return D == 32 || D == 96 || D == 128 || D == 256 || D == 768; }
static bool mfma_table_ok(const qinco_desc& d, const MlpInstance* inst) {
  return d.K == 256 && (builtin_table_dim(d.D) || (inst && inst->table));
}

// candidates pre-selected at step m (QincoSubstep._n_codes, qinco_base.py:108-112)
static int n_codes(const qinco_handle_s* h, int m) {
  if (h->A > 0 && h->d.ivf_K > 0 && m == 1) return h->A > h->B ? h->A : h->B;
  return h->A;
}

static double mlp_flops_per_row(const qinco_handle_s* h) {
  // R_mlp = [De != D] 4 D De + 2 (De + D) De + 4 L De Dh
  const qinco_desc& d = h->user;
  double f = 2.0 * (d.De + d.D) * d.De + 4.0 * d.L * (double)d.De * d.Dh;
  if (d.De != d.D) f += 4.0 * d.D * d.De;
  return f;
}

// FLOPs the matrix pipe EXECUTES for a fused-MLP launch of R rows in G groups (model dimensions, not the padded ones):
//   un-folded instance (decode twin):      the algorithmic count, R x R_mlp;
//   FOLD:   R x (4 L De Dh + [De != D] 2 De D)  +  G x 2 D De                       (U = W_x xhat once per group);
//   FOLD2:  R x ((4 L - 2) De Dh + [De != D] 2 De D)  +  G x (2 D De + 2 De Dh)     (Q = W_up[0] U too);
//   decode in one launch of the small form: every row is its own group, U and Q are computed per row.
static double mlp_flops_executed(const qinco_handle_s* h, double R, double G, bool folded, bool fold2, bool khead = false, bool khead_epilogue = false) {
  const qinco_desc& d = h->user;
  const double L = h->d.L;   // (a model without FFN blocks runs one all-zero block)
  if (!folded) return R * (2.0 * (d.De + d.D) * d.De + 4.0 * L * d.De * d.Dh + (d.De != d.D ? 4.0 * d.D * d.De : 0.0));
  double per_row = 4.0 * L * d.De * d.Dh + (d.De != d.D ? 2.0 * d.De * d.D : 0.0);
  double per_group = 2.0 * d.D * d.De;
  if (fold2) {
    per_row -= 2.0 * d.De * d.Dh;
    per_group += 2.0 * d.De * d.Dh;
  }
  if (khead && R > 0) {
    // KHEAD: the per-group rows of the head are added by one 32x32x2 MFMA per 32-feature block and pair of groups of a wave's
    // 32 rows (2 * 2 FLOPs per row and feature; padded dimensions: this is what the matrix pipe executes)
    const double rows_per_group = R / (G > 0 ? G : 1);
    int ng = rows_per_group >= 32 ? 1 : (int)((32 + rows_per_group - 1) / rows_per_group);
    if (rows_per_group < 32 && ((int)rows_per_group == 0 || 32 % (int)rows_per_group != 0)) ng += 1;
    per_row += 4.0 * (h->d.De + h->d.Dh) * ((ng + 1) / 2);
    // ... and, without projections, its epilogue adds xhat and lays x out the same way: 2 x one MFMA per block (encode: x is given)
    if (khead_epilogue && h->d.De == h->d.D) per_row += 8.0 * h->d.D;   // (not with the selection in the epilogue: that one gathers its rows)
  }
  return R * per_row + G * per_group;
}

template <class T>
static int dev_alloc(qinco_handle_s* h, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
  h->owned.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

static void dev_free(qinco_handle_s* h, void* p) {
  if (!p) return;
  for (auto& q : h->owned)
    if (q == p) { q = nullptr; break; }
  (void)hipFree(p);
}

// ---------------------------------------------------------------------------------------------
// Host-pointer entry points: a two-deep pipeline.  The reference's loop is H2D -> encode -> D2H per batch (search_tasks.py:107-116); done
// one after the other the GPU idles during both copies (and pageable copies are staged by the runtime at a few GB/s).  Here pass
// k + 1 is packed into a PINNED host buffer and copied in on its own stream, and the codes of pass k - 1 are copied out on a
// third, while the kernels of pass k run: two buffers of each kind, three streams, events between them.  Three slots per side:
//   in:  host hx[b] (pinned) -> device dx[b]      out A (codes / decode input):  device da[b] <-> host ha[b] (pinned)
//   out B (xhat of encode / x of decode): device db_[b] -> host hb[b] (pinned)
// ---------------------------------------------------------------------------------------------
struct HostPipe {
  hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
  hipEvent_t in_ready[2] = {nullptr, nullptr}, comp_done[2] = {nullptr, nullptr}, out_ready[2] = {nullptr, nullptr};
  void* hbuf[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // pinned host: [x | codes | floats][slot]
  void* dbuf[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // device
  size_t cap[3] = {0, 0, 0};
};

static void host_pipe_destroy(HostPipe* p) {
  if (!p) return;
  for (int k = 0; k < 3; ++k)
    for (int b = 0; b < 2; ++b) {
      if (p->hbuf[k][b]) (void)hipHostFree(p->hbuf[k][b]);
      if (p->dbuf[k][b]) (void)hipFree(p->dbuf[k][b]);
    }
  for (int b = 0; b < 2; ++b) {
    if (p->in_ready[b]) (void)hipEventDestroy(p->in_ready[b]);
    if (p->comp_done[b]) (void)hipEventDestroy(p->comp_done[b]);
    if (p->out_ready[b]) (void)hipEventDestroy(p->out_ready[b]);
  }
  if (p->s_in) (void)hipStreamDestroy(p->s_in);
  if (p->s_comp) (void)hipStreamDestroy(p->s_comp);
  if (p->s_out) (void)hipStreamDestroy(p->s_out);
  delete p;
}

// streams, events and the three buffer kinds at (at least) the given sizes
static int host_pipe_ensure(HostPipe** pp, const size_t (&need)[3]) {
  if (!*pp) {
    HostPipe* p = new HostPipe();
    *pp = p;
    HIP_TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p->s_comp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIP_TRY(hipEventCreateWithFlags(&p->in_ready[b], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&p->comp_done[b], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&p->out_ready[b], hipEventDisableTiming));
    }
  }
  HostPipe* p = *pp;
  for (int k = 0; k < 3; ++k) {
    if (need[k] <= p->cap[k]) continue;
    HIP_TRY(hipDeviceSynchronize());
    for (int b = 0; b < 2; ++b) {
      if (p->hbuf[k][b]) (void)hipHostFree(p->hbuf[k][b]);
      if (p->dbuf[k][b]) (void)hipFree(p->dbuf[k][b]);
      p->hbuf[k][b] = p->dbuf[k][b] = nullptr;
    }
    p->cap[k] = 0;
    for (int b = 0; b < 2; ++b) {
      HIP_TRY(hipHostMalloc(&p->hbuf[k][b], need[k], hipHostMallocDefault));
      HIP_TRY(hipMalloc(&p->dbuf[k][b], need[k]));
    }
    p->cap[k] = need[k];
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// weight packing: reference Linear weight W (O x I, row-major; y = W x) -> stream of 1 KiB fragments.
// Fragment (ob, ib, q): lane l, component e = W[ob*32 + (l&31)][ib*32 + 8q + 4(l>>5) + e], i.e. the A
// operands of the 4 MFMAs k-steps r = 4q..4q+3 whose B operand is register r of input block ib.
// ---------------------------------------------------------------------------------------------
static void put_frag(std::vector<float>& s, const float* W, int I, int ob, int ib, int q) {
  size_t base = s.size();
  s.resize(base + 256);
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e)
      s[base + l * 4 + e] = W[(size_t)(ob * 32 + (l & 31)) * I + ib * 32 + 8 * q + 4 * (l >> 5) + e];
}

static void pad_to(std::vector<float>& s, size_t frags_from, int T) {
  size_t want = frags_from + (size_t)T * 256;
  if (s.size() > want) abort();
  s.resize(want, 0.f);
}

static void pack_kouter(std::vector<float>& s, const float* W, int O, int I, int T) {
  size_t start = s.size();
  for (int ib = 0; ib < I / 32; ++ib)
    for (int q = 0; q < 4; ++q)
      for (int ob = 0; ob < O / 32; ++ob) put_frag(s, W, I, ob, ib, q);
  pad_to(s, start, T);
}

static void pack_obouter(std::vector<float>& s, const float* W, int O, int I, int T) {
  size_t start = s.size();
  for (int ob = 0; ob < O / 32; ++ob)
    for (int ib = 0; ib < I / 32; ++ib)
      for (int q = 0; q < 4; ++q) put_frag(s, W, I, ob, ib, q);
  pad_to(s, start, T);
}

static void pack_bias(std::vector<float>& s, const float* b, int O, int T) {
  size_t start = s.size();
  for (int ob = 0; ob < O / 32; ++ob)
    for (int q = 0; q < 4; ++q) {
      size_t base = s.size();
      s.resize(base + 256);
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) s[base + l * 4 + e] = b[ob * 32 + 8 * q + 4 * (l >> 5) + e];
    }
  pad_to(s, start, T);
}

// 16-row tile form (mlp16_kernel.hpp): fragment (ob, ib) = the 16 x 16 weight block, lane l, component r =
// W[ob*16 + (l & 15)][ib*16 + 4 (l >> 4) + r]; every section K-outer (for ib: for ob).
// (ld: row stride of W when it is a column range of a wider matrix; 0 = I)
static void pack16_kouter(std::vector<float>& s, const float* W, int O, int I, int T, int ld = 0) {
  size_t start = s.size();
  if (ld == 0) ld = I;
  for (int ib = 0; ib < I / 16; ++ib)
    for (int ob = 0; ob < O / 16; ++ob) {
      size_t base = s.size();
      s.resize(base + 256);
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) s[base + l * 4 + r] = W[(size_t)(ob * 16 + (l & 15)) * ld + ib * 16 + 4 * (l >> 4) + r];
    }
  pad_to(s, start, T);
}

static void pack16_bias(std::vector<float>& s, const float* b, int O, int T) {
  size_t start = s.size();
  for (int ob = 0; ob < O / 16; ++ob) {
    size_t base = s.size();
    s.resize(base + 256);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) s[base + l * 4 + r] = b[ob * 16 + 4 * (l >> 4) + r];
  }
  pad_to(s, start, T);
}

// Small-launch form (mlp_small_kernel.hpp): 16-feature blocks whose features sit in the 32-row kernel's contraction order.
// Fragment (ob, ib) = the A operands of the 4 MFMAs (v_mfma_f32_16x16x4_f32) of one 16 x 16 weight block: lane l, component e =
// W[16 ob + small_feat(m & 3, m >> 2)][16 ib + small_feat_at(4 e + (l >> 4))] with m = l & 15 -- MFMA output row m = 4 kg + r lands in
// lane group kg, register r, which is where the next layer's B operand wants feature small_feat(r, kg).  A section is, per wave w,
// [input block][j] with output block NW j + w; the NW = kSmallWaves waves' fragments are interleaved, blocks past the end are zero
// fragments.
static void pack_small_section(std::vector<float>& s, const float* W, int O, int I, int ld) {
  const int NW = kSmallWaves, NOB = O / 16, NIB = I / 16, NOW = (NOB + NW - 1) / NW;
  for (int ib = 0; ib < NIB; ++ib)
    for (int j = 0; j < NOW; ++j)
      for (int w = 0; w < NW; ++w) {
        const int ob = NW * j + w;
        const size_t base = s.size();
        s.resize(base + 256, 0.f);
        if (ob >= NOB) continue;
        for (int l = 0; l < 64; ++l) {
          const int m = l & 15, kga = l >> 4;
          const float* wr = W + (size_t)(16 * ob + small_feat(m & 3, m >> 2)) * ld + 16 * ib;
          for (int e = 0; e < 4; ++e) s[base + l * 4 + e] = wr[small_feat_at(4 * e + kga)];
        }
      }
}

// Split-fp16 form (mlp_split_kernel.hpp): fragment = the A operand of one v_mfma_f32_32x32x16_f16, 32 output features x 16
// inputs as fp16: lane l, element i = fp16 part of  scale * W[o*32 + (l & 31)][ib*32 + (i & 3) + 8 (2c + (i >> 2)) + 4 (l >> 5)]
// -- K-chunk c of input block ib in the register order of the 32x32 C/D layout (registers 8c .. 8c+7 of a lane).  hi = fp16(v)
// (round to nearest even), lo = fp16(v - hi).
static void put_split_frags(std::vector<float>& s, const float* W, int I, int o, int ib, int c, float scale) {
  size_t base = s.size();
  s.resize(base + 512);   // hi fragment, then lo fragment
  _Float16* hi = reinterpret_cast<_Float16*>(s.data() + base);
  _Float16* lo = reinterpret_cast<_Float16*>(s.data() + base + 256);
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 8; ++i) {
      const float v = scale * W[(size_t)(o * 32 + (l & 31)) * I + ib * 32 + (i & 3) + 8 * (2 * c + (i >> 2)) + 4 * (l >> 5)];
      const _Float16 h = (_Float16)v;
      hi[l * 8 + i] = h;
      lo[l * 8 + i] = (_Float16)(v - (float)h);
    }
}

// K-outer sections (up-projection Dh x De; out_proj D x De in passes of `no` output blocks): for ib: for c: for each output
// block of [o0, o0 + no): hi, lo
static void pack_split_kouter(std::vector<float>& s, const float* W, int ld, int K, int o0, int no, float scale) {
  for (int ib = 0; ib < K / 32; ++ib)   // (ld = row stride of W, K = its input features: W may be a column block of a wider matrix)
    for (int c = 0; c < 2; ++c)
      for (int o = 0; o < no; ++o) put_split_frags(s, W, ld, o0 + o, ib, c, scale);
}
static void pack_split_up(std::vector<float>& s, const float* W, int O, int I, int hh, int nhs, float scale, int T) {
  size_t start = s.size();
  const int noh = O / 32 / nhs;
  pack_split_kouter(s, W, I, I, hh * noh, noh, scale);
  pad_to(s, start, T);
}

// down-projection (De x Dh), hidden half hh: for each pair of output blocks: for ib (of the half): for c: hi, lo, hi, lo
static void pack_split_down(std::vector<float>& s, const float* W, int O, int I, int hh, int nhs, float scale, int T) {
  size_t start = s.size();
  const int nih = I / 32 / nhs;
  for (int og = 0; og < O / 64; ++og)
    for (int ib = 0; ib < nih; ++ib)
      for (int c = 0; c < 2; ++c)
        for (int o = 2 * og; o < 2 * og + 2; ++o) put_split_frags(s, W, I, o, hh * nih + ib, c, scale);
  pad_to(s, start, T);
}

// 2^e with max |scale * W| in [512, 1024): every weight down to 2^-11 of the largest keeps a normal fp16 lo part (>= 2^-14)
// and nothing comes near the fp16 maximum (65504).
static float split_weight_scale(const float* W, size_t n, size_t cols = 0, size_t ld = 0) {
  float mx = 0.f;   // (cols / ld given: the first `cols` columns of rows that are `ld` apart; n = rows * cols)
  for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(cols ? W[(i / cols) * ld + i % cols] : W[i]));
  if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
  return ldexpf(1.f, 9 - ilogbf(mx));
}

static int upload(qinco_handle_s* h, float** dst, const float* src, size_t count) {
  int rc = dev_alloc(h, dst, count);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*dst, src, count * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// A (rows, W) table with the features of every 16-block grouped by the lane group that holds them in the small-launch form
// (mlp_args.hpp small_feat): entry 4 kg + r of a block is feature small_feat(r, kg), so lane group kg of mlp_small_kernel finds its
// four registers of a block in ONE 16-byte load.
static int upload_block_layout(qinco_handle_s* h, const float* src, int rows, int W, float** dst) {
  std::vector<float> t((size_t)rows * W);
  for (int row = 0; row < rows; ++row)
    for (int b = 0; b < W / 16; ++b)
      for (int kg = 0; kg < 4; ++kg)
        for (int r = 0; r < 4; ++r) t[(size_t)row * W + 16 * b + 4 * kg + r] = src[(size_t)row * W + 16 * b + small_feat(r, kg)];
  return upload(h, dst, t.data(), t.size());
}

// codebook (K, D) in MFMA fragment order: (block of 32 codewords, feature block, q) -> 1 KiB
static int upload_fragments(qinco_handle_s* h, const float* cb, int K, int D, f32x4** out) {
  std::vector<float> s;
  s.reserve((size_t)K * D);
  for (int kb = 0; kb < K / 32; ++kb)
    for (int ib = 0; ib < D / 32; ++ib)
      for (int q = 0; q < 4; ++q) put_frag(s, cb, D, kb, ib, q);
  s.resize(s.size() + (size_t)16 * 256, 0.f);  // the table kernels prefetch up to 16 fragments past the end
  float* ds = nullptr;
  int rc = upload(h, &ds, s.data(), s.size());
  *out = reinterpret_cast<f32x4*>(ds);
  return rc;
}

// fp16 copy of the IVF centroids for the filter passes (ivf_f16_kernel.hpp): fragment (block of 32 centroids, k-step of
// 16 features): lane l, 8 halfs = C[block*32 + (l & 31)][k*16 + 8 (l >> 5) + 0..7], rounded to nearest even.
static int build_ivf_f16(qinco_handle_s* h, const float* cb, int real_rows) {
  const int K = h->d.ivf_K, D = h->d.D, NK = D / 16;
  float amax = 0.f;
  double n2max = 0.0;
  std::vector<float> nh(K);
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    double s2 = 0.0;
    for (int j = 0; j < D; ++j) {
      const float v = cb[(size_t)k * D + j];
      s = fmaf(v, v, s);
      s2 += (double)v * v;
      amax = fmaxf(amax, fabsf(v));
    }
    nh[k] = -0.5f * s;  // same |c|^2 as upload_with_norms, halved exactly and negated: the filter accumulators' start value
    if (k >= real_rows) nh[k] = -0.5f * 1e30f;   // an added row (zeros): as far away as upload_with_norms puts it
    if (s2 > n2max) n2max = s2;
  }
  if (!(amax < 60000.f)) return 0;  // outside the fp16 range: the exact fp32 kernel is used on its own
  // Stream order: a SAMPLE of the blocks first (every (nblocks / nsample)-th block, 1/8 of them), then the others.  Pass A of the
  // filter runs over the sample only: its minimum is an upper bound of the true minimum, so the threshold it gives pass B keeps
  // the argmin among the candidates (ivf_f16_kernel.hpp) -- at one eighth of the cost of a full pass.
  const int nblocks = K / 32;
  const int nsample = nblocks >= 8 ? nblocks / 8 : nblocks, sstride = nblocks / nsample;
  std::vector<int> perm;
  perm.reserve(nblocks);
  std::vector<char> taken(nblocks, 0);
  for (int i = 0; i < nsample; ++i) {
    perm.push_back(i * sstride);
    taken[i * sstride] = 1;
  }
  for (int b = 0; b < nblocks; ++b)
    if (!taken[b]) perm.push_back(b);
  std::vector<float> nhs(K);
  std::vector<_Float16> s((size_t)K * D + (size_t)32 * 512);   // the filter kernels' ring prefetches up to 16 fragments past the end
  size_t o = 0;
  for (int p = 0; p < nblocks; ++p) {
    const int b = perm[p];
    for (int i = 0; i < 32; ++i) nhs[(size_t)p * 32 + i] = nh[(size_t)b * 32 + i];
    for (int k = 0; k < NK; ++k)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) s[o++] = (_Float16)cb[(size_t)(b * 32 + (l & 31)) * D + k * 16 + 8 * (l >> 5) + e];
  }
  for (; o < s.size(); ++o) s[o] = (_Float16)0.f;
  nh.swap(nhs);   // (stream order from here on)
  int* dperm = nullptr;
  int rc0 = dev_alloc(h, &dperm, (size_t)nblocks);
  if (rc0) return rc0;
  HIP_TRY(hipMemcpy(dperm, perm.data(), (size_t)nblocks * sizeof(int), hipMemcpyHostToDevice));
  h->ivf_perm = dperm;
  h->ivf_sample_blocks = nsample;
  float* dh = nullptr;
  int rc = dev_alloc(h, &dh, (s.size() + 1) / 2);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(dh, s.data(), s.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  h->ivf_h16 = dh;
  if ((rc = upload(h, &h->ivf_cnorm_half, nh.data(), K))) return rc;
  h->ivf_cmax = (float)(std::sqrt(n2max) * 1.000001);
  h->ivf_f16 = true;
  return 0;
}

// real_rows >= 0: rows from there on were added to fill a block of 32 (all zeros) and must never win an arg-min
static constexpr float kNeverNorm = 1e30f;
static int upload_with_norms(qinco_handle_s* h, const float* cb, int K, int D, float** d_cb, float** d_norm, int real_rows = -1) {
  int rc = upload(h, d_cb, cb, (size_t)K * D);
  if (rc) return rc;
  std::vector<float> nrm(K);
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int j = 0; j < D; ++j) s = fmaf(cb[(size_t)k * D + j], cb[(size_t)k * D + j], s);
    nrm[k] = (real_rows >= 0 && k >= real_rows) ? kNeverNorm : s;
  }
  return upload(h, d_norm, nrm.data(), K);
}

// ---------------------------------------------------------------------------------------------
// scratch
// ---------------------------------------------------------------------------------------------
static int ensure_scratch(qinco_handle_s* h) {
  const qinco_desc& d = h->d;
  if (h->cap_n == d.max_batch && h->cap_A == h->A && h->cap_B == h->B) return 0;
  void* old[] = {h->xn, h->xhat[0], h->xhat[1], h->hist[0], h->hist[1], h->top_ids, h->cand, h->dist, h->ivf_best,
                 h->uproj, h->ivf_amin, h->ivf_cand};
  HIP_TRY(hipDeviceSynchronize());
  for (void* p : old) dev_free(h, p);
  h->xn = h->xhat[0] = h->xhat[1] = h->cand = h->dist = nullptr;
  h->hist[0] = h->hist[1] = h->top_ids = nullptr;
  h->ivf_best = nullptr;
  h->ivf_amin = nullptr;
  h->ivf_cand = nullptr;
  h->uproj = nullptr;
  h->cap_n = 0;
  const size_t n = (size_t)d.max_batch;
  // Widest beam and widest candidate set over the steps, walked exactly like encode_chunk does: beam_0 = min(B, K) (1 for
  // IVF / single-step models), then F <- min(B or 1, F * Ae).  (B > K: the beam grows past beam_0 after step 1.)
  size_t Bm = 1, Cm = 1;
  {
    size_t F = (d.M == 1 || d.ivf_K > 0) ? 1 : (size_t)(h->B < d.K ? h->B : d.K);
    Bm = F;
    for (int m = 1; m < d.M; ++m) {
      const int nc = n_codes(h, m) < d.K ? n_codes(h, m) : d.K;
      const size_t Ae = (size_t)(h->A > 0 ? nc : d.K);
      const size_t C = F * Ae;
      if (C > Cm) Cm = C;
      const size_t Fo = (size_t)((m < d.M - 1) ? h->B : 1);
      F = Fo < C ? Fo : C;
      if (F > Bm) Bm = F;
    }
  }
  int rc = 0;
  if ((rc = dev_alloc(h, &h->xn, n * d.D))) return rc;
  for (int i = 0; i < 2; ++i) {
    if ((rc = dev_alloc(h, &h->xhat[i], n * Bm * d.D))) return rc;
    if ((rc = dev_alloc(h, &h->hist[i], n * Bm * d.M))) return rc;
  }
  if ((rc = dev_alloc(h, &h->top_ids, n * (Cm > Bm ? Cm : Bm)))) return rc;
  if ((rc = dev_alloc(h, &h->cand, n * Cm * d.D))) return rc;
  if ((rc = dev_alloc(h, &h->dist, n * Cm))) return rc;
  if (d.ivf_K > 0 && (rc = dev_alloc(h, &h->ivf_best, n))) return rc;
  if (h->ivf_f16) {
    h->ivf_cand_cap = (int)(40 * n + 4096);   // pass B keeps ~8 candidates per vector (threshold from a 1/8 sample)
    if ((rc = dev_alloc(h, &h->ivf_amin, n))) return rc;
    if ((rc = dev_alloc(h, &h->ivf_cand, 4 + 2 * (size_t)h->ivf_cand_cap))) return rc;
  }
  if (h->fold && (rc = dev_alloc(h, &h->uproj, n * Bm * (d.De + (h->fold2 ? d.Dh : 0))))) return rc;
  h->cap_n = d.max_batch;
  h->cap_A = h->A;
  h->cap_B = h->B;
  return 0;
}

// FOLD (mlp_kernel.hpp): the row-independent head of the MLP.
//   T[k] = z_k + (b + W_cat[:, :De] z_k),  z_k = in_proj(c_k) (or c_k when De == D)     -- per codeword
//   wx   = W_cat[:, De:] as MFMA fragments (ob, ib, q)                                   -- for xproj_kernel
static int build_fold_tables(qinco_handle_s* h, const qinco_weights* w, int m) {
  const qinco_desc& d = h->d;
  const int D = d.D, De = d.De, K = d.K, I = De + D;
  const float* cb = w->codebook[m];
  const float* win = (De != D) ? w->in_proj[m] : nullptr;
  const float* wc = w->cat_w[m];
  const float* bc = w->cat_b[m];
  std::vector<float> T((size_t)K * De), z(De);
  for (int k = 0; k < K; ++k) {
    const float* c = cb + (size_t)k * D;
    for (int i = 0; i < De; ++i) {
      if (win) {
        float a = 0.f;
        for (int j = 0; j < D; ++j) a = fmaf(win[(size_t)i * D + j], c[j], a);
        z[i] = a;
      } else {
        z[i] = c[i];
      }
    }
    for (int i = 0; i < De; ++i) {
      float a = bc[i];
      const float* wr = wc + (size_t)i * I;
      for (int j = 0; j < De; ++j) a = fmaf(wr[j], z[j], a);
      T[(size_t)k * De + i] = z[i] + a;
    }
  }
  int rc = upload(h, &h->ttab[m], T.data(), T.size());
  if (rc) return rc;
  if (h->want_small && ((rc = upload_block_layout(h, T.data(), K, De, &h->ttab_s[m])) || (rc = upload_block_layout(h, cb, K, D, &h->cb_s[m]))))
    return rc;
  std::vector<float> s;
  s.reserve((size_t)De * D);
  if (h->inst->var & 128) {   // 16-row form: the projection runs through mlp16_kernel's own ring (MODE 1), K-outer fragments
    const int P = h->inst->P;
    pack16_kouter(s, wc + De, De, D, qinco::round_up((De / 16) * (D / 16), P), I);
    s.resize(s.size() + (size_t)P * 256, 0.f);   // the ring prefetches P fragments past the end
  } else {
    for (int ob = 0; ob < De / 32; ++ob)
      for (int ib = 0; ib < D / 32; ++ib)
        for (int q = 0; q < 4; ++q) put_frag(s, wc + De, I, ob, ib, q);
  }
  float* ds = nullptr;
  if ((rc = upload(h, &ds, s.data(), s.size()))) return rc;
  h->wx_stream[m] = reinterpret_cast<f32x4*>(ds);
  if (h->split16) {   // xproj_split_kernel: one stream [U section][Q section] of fp16 hi / lo fragments + its scalings
    const int Dh = d.Dh, P = h->inst->P;
    const float* up0 = w->up[(size_t)m * d.L];
    if (!up0) return fail(QINCO_ERR_INVALID, "qinco_create: FFN weights[%d][0] null", m);
    const float su = split_weight_scale(wc + De, (size_t)De * D, D, I), sq = split_weight_scale(up0, (size_t)Dh * De);
    std::vector<float> xs;
    size_t start = 0;
    pack_split_kouter(xs, wc + De, I, D, 0, De / 32, su);
    pad_to(xs, start, qinco::round_up(4 * (D / 32) * (De / 32), P));
    start = xs.size();
    pack_split_kouter(xs, up0, De, De, 0, Dh / 32, sq);
    pad_to(xs, start, qinco::round_up(4 * (De / 32) * (Dh / 32), P));
    xs.resize(xs.size() + (size_t)P * 256, 0.f);   // the ring prefetches P fragments past the end
    float* dxs = nullptr;
    if ((rc = upload(h, &dxs, xs.data(), xs.size()))) return rc;
    dev_free(h, ds);                               // (the fp32 fragments of W_cat[:, De:] are not used in this mode)
    h->wx_stream[m] = reinterpret_cast<f32x4*>(dxs);
    const float cx = 8.f, cu = 8.f;                // xhat' = 8 xhat, U' = 8 U (see the FFN blocks' scalings in qinco_create)
    const float sm[4] = {cx, 1.f / (cx * su), cu, 1.f / (cu * sq)};
    if ((rc = upload(h, &h->xsmul[m], sm, 4))) return rc;
  }
  if (h->fold2) {  // P_k = W_up[0] T_k  and  W_up[0] as fragments (ob, ib, q) for Q_g = W_up[0] U_g
    const int Dh = d.Dh;
    const float* up0 = w->up[(size_t)m * d.L];
    if (!up0) return fail(QINCO_ERR_INVALID, "qinco_create: FFN weights[%d][0] null", m);
    std::vector<float> Pt((size_t)K * Dh);
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < Dh; ++i) {
        float a = 0.f;
        const float* wr = up0 + (size_t)i * De;
        const float* t = T.data() + (size_t)k * De;
        for (int j = 0; j < De; ++j) a = fmaf(wr[j], t[j], a);
        Pt[(size_t)k * Dh + i] = a;
      }
    if ((rc = upload(h, &h->ptab[m], Pt.data(), Pt.size()))) return rc;
    if (h->want_small && (rc = upload_block_layout(h, Pt.data(), K, Dh, &h->ptab_s[m]))) return rc;
    std::vector<float> sq;
    sq.reserve((size_t)Dh * De);
    for (int ob = 0; ob < Dh / 32; ++ob)
      for (int ib = 0; ib < De / 32; ++ib)
        for (int q = 0; q < 4; ++q) put_frag(sq, up0, De, ob, ib, q);
    float* dq = nullptr;
    if ((rc = upload(h, &dq, sq.data(), sq.size()))) return rc;
    h->wq_stream[m] = reinterpret_cast<f32x4*>(dq);
  }
  return 0;
}

// Decode processes up to kDecodeChunk rows per pass so that even small-max_batch handles fill the chip
// (a pass needs rows/128 workgroups; 8192 rows would occupy only 64 of the 256 CUs).
static constexpr int64_t kDecodeChunk = 262144;

static int ensure_decode_scratch(qinco_handle_s* h, int64_t n) {
  int64_t want = n < kDecodeChunk ? n : kDecodeChunk;
  if (want < h->d.max_batch && n >= h->d.max_batch) want = h->d.max_batch;
  if (want <= h->dec_cap) return 0;
  HIP_TRY(hipDeviceSynchronize());
  dev_free(h, h->dxhat[0]);
  dev_free(h, h->dxhat[1]);
  dev_free(h, h->codes_t);
  dev_free(h, h->duproj);
  h->duproj = nullptr;
  h->dxhat[0] = h->dxhat[1] = nullptr;
  h->codes_t = nullptr;
  h->dec_cap = 0;
  int rc;
  for (int i = 0; i < 2; ++i)
    if ((rc = dev_alloc(h, &h->dxhat[i], (size_t)want * h->d.D))) return rc;
  if ((rc = dev_alloc(h, &h->codes_t, (size_t)want * h->d.M))) return rc;
  if (h->fold && !h->dec_inst && (rc = dev_alloc(h, &h->duproj, (size_t)want * (h->d.De + (h->fold2 ? h->d.Dh : 0))))) return rc;
  h->dec_cap = want;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------------------------
extern "C" int qinco_shape_supported(int32_t D, int32_t De, int32_t Dh) {
  if (D <= 0 || De <= 0 || Dh <= 0) return 0;
  int Dp, Dep, Dhp;
  padded_geometry(D, De, Dh, &Dp, &Dep, &Dhp);
  return find_mlp_instance(Dp, Dep, Dhp, -1, -1) != nullptr;
}

extern "C" int qinco_padded_shape(int32_t D, int32_t De, int32_t Dh, int32_t* out3) {
  if (!out3 || D <= 0 || De <= 0 || Dh <= 0) return fail(QINCO_ERR_INVALID, "qinco_padded_shape: bad argument");
  int Dp, Dep, Dhp;
  padded_geometry(D, De, Dh, &Dp, &Dep, &Dhp);
  out3[0] = Dp;
  out3[1] = Dep;
  out3[2] = Dhp;
  return QINCO_OK;
}

extern "C" int qinco_load_instance(const char* path) {
  if (!path) return fail(QINCO_ERR_INVALID, "qinco_load_instance: null path");
  void* so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!so) return fail(QINCO_ERR_INVALID, "qinco_load_instance: %s", dlerror());
  typedef int (*info_fn)(int32_t*, void**, int);
  info_fn info = reinterpret_cast<info_fn>(dlsym(so, "qinco_instance_info"));
  if (!info) {
    dlclose(so);
    return fail(QINCO_ERR_INVALID, "qinco_load_instance: %s does not export qinco_instance_info (not built from csrc/mlp_inst.hip "
                                   "with -DQINCO_INSTANCE_MODULE?)", path);
  }
  int32_t v[6] = {0, 0, 0, 0, 0, 0};
  // (a module from before the capacity argument ignores it and writes its own launcher count: the array is longer than any
  // count ever exported, and its abi word -- launcher count included since version 4 -- is refused below)
  void* fns[kInstanceNFns + 8] = {};
  const int abi = info(v, fns, kInstanceNFns), want_abi = instance_abi();
  if (abi != want_abi || !fns[0] || !fns[1]) {
    dlclose(so);
    return fail(QINCO_ERR_INVALID, "qinco_load_instance: %s was built against another version of csrc/mlp_args.hpp (0x%x vs 0x%x)", path,
                abi, want_abi);
  }
  for (const MlpInstance& i : g_loaded)
    if (i.D == v[0] && i.De == v[1] && i.Dh == v[2] && i.P == v[3] && i.var == v[4]) return QINCO_OK;   // already there
  g_loaded.push_back(MlpInstance{v[0], v[1], v[2], v[3], v[4], reinterpret_cast<mlp_launch_fn>(fns[0]),
                                 reinterpret_cast<xproj_launch_fn>(fns[1]), reinterpret_cast<table_launch_fn>(fns[2]),
                                 reinterpret_cast<ivf_launch_fn>(fns[3]), reinterpret_cast<small_launch_fn>(fns[4])});
  return QINCO_OK;
}

// Zero-padded copies of a model's tensors for a geometry that is not made of 32-feature blocks (padded_geometry).
struct PaddedWeights {
  std::deque<std::vector<float>> store;
  std::vector<const float*> cb, sub, inp, outp, cw, cbias, up, down;
  qinco_weights w{};

  const float* pad2d(const float* src, long rows, int cols, long rows_p, int cols_p) {
    if (!src) return nullptr;
    store.emplace_back((size_t)rows_p * cols_p, 0.f);
    float* dst = store.back().data();
    for (long r = 0; r < rows; ++r) std::memcpy(dst + (size_t)r * cols_p, src + (size_t)r * cols, (size_t)cols * sizeof(float));
    return dst;
  }
  // returns nullptr, or what is missing
  const char* build(const qinco_desc& d, const qinco_weights& u, int Dp, int Dep, int Dhp, int ivf_Kp) {
    const int M = d.M, L = d.L, D = d.D, De = d.De, Dh = d.Dh;
    if (!u.data_mean || !u.codebook) return "missing data_mean / codebook";
    w = u;
    w.data_mean = pad2d(u.data_mean, 1, D, 1, Dp);
    cb.assign(M, nullptr);
    sub.assign(M, nullptr);
    inp.assign(M, nullptr);
    outp.assign(M, nullptr);
    cw.assign(M, nullptr);
    cbias.assign(M, nullptr);
    up.assign((size_t)M * (L > 0 ? L : 1), nullptr);
    down.assign((size_t)M * (L > 0 ? L : 1), nullptr);
    for (int m = 0; m < M; ++m) {
      const long rows = (m == 0 && d.ivf_K > 0) ? d.ivf_K : d.K;
      cb[m] = pad2d(u.codebook[m], rows, D, (m == 0 && d.ivf_K > 0) ? ivf_Kp : rows, Dp);   // (added centroid rows: see ivf_real)
      if (m == 0) continue;
      if (u.sub_codebook) sub[m] = pad2d(u.sub_codebook[m], d.K, D, d.K, Dp);
      if (u.in_proj) inp[m] = pad2d(u.in_proj[m], De, D, Dep, Dp);
      if (u.out_proj) outp[m] = pad2d(u.out_proj[m], D, De, Dp, Dep);
      if (u.cat_b) cbias[m] = pad2d(u.cat_b[m], 1, De, 1, Dep);
      if (u.cat_w && u.cat_w[m]) {   // (De, De + D): the z columns and the xhat columns are padded separately
        store.emplace_back((size_t)Dep * (Dep + Dp), 0.f);
        float* dst = store.back().data();
        for (int r = 0; r < De; ++r) {
          std::memcpy(dst + (size_t)r * (Dep + Dp), u.cat_w[m] + (size_t)r * (De + D), (size_t)De * sizeof(float));
          std::memcpy(dst + (size_t)r * (Dep + Dp) + Dep, u.cat_w[m] + (size_t)r * (De + D) + De, (size_t)D * sizeof(float));
        }
        cw[m] = dst;
      }
      for (int l = 0; l < L; ++l) {
        if (u.up) up[(size_t)m * L + l] = pad2d(u.up[(size_t)m * L + l], Dh, De, Dhp, Dep);
        if (u.down) down[(size_t)m * L + l] = pad2d(u.down[(size_t)m * L + l], De, Dh, Dep, Dhp);
      }
    }
    w.codebook = cb.data();
    w.sub_codebook = u.sub_codebook ? sub.data() : nullptr;
    w.in_proj = u.in_proj ? inp.data() : nullptr;
    w.out_proj = u.out_proj ? outp.data() : nullptr;
    w.cat_w = u.cat_w ? cw.data() : nullptr;
    w.cat_b = u.cat_b ? cbias.data() : nullptr;
    w.up = u.up ? up.data() : nullptr;
    w.down = u.down ? down.data() : nullptr;
    return nullptr;
  }
};

// Diagnostic knobs of a handle (qinco_options in the ABI).  Experiment builds (-DQINCO_EXPERIMENT, scripts/) also read them from
// the environment so that an unmodified caller can be A/B-ed; the shipping library has no environment switches.
struct CreateOpts {
  bool calibrate = true;   // split form: compare with an fp32 twin on a calibration batch at create (never for the twin itself)
  int flags = 0;
  int mlp_P = -1, mlp_var = -1;
  long table_coop_max = -1;
};
static const int kCreateFlagMask = QINCO_CREATE_SPLIT_F16 | QINCO_CREATE_IVF_FP32 | QINCO_CREATE_TABLE_VALU | QINCO_CREATE_DECODE_FOLDED |
                                   QINCO_CREATE_TABLE_NO_COOP | QINCO_CREATE_SPLIT_NO_CALIBRATION | QINCO_CREATE_NO_PRESEL_FUSION |
                                   QINCO_CREATE_NO_SMALL_LAUNCH | QINCO_CREATE_EPILOGUE_SELECT | QINCO_CREATE_NO_EPILOGUE_SELECT;

static void env_opts(CreateOpts& o) {
#ifdef QINCO_EXPERIMENT
  if (const char* e = getenv("QINCO_SPLIT_F16")) if (atoi(e) > 0) o.flags |= QINCO_CREATE_SPLIT_F16;
  if (getenv("QINCO_IVF_FP32")) o.flags |= QINCO_CREATE_IVF_FP32;
  if (getenv("QINCO_TABLE_VALU")) o.flags |= QINCO_CREATE_TABLE_VALU;
  if (getenv("QINCO_DECODE_FOLDED")) o.flags |= QINCO_CREATE_DECODE_FOLDED;
  if (getenv("QINCO_TABLE_NO_COOP")) o.flags |= QINCO_CREATE_TABLE_NO_COOP;
  if (getenv("QINCO_NO_PRESEL_FUSION")) o.flags |= QINCO_CREATE_NO_PRESEL_FUSION;
  if (getenv("QINCO_NO_SMALL_LAUNCH")) o.flags |= QINCO_CREATE_NO_SMALL_LAUNCH;
  if (getenv("QINCO_EPILOGUE_SELECT")) o.flags |= QINCO_CREATE_EPILOGUE_SELECT;
  if (const char* e = getenv("QINCO_TABLE_COOP_MAX")) o.table_coop_max = atol(e);
  if (const char* e = getenv("QINCO_MLP_VARIANT")) sscanf(e, "%d,%d", &o.mlp_P, &o.mlp_var);
#else
  (void)o;
#endif
}

static int create_impl(const qinco_desc* desc, const qinco_weights* w, CreateOpts opt, qinco_handle* out);

extern "C" int qinco_create(const qinco_desc* desc, const qinco_weights* w, qinco_handle* out) {
  CreateOpts o;
  env_opts(o);
  return create_impl(desc, w, o, out);
}

extern "C" int qinco_create_ex(const qinco_desc* desc, const qinco_weights* w, int32_t create_flags, qinco_handle* out) {
  if (create_flags & ~kCreateFlagMask) return fail(QINCO_ERR_INVALID, "qinco_create_ex: unknown flag bits 0x%x", create_flags);
  CreateOpts o;
  env_opts(o);
  o.flags |= create_flags;
  return create_impl(desc, w, o, out);
}

extern "C" int qinco_create_opt(const qinco_desc* desc, const qinco_weights* w, const qinco_options* opt, qinco_handle* out) {
  CreateOpts o;
  env_opts(o);
  if (opt) {
    if (opt->struct_bytes != (int32_t)sizeof(qinco_options))
      return fail(QINCO_ERR_INVALID, "qinco_create_opt: struct_bytes %d, this library's qinco_options has %d", opt->struct_bytes,
                  (int)sizeof(qinco_options));
    if (opt->create_flags & ~kCreateFlagMask) return fail(QINCO_ERR_INVALID, "qinco_create_opt: unknown flag bits 0x%x", opt->create_flags);
    o.flags |= opt->create_flags;
    if (opt->mlp_P >= 0 || opt->mlp_var >= 0) { o.mlp_P = opt->mlp_P; o.mlp_var = opt->mlp_var; }
    if (opt->table_coop_max >= 0) o.table_coop_max = (long)opt->table_coop_max;
  }
  return create_impl(desc, w, o, out);
}

static int calibrate_split(qinco_handle_s* h, const qinco_desc* desc, const qinco_weights* w, const CreateOpts& opt);

static int create_impl(const qinco_desc* desc, const qinco_weights* w, CreateOpts opt, qinco_handle* out) {
  const int create_flags = opt.flags;
  const qinco_weights* const w_user = w;   // (w is re-pointed at padded copies below; the calibration twin starts from the user's)
  if (!desc || !w || !out) return fail(QINCO_ERR_INVALID, "qinco_create: null argument");
  // A model without FFN blocks (L = 0) runs as L = 1 with an all-zero block: z + W_down relu(W_up z) = z + 0 exactly, and every
  // kernel form (FOLD2 and the split form peel block 0, the 16-row tile form) then serves it without an instance of its own.
  qinco_desc dpad = *desc;
  qinco_weights wpad = *w;
  std::vector<float> zero_block;
  std::vector<const float*> zero_ptrs;
  if (desc->L == 0 && desc->M > 1 && desc->De > 0 && desc->Dh > 0) {
    zero_block.assign((size_t)desc->De * desc->Dh, 0.f);
    zero_ptrs.assign((size_t)desc->M, zero_block.data());
    dpad.L = 1;
    wpad.up = wpad.down = zero_ptrs.data();
    w = &wpad;
  }
  if (desc->D <= 0 || desc->De <= 0 || desc->Dh <= 0 || desc->M <= 0 || desc->K <= 0 || desc->L < 0 || desc->max_batch <= 0)
    return fail(QINCO_ERR_INVALID, "qinco_create: non-positive hyper-parameter");
  // Geometry that is not a multiple of 32 (or whose projections would vanish): zero-padded copies of every tensor.
  PaddedWeights padw;
  {
    int Dp, Dep, Dhp;
    padded_geometry(desc->D, desc->De, desc->Dh, &Dp, &Dep, &Dhp);
    // ... and the IVF kernels on blocks of 32 centroids: a coarse codebook of any other size gets all-zero rows up to the next
    // multiple whose squared norm is SET to 1e30 (upload_with_norms, build_ivf_f16), so that none of them is ever the arg-min
    const int ivf_Kp = desc->ivf_K > 0 ? round_up(desc->ivf_K, 32) : desc->ivf_K;
    if (Dp != desc->D || Dep != desc->De || Dhp != desc->Dh || ivf_Kp != desc->ivf_K) {
      if (const char* why = padw.build(dpad, *w, Dp, Dep, Dhp, ivf_Kp)) return fail(QINCO_ERR_INVALID, "qinco_create: %s", why);
      dpad.D = Dp;
      dpad.De = Dep;
      dpad.Dh = Dhp;
      dpad.ivf_K = ivf_Kp;
      w = &padw.w;
    }
  }
  const qinco_desc& d = dpad;
  const int ivf_real = desc->ivf_K;      // centroids the model has; d.ivf_K - ivf_real rows were added
  if (d.K > 1024)
    return fail(QINCO_ERR_UNSUPPORTED, "qinco_create: K=%d > 1024 not supported -- a limit of this engine, not of the reference: its codebooks are "
                "nn.Embedding(K, d) of any K (qinco_base.py:107, 229); the table / selection kernels here keep a codebook's K distances of "
                "a group in one wave's registers and LDS lists, sized for K <= 1024 (every preset of the reference has K = 256)", d.K);
  if (d.A < 0 || d.A > d.K || d.B < 1) return fail(QINCO_ERR_INVALID, "qinco_create: need 0 <= A <= K and B >= 1");
  if (d.ivf_K < 0 || d.ivf_K > (1 << 24))
    return fail(QINCO_ERR_UNSUPPORTED, "qinco_create: ivf_K=%d must be in [0, 2^24] -- a limit of this engine, not of the reference: IVFBook is "
                "nn.Embedding(ivf_K, D) of any size (qinco_base.py:138-139); the assignment kernels here carry the centroid id in the low "
                "24 bits of a (distance, id) key and int32 candidate lists (the reference's largest IVF has 2^20 centroids)", desc->ivf_K);
  if (d.ivf_K > 0 && d.M < 2) return fail(QINCO_ERR_INVALID, "qinco_create: an IVF model needs at least one QINCo step");
  if (d.ivf_K > 0 && d.A > 0 && d.B > d.K)
    return fail(QINCO_ERR_INVALID, "qinco_create: the first QINCo step of an IVF model pre-selects max(A, B) = %d of its K = %d codewords "
                "(qinco_base.py:108-112; the reference's topk raises)", d.B, d.K);
  if (d.ivf_K > 0 && !ivf_compiled_in(d.D) && !find_ivf_launcher(d.D))
    return fail(QINCO_ERR_UNSUPPORTED, "qinco_create: no IVF kernel instance for D=%d [the model's %d in 32-feature blocks]: build a module "
                "of that D on demand (qinco_amd.build.ensure_instance -> qinco_load_instance)", d.D, desc->D);
  if (!(w->data_std > 0.f)) return fail(QINCO_ERR_INVALID, "qinco_create: data_std must be > 0 (qinco_base.py:526)");
  const int want_P = opt.mlp_P, want_var = opt.mlp_var;  // diagnostics: a non-production instance of the shape
  const MlpInstance* fn = find_mlp_instance(d.D, d.De, d.Dh, want_P, want_var);
  if ((create_flags & QINCO_CREATE_SPLIT_F16) && d.M > 1) {
    // the split-fp16 instance of the shape (VAR bit 512); it peels FFN block 0 like FOLD2, so the model needs L >= 1
    const MlpInstance* sp = nullptr;
    for (int pp : {96, 64, 48})
      if (!sp) {
        const MlpInstance* c = find_mlp_instance(d.D, d.De, d.Dh, pp, 512 | 124);
        if (c && c->P == pp && c->var == (512 | 124)) sp = c;
      }
    if (!sp)
      return fail(QINCO_ERR_UNSUPPORTED, "qinco_create_ex: no split-fp16 kernel instance for (D=%d, De=%d, Dh=%d)", d.D, d.De, d.Dh);
    fn = sp;
  }
  if (!fn && d.M > 1)
    return fail(QINCO_ERR_UNSUPPORTED, "qinco_create: no fused-MLP kernel instance for (D=%d, De=%d, Dh=%d) [the model's (%d, %d, %d) in "
                "32-feature blocks]: build one on demand (qinco_amd.build.ensure_instance -> qinco_load_instance) or add it to csrc/shapes.def",
                d.D, d.De, d.Dh, desc->D, desc->De, desc->Dh);
  if (!w->data_mean || !w->codebook) return fail(QINCO_ERR_INVALID, "qinco_create: missing data_mean / codebook");
  if (d.M > 1 && (!w->cat_w || !w->cat_b || (d.L > 0 && (!w->up || !w->down))))
    return fail(QINCO_ERR_INVALID, "qinco_create: missing MLP weights");
  if (d.De != d.D && d.M > 1 && (!w->in_proj || !w->out_proj))
    return fail(QINCO_ERR_INVALID, "qinco_create: De != D requires in_proj / out_proj");
  if (d.A > 0 && d.M > 1 && !w->sub_codebook)
    return fail(QINCO_ERR_INVALID, "qinco_create: A > 0 requires the substep codebooks");

  qinco_handle_s* h = new qinco_handle_s();
  h->d = d;
  h->user = *desc;
  h->A = d.A;
  h->B = d.B;
  h->inst = fn;
  if (d.ivf_K > 0 && !ivf_compiled_in(d.D)) h->ivf_module = find_ivf_launcher(d.D);
  h->std_ = w->data_std;
  h->table_valu = (create_flags & QINCO_CREATE_TABLE_VALU) != 0;
  h->table_coop = !(create_flags & QINCO_CREATE_TABLE_NO_COOP);
  h->no_presel_fusion = (create_flags & QINCO_CREATE_NO_PRESEL_FUSION) != 0;
  if (opt.table_coop_max >= 0) h->table_coop_max = opt.table_coop_max;
  const int kRing = fn ? fn->P : 8;
  h->fold = fn && (fn->var & 16);
  h->fold2 = fn && (fn->var & 32);
  h->split16 = fn && (fn->var & 512);
  const bool tile16 = fn && (fn->var & 128);
  small_launch_fn small_fn = nullptr;
  if (d.M > 1 && !(create_flags & QINCO_CREATE_NO_SMALL_LAUNCH) && !h->split16) small_fn = find_small_launcher(fn);
  h->want_small = small_fn != nullptr;
  h->sd = stream_dims(d.D, d.De, d.Dh, kRing, h->fold, h->fold2, tile16 ? 16 : 32);
  // Decode through an un-folded twin of the instance: only where that is faster -- the two-workgroups-per-CU (short-MLP) shapes
  // (qinco2-S: 57.4 M vec/s against 54.5 M through the folded kernel).  On the 384-wide shapes the folded kernel wins once the
  // measurement is long enough to be warm (C2 2.12 M against 2.08 M, qinco2-M 7.46 M against 6.92 M; scripts/exp_decode_twin.py):
  // round 2 had it the other way round from 10 ms timing loops.
  if (h->fold && (fn->var & 256) && !(create_flags & QINCO_CREATE_DECODE_FOLDED)) {
    const MlpInstance* di = find_mlp_instance(d.D, d.De, d.Dh, fn->P, fn->var & ~(16 | 32 | 4096));
    if (di && !(di->var & (16 | 32 | 128)) && di->P == fn->P) {
      h->dec_inst = di;
      h->dec_sd = stream_dims(d.D, d.De, d.Dh, di->P, false, false, 32);
    }
  }
  if (fn && (fn->var & 4096)) {
    // KHEAD takes launches whose groups are A >= 32 or A == 16 rows (mlp_kernel.hpp); every other launch, and the epilogue
    // selection, runs on the twin without the bit, which reads block 0's down-projection ob-outer (alt_wstream)
    const MlpInstance* ai = find_mlp_instance(d.D, d.De, d.Dh, fn->P, fn->var & ~4096);
    if (!ai || ai->P != fn->P || ai->var != (fn->var & ~4096)) {
      delete h;
      return fail(QINCO_ERR_UNSUPPORTED, "qinco_create: kernel instance (P=%d, VAR=%d) needs its twin VAR=%d for group sizes it does not take",
                  fn->P, fn->var, fn->var & ~4096);
    }
    h->alt_inst = ai;
  }
  if (fn && !(create_flags & QINCO_CREATE_NO_EPILOGUE_SELECT) && !(fn->var & 2048) && d.De == d.D && want_var < 0) {
    // the step's top-T in the fused-MLP kernel's epilogue: by default with the instance's own KHEAD + SELEP form (it reads the production
    // stream), on request (QINCO_CREATE_EPILOGUE_SELECT) also with the twin's SELEP form (measured no faster than the two kernels)
    const int base = fn->var & ~4096;
    if (fn->var & 4096) {
      const MlpInstance* si = find_mlp_instance(d.D, d.De, d.Dh, fn->P, fn->var | 2048);
      if (si && si->P == fn->P && si->var == (fn->var | 2048)) h->sel_inst = si;
    }
    if (!h->sel_inst && (create_flags & QINCO_CREATE_EPILOGUE_SELECT)) {
      const MlpInstance* si = find_mlp_instance(d.D, d.De, d.Dh, fn->P, base | 2048);
      if (si && si->P == fn->P && si->var == (base | 2048)) h->sel_inst = si;
    }
  }
  int rc = 0;
  auto bail = [&](int code) {
    qinco_destroy(h);
    return code;
  };
  if (hipGetDevice(&h->device) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipGetDevice failed (no HIP device?)"));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->num_cu = cus;
  }
  if (h->split16) {   // the split-form kernel of the 384-wide shapes uses all 160 KiB of a gfx950 CU's LDS (ring + parked z')
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess || lds < 160 * 1024)
      return bail(fail(QINCO_ERR_UNSUPPORTED, "qinco_create_ex: the split-fp16 kernels need 160 KiB of LDS per workgroup (device: %d B)", lds));
  }
  if ((rc = upload(h, &h->mean, w->data_mean, d.D))) return bail(rc);

  h->codebook.assign(d.M, nullptr);
  h->sub_codebook.assign(d.M, nullptr);
  h->cnorm.assign(d.M, nullptr);
  h->sub_cnorm.assign(d.M, nullptr);
  h->wstream.assign(d.M, nullptr);
  h->dec_wstream.assign(d.M, nullptr);
  h->alt_wstream.assign(d.M, nullptr);
  h->cb_stream.assign(d.M, nullptr);
  h->sub_stream.assign(d.M, nullptr);
  h->ttab.assign(d.M, nullptr);
  h->ptab.assign(d.M, nullptr);
  h->ttab_s.assign(d.M, nullptr);
  h->ptab_s.assign(d.M, nullptr);
  h->cb_s.assign(d.M, nullptr);
  h->wx_stream.assign(d.M, nullptr);
  h->wq_stream.assign(d.M, nullptr);
  h->smul.assign(d.M, nullptr);
  h->xsmul.assign(d.M, nullptr);
  h->K0 = d.ivf_K > 0 ? ivf_real : d.K;
  std::vector<int> kv(d.M, d.K);
  kv[0] = h->K0;
  if ((rc = dev_alloc(h, &h->kvals, d.M))) return bail(rc);
  if (hipMemcpy(h->kvals, kv.data(), d.M * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(QINCO_ERR_HIP, "hipMemcpy(kvals) failed"));
  if ((rc = dev_alloc(h, &h->err_flag, 1))) return bail(rc);
  if (hipMemset(h->err_flag, 0, sizeof(int)) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipMemset failed"));
  if (h->split16) {
    if ((rc = dev_alloc(h, &h->split_stats, 2))) return bail(rc);
    if (hipMemset(h->split_stats, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipMemset failed"));
  }

  for (int m = 0; m < d.M; ++m) {
    if (!w->codebook[m]) return bail(fail(QINCO_ERR_INVALID, "qinco_create: codebook[%d] is null", m));
    if ((rc = upload_with_norms(h, w->codebook[m], (m == 0 && d.ivf_K > 0) ? d.ivf_K : d.K, d.D, &h->codebook[m], &h->cnorm[m],
                                (m == 0 && d.ivf_K > 0) ? ivf_real : -1)))
      return bail(rc);
    if (m == 0) {
      if (d.ivf_K > 0) {
        if ((rc = upload_fragments(h, w->codebook[0], d.ivf_K, d.D, &h->ivf_stream))) return bail(rc);
        if (!(create_flags & QINCO_CREATE_IVF_FP32) && !h->ivf_module && (rc = build_ivf_f16(h, w->codebook[0], ivf_real))) return bail(rc);
      } else if (mfma_table_ok(d, h->inst)) {
        if ((rc = upload_fragments(h, w->codebook[0], d.K, d.D, &h->cb_stream[0]))) return bail(rc);
      }
      continue;
    }
    if (d.A > 0) {
      if (!w->sub_codebook[m]) return bail(fail(QINCO_ERR_INVALID, "qinco_create: sub_codebook[%d] is null", m));
      if ((rc = upload_with_norms(h, w->sub_codebook[m], d.K, d.D, &h->sub_codebook[m], &h->sub_cnorm[m]))) return bail(rc);
      if (mfma_table_ok(d, h->inst) && (rc = upload_fragments(h, w->sub_codebook[m], d.K, d.D, &h->sub_stream[m]))) return bail(rc);
    }
    // packed stream, in the order mlp_kernel consumes it
    const StreamDims& sd = h->sd;
    std::vector<float> s;
    s.reserve((size_t)(sd.total(d.L) + kRing) * 256);
    if (sd.PROJ && (!w->in_proj[m] || !w->out_proj[m]))
      return bail(fail(QINCO_ERR_INVALID, "qinco_create: in/out_proj[%d] is null", m));
    if (!w->cat_w[m] || !w->cat_b[m]) return bail(fail(QINCO_ERR_INVALID, "qinco_create: concat weights[%d] null", m));
    if (tile16) {
      if (h->fold) {
        if ((rc = build_fold_tables(h, w, m))) return bail(rc);
      } else {
        if (sd.PROJ) pack16_kouter(s, w->in_proj[m], d.De, d.D, sd.T_IN);
        pack16_bias(s, w->cat_b[m], d.De, sd.T_BIAS);
        pack16_kouter(s, w->cat_w[m], d.De, d.De + d.D, sd.T_CAT);
      }
      for (int l = 0; l < d.L; ++l) {
        const float* up = w->up[(size_t)m * d.L + l];
        const float* dn = w->down[(size_t)m * d.L + l];
        if (!up || !dn) return bail(fail(QINCO_ERR_INVALID, "qinco_create: FFN weights[%d][%d] null", m, l));
        pack16_kouter(s, up, d.Dh, d.De, sd.T_UP);
        pack16_kouter(s, dn, d.De, d.Dh, sd.T_DOWN);
      }
      if (sd.PROJ) pack16_kouter(s, w->out_proj[m], d.D, d.De, sd.T_OUT);
    } else if (h->fold) {
      if ((rc = build_fold_tables(h, w, m))) return bail(rc);
    } else {
      if (sd.PROJ) pack_kouter(s, w->in_proj[m], d.De, d.D, sd.T_IN);
      pack_bias(s, w->cat_b[m], d.De, sd.T_BIAS);
      pack_kouter(s, w->cat_w[m], d.De, d.De + d.D, sd.T_CAT);
    }
    std::vector<float> smul;
    if (h->split16) {
      // z' = 2^c z, h' = 2^a h (c = a = 3: values of O(1) sit around 8, their fp16 lo parts around 2^-9 -- normal --, and
      // nothing below |v| = 2^16 / 8 overflows fp16); W' = scale * W per layer.  Epilogue multipliers (exact powers of two):
      //   up:   h' = m_up relu(acc),   acc = W_up' z' = 2^(c+d) W_up z        -> m_up = 2^(a-c-d)     (block 0: h' = 2^a relu(P+Q))
      //   down: z' += m_down acc,      acc = W_down' h' = 2^(a+b) W_down h    -> m_down = 2^(c-a-b)
      const float zc = 8.f;
      smul = {zc, 1.f / zc};
    }
    for (int l = 0; l < d.L && !tile16; ++l) {
      const float* up = w->up[(size_t)m * d.L + l];
      const float* dn = w->down[(size_t)m * d.L + l];
      if (!up || !dn) return bail(fail(QINCO_ERR_INVALID, "qinco_create: FFN weights[%d][%d] null", m, l));
      if (h->split16) {
        const int nhs = 1;
        const float su = split_weight_scale(up, (size_t)d.Dh * d.De), sdn = split_weight_scale(dn, (size_t)d.De * d.Dh);
        const float zc = smul[0], ha = 8.f;
        smul.push_back(l == 0 ? ha : ha / (zc * su));
        smul.push_back(zc / (ha * sdn));
        for (int hh = 0; hh < nhs; ++hh) {
          if (l > 0) pack_split_up(s, up, d.Dh, d.De, hh, nhs, su, sd.T_UP / nhs);
          pack_split_down(s, dn, d.De, d.Dh, hh, nhs, sdn, sd.T_DOWN / nhs);
        }
        continue;
      }
      if (!(h->fold2 && l == 0)) pack_obouter(s, up, d.Dh, d.De, sd.T_UP);
      if (h->fold2 && l == 0 && (fn->var & 4096)) pack_kouter(s, dn, d.De, d.Dh, sd.T_DOWN);   // KHEAD: block 0's down-projection K-outer
      else pack_obouter(s, dn, d.De, d.Dh, sd.T_DOWN);
    }
    if (h->split16 && sd.PROJ && (d.D / 32) % 2 == 0) {   // = split_out_proj(D, De): out_proj in the split form, K-outer passes
      const float so = split_weight_scale(w->out_proj[m], (size_t)d.D * d.De);
      const int og = d.D / 32 < d.Dh / 32 ? d.D / 32 : d.Dh / 32;   // = split_out_group(D, Dh)
      const size_t start = s.size();
      for (int o0 = 0; o0 < d.D / 32; o0 += og) pack_split_kouter(s, w->out_proj[m], d.De, d.De, o0, og, so);
      pad_to(s, start, sd.T_OUT);
      smul.push_back(1.f / (smul[0] * so));   // o = m_out acc,  acc = W_out' z' = 2^(c+s) W_out z
    } else if (sd.PROJ && !tile16) {
      pack_obouter(s, w->out_proj[m], d.D, d.De, sd.T_OUT);
    }
    if ((long)(s.size() / 256) != sd.total(d.L)) return bail(fail(QINCO_ERR_INVALID, "internal: stream size mismatch"));
    if (h->split16 && (rc = upload(h, &h->smul[m], smul.data(), smul.size()))) return bail(rc);
    s.resize(s.size() + (size_t)kRing * 256, 0.f);  // the ring prefetches P fragments past the end
    float* ds = nullptr;
    if ((rc = upload(h, &ds, s.data(), s.size()))) return bail(rc);
    h->wstream[m] = reinterpret_cast<f32x4*>(ds);
    if (h->alt_inst) {   // KHEAD's twin: block 0's down-projection (the stream's first section) ob-outer, the rest as it is
      std::vector<float> t;
      t.reserve(s.size());
      pack_obouter(t, w->down[(size_t)m * d.L], d.De, d.Dh, sd.T_DOWN);
      t.insert(t.end(), s.begin() + (size_t)sd.T_DOWN * 256, s.end());
      float* dt = nullptr;
      if ((rc = upload(h, &dt, t.data(), t.size()))) return bail(rc);
      h->alt_wstream[m] = reinterpret_cast<f32x4*>(dt);
    }
    if (h->dec_inst) {   // the complete stream (in_proj, bias, concat, every FFN block, out_proj) for the un-folded decode kernel
      const StreamDims& ds_ = h->dec_sd;
      std::vector<float> t;
      t.reserve((size_t)(ds_.total(d.L) + kRing) * 256);
      if (ds_.PROJ) pack_kouter(t, w->in_proj[m], d.De, d.D, ds_.T_IN);
      pack_bias(t, w->cat_b[m], d.De, ds_.T_BIAS);
      pack_kouter(t, w->cat_w[m], d.De, d.De + d.D, ds_.T_CAT);
      for (int l = 0; l < d.L; ++l) {
        pack_obouter(t, w->up[(size_t)m * d.L + l], d.Dh, d.De, ds_.T_UP);
        pack_obouter(t, w->down[(size_t)m * d.L + l], d.De, d.Dh, ds_.T_DOWN);
      }
      if (ds_.PROJ) pack_obouter(t, w->out_proj[m], d.D, d.De, ds_.T_OUT);
      if ((long)(t.size() / 256) != ds_.total(d.L)) return bail(fail(QINCO_ERR_INVALID, "internal: decode stream size mismatch"));
      t.resize(t.size() + (size_t)kRing * 256, 0.f);
      float* dt = nullptr;
      if ((rc = upload(h, &dt, t.data(), t.size()))) return bail(rc);
      h->dec_wstream[m] = reinterpret_cast<f32x4*>(dt);
    }
  }
  // ---- small-launch form: one contiguous stream over the steps 1 .. M-1 (decode walks through all of them in one launch) ----
  if (h->want_small) {
    if (small_launch_fn sf = small_fn) {
      h->small_max_nt[0] = (int)sf(nullptr, 0, 0, nullptr);
      h->small_max_nt[1] = (int)sf(nullptr, 1, 0, nullptr);
    if (h->small_max_nt[0] > 0 || h->small_max_nt[1] > 0) {
      h->small = sf;
      h->ssd = small_dims(d.D, d.De, d.Dh, h->fold2);
      const SmallDims& S = h->ssd;
      std::vector<float> t;
      t.reserve((size_t)((d.M - 1) * S.step(d.L) + 64) * kSmallWaves * 256);
      for (int m = 1; m < d.M; ++m) {
        const size_t start = t.size();
        pack_small_section(t, w->cat_w[m] + d.De, d.De, d.D, d.De + d.D);
        if (h->fold2) pack_small_section(t, w->up[(size_t)m * d.L], d.Dh, d.De, d.De);
        for (int l = 0; l < d.L; ++l) {
          if (!(h->fold2 && l == 0)) pack_small_section(t, w->up[(size_t)m * d.L + l], d.Dh, d.De, d.De);
          pack_small_section(t, w->down[(size_t)m * d.L + l], d.De, d.Dh, d.Dh);
        }
        if (S.PROJ) pack_small_section(t, w->out_proj[m], d.D, d.De, d.De);
        if ((long)((t.size() - start) / (256 * kSmallWaves)) != S.step(d.L)) return bail(fail(QINCO_ERR_INVALID, "internal: small-form stream size mismatch"));
      }
      t.resize(t.size() + (size_t)64 * kSmallWaves * 256, 0.f);   // the rings prefetch up to PW <= 64 fragments per wave past the end
      float* dt = nullptr;
      if ((rc = upload(h, &dt, t.data(), t.size()))) return bail(rc);
      h->small_stream = reinterpret_cast<f32x4*>(dt);
      std::vector<SmallStep> ss(d.M);
      for (int m = 1; m < d.M; ++m) ss[m] = SmallStep{h->ttab_s[m], h->ptab_s[m], h->cb_s[m]};
      if ((rc = dev_alloc(h, &h->small_steps, (size_t)d.M))) return bail(rc);
      if (hipMemcpy(h->small_steps, ss.data(), ss.size() * sizeof(SmallStep), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(QINCO_ERR_HIP, "hipMemcpy(small_steps) failed"));
    }
    }
  }
  if ((rc = ensure_scratch(h))) return bail(rc);
  if (h->split16 && opt.calibrate && !(create_flags & QINCO_CREATE_SPLIT_NO_CALIBRATION) && d.M > 1) {
    if ((rc = calibrate_split(h, desc, w_user, opt))) return bail(rc);
  }
  *out = h;
  return QINCO_OK;
}

// Create-time calibration of the split-fp16 form: the same model as an fp32 twin, a batch of vectors drawn around the model's
// own codebooks (sum of one random codeword per step + noise, in the normalised space), both encoded; recorded: code rows that
// differ, the largest relative difference of the reconstructions on rows with equal codes, non-finite outputs.  A split form
// that is not in the fp32 path's error class on this model (static power-of-two scalings chosen from the weights, fp16's
// five exponent bits) is REFUSED here, at create, with QINCO_ERR_RANGE -- not discovered later in somebody's recall numbers.
static int calibrate_split(qinco_handle_s* h, const qinco_desc* desc, const qinco_weights* w, const CreateOpts& opt) {
  const int n = 512, D = desc->D, M = desc->M;
  CreateOpts o2 = opt;
  o2.flags &= ~QINCO_CREATE_SPLIT_F16;
  o2.mlp_P = o2.mlp_var = -1;
  o2.calibrate = false;
  qinco_desc d2 = *desc;
  d2.max_batch = n;
  qinco_handle twin = nullptr;
  int rc = create_impl(&d2, w, o2, &twin);
  if (rc) return rc;
  struct Lcg {
    unsigned long long s;
    unsigned next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); }
    float gauss() {   // sum of 12 uniforms - 6
      float a = 0.f;
      for (int i = 0; i < 12; ++i) a += (float)(next() & 0xffffff) / 16777216.f;
      return a - 6.f;
    }
  } rng{0x51ED2701ull};
  std::vector<float> x((size_t)n * D), xs((size_t)n * D), xf((size_t)n * D);
  double energy = 0.0;
  for (int i = 0; i < n; ++i) {
    float* xi = &x[(size_t)i * D];
    for (int m = 0; m < M; ++m) {
      const long rows = (m == 0 && desc->ivf_K > 0) ? desc->ivf_K : desc->K;
      const float* c = w->codebook[m] + (size_t)(rng.next() % rows) * D;
      for (int j = 0; j < D; ++j) xi[j] += c[j];
    }
    for (int j = 0; j < D; ++j) energy += (double)xi[j] * xi[j];
  }
  const float noise = 0.1f * (float)std::sqrt(energy / ((double)n * D) + 1e-30);
  for (float& v : x) v += noise * rng.gauss();
  std::vector<int> cs((size_t)n * M), cf((size_t)n * M);
  rc = qinco_encode_host(h, x.data(), QINCO_X_F32, 0, n, cs.data(), QINCO_CODE_I32, xs.data(), QINCO_FLAG_NORMALISED);
  const bool range = rc == QINCO_ERR_RANGE;
  if (rc && !range) { qinco_destroy(twin); return rc; }
  rc = qinco_encode_host(twin, x.data(), QINCO_X_F32, 0, n, cf.data(), QINCO_CODE_I32, xf.data(), QINCO_FLAG_NORMALISED);
  qinco_destroy(twin);
  if (rc) return rc;
  qinco_split_report& r = h->calib;
  r.calibrated = 1;
  r.calib_vectors = n;
  r.calib_rows_differing = 0;
  double scale = 0.0, worst = 0.0;
  for (float v : xf) scale = std::fmax(scale, std::fabs((double)v));
  bool finite = !range;
  for (int i = 0; i < n; ++i) {
    const bool same = std::memcmp(&cs[(size_t)i * M], &cf[(size_t)i * M], (size_t)M * sizeof(int)) == 0;
    if (!same) { r.calib_rows_differing++; continue; }
    for (int j = 0; j < D; ++j) {
      const double e = std::fabs((double)xs[(size_t)i * D + j] - (double)xf[(size_t)i * D + j]);
      if (!(e == e) || !std::isfinite(xs[(size_t)i * D + j])) finite = false;
      else worst = std::fmax(worst, e);
    }
  }
  r.calib_max_rel_err = scale > 0.0 ? (float)(worst / scale) : 0.f;
  (void)hipMemset(h->split_stats, 0, 2 * sizeof(unsigned long long));   // the counters report production work only
  if (!finite)
    return fail(QINCO_ERR_RANGE, "qinco_create: the split-fp16 form overflows on this model (calibration batch of %d vectors around its "
                "codebooks): use the fp32 path", n);
  if (r.calib_max_rel_err > 1e-4f || r.calib_rows_differing > n / 20)
    return fail(QINCO_ERR_RANGE, "qinco_create: the split-fp16 form is not in the fp32 path's error class on this model (calibration: %d of "
                "%d code rows differ, reconstructions differ by %.2e relative): use the fp32 path", r.calib_rows_differing, n,
                (double)r.calib_max_rel_err);
  return 0;
}

extern "C" int qinco_split_stats(qinco_handle h, qinco_split_report* out) {
  if (!h || !out) return fail(QINCO_ERR_INVALID, "qinco_split_stats: null argument");
  *out = h->calib;
  out->split_form = h->split16 ? 1 : 0;
  if (!h->split16) return QINCO_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long v[2] = {0, 0};
  HIP_TRY(hipMemcpy(v, h->split_stats, sizeof(v), hipMemcpyDeviceToHost));
  out->lo_sampled = (int64_t)v[0];
  out->lo_subnormal = (int64_t)v[1];
  out->overflowed = h->ever_overflowed ? 1 : 0;
  return QINCO_OK;
}

extern "C" int qinco_destroy(qinco_handle h) {
  if (!h) return QINCO_OK;
  (void)hipDeviceSynchronize();
  for (void* p : h->owned)
    if (p) (void)hipFree(p);
  host_pipe_destroy(h->pipe);
  if (h->scratch_done) (void)hipEventDestroy(h->scratch_done);
  for (auto& e : h->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  delete h;
  return QINCO_OK;
}

extern "C" int qinco_set_beam(qinco_handle h, int32_t A, int32_t B) {
  if (!h) return fail(QINCO_ERR_INVALID, "qinco_set_beam: null handle");
  if (B < 1) return fail(QINCO_ERR_INVALID, "qinco_set_beam: B must be >= 1");
  if (A < 0 || A > h->d.K) return fail(QINCO_ERR_INVALID, "qinco_set_beam: need 0 <= A <= K");
  if (A > 0 && h->d.A == 0)
    return fail(QINCO_ERR_INVALID,
                "Can't evaluate a model trained with A=0 (no candidates pre-selection) using a non-zero A value.");
  if (h->d.ivf_K > 0 && A > 0 && B > h->d.K)
    return fail(QINCO_ERR_INVALID, "qinco_set_beam: the first QINCo step of an IVF model pre-selects max(A, B) = %d of its K = %d codewords", B,
                h->d.K);
  h->A = A;
  h->B = B;
  return ensure_scratch(h);
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
static unsigned ew_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// One step's fused MLP over a.R rows.  FOLD: a.uproj names the scratch for U (encode: h->uproj, decode: h->duproj);
// it is filled here by xproj_kernel for the R/A groups of this launch, and T comes from the step's table.
struct PreselJob {   // the step's pre-selection, when it rides in the xproj launch (small launches, presel_kernel.hpp)
  const float* x = nullptr;
  int F = 1, T = 0;
  const f32x4* cstream = nullptr;
  const float* cnorm = nullptr;
  int* ids = nullptr;
};

// the fused pre-selection + xproj kernel serves this launch: K = 256 MFMA table, a folded fp32 instance whose block counts split
// over four waves, a launch small enough for the cooperative form
static bool presel_fused(const qinco_handle_s* h, long G) {
  return h->inst && h->table_coop && !h->table_valu && !h->split16 && h->fold && G <= h->table_coop_max && h->d.K == 256 &&
         mfma_table_ok(h->d, h->inst) && presel_coop_ok(h->d.De, h->d.Dh, h->inst->var) && !h->no_presel_fusion;
}

// Which form serves a launch of R rows: NT (row tiles of 16 per workgroup) of the small-launch form, or 0 = the 128-row kernels.
// Cost in "one 16-row tile on one CU": a launch takes ceil(workgroups / CUs) rounds; a round of the 128-row kernel costs 8 tiles, a
// round of the small form NT tiles plus its per-workgroup fixed part (ring prologue, head gathers, barriers).  Larger NT = fewer weight
// bytes per row, so ties go to the larger NT.
static int small_nt(const qinco_handle_s* h, long R, bool dec) {
  const int mx = h->small ? h->small_max_nt[dec ? 1 : 0] : 0;
  if (mx <= 0 || R <= 0) return 0;
  const long cus = h->num_cu;
  auto rounds = [&](long wgs) { return (double)((wgs + cus - 1) / cus); };
  double best_cost = rounds((R + 127) / 128) * 8.0 * 0.97;
  int best = 0;
  for (int nt = mx; nt >= 1; --nt) {
    const double c = rounds((R + 16 * nt - 1) / (16 * nt)) * (nt + 0.35);
    if (c < best_cost - 1e-9) {
      best_cost = c;
      best = nt;
    }
  }
  return best;
}

// position of step m's fragments in the small-form stream (f32x4 units); body = past the in-kernel head sections
static const f32x4* small_stream_at(const qinco_handle_s* h, int m, bool body) {
  const long frags = (long)(m - 1) * h->ssd.step(h->d.L) + (body ? h->ssd.head() : 0);
  return h->small_stream + frags * kSmallWaves * 64;
}

static int launch_mlp(qinco_handle_s* h, MlpArgs a, int m, hipStream_t st, bool decode = false, const PreselJob* pj = nullptr,
                      bool* did_select = nullptr) {
  const bool unfolded = decode && h->dec_inst;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->prof) {   // the bracket covers xproj + mlp: all the work the algorithmic FLOP count stands for
    if (h->ev_used == h->ev_pool.size()) {
      hipEvent_t a0, a1;
      HIP_TRY(hipEventCreate(&a0));
      HIP_TRY(hipEventCreate(&a1));
      h->ev_pool.emplace_back(a0, a1);
    }
    e0 = h->ev_pool[h->ev_used].first;
    e1 = h->ev_pool[h->ev_used].second;
    h->ev_used++;
    HIP_TRY(hipEventRecord(e0, st));
  }
  if (unfolded) a.wstream = h->dec_wstream[m];
  if (h->fold && !unfolded) {
    XprojArgs xa{};
    xa.wx = h->wx_stream[m];
    xa.xhat = a.xhat;
    xa.uproj = const_cast<float*>(a.uproj);
    xa.G = a.R / a.A;
    if (h->fold2) {  // Q lives behind U in the same scratch
      xa.wq = h->wq_stream[m];
      xa.qproj = xa.uproj + (size_t)xa.G * h->d.De;
      a.qproj = xa.qproj;
      a.ptab = h->ptab[m];
    }
    xa.smul = h->split16 ? h->xsmul[m] : nullptr;
    if (pj) {
      xa.x = pj->x;
      xa.F = pj->F;
      xa.cstream = pj->cstream;
      xa.cnorm = pj->cnorm;
      xa.T = pj->T;
      xa.ids_out = pj->ids;
    }
    HIP_TRY(h->inst->xproj(&xa, st));
    a.ttab = h->ttab[m];
  }
  a.smul = h->split16 ? h->smul[m] : nullptr;
  a.err = h->split16 ? h->err_flag : nullptr;
  a.stats = h->split16 ? h->split_stats : nullptr;
#ifdef QINCO_TIMELINE
  {
    const size_t tiles = (size_t)((a.R + 31) / 32 + 4);
    if (tiles > h->tl_cap) {
      if (h->tl) (void)hipFree(h->tl);
      HIP_TRY(hipMalloc(&h->tl, tiles * 8 * sizeof(unsigned long long)));
      h->tl_cap = tiles;
    }
    HIP_TRY(hipMemsetAsync(h->tl, 0, tiles * 8 * sizeof(unsigned long long), st));
    h->tl_tiles = tiles;
    a.timeline = h->tl;
  }
#endif
  const int nt = decode ? 0 : small_nt(h, a.R, false);
  if (nt > 0 || !h->sel_inst || decode) a.sel_T = 0;   // (the epilogue selection lives in the 128-row encode kernel)
  // the 128-row instance this launch takes; KHEAD only for groups of A >= 32 or A == 16 rows (a wave's 32 rows span at most two)
  const MlpInstance* ran = unfolded ? h->dec_inst : (a.sel_T > 0 ? h->sel_inst : h->inst);
  if (!unfolded && h->alt_inst) {
    const bool khead_ok = a.A >= 32 || a.A == 16;
    if (a.sel_T > 0) {
      if (h->sel_inst->var & 4096) {       // KHEAD + SELEP: only for the groups KHEAD takes, else the plain two-kernel form on the twin
        if (!khead_ok) {
          a.sel_T = 0;
          if (did_select) *did_select = false;
          ran = h->alt_inst;
          a.wstream = h->alt_wstream[m];
        }
      } else {
        a.wstream = h->alt_wstream[m];     // (the twin's SELEP instance reads the ob-outer stream)
      }
    } else if (!khead_ok) {
      ran = h->alt_inst;
      a.wstream = h->alt_wstream[m];
    }
  }
  if (did_select) *did_select = a.sel_T > 0;
  if (nt > 0) {   // small launch: workgroups of 16 nt rows, same head (T + U, relu(P + Q)) and the same products in the same order
    SmallArgs sa{};
    sa.wstream = small_stream_at(h, m, true);
    sa.steps = h->small_steps;
    sa.m_first = m;
    sa.m_count = 1;
    sa.L = a.L;
    sa.add_c = a.add_c;
    sa.R = a.R;
    sa.cand_ids = a.cand_ids;
    sa.A = a.A;
    sa.F = a.F;
    sa.xhat = a.xhat;
    sa.x = a.x;
    sa.uproj = a.uproj;
    sa.qproj = a.qproj;
    sa.cand_out = a.cand_out;
    sa.dist_out = a.dist_out;
    HIP_TRY(h->small(&sa, 0, nt, st));
  } else {
    HIP_TRY(ran->fn(&a, st));
  }
  if (h->prof) {
    HIP_TRY(hipEventRecord(e1, st));
    h->prof_flops += (double)a.R * mlp_flops_per_row(h);
    h->prof_flops_exec += mlp_flops_executed(h, (double)a.R, (double)(a.R / a.A), h->fold && !unfolded, h->fold2 && !unfolded,
                                             nt == 0 && (ran->var & 4096), nt == 0 && (ran->var & 4096) && a.sel_T == 0);
  }
  return 0;
}

static int launch_dist_topk(qinco_handle_s* h, const float* x, const float* xhat, int F, const float* cb,
                            const f32x4* cstream, const float* cn, long G, int T, int* ids, hipStream_t st) {
  const qinco_desc& d = h->d;
  if (cstream && !h->table_valu) {
    TableArgs ta{x, xhat, F, cstream, cn, G, T, ids, (G <= h->table_coop_max && h->table_coop) ? 1 : 0};
    hipError_t e;
    switch (builtin_table_dim(d.D) ? d.D : 0) {
      case 32: e = launch_table_kernels<32>(ta, st); break;
      case 96: e = launch_table_kernels<96>(ta, st); break;
      case 128: e = launch_table_kernels<128>(ta, st); break;
      case 256: e = launch_table_kernels<256>(ta, st); break;
      case 768: e = launch_table_kernels<768>(ta, st); break;
      default: e = h->inst->table(&ta, st); break;   // (cstream is only packed when mfma_table_ok: the module has a launcher)
    }
    HIP_TRY(e);
    return 0;
  }
  size_t lds = ((size_t)DT_TG * d.D + 256 * DT_CP + (size_t)DT_TG * d.K + DT_TG) * sizeof(float);
  if (lds > 160 * 1024) return fail(QINCO_ERR_UNSUPPORTED, "pre-selection table: D=%d, K=%d do not fit the 160 KiB of LDS", d.D, d.K);
  if (lds > 64 * 1024 && lds > h->table_lds_max) {   // above the default dynamic-LDS limit (wide D without an MFMA table instance)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dist_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->table_lds_max = lds;
  }
  unsigned grid = (unsigned)((G + DT_TG - 1) / DT_TG);
  hipLaunchKernelGGL(dist_topk_kernel, dim3(grid), dim3(256), lds, st, x, xhat, F, cb, cn, d.K, d.D, G, T, ids);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int D>
static void launch_ivf_inst(qinco_handle_s* h, long n, int nblocks, int bps, int slices, const int* only_if, hipStream_t st) {
  hipLaunchKernelGGL(ivf_assign_kernel<D>, dim3((unsigned)((n + 127) / 128), (unsigned)slices), dim3(256), 0, st,
                     h->ivf_stream, h->cnorm[0], nblocks, bps, h->xn, n, h->ivf_best, only_if);
}

static void ivf_grid(long tiles, int nblocks, int target_wgs, int* bps, long* slices);

template <int D>
static void launch_ivf_f16_inst(const IvfF16Args& a, long tiles, int slices, int sample_blocks, hipStream_t st) {
  IvfF16Args sa = a;   // pass A: the sample at the head of the stream
  sa.nblocks = sample_blocks;
  long ss;
  ivf_grid(tiles, sample_blocks, 2048, &sa.blocks_per_slice, &ss);
  hipLaunchKernelGGL((ivf_f16_kernel<D, 0>), dim3((unsigned)tiles, (unsigned)ss), dim3(256), 0, st, sa);
  hipLaunchKernelGGL((ivf_f16_kernel<D, 1>), dim3((unsigned)tiles, (unsigned)slices), dim3(256), 0, st, a);
}

static void ivf_grid(long tiles, int nblocks, int target_wgs, int* bps, long* slices) {
  long s = (target_wgs + tiles - 1) / tiles;
  if (s > nblocks) s = nblocks;
  if (s < 1) s = 1;
  *bps = (int)((nblocks + s - 1) / s);
  *slices = (nblocks + *bps - 1) / *bps;
}

// IVF step 0: codes0 = argmin over the ivf_K centroids (IVFBook.quantize, qinco_base.py:146-163) -> top_ids[n].
// fp16 filter passes A, B + exact pass C (ivf_f16_kernel.hpp), then the exact fp32 table kernel, which returns at once
// unless the filter raised its overflow flag; without the fp16 copy the fp32 kernel does the whole job.
static int launch_ivf_assign(qinco_handle_s* h, long n, hipStream_t st) {
  const qinco_desc& d = h->d;
  const int nblocks = d.ivf_K / 32;
  HIP_TRY(hipMemsetAsync(h->ivf_best, 0xFF, (size_t)n * sizeof(unsigned long long), st));
  const int* only_if = nullptr;
  if (h->ivf_f16) {
    HIP_TRY(hipMemsetAsync(h->ivf_amin, 0xFF, (size_t)n * sizeof(unsigned), st));
    HIP_TRY(hipMemsetAsync(h->ivf_cand, 0, 4 * sizeof(int), st));
    IvfF16Args a{};
    a.cstream = reinterpret_cast<const h16x8*>(h->ivf_h16);
    a.cnorm_half = h->ivf_cnorm_half;
    a.nblocks = nblocks;
    a.x = h->xn;
    a.N = n;
    a.approx_min = h->ivf_amin;
    a.cmax = h->ivf_cmax;
    a.cand_count = h->ivf_cand;
    a.overflow = h->ivf_cand + 1;
    a.cand_cap = h->ivf_cand_cap;
    a.cand_vec = h->ivf_cand + 4;
    a.cand_id = h->ivf_cand + 4 + h->ivf_cand_cap;
    a.perm = h->ivf_perm;
    const int vs = QINCO_IVF_VS(d.D);
    const long tiles = (n + 128 * vs - 1) / (128 * vs);
    long slices;
    ivf_grid(tiles, nblocks, 2048, &a.blocks_per_slice, &slices);
    switch (d.D) {
      case 32: launch_ivf_f16_inst<32>(a, tiles, (int)slices, h->ivf_sample_blocks, st); break;
      case 96: launch_ivf_f16_inst<96>(a, tiles, (int)slices, h->ivf_sample_blocks, st); break;
      case 128: launch_ivf_f16_inst<128>(a, tiles, (int)slices, h->ivf_sample_blocks, st); break;
      case 256: launch_ivf_f16_inst<256>(a, tiles, (int)slices, h->ivf_sample_blocks, st); break;
      case 768: launch_ivf_f16_inst<768>(a, tiles, (int)slices, h->ivf_sample_blocks, st); break;
      default: return fail(QINCO_ERR_UNSUPPORTED, "no IVF kernel instance for D=%d", d.D);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ivf_exact_kernel, dim3(512), dim3(256), 0, st, a.cand_count, a.cand_cap, a.overflow, a.cand_vec, a.cand_id, h->xn,
                       h->codebook[0], h->cnorm[0], d.D, h->ivf_best);
    HIP_TRY(hipGetLastError());
    only_if = a.overflow;
  }
  {
    const long tiles = (n + 127) / 128;
    int bps;
    long slices;
    ivf_grid(tiles, nblocks, 2048, &bps, &slices);  // aim at >= 2048 workgroups
    if (h->ivf_module) {
      IvfArgs ia{h->ivf_stream, h->cnorm[0], nblocks, bps, (int)slices, h->xn, n, h->ivf_best, only_if};
      if (h->ivf_module(&ia, st) != hipSuccess) return fail(QINCO_ERR_HIP, "IVF module launch failed");
    } else switch (d.D) {
      case 32: launch_ivf_inst<32>(h, n, nblocks, bps, (int)slices, only_if, st); break;
      case 96: launch_ivf_inst<96>(h, n, nblocks, bps, (int)slices, only_if, st); break;
      case 128: launch_ivf_inst<128>(h, n, nblocks, bps, (int)slices, only_if, st); break;
      case 256: launch_ivf_inst<256>(h, n, nblocks, bps, (int)slices, only_if, st); break;
      case 768: launch_ivf_inst<768>(h, n, nblocks, bps, (int)slices, only_if, st); break;
      default: return fail(QINCO_ERR_UNSUPPORTED, "no IVF kernel instance for D=%d", d.D);
    }
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(ivf_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->ivf_best, n, h->top_ids);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int encode_chunk(qinco_handle_s* h, const void* x, int x_dtype, int64_t stride, int64_t n, void* codes_out,
                        int code_dtype, float* xhat_out, int flags, hipStream_t st) {
  const qinco_desc& d = h->d;
  const int A = h->A, B = h->B, K = d.K, D = d.D, M = d.M;
  hipLaunchKernelGGL(normalize_kernel, dim3(ew_grid(n * D)), dim3(256), 0, st, x, x_dtype, (long)stride,
                     (flags & QINCO_FLAG_NORMALISED) ? (const float*)nullptr : h->mean, h->std_, h->xn, (long)n, h->user.D, D);
  HIP_TRY(hipGetLastError());
  // step 0: plain codebook, beam_0 = min(B, K0), or 1 with IVF (qinco_inference.py:237-246); a single-step model
  // ends at F = 1
  int F = (M == 1 || d.ivf_K > 0) ? 1 : (B < K ? B : K);
  int rc;
  if (d.ivf_K > 0) {
    if ((rc = launch_ivf_assign(h, n, st))) return rc;
  } else if ((rc = launch_dist_topk(h, h->xn, nullptr, 1, h->codebook[0], h->cb_stream[0], h->cnorm[0], n, F, h->top_ids, st))) {
    return rc;
  }
  int cur = 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(n * F * (D / 4))), dim3(256), 0, st, h->codebook[0], h->top_ids,
                     (long)n * F, D, h->xhat[cur], h->hist[cur], M);
  HIP_TRY(hipGetLastError());
  for (int m = 1; m < M; ++m) {
    const int Fout_cfg = (m < M - 1) ? B : 1;  // qinco_inference.py:152
    const int Am = n_codes(h, m) < K ? n_codes(h, m) : K;
    const int Ae = A > 0 ? Am : K;
    const long G = (long)n * F;
    const int* cand_ids = nullptr;
    PreselJob pj;
    const bool fused = A > 0 && presel_fused(h, G);
    if (fused) {   // small launch: the table + top-A ride in the xproj launch
      pj.x = h->xn;
      pj.F = F;
      pj.T = Am;
      pj.cstream = h->sub_stream[m];
      pj.cnorm = h->sub_cnorm[m];
      pj.ids = h->top_ids;
      cand_ids = h->top_ids;
    } else if (A > 0) {
      if ((rc = launch_dist_topk(h, h->xn, h->xhat[cur], F, h->sub_codebook[m], h->sub_stream[m], h->sub_cnorm[m], G, Am, h->top_ids,
                                 st)))
        return rc;
      cand_ids = h->top_ids;
    }
    MlpArgs a{};
    a.wstream = h->wstream[m];
    a.L = d.L;
    a.codebook = h->codebook[m];
    a.cand_ids = cand_ids;
    a.A = Ae;
    a.F = F;
    a.xhat = h->xhat[cur];
    a.x = h->xn;
    a.R = G * Ae;
    a.cand_out = h->cand;
    a.dist_out = h->dist;
    a.add_c = d.qinco1_mode ? 0 : 1;
    a.uproj = h->uproj;
    const int C = F * Ae;
    const int T = Fout_cfg < C ? Fout_cfg : C;
    // the step's per-vector top-T in the fused-MLP kernel's epilogue (SELEP instance): a vector's candidates inside one workgroup
    if (h->sel_inst && C <= 128 && 128 % C == 0) {
      a.sel_T = T;
      a.sel_m = m;
      a.sel_M = M;
      a.sel_hist_in = h->hist[cur];
      a.sel_hist_out = h->hist[cur ^ 1];
      a.sel_xhat_out = h->xhat[cur ^ 1];
    }
    bool selected = false;
    if ((rc = launch_mlp(h, a, m, st, false, fused ? &pj : nullptr, &selected))) return rc;
    if (!selected) {
      const size_t lds = (size_t)4 * (((C + T + 3) & ~3) + 2 * SEL_SURV) * sizeof(float);
      if (lds > 160 * 1024)
        return fail(QINCO_ERR_UNSUPPORTED, "beam_select: %d candidates per vector do not fit the 160 KiB of LDS", C);
      if (lds > 64 * 1024 && lds > h->beam_lds_max) {   // above the default dynamic-LDS limit: raise it once
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(beam_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        h->beam_lds_max = lds;
      }
      hipLaunchKernelGGL(beam_select_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, st, h->dist, h->cand, cand_ids,
                         (long)n, F, Ae, D, T, m, M, h->hist[cur], h->hist[cur ^ 1], h->xhat[cur ^ 1]);
      HIP_TRY(hipGetLastError());
    }
    cur ^= 1;
    F = T;
  }
  hipLaunchKernelGGL(emit_codes_kernel, dim3(ew_grid(n * M)), dim3(256), 0, st, h->hist[cur], (long)n, F, M, codes_out,
                     code_dtype);
  HIP_TRY(hipGetLastError());
  if (xhat_out) {
    // F == 1 here for M > 1; for M == 1 beam 0 of each vector
    HIP_TRY(hipMemcpy2DAsync(xhat_out, (size_t)h->user.D * 4, h->xhat[cur], (size_t)F * D * 4, (size_t)h->user.D * 4, (size_t)n,
                             hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

static size_t code_size(int dt) { return dt == QINCO_CODE_I64 ? 8 : dt == QINCO_CODE_I32 ? 4 : 1; }

static int check_common(qinco_handle h, const void* a, const void* b, int64_t n, int code_dtype, const char* who) {
  if (!h) return fail(QINCO_ERR_INVALID, "%s: null handle", who);
  if (n < 0) return fail(QINCO_ERR_INVALID, "%s: n < 0", who);
  if (n > 0 && (!a || !b)) return fail(QINCO_ERR_INVALID, "%s: null buffer", who);
  if (code_dtype < 0 || code_dtype > 2) return fail(QINCO_ERR_INVALID, "%s: bad code dtype %d", who, code_dtype);
  if (code_dtype == QINCO_CODE_U8 && (h->d.K > 256 || h->user.ivf_K > 256))
    return fail(QINCO_ERR_INVALID, "%s: uint8 codes need K <= 256 (and no IVF column)", who);
  return 0;
}

static int scratch_enter(qinco_handle_s* h, hipStream_t st) {
  if (h->scratch_used && h->scratch_stream != st) HIP_TRY(hipStreamWaitEvent(st, h->scratch_done, 0));
  return 0;
}
static int scratch_leave(qinco_handle_s* h, hipStream_t st) {
  if (!h->scratch_done) HIP_TRY(hipEventCreateWithFlags(&h->scratch_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(h->scratch_done, st));
  h->scratch_stream = st;
  h->scratch_used = true;
  return 0;
}

// (the host forms: whatever way they return -- an error between two passes included -- the work they enqueued on their private
// stream is recorded as the scratch's last user, so the next call on another stream waits for it)
struct ScratchLeaveOnExit {
  qinco_handle_s* h;
  hipStream_t st;
  ~ScratchLeaveOnExit() { (void)scratch_leave(h, st); }
};

extern "C" int qinco_encode(qinco_handle h, const void* x, int x_dtype, int64_t stride, int64_t n, void* codes_out,
                            int code_dtype, float* xhat_out, int flags, void* stream) {
  int rc = check_common(h, x, codes_out, n, code_dtype, "qinco_encode");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));  // a handle is bound to the device it was created on
  if (x_dtype != QINCO_X_F32 && x_dtype != QINCO_X_U8) return fail(QINCO_ERR_INVALID, "qinco_encode: bad x dtype %d", x_dtype);
  const size_t esz = x_dtype == QINCO_X_F32 ? 4 : 1;
  if (stride == 0) stride = (int64_t)(h->user.D * esz);
  if (stride < (int64_t)(h->user.D * esz)) return fail(QINCO_ERR_INVALID, "qinco_encode: row stride smaller than a row");
  if ((rc = ensure_scratch(h))) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n > 0 && (rc = scratch_enter(h, st))) return rc;
  for (int64_t i0 = 0; i0 < n; i0 += h->d.max_batch) {
    int64_t nb = n - i0 < h->d.max_batch ? n - i0 : h->d.max_batch;
    const char* xp = reinterpret_cast<const char*>(x) + i0 * stride;
    char* cp = reinterpret_cast<char*>(codes_out) + (size_t)i0 * h->d.M * code_size(code_dtype);
    float* xo = xhat_out ? xhat_out + (size_t)i0 * h->user.D : nullptr;
    if ((rc = encode_chunk(h, xp, x_dtype, stride, nb, cp, code_dtype, xo, flags, st))) {
      (void)scratch_leave(h, st);   // (what was launched before the failure still owns the scratch)
      return rc;
    }
  }
  return n > 0 ? scratch_leave(h, st) : QINCO_OK;
}

static int decode_chunk(qinco_handle_s* h, const void* codes, int code_dtype, int64_t n, float* out, int flags,
                        hipStream_t st) {
  const qinco_desc& d = h->d;
  const int D = d.D, M = d.M;
  hipLaunchKernelGGL(import_codes_kernel, dim3(ew_grid(n * M)), dim3(256), 0, st, codes, code_dtype, (long)n, M, h->kvals,
                     h->codes_t, h->err_flag);
  HIP_TRY(hipGetLastError());
  if (const int nt = M > 1 ? small_nt(h, n, true) : 0) {
    // small call (the reference decodes 1024 / 12 288 rows per call): every step of a row tile in ONE launch of the small form --
    // xhat stays in the workgroup, the head is computed in the kernel in the folded association (T[code] + W_x xhat)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
      if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t a0, a1;
        HIP_TRY(hipEventCreate(&a0));
        HIP_TRY(hipEventCreate(&a1));
        h->ev_pool.emplace_back(a0, a1);
      }
      e0 = h->ev_pool[h->ev_used].first;
      e1 = h->ev_pool[h->ev_used].second;
      h->ev_used++;
      HIP_TRY(hipEventRecord(e0, st));
    }
    SmallArgs sa{};
    sa.wstream = small_stream_at(h, 1, false);
    sa.steps = h->small_steps;
    sa.m_first = 1;
    sa.m_count = M - 1;
    sa.L = d.L;
    sa.add_c = d.qinco1_mode ? 0 : 1;
    sa.R = n;
    sa.codes_t = h->codes_t;
    sa.codebook0 = h->codebook[0];
    sa.out = out;
    sa.mean = (flags & QINCO_FLAG_NORMALISED) ? nullptr : h->mean;
    sa.std_ = h->std_;
    sa.Duser = h->user.D;
    HIP_TRY(h->small(&sa, 1, nt, st));
    if (h->prof) {
      HIP_TRY(hipEventRecord(e1, st));
      h->prof_flops += (double)n * (M - 1) * mlp_flops_per_row(h);
      h->prof_flops_exec += mlp_flops_executed(h, (double)n * (M - 1), (double)n * (M - 1), true, h->fold2);
    }
    return 0;
  }
  int cur = 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(n * (D / 4))), dim3(256), 0, st, h->codebook[0], h->codes_t, (long)n, D,
                     h->dxhat[cur], (int*)nullptr, M);
  HIP_TRY(hipGetLastError());
  for (int m = 1; m < M; ++m) {
    MlpArgs a{};
    a.wstream = h->wstream[m];
    a.L = d.L;
    a.codebook = h->codebook[m];
    a.cand_ids = h->codes_t + (size_t)m * n;
    a.A = 1;
    a.F = 1;
    a.xhat = h->dxhat[cur];
    a.x = nullptr;
    a.R = n;
    a.cand_out = h->dxhat[cur ^ 1];  // xhat += f_m(c, xhat)  (qinco_inference.py:72-74)
    a.dist_out = nullptr;
    a.add_c = d.qinco1_mode ? 0 : 1;
    a.uproj = h->duproj;
    int rc = launch_mlp(h, a, m, st, /*decode=*/true);
    if (rc) return rc;
    cur ^= 1;
  }
  hipLaunchKernelGGL(denormalize_kernel, dim3(ew_grid(n * D)), dim3(256), 0, st, h->dxhat[cur],
                     (flags & QINCO_FLAG_NORMALISED) ? (const float*)nullptr : h->mean, h->std_, out, (long)n, h->user.D, D);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int qinco_decode(qinco_handle h, const void* codes, int code_dtype, int64_t n, float* out, int flags,
                            void* stream) {
  int rc = check_common(h, codes, out, n, code_dtype, "qinco_decode");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  if ((rc = ensure_decode_scratch(h, n))) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n > 0 && (rc = scratch_enter(h, st))) return rc;
  for (int64_t i0 = 0; i0 < n; i0 += h->dec_cap) {
    int64_t nb = n - i0 < h->dec_cap ? n - i0 : h->dec_cap;
    const char* cp = reinterpret_cast<const char*>(codes) + (size_t)i0 * h->d.M * code_size(code_dtype);
    if ((rc = decode_chunk(h, cp, code_dtype, nb, out + (size_t)i0 * h->user.D, flags, st))) {
      (void)scratch_leave(h, st);
      return rc;
    }
  }
  return n > 0 ? scratch_leave(h, st) : QINCO_OK;
}

// ---------------------------------------------------------------------------------------------
// host-pointer forms
// ---------------------------------------------------------------------------------------------
static int ensure_stage(void** p, size_t* cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  HIP_TRY(hipMalloc(p, need ? need : 16));
  *cap = need;
  return 0;
}

static int check_decode_range(qinco_handle_s* h);

// rows of `rowb` bytes, `stride` apart -> packed
static void pack_rows(void* dst, const void* src, size_t rowb, size_t stride, int64_t rows) {
  if (stride == rowb) {
    std::memcpy(dst, src, rowb * (size_t)rows);
    return;
  }
  for (int64_t r = 0; r < rows; ++r)
    std::memcpy(reinterpret_cast<char*>(dst) + (size_t)r * rowb, reinterpret_cast<const char*>(src) + (size_t)r * stride, rowb);
}

extern "C" int qinco_encode_host(qinco_handle h, const void* x, int x_dtype, int64_t stride, int64_t n, void* codes_out,
                                 int code_dtype, float* xhat_out, int flags) {
  int rc = check_common(h, x, codes_out, n, code_dtype, "qinco_encode_host");
  if (rc) return rc;
  if (x_dtype != QINCO_X_F32 && x_dtype != QINCO_X_U8) return fail(QINCO_ERR_INVALID, "qinco_encode_host: bad x dtype");
  if (n == 0) return QINCO_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t esz = x_dtype == QINCO_X_F32 ? 4 : 1;
  const size_t rowb = (size_t)h->user.D * esz;
  if (stride == 0) stride = (int64_t)rowb;
  if (stride < (int64_t)rowb) return fail(QINCO_ERR_INVALID, "qinco_encode_host: row stride smaller than a row");
  // passes of max_batch rows, so a whole database (search_tasks.py:107-116 feeds 1e9 rows) never has to fit the device twice
  const int64_t pass = h->d.max_batch;
  const size_t crow = (size_t)h->d.M * code_size(code_dtype), orow = (size_t)h->user.D * 4;
  const int64_t cap = n < pass ? n : pass;
  if ((rc = ensure_scratch(h))) return rc;
  const size_t need[3] = {(size_t)cap * rowb, (size_t)cap * crow, xhat_out ? (size_t)cap * orow : 0};
  if ((rc = host_pipe_ensure(&h->pipe, need))) return rc;
  HostPipe& p = *h->pipe;
  // the private compute stream is non-blocking: it is ordered behind earlier device-pointer calls on this handle (torch's stream,
  // the null stream) explicitly -- they work in the same scratch and raise the same flag
  if ((rc = scratch_enter(h, p.s_comp))) return rc;
  ScratchLeaveOnExit leave{h, p.s_comp};
  // this call reports its own work only: a flag left behind by an unchecked device-pointer call is dropped
  if (h->split16) HIP_TRY(hipMemsetAsync(h->err_flag, 0, sizeof(int), p.s_comp));
  const int64_t P = (n + pass - 1) / pass;
  auto rows_of = [&](int64_t k) { return k * pass + pass <= n ? pass : n - k * pass; };
  auto stage_in = [&](int64_t k) -> int {   // pack pass k into its pinned buffer, start its copy to the device
    const int b = (int)(k & 1);
    if (k >= 2) HIP_TRY(hipEventSynchronize(p.in_ready[b]));           // the copy of pass k - 2 out of this host buffer has finished
    pack_rows(p.hbuf[0][b], reinterpret_cast<const char*>(x) + (size_t)(k * pass) * (size_t)stride, rowb, (size_t)stride, rows_of(k));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(p.s_in, p.comp_done[b], 0));   // ... and the kernels of pass k - 2 have read dx[b]
    HIP_TRY(hipMemcpyAsync(p.dbuf[0][b], p.hbuf[0][b], (size_t)rows_of(k) * rowb, hipMemcpyHostToDevice, p.s_in));
    HIP_TRY(hipEventRecord(p.in_ready[b], p.s_in));
    return 0;
  };
  auto finish = [&](int64_t k) -> int {     // the codes (and xhat) of pass k have reached their pinned buffers: hand them to the caller
    const int b = (int)(k & 1);
    HIP_TRY(hipEventSynchronize(p.out_ready[b]));
    std::memcpy(reinterpret_cast<char*>(codes_out) + (size_t)(k * pass) * crow, p.hbuf[1][b], (size_t)rows_of(k) * crow);
    if (xhat_out) std::memcpy(xhat_out + (size_t)(k * pass) * h->user.D, p.hbuf[2][b], (size_t)rows_of(k) * orow);
    return 0;
  };
  if ((rc = stage_in(0))) return rc;
  for (int64_t k = 0; k < P; ++k) {
    const int b = (int)(k & 1);
    const int64_t nb = rows_of(k);
    HIP_TRY(hipStreamWaitEvent(p.s_comp, p.in_ready[b], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(p.s_comp, p.out_ready[b], 0));   // the copy-out of pass k - 2 has read dc[b]
    if ((rc = qinco_encode(h, p.dbuf[0][b], x_dtype, 0, nb, p.dbuf[1][b], code_dtype, xhat_out ? (float*)p.dbuf[2][b] : nullptr, flags,
                           p.s_comp)))
      return rc;
    HIP_TRY(hipEventRecord(p.comp_done[b], p.s_comp));
    HIP_TRY(hipStreamWaitEvent(p.s_out, p.comp_done[b], 0));
    if (k >= 2) HIP_TRY(hipEventSynchronize(p.out_ready[b]));   // (finish(k - 2) ran already: the host buffer is free; keeps the order explicit)
    HIP_TRY(hipMemcpyAsync(p.hbuf[1][b], p.dbuf[1][b], (size_t)nb * crow, hipMemcpyDeviceToHost, p.s_out));
    if (xhat_out) HIP_TRY(hipMemcpyAsync(p.hbuf[2][b], p.dbuf[2][b], (size_t)nb * orow, hipMemcpyDeviceToHost, p.s_out));
    HIP_TRY(hipEventRecord(p.out_ready[b], p.s_out));
    if (k + 1 < P && (rc = stage_in(k + 1))) return rc;   // ... under the kernels of pass k
    if (k >= 1 && (rc = finish(k - 1))) return rc;
  }
  if ((rc = finish(P - 1))) return rc;
  if (!h->split16) return QINCO_OK;
  HIP_TRY(hipStreamSynchronize(p.s_comp));
  return check_decode_range(h);   // (the split form's overflow flag)
}

// The range flag is sticky on the device: import_codes_kernel raises it (and decodes the offending code as 0), the
// next qinco_check / qinco_decode_host reads and clears it.
static int check_decode_range(qinco_handle_s* h) {
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIP_TRY(hipMemset(h->err_flag, 0, sizeof(int)));
    if (flag == 2) h->ever_overflowed = true;
    if (flag == 2)
      return fail(QINCO_ERR_RANGE, "split-fp16 form: an activation of the codeword MLP left the fp16 range (or the input is not finite); "
                                   "create the handle without QINCO_CREATE_SPLIT_F16 for this model / data");
    return fail(QINCO_ERR_RANGE, "qinco_decode: a code is outside [0, K)");
  }
  return 0;
}

extern "C" int qinco_check(qinco_handle h, void* stream) {
  if (!h) return fail(QINCO_ERR_INVALID, "qinco_check: null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  return check_decode_range(h);
}

extern "C" int qinco_decode_host(qinco_handle h, const void* codes, int code_dtype, int64_t n, float* out, int flags) {
  int rc = check_common(h, codes, out, n, code_dtype, "qinco_decode_host");
  if (rc) return rc;
  if (n == 0) return QINCO_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t crow = (size_t)h->d.M * code_size(code_dtype);
  const size_t orow = (size_t)h->user.D * 4;
  const int64_t pass = kDecodeChunk;
  const int64_t cap = n < pass ? n : pass;
  const size_t need[3] = {0, (size_t)cap * crow, (size_t)cap * orow};
  if ((rc = host_pipe_ensure(&h->pipe, need))) return rc;
  if ((rc = ensure_decode_scratch(h, cap))) return rc;
  HostPipe& p = *h->pipe;
  if ((rc = scratch_enter(h, p.s_comp))) return rc;   // (see qinco_encode_host)
  ScratchLeaveOnExit leave{h, p.s_comp};
  // this call reports its own codes only: a flag left behind by an unchecked device-pointer decode is dropped
  HIP_TRY(hipMemsetAsync(h->err_flag, 0, sizeof(int), p.s_comp));
  const int64_t P = (n + pass - 1) / pass;
  auto rows_of = [&](int64_t k) { return k * pass + pass <= n ? pass : n - k * pass; };
  auto stage_in = [&](int64_t k) -> int {
    const int b = (int)(k & 1);
    if (k >= 2) HIP_TRY(hipEventSynchronize(p.in_ready[b]));
    std::memcpy(p.hbuf[1][b], reinterpret_cast<const char*>(codes) + (size_t)(k * pass) * crow, (size_t)rows_of(k) * crow);
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(p.s_in, p.comp_done[b], 0));
    HIP_TRY(hipMemcpyAsync(p.dbuf[1][b], p.hbuf[1][b], (size_t)rows_of(k) * crow, hipMemcpyHostToDevice, p.s_in));
    HIP_TRY(hipEventRecord(p.in_ready[b], p.s_in));
    return 0;
  };
  auto finish = [&](int64_t k) -> int {
    const int b = (int)(k & 1);
    HIP_TRY(hipEventSynchronize(p.out_ready[b]));
    std::memcpy(out + (size_t)(k * pass) * h->user.D, p.hbuf[2][b], (size_t)rows_of(k) * orow);
    return 0;
  };
  if ((rc = stage_in(0))) return rc;
  for (int64_t k = 0; k < P; ++k) {
    const int b = (int)(k & 1);
    const int64_t nb = rows_of(k);
    HIP_TRY(hipStreamWaitEvent(p.s_comp, p.in_ready[b], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(p.s_comp, p.out_ready[b], 0));
    if ((rc = qinco_decode(h, p.dbuf[1][b], code_dtype, nb, (float*)p.dbuf[2][b], flags, p.s_comp))) return rc;
    HIP_TRY(hipEventRecord(p.comp_done[b], p.s_comp));
    HIP_TRY(hipStreamWaitEvent(p.s_out, p.comp_done[b], 0));
    HIP_TRY(hipMemcpyAsync(p.hbuf[2][b], p.dbuf[2][b], (size_t)nb * orow, hipMemcpyDeviceToHost, p.s_out));
    HIP_TRY(hipEventRecord(p.out_ready[b], p.s_out));
    if (k + 1 < P && (rc = stage_in(k + 1))) return rc;
    if (k >= 1 && (rc = finish(k - 1))) return rc;
  }
  if ((rc = finish(P - 1))) return rc;
  return check_decode_range(h);
}

// ---------------------------------------------------------------------------------------------
// profiling / accounting
// ---------------------------------------------------------------------------------------------
extern "C" int qinco_profile_enable(qinco_handle h, int enable) {
  if (!h) return fail(QINCO_ERR_INVALID, "qinco_profile_enable: null handle");
  h->prof = enable != 0;
  return QINCO_OK;
}

extern "C" int qinco_profile_read2(qinco_handle h, double* mlp_ms, int64_t* mlp_launches, double* mlp_flops, double* mlp_flops_executed_out) {
  if (!h) return fail(QINCO_ERR_INVALID, "qinco_profile_read: null handle");
  HIP_TRY(hipDeviceSynchronize());
  double ms = 0.0;
  for (size_t i = 0; i < h->ev_used; ++i) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second));
    ms += t;
  }
  if (mlp_ms) *mlp_ms = ms;
  if (mlp_launches) *mlp_launches = (int64_t)h->ev_used;
  if (mlp_flops) *mlp_flops = h->prof_flops;
  if (mlp_flops_executed_out) *mlp_flops_executed_out = h->prof_flops_exec;
  h->ev_used = 0;
  h->prof_flops = 0.0;
  h->prof_flops_exec = 0.0;
  return QINCO_OK;
}

extern "C" int qinco_profile_read(qinco_handle h, double* mlp_ms, int64_t* mlp_launches, double* mlp_flops) {
  return qinco_profile_read2(h, mlp_ms, mlp_launches, mlp_flops, nullptr);
}

extern "C" double qinco_flops_per_vector_encode(qinco_handle h) {
  if (!h) return 0.0;
  const qinco_desc& d = h->user;
  const double rm = mlp_flops_per_row(h);
  double total = 2.0 * d.D * (d.ivf_K > 0 ? d.ivf_K : d.K);  // step 0 table
  int F = (d.M == 1 || d.ivf_K > 0) ? 1 : (h->B < d.K ? h->B : d.K);
  for (int m = 1; m < d.M; ++m) {
    const double Ae = h->A > 0 ? n_codes(h, m) : d.K;
    total += F * Ae * rm;                              // MLP
    if (h->A > 0) total += (double)F * d.K * 2.0 * d.D;  // pre-selection table
    total += F * Ae * 2.0 * d.D;                        // candidate distances
    int Fout = (m < d.M - 1) ? h->B : 1;
    F = Fout < F * Ae ? Fout : (int)(F * Ae);
  }
  return total;
}

extern "C" int qinco_describe(qinco_handle h, char* buf, int32_t cap) {
  if (!h) return fail(QINCO_ERR_INVALID, "qinco_describe: null handle");
  char tmp[512];
  const MlpInstance* i = h->inst;
  const int n = snprintf(tmp, sizeof(tmp), "model=%dx%dx%d mlp=%dx%dx%d P=%d var=%d decode_var=%d form=%s tile=%d table=%s ivf=%s", h->user.D,
                         h->user.De, h->user.Dh, h->d.D, h->d.De, h->d.Dh, i ? i->P : 0, i ? i->var : -1, h->dec_inst ? h->dec_inst->var : -1, h->split16 ? "split-fp16" : "fp32",
                         (i && (i->var & 128)) ? 16 : 32, (mfma_table_ok(h->d, h->inst) && !h->table_valu) ? "mfma" : "valu",
                         h->d.ivf_K == 0 ? "none" : (h->ivf_f16 ? "fp16-filter+fp32" : "fp32"));
  if (buf && cap > 0) {
    strncpy(buf, tmp, (size_t)cap - 1);
    buf[cap - 1] = 0;
  }
  return n + 1;
}

extern "C" int qinco_ivf_last_stats(qinco_handle h, int64_t* candidates, int32_t* fell_back) {
  if (!h || !candidates || !fell_back) return fail(QINCO_ERR_INVALID, "qinco_ivf_last_stats: null argument");
  *candidates = 0;
  *fell_back = 0;
  if (!h->ivf_f16 || !h->ivf_cand) return QINCO_OK;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  int v[2] = {0, 0};
  HIP_TRY(hipMemcpy(v, h->ivf_cand, sizeof(v), hipMemcpyDeviceToHost));
  *candidates = v[0] < h->ivf_cand_cap ? v[0] : h->ivf_cand_cap;
  *fell_back = v[1] != 0;
  return QINCO_OK;
}

extern "C" double qinco_flops_per_vector_decode(qinco_handle h) {
  if (!h) return 0.0;
  return (double)(h->d.M - 1) * mlp_flops_per_row(h);
}

// ---------------------------------------------------------------------------------------------
// look-up decoders (f4)
// ---------------------------------------------------------------------------------------------
struct qinco_lut_s {
  int device = 0;
  LutArgs a{};
  float* tables = nullptr;
  int* err_flag = nullptr;
  void* stage_codes = nullptr;
  size_t stage_codes_bytes = 0;
  float* stage_out = nullptr;
  size_t stage_out_bytes = 0;
};

extern "C" int qinco_lut_create(const float* tables, int32_t J, int64_t Kt, int32_t D, const int32_t* a, const int32_t* b,
                                int64_t mul, qinco_lut* out) {
  if (!tables || !a || !out) return fail(QINCO_ERR_INVALID, "qinco_lut_create: null argument");
  if (J < 1 || J > kLutMaxJ || Kt < 1 || D < 4 || D % 4 || mul < 1)
    return fail(QINCO_ERR_INVALID, "qinco_lut_create: need 1 <= J <= %d, Kt >= 1, D %% 4 == 0, mul >= 1", kLutMaxJ);
  qinco_lut_s* l = new qinco_lut_s();
  auto bail = [&](int code) {
    qinco_lut_destroy(l);
    return code;
  };
  if (hipGetDevice(&l->device) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipGetDevice failed (no HIP device?)"));
  const size_t bytes = (size_t)J * Kt * D * sizeof(float);
  if (hipMalloc((void**)&l->tables, bytes) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipMalloc(%zu) failed", bytes));
  if (hipMemcpy(l->tables, tables, bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(QINCO_ERR_HIP, "hipMemcpy failed"));
  if (hipMalloc((void**)&l->err_flag, sizeof(int)) != hipSuccess || hipMemset(l->err_flag, 0, sizeof(int)) != hipSuccess)
    return bail(fail(QINCO_ERR_HIP, "hipMalloc failed"));
  l->a.tables = l->tables;
  l->a.J = J;
  l->a.D = D;
  l->a.Kt = Kt;
  l->a.mul = mul;
  l->a.err_flag = l->err_flag;
  for (int j = 0; j < J; ++j) {
    l->a.a[j] = a[j];
    l->a.b[j] = b ? b[j] : -1;
    if (a[j] < 0) return bail(fail(QINCO_ERR_INVALID, "qinco_lut_create: a[%d] < 0", j));
  }
  *out = l;
  return QINCO_OK;
}

extern "C" int qinco_lut_destroy(qinco_lut l) {
  if (!l) return QINCO_OK;
  (void)hipDeviceSynchronize();
  if (l->tables) (void)hipFree(l->tables);
  if (l->err_flag) (void)hipFree(l->err_flag);
  if (l->stage_codes) (void)hipFree(l->stage_codes);
  if (l->stage_out) (void)hipFree(l->stage_out);
  delete l;
  return QINCO_OK;
}

extern "C" int qinco_lut_decode(qinco_lut l, const void* codes, int code_dtype, int32_t Mc, int64_t n, float* out, void* stream) {
  if (!l) return fail(QINCO_ERR_INVALID, "qinco_lut_decode: null handle");
  if (n < 0 || code_dtype < 0 || code_dtype > 2 || Mc < 1) return fail(QINCO_ERR_INVALID, "qinco_lut_decode: bad argument");
  if (n == 0) return QINCO_OK;
  if (!codes || !out) return fail(QINCO_ERR_INVALID, "qinco_lut_decode: null buffer");
  for (int j = 0; j < l->a.J; ++j)
    if (l->a.a[j] >= Mc || l->a.b[j] >= Mc) return fail(QINCO_ERR_INVALID, "qinco_lut_decode: code column out of range (Mc=%d)", Mc);
  HIP_TRY(hipSetDevice(l->device));
  LutArgs a = l->a;
  a.codes = codes;
  a.code_dtype = code_dtype;
  a.Mc = Mc;
  a.n = n;
  a.out = out;
  hipLaunchKernelGGL(lut_decode_kernel, dim3(ew_grid(n * (a.D / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return QINCO_OK;
}

extern "C" int qinco_lut_decode_host(qinco_lut l, const void* codes, int code_dtype, int32_t Mc, int64_t n, float* out) {
  if (!l) return fail(QINCO_ERR_INVALID, "qinco_lut_decode_host: null handle");
  if (n == 0) return QINCO_OK;
  if (!codes || !out || code_dtype < 0 || code_dtype > 2 || Mc < 1) return fail(QINCO_ERR_INVALID, "qinco_lut_decode_host: bad argument");
  HIP_TRY(hipSetDevice(l->device));
  const size_t cb = (size_t)n * Mc * code_size(code_dtype), ob = (size_t)n * l->a.D * 4;
  int rc;
  if ((rc = ensure_stage(&l->stage_codes, &l->stage_codes_bytes, cb))) return rc;
  if ((rc = ensure_stage((void**)&l->stage_out, &l->stage_out_bytes, ob))) return rc;
  HIP_TRY(hipMemcpy(l->stage_codes, codes, cb, hipMemcpyHostToDevice));
  if ((rc = qinco_lut_decode(l, l->stage_codes, code_dtype, Mc, n, l->stage_out, nullptr))) return rc;
  HIP_TRY(hipMemcpy(out, l->stage_out, ob, hipMemcpyDeviceToHost));
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, l->err_flag, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIP_TRY(hipMemset(l->err_flag, 0, sizeof(int)));
    return fail(QINCO_ERR_RANGE, "qinco_lut_decode: a look-up index is outside its table");
  }
  return QINCO_OK;
}

// ---------------------------------------------------------------------------------------------
// device self-test of the in-wave sorting / selection primitives (aux_kernels.hpp)
// ---------------------------------------------------------------------------------------------
extern "C" int qinco_selftest(void) {
  struct Lcg {
    unsigned long long s;
    unsigned next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); }
  } rng{12345};
  // 1. wave_sort64
  const int NW = 256;
  std::vector<unsigned> hin((size_t)NW * 64), hout(hin.size());
  for (size_t i = 0; i < hin.size(); ++i) hin[i] = (i / 64) % 3 == 0 ? rng.next() % 7 : rng.next();   // every third wave: many duplicates
  unsigned *din = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc(&din, hin.size() * 4));
  HIP_TRY(hipMalloc(&dout, hin.size() * 4));
  HIP_TRY(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(selftest_sort_kernel, dim3(NW), dim3(64), 0, nullptr, din, dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
  (void)hipFree(din);
  (void)hipFree(dout);
  for (int w = 0; w < NW; ++w) {
    std::vector<unsigned> want(hin.begin() + w * 64, hin.begin() + (w + 1) * 64);
    std::sort(want.begin(), want.end());
    for (int l = 0; l < 64; ++l)
      if (hout[(size_t)w * 64 + l] != want[l])
        return fail(QINCO_ERR_HIP, "qinco_selftest: wave_sort64 wrong at wave %d lane %d (%u, want %u)", w, l, hout[(size_t)w * 64 + l], want[l]);
  }
  // 2. wave_select_smallest: (C, T) pairs, continuous values / heavy ties / NaN and inf
  const int cases[][2] = {{256, 16}, {256, 32}, {256, 2}, {128, 8}, {512, 32}, {100, 7}, {64, 64}, {2048, 8}, {37, 37}, {8192, 32}, {256, 48}};
  for (const auto& ct : cases) {
    const int C = ct[0], T = ct[1], NP = 96;
    std::vector<float> hd((size_t)NP * C);
    for (int p = 0; p < NP; ++p)
      for (int k = 0; k < C; ++k) {
        float v;
        const int kind = p % 4;
        if (kind == 0) v = (float)(rng.next() % 100000) * 1e-3f - 20.f;          // continuous, both signs
        else if (kind == 1) v = (float)(rng.next() % 40);                         // heavy ties
        else if (kind == 2) v = (float)(rng.next() % 1000) * 0.5f;                // some ties
        else {
          const unsigned r = rng.next() % 50;
          v = r == 0 ? __builtin_nanf("") : r == 1 ? __builtin_inff() : r == 2 ? -0.f : (float)(rng.next() % 5000) * 1e-2f;
        }
        hd[(size_t)p * C + k] = v;
      }
    float* dd = nullptr;
    int *dids = nullptr, *dfb = nullptr;
    HIP_TRY(hipMalloc(&dd, hd.size() * 4));
    HIP_TRY(hipMalloc(&dids, (size_t)NP * T * 4));
    HIP_TRY(hipMalloc(&dfb, NP * 4));
    HIP_TRY(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dids, 0xFF, (size_t)NP * T * 4));
    const size_t lds = (size_t)(((C + 3) & ~3) + 2 * SEL_SURV) * 4;
    hipLaunchKernelGGL(selftest_select_kernel, dim3(NP), dim3(64), lds, nullptr, dd, C, T, dids, dfb);
    HIP_TRY(hipGetLastError());
    std::vector<int> hids((size_t)NP * T), hfb(NP);
    HIP_TRY(hipMemcpy(hids.data(), dids, hids.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hfb.data(), dfb, NP * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dd);
    (void)hipFree(dids);
    (void)hipFree(dfb);
    for (int p = 0; p < NP; ++p) {
      std::vector<int> order(C);
      for (int k = 0; k < C; ++k) order[k] = k;
      const float* row = &hd[(size_t)p * C];
      auto key = [&](int k) -> unsigned long long {   // the device's ordering: NaN last, -0 == +0, ties -> lower index
        const float v = row[k];
        unsigned u;
        if (v != v) u = 0xffffffffu;
        else {
          const float z = v + 0.f;
          std::memcpy(&u, &z, 4);
          u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        }
        return ((unsigned long long)u << 32) | (unsigned)k;
      };
      std::sort(order.begin(), order.end(), [&](int x, int y) { return key(x) < key(y); });
      if (hfb[p]) {
        // expected only for massive ties, or when more than 64 elements pass the threshold (T > 32: about 1.4 T survivors)
        if (p % 4 == 0 && T <= 32) return fail(QINCO_ERR_HIP, "qinco_selftest: selection fell back on continuous data (C=%d, T=%d)", C, T);
        continue;
      }
      for (int t = 0; t < T; ++t)
        if (hids[(size_t)p * T + t] != order[t])
          return fail(QINCO_ERR_HIP, "qinco_selftest: wave_select_smallest wrong (C=%d, T=%d, problem %d, rank %d: %d, want %d)", C, T, p, t,
                      hids[(size_t)p * T + t], order[t]);
    }
  }
  // 3. pair_top_t: the lane-pair selection of the pre-selection table kernels, every T they can be asked for, both list layouts;
  //    rows of continuous values, heavy / some / massive exact ties, NaN / inf / -0, constant rows, a row sorted the wrong way round
  {
    const int NT = 24, ROWS = NT * 32, K = 256;
    std::vector<float> hd((size_t)ROWS * K);
    for (int p = 0; p < ROWS; ++p)
      for (int k = 0; k < K; ++k) {
        float v;
        const int kind = p % 8;
        if (kind == 0 || kind == 5) v = (float)(rng.next() % 1000000) * 1e-4f - 20.f;     // continuous, both signs
        else if (kind == 1) v = (float)(rng.next() % 40);                                  // heavy ties
        else if (kind == 2) v = (float)(rng.next() % 1000) * 0.5f;                         // some ties
        else if (kind == 3) {
          const unsigned r = rng.next() % 50;
          v = r == 0 ? __builtin_nanf("") : r == 1 ? __builtin_inff() : r == 2 ? -0.f : r == 3 ? -__builtin_inff() : (float)(rng.next() % 5000) * 1e-2f;
        } else if (kind == 4) v = (p % 16 == 4) ? 3.25f : (k % 7 == 0 ? 1.f : 2.f);       // constant row / two values
        else if (kind == 6) v = (float)(K - k) + (p % 3 == 0 ? 0.f : 1e-3f * (float)(rng.next() % 100));   // descending in k
        else v = (rng.next() % 4 == 0) ? __builtin_nanf("") : (float)(rng.next() % 3);    // a quarter NaN, the rest from three values
        hd[(size_t)p * K + k] = v;
      }
    float* dd = nullptr;
    int* dids = nullptr;
    HIP_TRY(hipMalloc(&dd, hd.size() * 4));
    HIP_TRY(hipMalloc(&dids, (size_t)ROWS * 64 * 4));
    HIP_TRY(hipMemcpy(dd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    std::vector<std::vector<int>> order(ROWS, std::vector<int>(K));
    for (int p = 0; p < ROWS; ++p) {
      const float* row = &hd[(size_t)p * K];
      auto key = [&](int k) -> unsigned long long {   // NaN last, -0 == +0, ties -> lower index
        const float v = row[k];
        unsigned u;
        if (v != v) u = 0xffffffffu;
        else {
          const float z = v + 0.f;
          std::memcpy(&u, &z, 4);
          u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        }
        return ((unsigned long long)u << 32) | (unsigned)k;
      };
      for (int k = 0; k < K; ++k) order[p][k] = k;
      std::sort(order[p].begin(), order[p].end(), [&](int x, int y) { return key(x) < key(y); });
    }
    const int Ts[] = {16, 8, 32, 1, 2, 3, 4, 12, 15, 17, 24, 31, 33, 64};
    std::vector<int> hids((size_t)ROWS * 64), hrounds(ROWS);
    int* drounds = nullptr;
    HIP_TRY(hipMalloc(&drounds, ROWS * 4));
    {   // the exchange with lane ^ 32 itself (v_permlane32_swap): 32-bit and 64-bit
      hipLaunchKernelGGL(selftest_pair_kernel, dim3(1), dim3(64), 0, nullptr, dd, 0, dids, 0, drounds);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpy(hids.data(), dids, 128 * 4, hipMemcpyDeviceToHost));
      for (int l = 0; l < 64; ++l)
        if (hids[l] != (l ^ 32) + 1000 || hids[64 + l] != (int)((double)(l ^ 32) * 0.5 + 7.0)) {
          (void)hipFree(dd);
          (void)hipFree(dids);
          (void)hipFree(drounds);
          return fail(QINCO_ERR_HIP, "qinco_selftest: pair_partner hands lane %d the values %d / %d (want %d / %d)", l, hids[l], hids[64 + l],
                      (l ^ 32) + 1000, (int)((double)(l ^ 32) * 0.5 + 7.0));
        }
    }
    for (int coop = 0; coop < 2; ++coop)
      for (int T : Ts) {
        HIP_TRY(hipMemset(dids, 0xFF, (size_t)ROWS * T * 4));
        HIP_TRY(hipMemset(drounds, 0, ROWS * 4));
        hipLaunchKernelGGL(selftest_pair_kernel, dim3(coop ? ROWS / 8 : NT), dim3(64), 0, nullptr, dd, T, dids, coop, drounds);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(hids.data(), dids, (size_t)ROWS * T * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hrounds.data(), drounds, ROWS * 4, hipMemcpyDeviceToHost));
        // the fast path must be what runs on continuous data (kinds 0 and 5) for 2 <= T <= 16: a selection that silently went to
        // the exact rounds every time would pass every parity test at a tenth of the speed
        if (T >= 2 && T <= 16) {
          int went = 0, rows = 0;
          for (int p = 0; p < ROWS; ++p)
            if (p % 8 == 0 || p % 8 == 5) {
              ++rows;
              went += (hrounds[p] & 3) != 0;
            }
          if (went * 50 > rows) {
            (void)hipFree(dd);
            (void)hipFree(dids);
            (void)hipFree(drounds);
            return fail(QINCO_ERR_HIP, "qinco_selftest: pair_top_t sent %d of %d continuous rows to the rounds (T=%d, coop=%d)", went, rows, T, coop);
          }
        }
        for (int p = 0; p < ROWS; ++p)
          for (int t = 0; t < T; ++t)
            if (hids[(size_t)p * T + t] != order[p][t]) {
              (void)hipFree(dd);
              (void)hipFree(dids);
              return fail(QINCO_ERR_HIP, "qinco_selftest: pair_top_t wrong (T=%d, coop=%d, row %d of kind %d, rank %d: %d, want %d)", T, coop, p,
                          p % 8, t, hids[(size_t)p * T + t], order[p][t]);
            }
      }
    (void)hipFree(dd);
    (void)hipFree(dids);
    (void)hipFree(drounds);
  }
  return QINCO_OK;
}

// Diagnostics (scripts/exp_pair_select.py; not part of include/qinco_hip.h): the lane-pair selection alone on `rows` (a multiple of 32)
// rows of 256 distances in device memory -> ids (rows, T), rounds (rows): 1 where a row went to the exact arg-min rounds.
extern "C" __attribute__((visibility("default"))) int qinco_debug_pair_select(const float* d, long rows, int T, int coop, int* ids,
                                                                               int* rounds, void* stream) {
  if (!d || !ids || !rounds || rows <= 0 || rows % 32 || T < 1 || T > 64) return fail(QINCO_ERR_INVALID, "qinco_debug_pair_select: bad argument");
  hipLaunchKernelGGL(selftest_pair_kernel, dim3((unsigned)(coop ? rows / 8 : rows / 32)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), d, T,
                     ids, coop, rounds);
  HIP_TRY(hipGetLastError());
  return QINCO_OK;
}

#ifdef QINCO_TIMELINE
// experiment builds only: the cycle stamps of the handle's LAST fused-MLP launch, (tiles, 8) uint64; returns the tile count
extern "C" __attribute__((visibility("default"))) long qinco_debug_timeline(qinco_handle h, unsigned long long* out, long cap_tiles) {
  if (!h || !h->tl) return 0;
  (void)hipDeviceSynchronize();
  const long n = (long)h->tl_tiles < cap_tiles ? (long)h->tl_tiles : cap_tiles;
  if (out && hipMemcpy(out, h->tl, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long)h->tl_tiles;
}
#endif

extern "C" const char* qinco_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* qinco_version(void) { return "qinco_hip 0.1 (gfx950)"; }
