/*
 * qinco_hip.h -- C ABI of libqinco_hip.so, the MI355X (gfx950) QINCo / QINCo2 encode-decode engine.
 *
 * The reference (facebookresearch/Qinco) is pure Python over ATen; this ABI is what a Python (ctypes),
 * C++ or any-FFI host binds in place of the reference's model object for the encode/decode path:
 *
 *   qinco_create        <-  QINCo(cfg) + load_state_dict + QINCoInferenceWrapper.build()
 *                           (qinco/qinco_tasks.py:302-309, qinco/model/qinco_inference.py:285-294)
 *   qinco_set_beam      <-  the CLI override of A / B over the checkpoint's values (qinco/utils.py:166-172)
 *   qinco_encode[_host] <-  model(x, step="encode")   (qinco_inference.py:272-279, 239-254, 156-224, 78-140)
 *   qinco_decode[_host] <-  model(codes, step="decode") (qinco_inference.py:280-281, 66-75)
 *   qinco_destroy       <-  del model
 *
 * Conventions: plain pointers and sizes only.  Codes cross the boundary as (n, M) row-major (the reference
 * returns (M, n); its on-disk format and every consumer use (n, M): search_tasks.py:115).  All entry points
 * return 0 on success or a negative qinco_status; qinco_last_error() gives the message of the last failure on
 * the calling thread.  A handle is bound to the HIP device that was current at qinco_create; it is not
 * thread-safe (one handle per host thread / stream), different handles are independent.
 */
#ifndef QINCO_HIP_H
#define QINCO_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define QINCO_API __attribute__((visibility("default")))
#else
#define QINCO_API
#endif

typedef struct qinco_handle_s* qinco_handle;

typedef enum {
  QINCO_OK = 0,
  QINCO_ERR_INVALID = -1,      /* bad argument / unsupported hyper-parameters (reference: assert / ValueError) */
  QINCO_ERR_HIP = -2,          /* HIP runtime failure */
  QINCO_ERR_UNSUPPORTED = -3,  /* (D, De, Dh) has no kernel instance (compiled in or loaded: qinco_load_instance) */
  QINCO_ERR_RANGE = -4         /* a code >= K was passed to decode (reference: index error) */
} qinco_status;

/* element types of the x / codes buffers */
enum { QINCO_X_F32 = 0, QINCO_X_U8 = 1 };
enum { QINCO_CODE_I64 = 0, QINCO_CODE_I32 = 1, QINCO_CODE_U8 = 2 };
/* flags: operate in the model's normalised space, i.e. the .encode(x_norm) / .decode(codes) methods of
 * QINCoInferenceWrapper (qinco_inference.py:330-350) instead of forward()'s (x - mean) / std and * std + mean */
enum { QINCO_FLAG_NORMALISED = 1 };

/* Hyper-parameters: checkpoint["parameters"] + checkpoint["data_dim"]  (qinco/utils.py:100-137). */
typedef struct {
  int32_t D;            /* data dimension (cfg._D) */
  int32_t De;           /* cfg.de, or D when de is None (QINCo1) */
  int32_t Dh;           /* cfg.dh */
  int32_t L;            /* residual blocks per step */
  int32_t M;            /* number of steps = columns of the code matrix (cfg._M_ivf: counts the IVF step) */
  int32_t K;            /* codebook size of every QINCo step */
  int32_t A;            /* candidates pre-selected per beam (0 = all K, QINCo1) */
  int32_t B;            /* beam size */
  int32_t qinco1_mode;  /* 1: res_codeword_coeff = 0 (qinco_inference.py:29) */
  int32_t ivf_K;        /* 0, or IVF-QINCo: step 0 is an IVFBook of ivf_K centroids (any count up to 2^24; the kernels work on
                           blocks of 32 centroids, rows added to fill the last block can never be the arg-min);
                           beam_0 = 1 and the first QINCo step pre-selects max(A, B) (qinco_base.py:108-112, 128-196) */
  int64_t max_batch;    /* vectors processed per internal pass (scratch is sized for it) */
} qinco_desc;

/* Host pointers to contiguous fp32 tensors in the reference state_dict layout (qinco_base.py:229-260, 432-445).
 * Arrays are indexed by step m = 0..M-1; entry 0 of the per-MLP arrays is ignored (step 0 is codebook only). */
typedef struct {
  const float* data_mean;           /* (D) */
  float data_std;                   /* () must be > 0 (qinco_base.py:526) */
  const float* const* codebook;     /* [M] -> (K, D)   steps.m.codebook.weight; with ivf_K > 0 entry 0 is
                                       steps.0.ivf_centroids.weight (ivf_K, D) */
  const float* const* sub_codebook; /* [M] -> (K, D)   steps.m.substep.codebook.weight; NULL entries / NULL if A == 0 */
  const float* const* in_proj;      /* [M] -> (De, D)  steps.m.in_proj.weight;  NULL if De == D */
  const float* const* out_proj;     /* [M] -> (D, De)  steps.m.out_proj.weight; NULL if De == D */
  const float* const* cat_w;        /* [M] -> (De, De + D) steps.m.concat.mlp.weight (cat order: z then xhat) */
  const float* const* cat_b;        /* [M] -> (De)     steps.m.concat.mlp.bias */
  const float* const* up;           /* [M*L] -> (Dh, De) steps.m.residual_blocks.l.up_proj.weight   at [m*L + l] */
  const float* const* down;         /* [M*L] -> (De, Dh) steps.m.residual_blocks.l.down_proj.weight at [m*L + l] */
} qinco_weights;

QINCO_API int qinco_create(const qinco_desc* desc, const qinco_weights* weights, qinco_handle* out);

/* qinco_create with options.  QINCO_CREATE_SPLIT_F16: evaluate the residual FFN blocks of the codeword MLP (94 % of the
 * FLOPs of qinco2-L) on the fp16 matrix pipe with every fp32 operand split into two fp16 values (hi + lo, three MFMAs per
 * product, fp32 accumulation; csrc/mlp_split_kernel.hpp) instead of the fp32-in MFMA, which runs at 1/16 of the fp16 rate on
 * gfx950.  Same function and an error of the same class as the fp32 path against a float64 evaluation, but not the
 * same bits: opt-in, off by default; QINCO_ERR_UNSUPPORTED if the shape has no split instance (csrc/shapes.def).
 * fp16 has a range: activations beyond |z| ~ 8000 (never seen with std-normalised data) overflow; the kernel then raises the
 * sticky device flag and qinco_encode_host / qinco_decode_host / qinco_check return QINCO_ERR_RANGE instead of results computed
 * from NaNs.  The library reads NO environment variables: every switch is an argument.
 * The other flags are diagnostics (A/B measurements, the race-detector tests): the exact fp32 IVF table without the fp16
 * filter; the VALU pre-selection table; decode through the folded encode instance; no cooperative table kernel; no fusion of
 * the pre-selection into the xproj launch on small launches; no small-launch form of the fused MLP (csrc/mlp_small_kernel.hpp:
 * launches below ~one 128-row workgroup per CU -- every decode call at the reference's batch sizes, greedy encode steps --
 * otherwise run on workgroups of 16 * NT rows whose waves split the features, with the same products in the same order). */
enum {
  QINCO_CREATE_SPLIT_F16 = 1,
  QINCO_CREATE_IVF_FP32 = 2,
  QINCO_CREATE_TABLE_VALU = 4,
  QINCO_CREATE_DECODE_FOLDED = 8,
  QINCO_CREATE_TABLE_NO_COOP = 16,
  QINCO_CREATE_SPLIT_NO_CALIBRATION = 32,  /* skip the create-time comparison with the fp32 instance (below) */
  QINCO_CREATE_NO_PRESEL_FUSION = 64,      /* diagnostics: pre-selection table and xproj as two launches at every launch size */
  QINCO_CREATE_NO_SMALL_LAUNCH = 128,      /* diagnostics: the 128-rows-per-workgroup kernels at every launch size */
  QINCO_CREATE_EPILOGUE_SELECT = 256,      /* identity-projection models whose F * A candidates per vector fit a 128-row workgroup take the step's
                                              per-vector top-B in the fused-MLP kernel's epilogue (csrc/mlp_kernel.hpp SELEP): no candidate /
                                              distance write-back, no beam_select launch, bit-identical codes.  On by default where the shape
                                              has the KHEAD + SELEP instance (qinco2-S / QINCo1 with A > 0 on 128-d data: +2 %);
                                              this flag asks for it on the other identity-projection shapes that have a SELEP instance */
  QINCO_CREATE_NO_EPILOGUE_SELECT = 512    /* diagnostics: never (the two-kernel form: candidate write-back + beam_select_kernel) */
};

/* The split form checks itself.  (1) At create, unless QINCO_CREATE_SPLIT_NO_CALIBRATION: the model is also built as an fp32
 * twin, 512 vectors drawn around its own codebooks are encoded by both, and a split form outside the fp32 path's error class
 * on THIS model (more than 5 % of the code rows differ, reconstructions differ by more than 1e-4 relative, or anything
 * overflows) makes qinco_create_ex fail with QINCO_ERR_RANGE.  (2) At run time: overflow raises the sticky flag
 * (qinco_check); underflow is silent by nature, so every 64th workgroup counts how many of the activations' fp16 lo parts
 * are subnormal (have lost bits).  qinco_split_stats reports both.  A subnormal lo part costs at most 2^-25 absolute per element, so
 * lo_subnormal / lo_sampled is a drift indicator, not an error: measured 0.02-0.1 on the synthetic models and 0.32 on a
 * checkpoint trained by the reference (its later steps quantise small residuals, |z| ~ 0.05: calibration error 2.9e-7, no
 * code row changed); it approaches 1 when the activations sit orders of magnitude below the design range -- the regime in
 * which the calibration starts to fail (tests/test_hip_parity.py::test_split_f16_checks_itself_*). */
typedef struct {
  int32_t split_form;            /* 1 if the handle runs the split-fp16 kernels */
  int32_t calibrated;            /* 1 if the create-time calibration ran */
  int32_t calib_vectors;
  int32_t calib_rows_differing;  /* code rows that differ between the split and the fp32 instance */
  float calib_max_rel_err;       /* max |xhat_split - xhat_fp32| / max |xhat_fp32| over the rows with equal codes */
  int32_t overflowed;            /* 1 if an overflow was ever reported through qinco_check / the host entry points */
  int64_t lo_sampled;            /* since create: non-zero activation elements inspected */
  int64_t lo_subnormal;          /* ... of which the fp16 lo part is subnormal */
} qinco_split_report;
QINCO_API int qinco_split_stats(qinco_handle h, qinco_split_report* out);
QINCO_API int qinco_create_ex(const qinco_desc* desc, const qinco_weights* weights, int32_t create_flags, qinco_handle* out);

/* ... and with the remaining diagnostic knobs.  struct_bytes = sizeof(qinco_options) (lets the struct grow); mlp_P / mlp_var
 * select a non-production fused-MLP kernel instance of the model's shape (csrc/shapes.def: QINCO_SHAPE(D, De, Dh, P, VAR)),
 * -1 / -1 = the production instance; table_coop_max = largest launch (in groups) that takes the cooperative pre-selection
 * kernel, -1 = default.  opt == NULL is qinco_create. */
typedef struct {
  int32_t struct_bytes;
  int32_t create_flags;
  int32_t mlp_P, mlp_var;
  int64_t table_coop_max;
} qinco_options;
QINCO_API int qinco_create_opt(const qinco_desc* desc, const qinco_weights* weights, const qinco_options* opt, qinco_handle* out);
QINCO_API int qinco_destroy(qinco_handle h);

/* Change the search width.  A must be 0 iff the model was created with A == 0 (utils.py:169-172); 0 < A <= K. */
QINCO_API int qinco_set_beam(qinco_handle h, int32_t A, int32_t B);

/* Device-pointer entry points: x / codes / out are device (or device-accessible) buffers; work is enqueued on
 * `stream` (a hipStream_t; NULL = default stream) and NOT synchronised.  x rows are `x_row_stride_bytes`
 * apart (0 = tightly packed).  codes_out is (n, M) of `code_dtype`; xhat_out (nullable) receives the
 * NORMALISED reconstruction (n, D) that QINCoInferenceWrapper.encode returns next to the codes. */
QINCO_API int qinco_encode(qinco_handle h, const void* x, int x_dtype, int64_t x_row_stride_bytes, int64_t n,
                 void* codes_out, int code_dtype, float* xhat_out, int flags, void* stream);
QINCO_API int qinco_decode(qinco_handle h, const void* codes, int code_dtype, int64_t n, float* out, int flags,
                 void* stream);

/* Out-of-range codes on the device-pointer path: qinco_decode cannot fail synchronously (nothing is synchronised), so a
 * code outside [0, K_m) raises a sticky flag on the device and is decoded as code 0.  qinco_check waits for `stream`
 * and returns QINCO_ERR_RANGE if any qinco_decode since the last check / host decode saw such a code (the reference
 * raises torch's index error at qinco_inference.py:70), then clears the flag.  qinco_decode_host checks by itself. */
QINCO_API int qinco_check(qinco_handle h, void* stream);

/* Host-pointer convenience forms (stage through device buffers owned by the handle; synchronous). */
QINCO_API int qinco_encode_host(qinco_handle h, const void* x, int x_dtype, int64_t x_row_stride_bytes, int64_t n,
                      void* codes_out, int code_dtype, float* xhat_out, int flags);
QINCO_API int qinco_decode_host(qinco_handle h, const void* codes, int code_dtype, int64_t n, float* out, int flags);

/* Roofline instrumentation: when enabled, every launch of the fused-MLP kernel is bracketed by HIP events on
 * its own stream.  qinco_profile_read synchronises, returns the totals since the last read and resets them. */
QINCO_API int qinco_profile_enable(qinco_handle h, int enable);
QINCO_API int qinco_profile_read(qinco_handle h, double* mlp_ms, int64_t* mlp_launches, double* mlp_flops);
/* ... and the FLOPs the matrix pipe EXECUTED for those launches.  mlp_flops is the reference algorithm's count (rows x R_mlp,
 * SURVEY.md 8d); the kernels take the row-independent head of the MLP out of the per-row work (a table per codeword, one small GEMM
 * per (vector, beam) group: csrc/mlp_kernel.hpp FOLD / FOLD2), so they execute fewer: executed / time / peak is a pipe utilisation
 * (<= 1 by construction), algorithmic / time / peak is not.  Counted per launch from the kernel form that ran. */
QINCO_API int qinco_profile_read2(qinco_handle h, double* mlp_ms, int64_t* mlp_launches, double* mlp_flops, double* mlp_flops_executed);

/* Algorithmic FLOPs of one vector's encode / decode at the handle's current A, B. */
QINCO_API double qinco_flops_per_vector_encode(qinco_handle h);
QINCO_API double qinco_flops_per_vector_decode(qinco_handle h);

/* IVF models: statistics of the most recent IVF assignment of the handle (the last chunk of the last encode call).
 * The assignment runs two fp16 matrix-core filter passes and an exact fp32 pass over the surviving candidates
 * (csrc/ivf_f16_kernel.hpp); *candidates = pairs the exact pass evaluated, *fell_back = 1 if the exact fp32 table
 * kernel had to redo the batch (candidate list full, or inputs outside the fp16 range).  Both 0 when the handle has no
 * fp16 copy (QINCO_IVF_FP32=1, or centroids outside the fp16 range).  Synchronises the device. */
QINCO_API int qinco_ivf_last_stats(qinco_handle h, int64_t* candidates, int32_t* fell_back);

/* One line about the handle for logs and bench records: the fused-MLP kernel instance serving encode ("mlp=DxDexDh P=.. var=..",
 * csrc/shapes.def), the instance serving decode (decode_var=-1: the encode instance), the arithmetic form, the table kernels.
 * Writes at most `cap` bytes including the terminator; returns the length the full text needs. */
QINCO_API int qinco_describe(qinco_handle h, char* buf, int32_t cap);

/* Model geometry.  The reference builds any (D, de, dh) (qinco/model/qinco_base.py:229-260).  The kernels work on 32-feature
 * blocks: qinco_create zero-pads every tensor of another geometry up to the next multiple of 32 (padding features are exact
 * zeros and contribute exact zeros; a model with in/out projections keeps De != D), and looks for a fused-MLP kernel instance of
 * the padded shape -- compiled in (csrc/shapes.def: every preset of the reference on every dataset dimension) or loaded:
 *   qinco_padded_shape   out3 = {D, De, Dh} as the kernels will see them
 *   qinco_shape_supported  1 if an instance for the padded shape exists right now
 *   qinco_load_instance  register the instance in the shared object `path`: ONE translation unit of csrc/mlp_inst.hip built
 *                        with -DQD= -DQDE= -DQDH= -DQP= -DQVAR= -DQINCO_INSTANCE_MODULE (hipcc -shared --offload-arch=gfx950;
 *                        the Python host does this on demand: qinco_amd.build.ensure_instance).  A module also brings the
 *                        pre-selection table and the exact IVF coarse assignment for its D (compiled in for D = 32, 96, 128,
 *                        256, 768 only -- there with the fp16 filter in front).  Loaded instances stay for the
 *                        life of the process.  QINCO_ERR_INVALID if the file cannot be loaded or was built from other sources. */
QINCO_API int qinco_shape_supported(int32_t D, int32_t De, int32_t Dh);
QINCO_API int qinco_padded_shape(int32_t D, int32_t De, int32_t Dh, int32_t* out3);
QINCO_API int qinco_load_instance(const char* path);

/* ---- multi-GPU: the one collective of the path -------------------------------------------------------
 * Database encoding shards over vectors with no exchange (search_tasks.py:103-104); at the end of the job the ranks' codes go
 * to one root.  The reference writes per-rank part files instead (search_tasks.py:119-134); the Python host here gathers with
 * torch.distributed (qinco_amd/encode_db.py); this entry point is the same exchange for a C / C++ host that owns an RCCL
 * communicator: grouped ncclSend / ncclRecv of the raw code bytes (each peer reaches the root over its own xGMI link).
 *   codes_local  device (n_local, M) of code_dtype on this rank          counts[world]  rows of every rank (shards are uneven:
 *   out          device (sum counts, M) on `root` (ignored elsewhere)                   the last one takes the remainder)
 *   nccl_comm    the host's ncclComm_t (one rank per GPU); world == 1 needs none: a device copy.  A communicator of ONE rank,
 *                when given, is used: the shard goes through a grouped ncclSend-to-self + ncclRecv-from-self (the many-rank
 *                sequence on the real library -- what a 1-GPU box can execute of it)
 * Enqueued on `stream`, not synchronised.  RCCL is resolved at call time from the libraries already loaded in the process
 * (the one that made the communicator), else librccl.so.1; QINCO_ERR_UNSUPPORTED when there is none. */
QINCO_API int qinco_gather_codes(const void* codes_local, int64_t n_local, int32_t M, int code_dtype, void* out,
                                 const int64_t* counts, int32_t world, int32_t rank, int32_t root, void* nccl_comm, void* stream);
/* The shared object qinco_gather_codes' ncclSend resolved to (dladdr; at most cap bytes with the terminator): a PyTorch process
 * holds the wheel's own librccl.so next to /opt/rocm's, and a communicator works only with the library that created it.
 * (No reference counterpart: torch.distributed hides the choice, qinco/search/search_tasks.py:85-137 runs under accelerate.) */
QINCO_API int qinco_rccl_library(char* path, size_t cap);

/* ---- look-up decoders downstream of the hot path --------------------------------------------
 * out[n] = sum_j tables[j][ codes[n][a[j]] * mul + (b[j] >= 0 ? codes[n][b[j]] : 0) ]   (fp32, summed in j order)
 *   additive-quantiser decode  reconstruct_from_fixed_codebooks (qinco/search/search_utils.py:105-115):
 *       J = M, a[j] = j, b[j] = -1, mul = 1, Kt = K
 *   pairwise decoder  PairwiseDecoderIVF.forward + map_codes (qinco/search/pairwise_decoder.py:88-93, 126-130):
 *       J = M_target, a = combine_mvals_m[0], b = combine_mvals_m[1], mul = K_base, Kt = K_base^2; the caller
 *       appends ivf_code_map[ivf_codes] to the code columns first, like map_codes does.
 * tables: host pointer to (J, Kt, D) fp32, copied to the device at create.  J <= 64, D % 4 == 0. */
typedef struct qinco_lut_s* qinco_lut;
QINCO_API int qinco_lut_create(const float* tables, int32_t J, int64_t Kt, int32_t D, const int32_t* a, const int32_t* b,
                               int64_t mul, qinco_lut* out);
QINCO_API int qinco_lut_destroy(qinco_lut lut);
/* codes: device (n, Mc) of code_dtype; out: device (n, D) fp32; enqueued on `stream`, not synchronised. */
QINCO_API int qinco_lut_decode(qinco_lut lut, const void* codes, int code_dtype, int32_t Mc, int64_t n, float* out,
                               void* stream);
/* host buffers; synchronous; QINCO_ERR_RANGE if a look-up index falls outside its table. */
QINCO_API int qinco_lut_decode_host(qinco_lut lut, const void* codes, int code_dtype, int32_t Mc, int64_t n, float* out);

/* ---- evaluation stages either side of the hot path -----------------------------------------
 * Brute-force L2 top-k of run_search_full_direct_small_db (qinco/search/search_tasks.py:551-603):
 *   ids[q] = argsort_n( |queries[q]|^2 + |db[n]|^2 - 2 queries[q].db[n] )[:k]     (approx_pairwise_distance,
 *   qinco/utils.py:336-346, fp32; ties -> lower n, i.e. a stable argsort), dist = the matching distances ascending.
 * D in {32, 64, 96, 128, 256, 768}; 1 <= k <= min(n, 2048); n <= 2^31.  Scratch (a distance table of up to 8 GiB
 * per chunk of queries) is owned by the handle and grows on demand. */
typedef struct qinco_knn_s* qinco_knn;
QINCO_API int qinco_knn_create(int32_t D, qinco_knn* out);
QINCO_API int qinco_knn_destroy(qinco_knn knn);
/* db (n, D), queries (nq, D) fp32 row-major on the device; ids_out (nq, k) int64, dist_out (nq, k) fp32 or NULL on the
 * device; enqueued on `stream`, not synchronised. */
QINCO_API int qinco_knn_search(qinco_knn knn, const float* db, int64_t n, const float* queries, int64_t nq, int32_t k,
                               int64_t* ids_out, float* dist_out, void* stream);
/* Large databases take a FILTERED form that never writes the (queries x n) table (csrc/knn_kernel.hpp: thresholds from a strided
 * sample of the database, the whole table on the matrix pipe with only the pairs under their query's threshold kept, those sorted);
 * same ids and distances bit for bit; a chunk of queries whose candidate lists overflow is redone unfiltered on the device.
 * Options: QINCO_KNN_OPT_FILTER 0 = never / 1 = where it pays (default); QINCO_KNN_OPT_FILTER_MIN_N = smallest n that takes it
 * (default 65536).  qinco_knn_last_stats (synchronises the device): out3 = {chunks of queries of the last search, chunks that took
 * the filtered form, chunks that were redone unfiltered}. */
#define QINCO_KNN_OPT_FILTER 0
#define QINCO_KNN_OPT_FILTER_MIN_N 1
#define QINCO_KNN_OPT_QUERY_BYTES 2   /* bytes of query rows per chunk (default 1 MiB: the chunk's fragments stay in L2), <= 4096 rows */
#define QINCO_KNN_OPT_ROLES 3         /* 0 (default): every wave computes and filters (knn_table_kernel<D, true>); 1: D <= 128 runs the
                                         filtered table as one MFMA wave + one filter wave per SIMD (csrc/knn_roles_kernel.hpp) -- the
                                         round-6 experiment, measured slower, kept selectable.  Same candidate sets, same
                                         result bits either way. */
QINCO_API int qinco_knn_set_option(qinco_knn knn, int32_t option, int64_t value);
QINCO_API int qinco_knn_last_stats(qinco_knn knn, int64_t* out3);
/* Of the two-role kernel's launches since the last call (synchronises the device): out2 = {workgroups whose eight waves covered
 * all four SIMDs of their CU -- one MFMA wave per SIMD, taken by HW_ID --, workgroups that fell back to roles by wave index}. */
QINCO_API int qinco_knn_roles_stats(qinco_knn knn, int64_t* out2);
/* the same on host buffers; synchronous */
QINCO_API int qinco_knn_search_host(qinco_knn knn, const float* db, int64_t n, const float* queries, int64_t nq, int32_t k,
                                    int64_t* ids_out, float* dist_out);
/* *sum_out = sum_i (a[i] - b[i])^2 over `count` floats on the device (AnyVectMSE.update, qinco/metrics.py:43-50;
 * accumulated in fp64); synchronises `stream`. */
QINCO_API int qinco_sqerr_sum(const float* a, const float* b, int64_t count, double* sum_out, void* stream);

/* Re-rank of per-query shortlists -- the re-rank stages of run_search_ivf (qinco/search/search_tasks.py:447-472 with the pairwise
 * look-up decoder's reconstructions, :497-507 with QINCo's): for every query q the distances |xq[q]|^2 + |c|^2 - 2 xq[q].c to ITS ns
 * candidates cand[q] (compute_batch_distances(..., approx=True), utils.py:349-383), sorted ascending (ties -> the earlier shortlist
 * position), the first k kept: their shortlist positions, distances, database ids (ids_in (nq, ns) -> ids_out (nq, k)) and code rows
 * (codes_in (nq, ns, Mc) int32 -> codes_out (nq, k, Mc)).  Every output is optional (NULL).  Device pointers, asynchronous on
 * `stream`.  ns * 8 + D * 4 bytes of LDS per query: ns <= 16384. */
QINCO_API int qinco_rerank(const float* xq, const float* cand, int64_t nq, int32_t ns, int32_t D, int32_t k, const int64_t* ids_in,
                 const int32_t* codes_in, int32_t Mc, int64_t* pos_out, float* dist_out, int64_t* ids_out, int32_t* codes_out,
                 void* stream);

/* Device self-test of the in-wave sort / top-T selection primitives the table and beam kernels are built on, against a host
 * computation (random data, heavy ties, NaN / inf).  QINCO_OK, or QINCO_ERR_HIP with the first discrepancy in
 * qinco_last_error().  Runs on the current device; synchronous. */
QINCO_API int qinco_selftest(void);

QINCO_API const char* qinco_last_error(void);
QINCO_API const char* qinco_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QINCO_HIP_H */
