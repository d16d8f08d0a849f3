// One instance of the fused-MLP kernel: compiled once per QINCO_SHAPE with -DQD= -DQDE= -DQDH= -DQP= -DQVAR=.
#include <cstdlib>

#include "mlp_kernel.hpp"
#include "mlp16_kernel.hpp"
#include "mlp_split_kernel.hpp"
#include "presel_kernel.hpp"
#include "mlp_launch.hpp"

#define QINCO_CAT_(a, b, c, d, e, f) a##b##_##c##_##d##_##e##_##f
#define QINCO_CAT(a, b, c, d, e, f) QINCO_CAT_(a, b, c, d, e, f)

// Co-residency of the shared-ring kernels.  Round 1 padded every shared-ring launch with dummy dynamic LDS so that two
// workgroups could not share a CU: with small models the 16-row kernel fits 3 per CU and then ~1 wave in 100 computed with a
// wrong weight fragment.  Round 2 found the cause (mlp16_kernel.hpp, fragmm): hipcc moves the consuming MFMAs
// and the lgkmcnt wait of a fragment's LDS read below the s_barrier that licenses the refill of its ring slot.  With the
// fragments pinned where they are consumed the kernels are exact at 1-3 workgroups per CU and nothing is padded any more.
// Experiment builds (-DQINCO_EXPERIMENT): QINCO_RING_PAD_KIB=n still pads every shared-ring launch.
#ifdef QINCO_EXPERIMENT
static unsigned exclusive_lds() {
  static const int env = [] {
    const char* e = getenv("QINCO_RING_PAD_KIB");
    return e ? atoi(e) : -1;
  }();
  return ((QVAR & 64) && env >= 0) ? (unsigned)env * 1024u : 0u;
}
#define kExclusiveLds exclusive_lds()
#else
#define kExclusiveLds 0u
#endif

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_mlp_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::MlpArgs* a, hipStream_t stream) {
  if (a->R <= 0) return hipSuccess;
#if (QVAR & 512)  // split-fp16 FFN blocks (mlp_split_kernel.hpp)
  const long tiles = (a->R + 31) / 32;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp_split_kernel<QD, QDE, QDH, QP>), dim3(grid), dim3(256), 0, stream, *a);
#elif (QVAR & 128)  // 16-row tile form (mlp16_kernel.hpp)
  const long tiles = (a->R + 15) / 16;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp16_kernel<QD, QDE, QDH, QP, (QVAR & 1024) ? 2 : 1, (QVAR & 16) != 0>), dim3(grid), dim3(256), kExclusiveLds,
                     stream, *a);
#else
  const long tiles = (a->R + 31) / 32;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp_kernel<QD, QDE, QDH, QP, QVAR>), dim3(grid), dim3(256), kExclusiveLds, stream, *a);
#endif
  return hipGetLastError();
}

#ifdef QINCO_INSTANCE_MODULE
#include "ivf_kernel.hpp"
#include "table_kernel.hpp"
#if (QVAR & 16) && !(QVAR & (512 | 128))   // a folded fp32 32-row instance: the module also brings the small-launch form of its shape
#define QF2 ((QVAR & 32) ? 1 : 0)
#define QINCO_MODULE_HAS_SMALL 1
#define QINCO_SMALL_FN_NAME qinco_module_small_launch
#include "mlp_small_inst.hip"
#endif
extern "C" __attribute__((visibility("hidden"))) hipError_t qinco_module_ivf_launch(const qinco::IvfArgs* a, hipStream_t stream) {
  if (a->N <= 0) return hipSuccess;
  return qinco::launch_ivf_assign_kernel<QD>(*a, stream);
}
extern "C" __attribute__((visibility("hidden"))) hipError_t qinco_module_table_launch(const qinco::TableArgs* a, hipStream_t stream) {
  if (a->G <= 0) return hipSuccess;
  return qinco::launch_table_kernels<QD>(*a, stream);
}
// Built on demand as a shared object of its own (qinco_amd.build.ensure_instance) and registered with qinco_load_instance:
// v = {D, De, Dh, P, VAR, 0}, fns = {mlp launcher, xproj launcher, table launcher (K = 256 pre-selection for this D), IVF coarse assignment for this D,
// small-launch form or nullptr} -- kInstanceNFns entries, of which at most `cap` (the caller's array length) are written; returns
// instance_abi() (the sizes of the argument blocks and the launcher count) as the source-version check (a module built against
// another csrc/mlp_args.hpp must not be launched).
extern "C" hipError_t QINCO_CAT(qinco_xproj_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::XprojArgs* a, hipStream_t stream);
extern "C" __attribute__((visibility("default"))) int qinco_instance_info(int* v, void** fns, int cap) {
  v[0] = QD;
  v[1] = QDE;
  v[2] = QDH;
  v[3] = QP;
  v[4] = QVAR;
  void* all[qinco::kInstanceNFns] = {
      reinterpret_cast<void*>(&QINCO_CAT(qinco_mlp_launch_, QD, QDE, QDH, QP, QVAR)),
      reinterpret_cast<void*>(&QINCO_CAT(qinco_xproj_launch_, QD, QDE, QDH, QP, QVAR)),
      reinterpret_cast<void*>(&qinco_module_table_launch),
      reinterpret_cast<void*>(&qinco_module_ivf_launch),
#ifdef QINCO_MODULE_HAS_SMALL
      reinterpret_cast<void*>(&qinco_module_small_launch),
#else
      nullptr,
#endif
  };
  for (int i = 0; i < cap && i < qinco::kInstanceNFns; ++i) fns[i] = all[i];
  return qinco::instance_abi();
}
#endif

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_xproj_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::XprojArgs* a, hipStream_t stream) {
#if (QVAR & 512)
  if (a->G <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a->G + 127) / 128);
  hipLaunchKernelGGL((qinco::xproj_split_kernel<QD, QDE, QDH, QP>), dim3(grid), dim3(256), 0, stream, *a);
  return hipGetLastError();
#elif (QVAR & 128) && (QVAR & 16)   // folded 16-row form: U = W_cat[:, De:] xhat per group, the MLP kernel's MODE 1
  if (a->G <= 0) return hipSuccess;
  qinco::MlpArgs m{};
  m.wstream = a->wx;
  m.xhat = a->xhat;
  m.uproj = a->uproj;
  m.R = a->G;
  m.A = 1;
  m.F = 1;
  const unsigned grid = (unsigned)(((a->G + 15) / 16 + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp16_kernel<QD, QDE, QDH, QP, (QVAR & 1024) ? 2 : 1, true, 1>), dim3(grid), dim3(256), kExclusiveLds, stream, m);
  return hipGetLastError();
#elif (QVAR & 128)
  (void)a;
  (void)stream;
  return hipErrorNotSupported;  // (the un-folded 16-row form has no group projection)
#else
  if (a->G <= 0) return hipSuccess;
  if (a->cstream) {   // small launch with the step's pre-selection fused in (the host checked presel_coop_ok for this instance)
    if constexpr (qinco::presel_coop_ok(QDE, QDH, QVAR)) {
      hipLaunchKernelGGL((qinco::presel_xproj_coop_kernel<QD, QDE, QDH, 8>), dim3((unsigned)((a->G + 31) / 32)), dim3(256), 0, stream, *a);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  const unsigned grid = (unsigned)((a->G + 127) / 128);
  hipLaunchKernelGGL((qinco::xproj_kernel<QD, QDE, QDH>), dim3(grid), dim3(256), 0, stream, *a);
  return hipGetLastError();
#endif
}
