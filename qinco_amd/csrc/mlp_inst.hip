// One instance of the fused-MLP kernel: compiled once per QINCO_SHAPE with -DQD= -DQDE= -DQDH= -DQP= -DQVAR=.
#include <cstdlib>

#include "mlp_kernel.hpp"
#include "mlp16_kernel.hpp"
#include "mlp_launch.hpp"

#define QINCO_CAT_(a, b, c, d, e, f) a##b##_##c##_##d##_##e##_##f
#define QINCO_CAT(a, b, c, d, e, f) QINCO_CAT_(a, b, c, d, e, f)

// Co-residency of the shared-ring kernels.  With small models the 16-row kernel (mlp16_kernel.hpp) fits 3 workgroups per CU, and
// then single waves produce wrong rows now and then (about 1 wave in 100 at 1500 workgroups; exact and deterministic with
// one workgroup per CU).  Round-2 findings (scripts/ubench/ring_check.hip, scripts/exp_coresidency.py): the ring
// itself delivers the right bytes at 1-3 workgroups per CU, VGPR loads issued among the LDS-DMAs retire in order, and the
// 32-row kernels are exact at 2-3 workgroups per CU -- so only the 16-row kernel keeps the padding (its one production
// shape, De = D = 768, is exclusive by registers anyway).
static unsigned exclusive_lds() {
  // 16-row kernel: 48 KiB static + 36 = 84 > 80 -> one workgroup per CU.  The 32-row ring kernels share a CU freely (ring
  // checker, scripts/exp_coresidency.py and the bitwise variant tests at 2-3 workgroups per CU are clean).
  // QINCO_RING_PAD_KIB (experiments) overrides the padding of every shared-ring kernel.
  static const int env = [] {
    const char* e = getenv("QINCO_RING_PAD_KIB");
    return e ? atoi(e) : -1;
  }();
  if (!(QVAR & 64)) return 0u;
  if (env >= 0) return (unsigned)env * 1024u;
  return (QVAR & 128) ? 36u * 1024u : 0u;
}
#define kExclusiveLds exclusive_lds()

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_mlp_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::MlpArgs* a, hipStream_t stream) {
  if (a->R <= 0) return hipSuccess;
#if (QVAR & 128)  // 16-row tile form (mlp16_kernel.hpp)
  const long tiles = (a->R + 15) / 16;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp16_kernel<QD, QDE, QDH, QP>), dim3(grid), dim3(256), kExclusiveLds, stream, *a);
#else
  const long tiles = (a->R + 31) / 32;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp_kernel<QD, QDE, QDH, QP, QVAR>), dim3(grid), dim3(256), kExclusiveLds, stream, *a);
#endif
  return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_xproj_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::XprojArgs* a, hipStream_t stream) {
#if (QVAR & 128)
  (void)a;
  (void)stream;
  return hipErrorNotSupported;  // the 16-row form is never folded
#else
  if (a->G <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a->G + 127) / 128);
  hipLaunchKernelGGL((qinco::xproj_kernel<QD, QDE, QDH>), dim3(grid), dim3(256), 0, stream, *a);
  return hipGetLastError();
#endif
}
