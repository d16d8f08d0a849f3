// One shape of the small-launch fused-MLP kernel (mlp_small_kernel.hpp): compiled once per QINCO_SMALL_SHAPE of
// csrc/small_shapes.def with -DQD= -DQDE= -DQDH= -DQF2=, and inside every kernel-instance module built on demand.
#include "mlp_small_kernel.hpp"
#include "mlp_launch.hpp"

#define QINCO_SCAT_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define QINCO_SCAT(a, b, c, d, e) QINCO_SCAT_(a, b, c, d, e)
#ifndef QINCO_SMALL_FN_NAME   // (a module built on demand names its launcher itself: its FOLD2 is an expression, not a token)
#define QINCO_SMALL_FN_NAME QINCO_SCAT(qinco_small_launch_, QD, QDE, QDH, QF2)
#endif

namespace {
template <int NT, bool DEC>
hipError_t launch_small(const qinco::SmallArgs& a, hipStream_t st) {
  constexpr qinco::SmallPlan PL = qinco::small_plan(QD, QDE, QDH, NT, QF2 != 0, DEC);
  if constexpr (!PL.ok) {
    return hipErrorNotSupported;
  } else {
    auto kern = qinco::mlp_small_kernel<QD, QDE, QDH, NT, QF2 != 0, DEC>;
    // (per instantiation AND per device: the attribute applies to the current device's copy of the function, and a process may drive
    // several GPUs; setting it twice is harmless, so the flags need no lock)
    static bool raised[64] = {};
    int devid = 0;
    if (hipError_t e = hipGetDevice(&devid); e != hipSuccess) return e;
    if (PL.lds_bytes > 64 * 1024 && !raised[devid & 63]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL.lds_bytes);
      if (e != hipSuccess) return e;
      raised[devid & 63] = true;
    }
    const unsigned grid = (unsigned)((a.R + 16 * NT - 1) / (16 * NT));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * qinco::kSmallWaves), PL.lds_bytes, st, a);
    return hipGetLastError();
  }
}
template <bool DEC>
int max_nt() {
  int best = 0;
  if (qinco::small_plan(QD, QDE, QDH, 1, QF2 != 0, DEC).ok) best = 1;
  if (qinco::small_plan(QD, QDE, QDH, 2, QF2 != 0, DEC).ok) best = 2;
  if (qinco::small_plan(QD, QDE, QDH, 3, QF2 != 0, DEC).ok) best = 3;
  if (qinco::small_plan(QD, QDE, QDH, 4, QF2 != 0, DEC).ok) best = 4;
  return best;
}
}  // namespace

// dec: 1 = every decode step in one launch, 0 = one encode step.  NT = 0: returns the largest NT this shape has (as an "error"
// code: a query, nothing is launched).
extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_SMALL_FN_NAME(const qinco::SmallArgs* a, int dec, int NT, hipStream_t st) {
  if (NT == 0) return (hipError_t)(dec ? max_nt<true>() : max_nt<false>());
  if (a->R <= 0) return hipSuccess;
  switch (NT * 2 + (dec ? 1 : 0)) {
    case 2: return launch_small<1, false>(*a, st);
    case 3: return launch_small<1, true>(*a, st);
    case 4: return launch_small<2, false>(*a, st);
    case 5: return launch_small<2, true>(*a, st);
    case 6: return launch_small<3, false>(*a, st);
    case 7: return launch_small<3, true>(*a, st);
    case 8: return launch_small<4, false>(*a, st);
    case 9: return launch_small<4, true>(*a, st);
    default: return hipErrorNotSupported;
  }
}
