// One instance of the fused-MLP kernel: compiled once per QINCO_SHAPE with -DQD= -DQDE= -DQDH= -DQP= -DQVAR=.
#include "mlp_kernel.hpp"
#include "mlp_launch.hpp"

#define QINCO_CAT_(a, b, c, d, e, f) a##b##_##c##_##d##_##e##_##f
#define QINCO_CAT(a, b, c, d, e, f) QINCO_CAT_(a, b, c, d, e, f)

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_mlp_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::MlpArgs* a, hipStream_t stream) {
  if (a->R <= 0) return hipSuccess;
  const long tiles = (a->R + 31) / 32;
  const unsigned grid = (unsigned)((tiles + 3) / 4);
  hipLaunchKernelGGL((qinco::mlp_kernel<QD, QDE, QDH, QP, QVAR>), dim3(grid), dim3(256), 0, stream, *a);
  return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden")))
hipError_t QINCO_CAT(qinco_xproj_launch_, QD, QDE, QDH, QP, QVAR)(const qinco::XprojArgs* a, hipStream_t stream) {
  if (a->G <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a->G + 127) / 128);
  hipLaunchKernelGGL((qinco::xproj_kernel<QD, QDE, QDH>), dim3(grid), dim3(256), 0, stream, *a);
  return hipGetLastError();
}
