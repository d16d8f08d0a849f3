// Dispatch table for the fused-MLP kernel instances (see shapes.def).
#pragma once
#include <hip/hip_runtime.h>

namespace qinco {
struct MlpArgs;
struct XprojArgs;
struct TableArgs;
struct IvfArgs;
struct SmallArgs;
typedef hipError_t (*mlp_launch_fn)(const MlpArgs*, hipStream_t);
typedef hipError_t (*xproj_launch_fn)(const XprojArgs*, hipStream_t);
typedef hipError_t (*table_launch_fn)(const TableArgs*, hipStream_t);
typedef hipError_t (*ivf_launch_fn)(const IvfArgs*, hipStream_t);
// small-launch form (mlp_small_inst.hip): dec = 1 every decode step in one launch, 0 = one encode step; NT = row tiles of 16 per
// workgroup; NT = 0 is a query: nothing is launched, the largest NT the shape has comes back in place of the error code
typedef hipError_t (*small_launch_fn)(const SmallArgs*, int dec, int NT, hipStream_t);
struct MlpInstance {
  int D, De, Dh, P, var;
  mlp_launch_fn fn;
  xproj_launch_fn xproj;   // used when var has the FOLD bit (16)
  table_launch_fn table;   // modules built on demand: the matrix-core pre-selection table for their D; nullptr = the compiled-in set
  ivf_launch_fn ivf;       // modules built on demand: the exact fp32 coarse assignment for their D; nullptr = the compiled-in set
  small_launch_fn small;   // modules built on demand: the small-launch form for their shape; nullptr = look in csrc/small_shapes.def
};
// want_P / want_var < 0: the production (first listed) instance of the shape.
const MlpInstance* find_mlp_instance(int D, int De, int Dh, int want_P, int want_var);
}  // namespace qinco
