// Dispatch table for the per-shape fused-MLP kernel instances (see shapes.def).
#pragma once
#include <hip/hip_runtime.h>

namespace qinco {
struct MlpArgs;
constexpr int kRing = 8;  // weight prefetch ring depth (fragments of 1 KiB per wave)
typedef hipError_t (*mlp_launch_fn)(const MlpArgs*, hipStream_t);
mlp_launch_fn find_mlp_launcher(int D, int De, int Dh);
}  // namespace qinco
