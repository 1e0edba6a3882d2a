// Host-side dispatch hooks of halo.hip (halo-resident MFMA convolution for narrow channel counts).
#pragma once
#include <hip/hip_runtime.h>

namespace vp {
// 0 = not applicable (use the implicit-GEMM kernel), otherwise the configuration to launch
int halo_gather_kind(int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int halo_gather_launch(int kind, const void* big_split, const void* w_p0_split, const float* bias, float* out, int B, int Hs, int Ws,
                       int Cbig, int Csmall, int stride, int act, hipStream_t s);
int halo_scatter_kind(int B, int Hs, int Ws, int Csmall, int Cbig, int stride);
int halo_scatter_launch(int kind, const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                        int Cbig, hipStream_t s);
}  // namespace vp
