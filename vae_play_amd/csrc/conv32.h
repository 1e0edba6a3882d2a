// fp32 gather / scatter on the split-operand kernel's data path (conv32.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"
namespace vp {
bool f32_fast_gather_ok(const ConvGeom& g, const float* big, const float* w, int act);
bool f32_fast_scatter_ok(const ConvGeom& g, const float* small, const float* w);
// `stat`: per-workgroup BatchNorm statistics slab of the epilogue (igemm16.h epilogue_stats32), or nullptr
int f32_fast_gather(const float* big, const float* w_p0, const float* bias, float* out, const ConvGeom& g, int act, hipStream_t s, float* stat = nullptr);
int f32_fast_scatter(const float* small, const float* w_p1, float* out, const ConvGeom& g, hipStream_t s, float* stat = nullptr);
int stats_slab_finish(const float* slab, int groups, int tiles_m, int bm, long M, long R, int N, float eps, float momentum, float* mean,
                      float* rstd, float* rm, float* rv, hipStream_t stream);
}
