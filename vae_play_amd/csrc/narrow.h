// Host-side dispatch hooks of narrow.hip (edge layers with a 1- or 3-channel side).
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"

namespace vp {
bool narrow_gather_applicable(const ConvGeom& g, int act);
int narrow_gather_launch(const float* big, const float* w_p0, const float* bias, float* out, const ConvGeom& g, int act, hipStream_t s);
bool tapn_gather_applicable(const ConvGeom& g, int act);
int tapn_gather_launch(const float* big, const float* w_p0, const float* bias, float* out, const ConvGeom& g, int act, hipStream_t s);
int narrow_wgrad_kind(const ConvGeom& g);
size_t narrow_wgrad_ws_floats(const ConvGeom& g);
int narrow_wgrad_launch(const float* big, const float* small, float* dw_ref, const ConvGeom& g, float* ws, hipStream_t s);
int pack_w5_f32_launch(const float* w, float* p0, float* p1, int Cs, int Cb, hipStream_t s, int nt = kTaps);
int slab_reduce_launch(const float* slab, float* dw_ref, int Cs, int Cb, int nsplit, hipStream_t s, int nt = kTaps);
}  // namespace vp
