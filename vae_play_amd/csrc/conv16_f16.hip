// fp16-pair ("f16") entry points of the split-operand convolution families: their own translation unit, so that the fp16 problem
// descriptors (modes 1 and 2 of igemm16.h) are instantiated here and compile in parallel with conv16.hip's bf16 ones.
#define VP_PCFG_LIBRARY 1
#include "conv16_impl.h"

#define VP_PRODUCTS_OK(what) VP_REQUIRE(products == 2 || products == 3, what ": products must be 2 or 3")

extern "C" {

// ---- fp16-pair planes ("f16x2" plans): the same three families with two or three MFMAs per fragment pair (igemm16.h mfma_split) -------
// Operands are written by the *_fmt producers with format 1; `products` = 3: al*bh + ah*bl + ah*bh (forward layers: ~1e-6 relative),
// 2: (ah + al)*bh (backward layers: the weight operand of the gather / scatter families and the `big` operand of the weight gradient
// contribute their fp16 hi plane only, ~2e-4 relative per layer).  out_scale multiplies the accumulators (1 / the scale the producer
// of a gradient operand applied; 1 for activations and weights).
int vp_conv5_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs,
                        int Ws, int Cbig, int Csmall, int stride, int act, int products, float out_scale, vp_stream stream) {
  VP_PRODUCTS_OK("vp_conv5_gather_f16");
  return products == 2 ? gather16<2>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream, out_scale)
                       : gather16<3>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream, out_scale);
}
int vp_conv_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                       int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, int products, float out_scale, vp_stream stream) {
  VP_PRODUCTS_OK("vp_conv_gather_f16");
  return products == 2 ? gather16<2>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, stream, out_scale)
                       : gather16<3>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, stream, out_scale);
}
int vp_conv5_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                         int Cbig, int stride, int products, float out_scale, vp_stream stream) {
  VP_PRODUCTS_OK("vp_conv5_scatter_f16");
  return products == 2 ? scatter16<2>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream, out_scale)
                       : scatter16<3>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream, out_scale);
}
int vp_conv_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                        int Csmall, int Cbig, int ks, int stride, int products, float out_scale, vp_stream stream) {
  VP_PRODUCTS_OK("vp_conv_scatter_f16");
  return products == 2 ? scatter16<2>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, stream, out_scale)
                       : scatter16<3>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, stream, out_scale);
}
int vp_conv5_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                         int Csmall, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<2>(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream, out_scale);
}
int vp_conv5_wgrad_f16x2_cus(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                             int Csmall, int stride, float out_scale, int max_cus, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<2>(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream, out_scale, max_cus);
}

int vp_conv_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                        int Csmall, int ks, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<2>(big_split, small_split, dw_ref, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, ws, ws_bytes, stream, out_scale);
}

// the same on fp16-pair planes (forward layers: out_scale is 1)
size_t vp_conv5_stats_f16_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  const StatPlan sp = stat_plan(family, B, Hs, Ws, Cbig, Csmall, stride, true);
  return sp.ok ? (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float) : 0;
}

int vp_conv5_gather_stats_f16(const void* big_split, const void* w_p0_split, float* small_out, int B, int Hs, int Ws, int Cbig,
                              int Csmall, int stride, int products, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                              float* running_var, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(products == 2 || products == 3, "vp_conv5_gather_stats_f16: products must be 2 or 3");
  VP_REQUIRE(big_split && w_p0_split && small_out && mean && rstd && ws, "vp_conv5_gather_stats_f16: null pointer");
  const StatPlan sp = stat_plan(0, B, Hs, Ws, Cbig, Csmall, stride, true);
  VP_REQUIRE(sp.ok, "vp_conv5_gather_stats_f16: this shape cannot emit statistics (vp_conv5_stats_f16_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_gather_stats_f16: workspace too small");
  int rc = products == 2 ? gather16_t<ProbF16X>(big_split, w_p0_split, nullptr, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5,
                                                stride, VP_ACT_NONE, true, stream, (float*)ws)
                         : gather16_t<ProbF16H>(big_split, w_p0_split, nullptr, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5,
                                                stride, VP_ACT_NONE, true, stream, (float*)ws);
  if (rc) return rc;
  return vp16_stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
}

int vp_conv5_scatter_stats_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                               int Cbig, int stride, int products, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                               float* running_var, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(products == 2 || products == 3, "vp_conv5_scatter_stats_f16: products must be 2 or 3");
  VP_REQUIRE(small_split && w_p1_split && big_out && mean && rstd && ws, "vp_conv5_scatter_stats_f16: null pointer");
  const StatPlan sp = stat_plan(1, B, Hs, Ws, Cbig, Csmall, stride, true);
  VP_REQUIRE(sp.ok, "vp_conv5_scatter_stats_f16: this shape cannot emit statistics (vp_conv5_stats_f16_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_scatter_stats_f16: workspace too small");
  int rc = products == 2 ? scatter16_t<ProbT16X>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride,
                                                 true, stream, (float*)ws)
                         : scatter16_t<ProbT16H>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride,
                                                 true, stream, (float*)ws);
  if (rc) return rc;
  return vp16_stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
}

}
