// 5x5 convolution families on the MFMA implicit-GEMM kernel (igemm.h) + weight packing + wgrad reduce.
#include "common.h"
#include "igemm.h"
#include "narrow.h"
#include "conv32.h"
#include "wgrad5.h"

namespace vp {

// Row-of-taps weight gradient in exact fp32 (wgrad5.h: wgrad5f_kernel): the plain 5x5 stride-2 layers with 128 | Cs and
// Cb = 64 or 128 | Cb.  A/B knob VP_WGRAD5F=0 sends them back to igemm_kernel<ProbW>.
static int wgrad5f_kind(const ConvGeom& g, const float* big, const float* small) {
  const char* e = VP_GETENV("VP_WGRAD5F");
  if (e && atoi(e) == 0) return 0;
  if ((((uintptr_t)big | (uintptr_t)small) & 15) != 0) return 0;
  return wgrad5_bn(g);
}

}  // namespace vp

using namespace vp;

extern "C" {

int vp_pack_w_f32(const float* w_ref, float* p0, float* p1, int Csmall, int Cbig, int ks, vp_stream stream) {
  VP_REQUIRE(w_ref && (p0 || p1) && Csmall > 0 && Cbig > 0, "vp_pack_w_f32: bad arguments");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_pack_w_f32: kernel size must be 1, 3 or 5");
  VP_REQUIRE(Csmall <= 65535 && Cbig <= 65535, "vp_pack_w_f32: channel count too large");
  return pack_w5_f32_launch(w_ref, p0, p1, Csmall, Cbig, (hipStream_t)stream, ks * ks);
}

int vp_pack_w5_f32(const float* w_ref, float* p0, float* p1, int Csmall, int Cbig, vp_stream stream) {
  return vp_pack_w_f32(w_ref, p0, p1, Csmall, Cbig, 5, stream);
}

static int conv_check(const char* what, int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride) {
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig > 0 && Csmall > 0, "%s: bad shape", what);
  VP_REQUIRE(stride == 1 || stride == 2, "%s: stride must be 1 or 2", what);
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "%s: kernel size must be 1, 3 or 5", what);
  const int pad = (ks - 1) / 2;
  VP_REQUIRE((Hb + 2 * pad - ks) / stride + 1 == Hs && (Wb + 2 * pad - ks) / stride + 1 == Ws,
             "%s: small size must be floor((big + 2*pad - ks)/stride) + 1 (got big %dx%d small %dx%d)", what, Hb, Wb, Hs, Ws);
  VP_REQUIRE((long)B * Hb * Wb < (1L << 30) && (long)B * Hs * Ws < (1L << 30), "%s: pixel count overflows int", what);
  return VP_OK;
}

int vp_conv_gather_f32(const float* big, const float* w_p0, const float* bias, float* small_out, int B, int Hs, int Ws, int Hb,
                       int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream) {
  VP_REQUIRE(big && w_p0 && small_out, "vp_conv_gather_f32: null pointer");
  int rc = conv_check("vp_conv_gather_f32", B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride);
  if (rc) return rc;
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_SIGMOID, "vp_conv_gather_f32: epilogue supports none|sigmoid");
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  if (narrow_gather_applicable(g, act)) return narrow_gather_launch(big, w_p0, bias, small_out, g, act, (hipStream_t)stream);
  if (f32_fast_gather_ok(g, big, w_p0, act)) return f32_fast_gather(big, w_p0, bias, small_out, g, act, (hipStream_t)stream);
  ProbF p = make_probF(big, w_p0, bias, small_out, g, act);
  launch_igemm(p, p.M, p.N, 1, (hipStream_t)stream);
  return check_launch("vp_conv_gather_f32");
}

int vp_conv_scatter_f32(const float* small, const float* w_p1, float* big_out, int B, int Hs, int Ws, int Hb, int Wb, int Csmall,
                        int Cbig, int ks, int stride, vp_stream stream) {
  VP_REQUIRE(small && w_p1 && big_out, "vp_conv_scatter_f32: null pointer");
  int rc = conv_check("vp_conv_scatter_f32", B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride);
  if (rc) return rc;
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  if (f32_fast_scatter_ok(g, small, w_p1)) return f32_fast_scatter(small, w_p1, big_out, g, (hipStream_t)stream);
  ProbT p = make_probT(small, w_p1, big_out, g);
  launch_igemm(p, p.M, p.N, stride * stride, (hipStream_t)stream);
  return check_launch("vp_conv_scatter_f32");
}

size_t vp_conv_wgrad_workspace_bytes(int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride) {
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  if (narrow_wgrad_kind(g)) return narrow_wgrad_ws_floats(g) * sizeof(float);
  size_t n = wgrad_slab_floats(g, wgrad_nsplit(g));
  if (const int bn = wgrad5_bn(g)) {
    int kper = 0;
    const size_t n5 = wgrad5_slab_floats(g, bn, wgrad5_nsplit(g, bn, &kper));
    if (n5 > n) n = n5;
  }
  return n * sizeof(float);
}

static int conv_wgrad_f32(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                          int Csmall, int ks, int stride, int max_cus, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(big && small && dw_ref && ws, "vp_conv_wgrad_f32: null pointer");
  int rc = conv_check("vp_conv_wgrad_f32", B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride);
  if (rc) return rc;
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  if (ws_bytes < vp_conv_wgrad_workspace_bytes(B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride))
    return fail(VP_ERR_WORKSPACE, "vp_conv_wgrad_f32: workspace too small");
  if (narrow_wgrad_kind(g)) return narrow_wgrad_launch(big, small, dw_ref, g, (float*)ws, (hipStream_t)stream);
  if (const int bn = wgrad5f_kind(g, big, small)) {
    int kper = 0, slabs = 0;
    const int ns5 = wgrad5_nsplit(g, bn, &kper, max_cus);
    wgrad5f_launch(big, small, (float*)ws, g, bn, ns5, kper, (hipStream_t)stream, &slabs);
    rc = check_launch("vp_conv_wgrad_f32(rows of taps)");
    if (rc) return rc;
    return slab_reduce_launch((const float*)ws, dw_ref, Csmall, Cbig, slabs, (hipStream_t)stream, g.nt);
  }
  const int ns = wgrad_nsplit(g);
  ProbW p = make_probW(big, small, (float*)ws, g, ns);
  launch_igemm(p, p.M, p.N, g.nt * ns, (hipStream_t)stream);
  rc = check_launch("vp_conv_wgrad_f32(main)");
  if (rc) return rc;
  return slab_reduce_launch((const float*)ws, dw_ref, Csmall, Cbig, ns, (hipStream_t)stream, g.nt);
}

int vp_conv_wgrad_f32(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                      int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return conv_wgrad_f32(big, small, dw_ref, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, 0, ws, ws_bytes, stream);
}

int vp_conv5_wgrad_f32_cus(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Cbig, int Csmall,
                           int stride, int max_cus, void* ws, size_t ws_bytes, vp_stream stream) {
  return conv_wgrad_f32(big, small, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, max_cus, ws, ws_bytes, stream);
}

/* 5x5 wrappers (big = stride * small) */
int vp_conv5_gather_f32(const float* big, const float* w_p0, const float* bias, float* small_out, int B, int Hs, int Ws,
                        int Cbig, int Csmall, int stride, int act, vp_stream stream) {
  return vp_conv_gather_f32(big, w_p0, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream);
}

int vp_conv5_scatter_f32(const float* small, const float* w_p1, float* big_out, int B, int Hs, int Ws, int Csmall,
                         int Cbig, int stride, vp_stream stream) {
  return vp_conv_scatter_f32(small, w_p1, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream);
}

size_t vp_conv5_wgrad_workspace_bytes(int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  return vp_conv_wgrad_workspace_bytes(B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride);
}

int vp_conv5_wgrad_f32(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Cbig, int Csmall,
                       int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return vp_conv_wgrad_f32(big, small, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream);
}
}
