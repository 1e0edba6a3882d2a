// Exact-fp32 gather / scatter convolutions of the plain 5x5 layers on the split-operand kernel's data path (round 3).
//
// igemm.h (round 1) decomposes k -> (tap, channel) per 16-B load, predicates, walks K tap-major and never splits K: 41-70 % of the
// fp32-MFMA peak per layer (profiles/r03_*_f32).  igemm16_kernel already has what the 16-bit modes needed -- workgroup-uniform tap
// decomposition, the zero page, channel-chunk-major K order, XCD-aware tile order, 2-way split-K for the few-tile layers -- and its
// data path does not care what the 16-B chunks hold: MODE 3 (igemm16.h) stages ONE plane per operand, reads a fragment as 4 fp32 and
// contracts it with four v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: the reference's arithmetic,
// models/networks.py:14,38).  The fp32 tensors are addressed in 16-bit units: [pixel][C] fp32 = [pixel][2C] u16, a 64-deep K-tile =
// 32 floats, so the gathered channel count is doubled in the geometry and nothing else changes.
#include "common.h"
#include "igemm16.h"
#include "conv32.h"

namespace vp {

template <class P, int BKT>
static void launch32_bk(const P& p, long M, long N, int gz, hipStream_t s, const Tile16& t) {
  const dim3 block(256);
  auto grid = [&](int bm, int bn) { return dim3((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)gz); };
  if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((igemm16_kernel<P, 128, 128, 2, 2, BKT, true>), grid(128, 128), block, 0, s, p);
  else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((igemm16_kernel<P, 128, 64, 2, 2, BKT, true>), grid(128, 64), block, 0, s, p);
  else hipLaunchKernelGGL((igemm16_kernel<P, 64, 64, 2, 2, BKT, true>), grid(64, 64), block, 0, s, p);
}

// tile rule: the 16-bit kernels' (a K-tile is 2.7x more MFMA cycles here, so the same grids are at least as well fed)
static Tile16 tile32(long M, long N, int gz) {
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long)gz; };
  const long minb = 384;
  if (M >= 128 && N >= 128 && blocks(128, 128) >= minb) return {128, 128};
  if (M >= 128 && N >= 64 && blocks(128, 64) >= minb) return {128, 64};
  return {64, 64};
}

static int xcd32(long M, long N, const Tile16& t) {
  const long gx = (M + t.bm - 1) / t.bm, gy = (N + t.bn - 1) / t.bn;
  return (gx % 8 == 0 && gy >= 2) ? 1 : 0;
}

static bool f32_fast_on() {
  const char* e = VP_GETENV("VP_F32_FAST");      // A/B knob: 0 = igemm.h for every fp32 convolution
  return !(e && atoi(e) == 0);
}

// the fast path takes: plain 5x5 layers (big = stride * small), gathered channel count a multiple of 16 (32-deep K-tiles of u16
// = 16 floats; 64-deep from 32 channels on), at least 64 output columns, 16-B aligned operands, element offsets below 2^31 u16
bool f32_fast_gather_ok(const ConvGeom& g, const float* big, const float* w, int act) {
  return f32_fast_on() && g.ks == 5 && g.Hb == g.Hs * g.stride && g.Wb == g.Ws * g.stride && g.Cb % 16 == 0 && g.Cs >= 64 &&
         (act == ACT_NONE || act == ACT_SIGMOID) && (((uintptr_t)big | (uintptr_t)w) & 15) == 0 &&
         (size_t)g.B * g.Hb * g.Wb * g.Cb * 2 < ((size_t)1 << 31) && (size_t)g.Cs * 25 * g.Cb * 2 < ((size_t)1 << 31);
}
bool f32_fast_scatter_ok(const ConvGeom& g, const float* small, const float* w) {
  return f32_fast_on() && g.ks == 5 && g.Hb == g.Hs * g.stride && g.Wb == g.Ws * g.stride && g.Cs % 16 == 0 && g.Cb >= 64 &&
         (((uintptr_t)small | (uintptr_t)w) & 15) == 0 &&
         (size_t)g.B * g.Hs * g.Ws * g.Cs * 2 < ((size_t)1 << 31) && (size_t)g.Cb * 25 * g.Cs * 2 < ((size_t)1 << 31);
}

// split-K decisions, shared by the launchers and the statistics plan
static int gather_ns32(long M, int N, int K2, int C2, bool has_bias, int act) {
  // few output tiles and a long K (the 8x8-resolution layers): two K halves added atomically onto a zeroed output -- two addends,
  // so the result does not depend on their order (bit-reproducible)
  const long tiles = ((M + 127) / 128) * ((N + 63) / 64);
  return (!has_bias && act == ACT_NONE && C2 % 64 == 0 && tiles < 384 && K2 >= 4096) ? 2 : 1;
}
static int scatter_ns32(long M, int N, int C2, int phases) {
  const long tiles = ((M + 127) / 128) * ((N + 63) / 64) * phases;
  return (C2 % 64 == 0 && tiles < 384 && 4 * C2 >= 1024) ? 2 : 1;
}

int f32_fast_gather(const float* big, const float* w_p0, const float* bias, float* out, const ConvGeom& g0, int act, hipStream_t s, float* stat) {
  typedef ProbF16T<true, 3> P;
  P p;
  p.alpha = 1.f;
  p.zero = vp_zero_page();
  const int C2 = 2 * g0.Cb;                                   // gathered channels in 16-bit units
  p.g = make_geom(g0.B, g0.Hs, g0.Ws, g0.Cs, C2, g0.stride, 5, g0.Hb, g0.Wb);
  p.big = (const u16*)big; p.big_plane = 0;
  p.w = (const u16*)w_p0; p.w_plane = 0;
  p.bias = bias; p.out = out; p.act = act;
  p.M = g0.B * g0.Hs * g0.Ws; p.N = g0.Cs; p.K = 25 * C2;
  const Tile16 t0 = tile32(p.M, p.N, 1);
  p.nsplit = gather_ns32(p.M, p.N, p.K, C2, bias != nullptr, act);
  p.stat = stat;
  if (stat && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_gather_stats_f32: this shape splits K");
  p.k_per_split = p.nsplit == 2 ? ((p.K / 64 + 1) / 2) * 64 : p.K;
  if (p.nsplit == 2 && hipMemsetAsync(out, 0, (size_t)p.M * p.N * sizeof(float), s) != hipSuccess) return fail(VP_ERR_LAUNCH, "vp_conv_gather_f32: memset failed");
  const Tile16 t = p.nsplit == 2 ? tile32(p.M, p.N, 2) : t0;
  p.xcd_map = xcd32(p.M, p.N, t);
  if (C2 % 64 == 0) launch32_bk<P, 64>(p, p.M, p.N, p.nsplit, s, t);
  else launch32_bk<P, 32>(p, p.M, p.N, p.nsplit, s, t);
  return check_launch("vp_conv_gather_f32(fast)");
}

int f32_fast_scatter(const float* small, const float* w_p1, float* out, const ConvGeom& g0, hipStream_t s, float* stat) {
  typedef ProbT16T<true, 3> P;
  P p;
  p.alpha = 1.f;
  p.zero = vp_zero_page();
  const int C2 = 2 * g0.Cs;
  p.g = make_geom(g0.B, g0.Hs, g0.Ws, C2, g0.Cb, g0.stride, 5, g0.Hb, g0.Wb);
  p.small = (const u16*)small; p.small_plane = 0;
  p.w = (const u16*)w_p1; p.w_plane = 0;
  p.out = out; p.M = g0.B * g0.Hs * g0.Ws; p.N = g0.Cb;
  const int ph = g0.stride * g0.stride;
  p.nsplit = scatter_ns32(p.M, p.N, C2, ph);
  p.stat = stat;
  if (stat && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_scatter_stats_f32: this shape splits K");
  if (p.nsplit == 2 && hipMemsetAsync(out, 0, (size_t)g0.B * g0.Hb * g0.Wb * g0.Cb * sizeof(float), s) != hipSuccess)
    return fail(VP_ERR_LAUNCH, "vp_conv_scatter_f32: memset failed");
  const int gz = ph * p.nsplit;
  const Tile16 t = tile32(p.M, p.N, gz);
  p.xcd_map = xcd32(p.M, p.N, t);
  p.pair_phases = (g0.stride == 2 && p.nsplit == 1 && ((p.M + t.bm - 1) / t.bm) * ((p.N + t.bn - 1) / t.bn) * gz <= 512) ? 1 : 0;
  if (C2 % 64 == 0) launch32_bk<P, 64>(p, p.M, p.N, gz, s, t);
  else launch32_bk<P, 32>(p, p.M, p.N, gz, s, t);
  return check_launch("vp_conv_scatter_f32(fast)");
}


// ---- BatchNorm statistics from the epilogue (as vp_conv5_*_stats_bf16x3): {pivot, sum(x - pivot), sum((x - pivot)^2)} per
// (workgroup, channel) straight from the fp32 accumulators + one finaliser launch; removes bn_partial_kernel<0>'s read of the
// activation for the layers that do not split K ---------------------------------------------------------------------------------
struct Stat32 { int ok, bm, tiles_m, gz, N; long M, R; };

static Stat32 stat_plan32(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  Stat32 sp = {0, 0, 0, 0, 0, 0, 0};
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Cbig <= 0 || Csmall <= 0 || (stride != 1 && stride != 2) || !f32_fast_on()) return sp;
  sp.M = (long)B * Hs * Ws;
  if (family == 0) {
    if (Cbig % 16 != 0 || Csmall < 64) return sp;
    sp.N = Csmall;
    if (gather_ns32(sp.M, sp.N, 25 * 2 * Cbig, 2 * Cbig, false, ACT_NONE) != 1) return sp;
    sp.gz = 1; sp.R = sp.M;
  } else {
    if (Csmall % 16 != 0 || Cbig < 64) return sp;
    sp.N = Cbig;
    if (scatter_ns32(sp.M, sp.N, 2 * Csmall, stride * stride) != 1) return sp;
    sp.gz = stride * stride; sp.R = sp.M * sp.gz;
  }
  sp.bm = tile32(sp.M, sp.N, sp.gz).bm;
  sp.tiles_m = (int)((sp.M + sp.bm - 1) / sp.bm);
  sp.ok = 1;
  return sp;
}

}  // namespace vp

using namespace vp;
extern "C" {

size_t vp_conv5_stats_f32_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  const Stat32 sp = stat_plan32(family, B, Hs, Ws, Cbig, Csmall, stride);
  return sp.ok ? (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float) : 0;
}

int vp_conv5_gather_stats_f32(const float* big, const float* w_p0, float* small_out, int B, int Hs, int Ws, int Cbig, int Csmall, int stride,
                              float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var, void* ws,
                              size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(big && w_p0 && small_out && mean && rstd && ws, "vp_conv5_gather_stats_f32: null pointer");
  const Stat32 sp = stat_plan32(0, B, Hs, Ws, Cbig, Csmall, stride);
  const ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, 5);
  VP_REQUIRE(sp.ok && f32_fast_gather_ok(g, big, w_p0, VP_ACT_NONE), "vp_conv5_gather_stats_f32: this shape cannot emit statistics (vp_conv5_stats_f32_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_gather_stats_f32: workspace too small");
  int rc = f32_fast_gather(big, w_p0, nullptr, small_out, g, VP_ACT_NONE, (hipStream_t)stream, (float*)ws);
  if (rc) return rc;
  return stats_slab_finish((const float*)ws, sp.tiles_m * sp.gz, sp.tiles_m, sp.bm, sp.M, sp.R, sp.N, eps, momentum, mean, rstd, running_mean, running_var, (hipStream_t)stream);
}

int vp_conv5_scatter_stats_f32(const float* small, const float* w_p1, float* big_out, int B, int Hs, int Ws, int Csmall, int Cbig, int stride,
                               float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var, void* ws,
                               size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(small && w_p1 && big_out && mean && rstd && ws, "vp_conv5_scatter_stats_f32: null pointer");
  const Stat32 sp = stat_plan32(1, B, Hs, Ws, Cbig, Csmall, stride);
  const ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, 5);
  VP_REQUIRE(sp.ok && f32_fast_scatter_ok(g, small, w_p1), "vp_conv5_scatter_stats_f32: this shape cannot emit statistics (vp_conv5_stats_f32_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_scatter_stats_f32: workspace too small");
  int rc = f32_fast_scatter(small, w_p1, big_out, g, (hipStream_t)stream, (float*)ws);
  if (rc) return rc;
  return stats_slab_finish((const float*)ws, sp.tiles_m * sp.gz, sp.tiles_m, sp.bm, sp.M, sp.R, sp.N, eps, momentum, mean, rstd, running_mean, running_var, (hipStream_t)stream);
}

}
