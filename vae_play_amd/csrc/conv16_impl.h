// Shared by conv16.hip (bf16 pairs) and conv16_f16.hip (fp16 pairs): launch planning and the templated launchers of the three
// split-operand families.  Everything here is static / a template, so that each translation unit instantiates only the problem
// descriptors it dispatches (the fp16 modes live in their own object file: the two compile in parallel).
#pragma once
#include <type_traits>
#include "common.h"
#include "igemm16p.h"
#include "wgrad5.h"
#include "halo.h"
#include "narrow.h"
#include "split.h"

using namespace vp;


// XCD-aware tile order (igemm16.h): valid when the row-tile count is a multiple of 8 and there are >= 2 column tiles
static int xcd_map_for(long M, long N, int gz, int ctile = 0) {
  const Tile16 t = choose_tile16(M, N, gz, false, ctile);
  const long gx = (M + t.bm - 1) / t.bm, gy = (N + t.bn - 1) / t.bn;
  return (gx % 8 == 0 && gy >= 2) ? 1 : 0;
}

static bool halo_enabled() { return true; }
// Where the register-staged gather / scatter kernels run their v_mfma_f32_16x16x32_bf16 form (igemm16.h M16; VP_IGEMM16_M16=0: nowhere).
// Measured per layer in the step at 32 images (profiles/r03_notes.md section 11): the 128x128 tile gains 4 - 5 % in both families
// (dec1.fwd 180 -> 171 us, dec2.fwd 175 -> 166, dec3.dgrad 174 -> 167), the scatter family's 128x64 tile 2 % on its largest launch
// (dec3.fwd 239 -> 234); the gather family's 128x64 tile loses 8 % (dec1.dgrad) and the 64x64 tile up to 12 % (enc2.fwd): not taken.
static bool igemm16_m16_rule(bool scatter, long M, long N, int gz, int ctile) {
  const char* e = VP_GETENV("VP_IGEMM16_M16");
  if (e && atoi(e) == 0) return false;
  if (ctile <= 0 || ctile % 64 != 0) return false;
  const Tile16 t = choose_tile16(M, N, gz, false, ctile);
  if (t.bm != 128) return false;
  return t.bn == 128 || (scatter && t.bn == 64 && M >= 65536);
}

// layers with fewer output tiles than this split K in two
static long conv_split_tiles() { return 384; }

// Kernel choice for a plain 5x5 VAE layer on split planes: the pipelined LDS-DMA kernel (igemm16p.h, bit-identical results) takes
// the shapes where its 256x256 eight-wave tile fills the chip -- N a multiple of 256 and at least one workgroup per CU, i.e. the
// N >= 256 layers from ~128 images per GPU on (+5-10 % there, profiles/r02_notes.md); everything else stays on igemm16_kernel.
// It also takes the small layers and one big gather layer (see plan16).  VP_IGEMM16P=0 disables it, =2 keeps only the 256x256 rule (A/B).
struct Launch16 { int pcfg, bm, bn; bool m16; };
static bool pipelined_m16() {
  static const bool on = [] { const char* e = getenv("VP_IGEMM16P_M16"); return !e || atoi(e) != 0; }();
  return on;
}
static Launch16 plan16(long M, long N, int gz, int ctile, int nsplit, size_t plane_elems_a, size_t plane_elems_b, long kmin, bool stat = false) {
  const Tile16 t = choose_tile16(M, N, gz, false, ctile);
  Launch16 l = {PCFG_NONE, t.bm, t.bn, false};
  static const int mode = [] { const char* e = getenv("VP_IGEMM16P"); return e ? atoi(e) : 1; }();
  if (!mode || ctile % 32 != 0 || kmin / 32 < 4) return l;
  if (plane_elems_a >= ((size_t)1 << 29) || plane_elems_b >= ((size_t)1 << 29)) return l;      // 32-bit byte offsets of both planes
  int cfg = PCFG_NONE;
  if (nsplit == 1 && N % 256 == 0 && M >= 256 && ((M + 255) / 256) * (N / 256) * gz >= 256) cfg = PCFG_256x256_S2;
  // the small layers (8x8 / 16x16 resolution at 32 images: M*N <= 1 M outputs, where igemm16_kernel runs 64x64 / 128x64 tiles with
  // K split in two): the pipelined 128x64 tile, same arithmetic bit for bit, is 4-10 % faster there (r02_a_kbench_m16_b32.log: c7b)
  else if (mode != 2 && M * N <= (1L << 20) && N >= 128 && M >= 128) cfg = PCFG_128x64_S3;
  // round 3: the big gather layer where the pipelined kernel on the v_mfma_f32_16x16x32_bf16 form (`m16`) beats igemm16_kernel at 32
  // images (profiles/r02_a_kbench_m16_b32.log: dec2.dgrad 176.0 -> 168.0 us on the 256x128 tile).  Same FLOPs per cycle, but the chip
  // holds a higher clock on this MFMA shape (MI355X_MICROARCH.md, DVFS give-back 7); equal to the 32x32x16 form to rounding, not bit
  // for bit (tests/test_gpu_kbench.py).  Never with the statistics epilogue, which reads the 32x32 accumulator layout
  // (dec2.fwd, the other shape kbench favours, has one).  A/B knob VP_IGEMM16P_M16=0.
  else if (mode != 2 && !stat && pipelined_m16() && nsplit == 1 && gz == 1 && N == 256 && M >= 16384 && M <= 65536 && ctile == 128) { cfg = PCFG_256x128_S3; l.m16 = true; }
  if (cfg == PCFG_NONE) return l;
  int bm, bn;
  pcfg_tile(cfg, bm, bn);
  if (bn > N || bm > M) return l;
  l.pcfg = cfg; l.bm = bm; l.bn = bn;
  return l;
}
static int xcd_map_tile(long M, long N, int bm, int bn) {
  const long gx = (M + bm - 1) / bm, gy = (N + bn - 1) / bn;
  return (gx % 8 == 0 && gy >= 2) ? 1 : 0;
}

// split-K decisions (shared by the launchers and by the statistics plan)
static int gather_nsplit(long M, int N, int K, int Cbig, bool plain5, bool has_bias, int act) {
  // few output tiles and a long K (the 8x8-resolution layers: 256 workgroups = one per CU): split K in two and
  // accumulate both halves with fp32 atomics onto a zeroed output (two addends: the sum does not depend on order)
  const long tiles = ((M + 127) / 128) * ((N + 63) / 64);
  return (plain5 && !has_bias && act == VP_ACT_NONE && Cbig % 64 == 0 && tiles < conv_split_tiles() && K >= 4096) ? 2 : 1;
}
static int scatter_nsplit(long M, int N, int Csmall, int stride, bool plain5) {
  const long tiles = ((M + 127) / 128) * ((N + 63) / 64) * stride * stride;
  return (plain5 && Csmall % 64 == 0 && tiles < conv_split_tiles() && 4 * Csmall >= 1024) ? 2 : 1;
}

template <class PF>
static int gather16_t(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws, int Hb,
                      int Wb, int Cbig, int Csmall, int ks, int stride, int act, bool plain5, vp_stream stream, float* stat = nullptr,
                      float alpha = 1.f) {
  PF p;
  p.alpha = alpha;
  p.zero = vp_zero_page();
  p.g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  p.big = (const u16*)big_split; p.big_plane = (size_t)B * p.g.Hb * p.g.Wb * Cbig;
  p.w = (const u16*)w_p0_split; p.w_plane = (size_t)Csmall * Cbig * p.g.nt;
  p.bias = bias; p.out = small_out; p.act = act;
  p.M = B * Hs * Ws; p.N = Csmall; p.K = p.g.nt * Cbig;
  p.nsplit = gather_nsplit(p.M, p.N, p.K, Cbig, plain5, bias != nullptr, act);
  p.stat = stat;
  if (stat && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_gather_stats_bf16x3: this shape splits K");
  p.k_per_split = p.nsplit == 2 ? ((p.K / 64 + 1) / 2) * 64 : p.K;
  if (p.nsplit == 2 && hipMemsetAsync(small_out, 0, (size_t)p.M * p.N * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(VP_ERR_LAUNCH, "vp_conv5_gather_bf16x3: memset failed");
  if constexpr (std::is_same<PF, ProbF16>::value) {
    const Launch16 l = plan16(p.M, p.N, p.nsplit, Cbig, p.nsplit, p.big_plane, p.w_plane, p.k_per_split, stat != nullptr);
    if (l.pcfg != PCFG_NONE) {
      PF16 q;
      static_cast<ProbF16&>(q) = p;
      q.xcd_map = xcd_map_tile(p.M, p.N, l.bm, l.bn);
      launch_igemm16p(q, l.pcfg, p.M, p.N, p.nsplit, (hipStream_t)stream, Cbig, true, l.m16);
      return check_launch("vp_conv_gather_bf16x3(pipelined)");
    }
  }
  p.xcd_map = xcd_map_for(p.M, p.N, p.nsplit, Cbig);
  launch_igemm16(p, p.M, p.N, p.nsplit, (hipStream_t)stream, Cbig, std::is_same<PF, ProbF16>::value && igemm16_m16_rule(false, p.M, p.N, p.nsplit, Cbig));
  return check_launch("vp_conv_gather_bf16x3");
}

template <class PT>
static int scatter16_t(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb, int Csmall,
                       int Cbig, int ks, int stride, bool plain5, vp_stream stream, float* stat = nullptr, float alpha = 1.f) {
  PT p;
  p.alpha = alpha;
  p.zero = vp_zero_page();
  p.g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  p.small = (const u16*)small_split; p.small_plane = (size_t)B * Hs * Ws * Csmall;
  p.w = (const u16*)w_p1_split; p.w_plane = (size_t)Csmall * Cbig * p.g.nt;
  p.out = big_out; p.M = B * Hs * Ws; p.N = Cbig;
  p.nsplit = scatter_nsplit(p.M, p.N, Csmall, stride, plain5);
  p.stat = stat;
  if (stat && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_scatter_stats_bf16x3: this shape splits K");
  if (p.nsplit == 2 &&
      hipMemsetAsync(big_out, 0, (size_t)B * p.g.Hb * p.g.Wb * Cbig * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(VP_ERR_LAUNCH, "vp_conv5_scatter_bf16x3: memset failed");
  if constexpr (std::is_same<PT, ProbT16>::value) {
    const Launch16 l = plan16(p.M, p.N, stride * stride * p.nsplit, Csmall, p.nsplit, p.small_plane, p.w_plane, (long)stride * stride * Csmall, stat != nullptr);
    if (l.pcfg != PCFG_NONE && stride == 2) {
      PT16 q;
      static_cast<ProbT16&>(q) = p;
      q.xcd_map = xcd_map_tile(p.M, p.N, l.bm, l.bn);
      launch_igemm16p(q, l.pcfg, p.M, p.N, stride * stride * p.nsplit, (hipStream_t)stream, Csmall, true, l.m16);
      return check_launch("vp_conv_scatter_bf16x3(pipelined)");
    }
  }
  p.xcd_map = xcd_map_for(p.M, p.N, stride * stride * p.nsplit, Csmall);
  {
    const Tile16 t = choose_tile16(p.M, p.N, stride * stride * p.nsplit, false, Csmall);
    const long wgs = ((p.M + t.bm - 1) / t.bm) * ((p.N + t.bn - 1) / t.bn) * stride * stride * p.nsplit;
    p.pair_phases = (stride == 2 && p.nsplit == 1 && wgs <= 512) ? 1 : 0;
  }
  launch_igemm16(p, p.M, p.N, stride * stride * p.nsplit, (hipStream_t)stream, Csmall, std::is_same<PT, ProbT16>::value && igemm16_m16_rule(true, p.M, p.N, stride * stride * p.nsplit, Csmall));
  return check_launch("vp_conv_scatter_bf16x3");
}

// Tap pairs (ProbW16T<.., PAIR>, igemm16.h): a 5x5 or 3x3 layer whose big side has 32 or 64 channels contracts TWO taps per workgroup, the
// two taps' channels side by side in a 64- / 128-column tile: a 32-channel layer no longer pads half of a 64-column tile (and leaves the
// slow partial-tile loads), a 64-channel layer gets a 128-column tile's operand reuse.  (nt + 1) / 2 pairs instead of nt taps per split:
// the pixel range is split twice as deep where the workspace allows.  A 64-channel SMALL side takes the 64x128 tile (wave tiles
// 32x64) instead of 64x64 (32x32): with 128 | 256 | ... big channels directly (kind 4), with 64 big channels on tap pairs (kind 3) --
// the font U-Net's full-resolution layers, 110 -> ~170 TFLOP/s (tools/microbench_font_layers.py).
//   kind 1: pairs, 64x64 tile (Cb = 32) | 2: pairs, 128x128 (Cb = 64, Cs % 128 == 0) | 3: pairs, 64x128 (Cb = 64, Cs % 64 == 0)
//   kind 4: single taps, 64x128 tile (Cs % 128 == 64, Cb % 128 == 0)
static inline int wgrad_wide_kind(const ConvGeom& g) {
  const bool pair_nt = g.nt == 25 || g.nt == 9;
  if (pair_nt && g.Cb == 32 && g.Cs % 64 == 0) return 1;
  if (pair_nt && g.Cb == 64 && g.Cs % 128 == 0) return 2;
  if (pair_nt && g.Cb == 64 && g.Cs % 64 == 0) return 3;
  if (g.Cs % 128 == 64 && g.Cb % 128 == 0) return 4;
  return 0;
}
static inline bool wgrad_pair_applicable(const ConvGeom& g, bool /*plain5*/ = true) { const int k = wgrad_wide_kind(g); return k >= 1 && k <= 3; }
static inline int wgrad_pair_nsplit(const ConvGeom& g, int ns) {
  const long K = (long)g.B * g.Hs * g.Ws, maxs = (K + 511) / 512;
  long n2 = 2L * ns;
  if (n2 > maxs) n2 = maxs;
  if (n2 > 64) n2 = 64;
  return n2 > ns ? (int)n2 : ns;
}

template <bool K5>      // (a template so that only the bf16 translation unit instantiates its kernels)
static int wgrad16_wide(int kind, const void* big_split, const void* small_split, float* dw_ref, const ConvGeom& g, int ns, void* ws,
                        vp_stream stream) {
  const int per = ((long)g.B * g.Hs * g.Ws + ns - 1) / ns;
  auto fill = [&](auto& p, int N) {
    p.alpha = 1.f;
    p.zero = vp_zero_page();
    p.g = g;
    p.big = (const u16*)big_split; p.big_plane = (size_t)g.B * g.Hb * g.Wb * g.Cb;
    p.small = (const u16*)small_split; p.small_plane = (size_t)g.B * g.Hs * g.Ws * g.Cs;
    p.slab = (float*)ws; p.M = g.Cs; p.N = N; p.K = g.B * g.Hs * g.Ws;
    p.nsplit = ns;
    p.k_per_split = ((per + 31) / 32) * 32;
  };
  hipStream_t s = (hipStream_t)stream;
  if (kind == 4) {
    ProbW16T<K5, 0, false> p;
    fill(p, g.Cb);
    hipLaunchKernelGGL((igemm16_kernel<ProbW16T<K5, 0, false>, 64, 128, 2, 2, 32, true>), dim3(g.Cs / 64, g.Cb / 128, (unsigned)(g.nt * ns)), dim3(256), 0, s, p);
  } else {
    ProbW16T<K5, 0, true> p;
    fill(p, 2 * g.Cb);
    const unsigned gz = (unsigned)((g.nt + 1) / 2) * (unsigned)ns;
    if (kind == 1) hipLaunchKernelGGL((igemm16_kernel<ProbW16T<K5, 0, true>, 64, 64, 2, 2, 32, true>), dim3(g.Cs / 64, 1, gz), dim3(256), 0, s, p);
    else if (kind == 2) hipLaunchKernelGGL((igemm16_kernel<ProbW16T<K5, 0, true>, 128, 128, 2, 2, 32, true>), dim3(g.Cs / 128, 1, gz), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm16_kernel<ProbW16T<K5, 0, true>, 64, 128, 2, 2, 32, true>), dim3(g.Cs / 64, 1, gz), dim3(256), 0, s, p);
  }
  int rc = check_launch("vp_conv_wgrad_bf16x3(wide tiles / tap pairs)");
  if (rc) return rc;
  return slab_reduce_launch((const float*)ws, dw_ref, g.Cs, g.Cb, ns, s, g.nt);
}

// Row-of-taps weight gradient (wgrad5.h): the plain 5x5 stride-2 layers with 128 | Cs and Cb = 64 or 128 | Cb -- every block of the
// VAE / VAE-GAN except the 3-channel edge layers.  A/B knob VP_WGRAD5=0 sends them back to the one-tap-per-workgroup kernels.
static inline int wgrad5_kind(const ConvGeom& g) {
  const char* e = VP_GETENV("VP_WGRAD5");
  return (e && atoi(e) == 0) ? 0 : wgrad5_bn(g);
}

template <int MODE>
static int wgrad16_rows(int bn, const void* big_split, const void* small_split, float* dw_ref, const ConvGeom& g, void* ws, size_t ws_bytes,
                        vp_stream stream, float alpha, int max_cus) {
  int kper = 0;
  const int ns = wgrad5_nsplit(g, bn, &kper, max_cus);
  if (ws_bytes < wgrad5_slab_floats(g, bn, ns) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv_wgrad_bf16x3: workspace too small");
  int slabs = 0;
  wgrad5_launch<MODE>(big_split, small_split, (float*)ws, g, bn, ns, kper, alpha, (hipStream_t)stream, &slabs);
  int rc = check_launch("vp_conv_wgrad_bf16x3(rows of taps)");
  if (rc) return rc;
  return slab_reduce_launch((const float*)ws, dw_ref, g.Cs, g.Cb, slabs, (hipStream_t)stream, g.nt);
}

template <class PW>
static int wgrad16_t(const void* big_split, const void* small_split, float* dw_ref, const ConvGeom& g, int ns, void* ws, vp_stream stream,
                     float alpha = 1.f) {
  const int B = g.B, Hs = g.Hs, Ws = g.Ws, Cbig = g.Cb, Csmall = g.Cs;
  PW p;
  p.alpha = alpha;
  p.zero = vp_zero_page();
  p.g = g;
  p.big = (const u16*)big_split; p.big_plane = (size_t)B * g.Hb * g.Wb * Cbig;
  p.small = (const u16*)small_split; p.small_plane = (size_t)B * Hs * Ws * Csmall;
  p.slab = (float*)ws; p.M = Csmall; p.N = Cbig; p.K = B * Hs * Ws;
  p.nsplit = ns;
  const int per = (p.K + ns - 1) / ns;
  p.k_per_split = ((per + 31) / 32) * 32;
  launch_igemm16(p, p.M, p.N, g.nt * ns, (hipStream_t)stream);
  int rc = check_launch("vp_conv_wgrad_bf16x3(main)");
  if (rc) return rc;
  return slab_reduce_launch((const float*)ws, dw_ref, Csmall, Cbig, ns, (hipStream_t)stream, g.nt);
}


// ---- BatchNorm statistics from the convolution epilogue -----------------------------------------------------------------
// A 5x5 VAE layer on the split-bf16 kernels can emit {pivot, sum(x - pivot), sum((x - pivot)^2)} per (workgroup, output channel)
// from its accumulators (igemm16.h epilogue_stats32); one finaliser launch then produces mean / rstd / running statistics.
// This replaces bn_partial_kernel<0>, i.e. one full read of the activation per BatchNorm layer.
struct StatPlan { int ok, bm, tiles_m, gz, N; long M, R; };

static StatPlan stat_plan(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride, bool x2 = false) {
  StatPlan sp = {0, 0, 0, 0, 0, 0, 0};
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Cbig <= 0 || Csmall <= 0 || (stride != 1 && stride != 2)) return sp;
  if (Cbig % 8 != 0 || Csmall % 8 != 0) return sp;
  sp.M = (long)B * Hs * Ws;
  if (family == 0) {
    if (!x2 && halo_enabled() && halo_gather_kind(B, Hs, Ws, Cbig, Csmall, stride)) return sp;
    sp.N = Csmall;
    if (gather_nsplit(sp.M, sp.N, 25 * Cbig, Cbig, true, false, VP_ACT_NONE) != 1) return sp;
    sp.gz = 1;
    sp.R = sp.M;
  } else {
    if (!x2 && halo_enabled() && halo_scatter_kind(B, Hs, Ws, Csmall, Cbig, stride)) return sp;
    sp.N = Cbig;
    if (scatter_nsplit(sp.M, sp.N, Csmall, stride, true) != 1) return sp;
    sp.gz = stride * stride;
    sp.R = sp.M * stride * stride;
  }
  const size_t act_plane = family == 0 ? (size_t)sp.M * stride * stride * Cbig : (size_t)sp.M * Csmall;
  const Launch16 l = family == 0 ? plan16(sp.M, sp.N, sp.gz, Cbig, 1, act_plane, (size_t)Csmall * Cbig * 25, 25L * Cbig, true)
                                 : plan16(sp.M, sp.N, sp.gz, Csmall, 1, act_plane, (size_t)Csmall * Cbig * 25, (long)stride * stride * Csmall, true);
  sp.bm = (!x2 && l.pcfg != PCFG_NONE && !(family == 1 && stride != 2)) ? l.bm : choose_tile16(sp.M, sp.N, sp.gz, false, family == 0 ? Cbig : Csmall).bm;
  sp.tiles_m = (int)((sp.M + sp.bm - 1) / sp.bm);
  sp.ok = 1;
  return sp;
}

// ---- argument checks + dispatch of the three families; F16 = 0: bf16 pairs (three products), 2 | 3: fp16 pairs, products per fragment pair
template <int F16>
static int gather16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws, int Hb,
                    int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream, float alpha = 1.f) {
  VP_REQUIRE(big_split && w_p0_split && small_out, "vp_conv_gather_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig > 0 && Csmall > 0 && Cbig % 8 == 0, "vp_conv_gather_bf16x3: Cbig must be a multiple of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_gather_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_gather_bf16x3: kernel size must be 1, 3 or 5");
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_SIGMOID, "vp_conv_gather_bf16x3: epilogue supports none|sigmoid");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if constexpr (F16 != 0) {    // fp16-pair planes, f16 = products per fragment pair: always the implicit-GEMM kernels (the halo kernels read bf16 pairs)
    VP_REQUIRE(alpha > 0.f, "vp_conv_gather_f16: out_scale must be positive");
#define VP_G16(PF, P5) gather16_t<PF>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, P5, stream, nullptr, alpha)
    if constexpr (F16 == 2) return plain5 ? VP_G16(ProbF16X, true) : VP_G16(ProbF16KX, false);
    else return plain5 ? VP_G16(ProbF16H, true) : VP_G16(ProbF16KH, false);
#undef VP_G16
  } else {
  if (plain5 && halo_enabled())
    if (const int kind = halo_gather_kind(B, Hs, Ws, Cbig, Csmall, stride))
      return halo_gather_launch(kind, big_split, w_p0_split, bias, small_out, B, Hs, Ws, Cbig, Csmall, stride, act, (hipStream_t)stream);
  if (!plain5) return gather16_t<ProbF16K>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, false, stream);
  return gather16_t<ProbF16>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, true, stream);
  }
}

template <int F16>
static int scatter16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb, int Csmall,
                     int Cbig, int ks, int stride, vp_stream stream, float alpha = 1.f) {
  VP_REQUIRE(small_split && w_p1_split && big_out, "vp_conv_scatter_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig > 0 && Csmall > 0 && Csmall % 8 == 0, "vp_conv_scatter_bf16x3: Csmall must be a multiple of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_scatter_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_scatter_bf16x3: kernel size must be 1, 3 or 5");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if constexpr (F16 != 0) {
    VP_REQUIRE(alpha > 0.f, "vp_conv_scatter_f16: out_scale must be positive");
#define VP_S16(PT, P5) scatter16_t<PT>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, P5, stream, nullptr, alpha)
    if constexpr (F16 == 2) return plain5 ? VP_S16(ProbT16X, true) : VP_S16(ProbT16KX, false);
    else return plain5 ? VP_S16(ProbT16H, true) : VP_S16(ProbT16KH, false);
#undef VP_S16
  } else {
  if (plain5 && halo_enabled())
    if (const int kind = halo_scatter_kind(B, Hs, Ws, Csmall, Cbig, stride))
      return halo_scatter_launch(kind, small_split, w_p1_split, big_out, B, Hs, Ws, Csmall, Cbig, (hipStream_t)stream);
  if (!plain5) return scatter16_t<ProbT16K>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, false, stream);
  return scatter16_t<ProbT16>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, true, stream);
  }
}

template <int F16>
static int wgrad16(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                   int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream, float alpha = 1.f, int max_cus = 0) {
  VP_REQUIRE(big_split && small_split && dw_ref && ws, "vp_conv_wgrad_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig % 8 == 0 && Csmall % 8 == 0 && Cbig > 0 && Csmall > 0,
             "vp_conv_wgrad_bf16x3: channel counts must be multiples of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_wgrad_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_wgrad_bf16x3: kernel size must be 1, 3 or 5");
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  if (const int bn = wgrad5_kind(g)) {
    if constexpr (F16 != 0) {
      VP_REQUIRE(alpha > 0.f, "vp_conv_wgrad_f16x2: out_scale must be positive");
      return wgrad16_rows<1>(bn, big_split, small_split, dw_ref, g, ws, ws_bytes, stream, alpha, max_cus);
    } else {
      return wgrad16_rows<0>(bn, big_split, small_split, dw_ref, g, ws, ws_bytes, stream, 1.f, max_cus);
    }
  }
  const int ns = wgrad_nsplit(g);
  if (ws_bytes < wgrad_slab_floats(g, ns) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv_wgrad_bf16x3: workspace too small");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if constexpr (F16 != 0) {
    static_assert(F16 == 0 || F16 == 2, "weight gradients have no 3-product fp16 form");
    VP_REQUIRE(alpha > 0.f, "vp_conv_wgrad_f16x2: out_scale must be positive");
    if (!plain5) return wgrad16_t<ProbW16KX>(big_split, small_split, dw_ref, g, ns, ws, stream, alpha);
    return wgrad16_t<ProbW16X>(big_split, small_split, dw_ref, g, ns, ws, stream, alpha);
  } else {
  if (const int kind = wgrad_wide_kind(g)) {
    int n = ns;
    if (kind <= 3) {      // deeper split when the caller's workspace holds it (the *_workspace_bytes query asks for it)
      const int n2 = wgrad_pair_nsplit(g, ns);
      n = ws_bytes >= wgrad_slab_floats(g, n2) * sizeof(float) ? n2 : ns;
    }
    return plain5 ? wgrad16_wide<true>(kind, big_split, small_split, dw_ref, g, n, ws, stream)
                  : wgrad16_wide<false>(kind, big_split, small_split, dw_ref, g, n, ws, stream);
  }
  if (!plain5) return wgrad16_t<ProbW16K>(big_split, small_split, dw_ref, g, ns, ws, stream);
  return wgrad16_t<ProbW16>(big_split, small_split, dw_ref, g, ns, ws, stream);
  }
}

// finaliser of the epilogue statistics (kernel in conv16.hip)
int vp16_stats_finish(const StatPlan& sp, const float* slab, float eps, float momentum, float* mean, float* rstd, float* rm, float* rv,
                      vp_stream stream);
