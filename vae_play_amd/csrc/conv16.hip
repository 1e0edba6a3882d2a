// Split-bf16 ("bf16x3") convolution families: C ABI + the producers of split planes.
#include <type_traits>
#include "common.h"
#include "igemm16p.h"
#include "halo.h"
#include "narrow.h"
#include "split.h"

namespace vp {

__global__ void split_kernel(const float* __restrict__ x, u16_t* __restrict__ out, size_t n, int fmt, float scale) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const vp_f32x4 v = *reinterpret_cast<const vp_f32x4*>(x + i * 4);
    store_split4(out, n, i * 4, v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale, fmt);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    u16_t h, l;
    split_f32(x[i] * scale, h, l, fmt);
    out[i] = h;
    out[n + i] = l;
  }
}

// [npix][C] fp32 -> split planes [npix][Cpad] with zero channels C..Cpad-1 (Cpad a multiple of 8): lets a layer whose channel
// count is not a multiple of 8 (1 or 2 outputs, 32 + 2 coordinate channels, models/blocks.py:97-146) use the split-bf16 kernels.
__global__ void split_pad_kernel(const float* __restrict__ x, u16_t* __restrict__ out, size_t npix, int C, int Cpad) {
  const size_t n = npix * (size_t)Cpad, n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i * 4, pix = e / Cpad;
    const int c = (int)(e - pix * Cpad);            // Cpad % 4 == 0: the four elements share a pixel
    const float* src = x + pix * (size_t)C;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = c + j < C ? src[c + j] : 0.f;
    store_split4(out, n, e, v[0], v[1], v[2], v[3]);
  }
}

// Weight re-pack through LDS so that both the fp32 reads and the packed writes are contiguous.
//   MODE 0: p0[cs][t][cb] <- w[cs][cb][t]   (one cs, 64 cb per workgroup: 1600 contiguous floats in)
//   MODE 1: p1[cb][t][cs] <- w[cs][cb][t]   (one cb, 64 cs per workgroup: 64 x 100-B segments in)
// SPLIT: write bf16 hi/lo planes (plane stride n) instead of fp32.
// CsP >= Cs: MODE 1 may zero-pad the small-channel (inner) dimension of p1 to CsP (edge layers on the bf16x3 path).
template <int MODE, bool SPLIT>
__global__ void __launch_bounds__(256) pack_w5_tiled_kernel(const float* __restrict__ w, void* __restrict__ outv, int Cs, int Cb,
                                                            int CsP, int nt, int fmt = SPLIT_BF16) {
  __shared__ float tile[64][kTaps + 1];
  const size_t n = (size_t)(MODE == 1 ? CsP : Cs) * Cb * nt;
  const int fixed = blockIdx.x, j0 = blockIdx.y * 64;
  const int lim = (MODE == 0 ? Cb : Cs) - j0;           // valid entries of the 64-wide tile
  const int lim_out = (MODE == 0 ? Cb : CsP) - j0;      // entries written (zeros beyond lim)
  for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {
    const int j = idx / nt, t = idx - j * nt;
    if (j < lim) {
      const size_t src = MODE == 0 ? ((size_t)fixed * Cb + j0 + j) * nt + t : ((size_t)(j0 + j) * Cb + fixed) * nt + t;
      tile[j][t] = w[src];
    }
  }
  __syncthreads();
  const int inner = MODE == 0 ? Cb : CsP;
  for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {
    const int t = idx >> 6, j = idx & 63;
    if (j < lim_out) {
      const size_t o = ((size_t)fixed * nt + t) * inner + j0 + j;
      const float v = j < lim ? tile[j][t] : 0.f;
      if constexpr (SPLIT) {
        u16_t h, l;
        split_f32(v, h, l, fmt);
        ((u16_t*)outv)[o] = h;
        ((u16_t*)outv)[n + o] = l;
      } else {
        ((float*)outv)[o] = v;
      }
    }
  }
}

template <bool SPLIT>
static int pack_w5_launch(const float* w, void* p0, void* p1, int Cs, int Cb, hipStream_t s, const char* what, int CsP = 0,
                          int nt = kTaps) {
  if (CsP < Cs) CsP = Cs;
  if (p0) hipLaunchKernelGGL((pack_w5_tiled_kernel<0, SPLIT>), dim3(Cs, (Cb + 63) / 64), dim3(256), 0, s, w, p0, Cs, Cb, Cs, nt);
  if (p1) hipLaunchKernelGGL((pack_w5_tiled_kernel<1, SPLIT>), dim3(Cb, (CsP + 63) / 64), dim3(256), 0, s, w, p1, Cs, Cb, CsP, nt);
  return check_launch(what);
}

// ---- all conv weights of a step in ONE launch -------------------------------------------------------------
// A step re-packs ~10 weight tensors into 1-2 layouts each: 15-20 launches of 7-9 us whose cost is launch latency,
// not bytes.  The batch kernel walks a job table passed by value: job = (tensor, layout, fp32|split), blockIdx.x
// is a tile index into the concatenation of all jobs' tiles.
constexpr int kMaxPackJobs = 32;
struct PackJob { const float* w; void* out; int Cs, Cb, CsP, mode, split, tile_begin; };
struct PackTable { PackJob j[kMaxPackJobs]; int n; };

template <int MODE, bool SPLIT>
__device__ __forceinline__ void pack_tile(float (*tile)[kTaps + 1], const float* __restrict__ w, void* __restrict__ outv, int Cs,
                                          int Cb, int CsP, int fixed, int j0, int fmt) {
  constexpr int nt = kTaps;
  const size_t n = (size_t)(MODE == 1 ? CsP : Cs) * Cb * nt;
  const int lim = (MODE == 0 ? Cb : Cs) - j0, lim_out = (MODE == 0 ? Cb : CsP) - j0;
  for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {
    const int j = idx / nt, t = idx - j * nt;
    if (j < lim) tile[j][t] = w[MODE == 0 ? ((size_t)fixed * Cb + j0 + j) * nt + t : ((size_t)(j0 + j) * Cb + fixed) * nt + t];
  }
  __syncthreads();
  const int inner = MODE == 0 ? Cb : CsP;
  for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {
    const int t = idx >> 6, j = idx & 63;
    if (j < lim_out) {
      const size_t o = ((size_t)fixed * nt + t) * inner + j0 + j;
      const float v = j < lim ? tile[j][t] : 0.f;
      if constexpr (SPLIT) {
        u16_t h, l;
        split_f32(v, h, l, fmt);
        ((u16_t*)outv)[o] = h;
        ((u16_t*)outv)[n + o] = l;
      } else {
        ((float*)outv)[o] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(256) pack_w5_batch_kernel(const PackTable tab) {
  __shared__ float tile[64][kTaps + 1];
  int ji = 0;
  for (int k = 1; k < tab.n; ++k)
    if ((int)blockIdx.x >= tab.j[k].tile_begin) ji = k;     // tile_begin is increasing: the last job that starts at or before us
  const PackJob& jb = tab.j[ji];
  const int tl = blockIdx.x - jb.tile_begin;
  const int ntile = ((jb.mode == 0 ? jb.Cb : jb.CsP) + 63) / 64;
  const int fixed = tl / ntile, j0 = (tl - fixed * ntile) * 64;
  if (jb.mode == 0) {
    const int fmt = jb.split == 2 ? SPLIT_F16 : SPLIT_BF16;       // vp_pack_job.split: 0 fp32 | 1 bf16 pair | 2 fp16 pair
    if (jb.split) pack_tile<0, true>(tile, jb.w, jb.out, jb.Cs, jb.Cb, jb.CsP, fixed, j0, fmt);
    else pack_tile<0, false>(tile, jb.w, jb.out, jb.Cs, jb.Cb, jb.CsP, fixed, j0, fmt);
  } else {
    const int fmt = jb.split == 2 ? SPLIT_F16 : SPLIT_BF16;
    if (jb.split) pack_tile<1, true>(tile, jb.w, jb.out, jb.Cs, jb.Cb, jb.CsP, fixed, j0, fmt);
    else pack_tile<1, false>(tile, jb.w, jb.out, jb.Cs, jb.Cb, jb.CsP, fixed, j0, fmt);
  }
}

int pack_w5_f32_launch(const float* w, float* p0, float* p1, int Cs, int Cb, hipStream_t s, int nt) {
  return pack_w5_launch<false>(w, p0, p1, Cs, Cb, s, "vp_pack_w_f32", 0, nt);
}


// ---- first encoder conv (1 or 3 image channels, 5x5, stride 2) as a 1x1 convolution over a materialised im2col -------------
// K = 25*C = 75 is hostile to the implicit GEMM (scalar gathers, >= 90 % padded tiles: 47 us forward on the exact-f32 kernel,
// 152 us for the weight gradient on a VALU kernel at B = 32, 128x128).  The image is tiny (6 MB), so its im2col is simply
// written out once per step as split planes Xcol[pixel][KC] (KC = 96 for 3 channels: five kernel rows x 16 columns, of which
// 15 = (q, cin) are used; 50 MB) and BOTH the forward convolution and the weight gradient become 1x1 layers on the split-bf16
// MFMA kernels (vp_conv_gather_bf16x3 / vp_conv_wgrad_bf16x3 with ks = 1).  Column of (r, q, cin) = r*GW + q*C + cin.
static inline int im2col5_gw(int C) { return C == 3 ? 16 : (C == 1 ? 8 : 0); }
static inline int im2col5_kc(int C) { const int g = im2col5_gw(C); return g ? ((5 * g + 31) / 32) * 32 : 0; }

template <int C>
__global__ void __launch_bounds__(256) im2col5s2_split_kernel(const float* __restrict__ x, u16_t* __restrict__ out, int B, int Hb, int Wb,
                                                              int Hs, int Ws, int nchw, int fmt) {
  // blockIdx.x = output row (b, hs); a thread = (ws, 8-column group): 16-B stores into both planes, every division by a constant
  constexpr int GW = C == 3 ? 16 : 8, KC = ((5 * GW + 31) / 32) * 32, G = KC / 8;
  const size_t npix = (size_t)B * Hs * Ws, n = npix * KC;
  const int row = blockIdx.x, b = row / Hs, hs = row - b * Hs;
  for (int i = threadIdx.x; i < Ws * G; i += blockDim.x) {
    const int ws = i / G, c0 = (i - ws * G) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = c0 + j, r = col / GW, rem = col - r * GW;
      const int q = rem / C, cin = rem - q * C;
      const int h = 2 * hs - 2 + r, w_ = 2 * ws - 2 + q;
      const bool ok = r < 5 && q < 5 && h >= 0 && h < Hb && w_ >= 0 && w_ < Wb;
      v[j] = ok ? (nchw ? x[(((size_t)b * C + cin) * Hb + h) * Wb + w_] : x[(((size_t)b * Hb + h) * Wb + w_) * C + cin]) : 0.f;
    }
    const size_t o = ((size_t)row * Ws + ws) * KC + c0;
    store_split4(out, n, o, v[0], v[1], v[2], v[3], fmt);
    store_split4(out, n, o + 4, v[4], v[5], v[6], v[7], fmt);
  }
}

// w_ref [Cout][C][5][5] -> split planes [Cout][KC] in the im2col column order (zero columns where no tap lives)
__global__ void pack_w_im2col5_split_kernel(const float* __restrict__ w, u16_t* __restrict__ out, int Cout, int C, int GW, int KC,
                                            int fmt) {
  const size_t n = (size_t)Cout * KC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / KC), col = (int)(i - (size_t)co * KC);
    const int r = col / GW, rem = col - r * GW, q = rem / C, cin = rem - q * C;
    const float v = (r < 5 && q < 5) ? w[(((size_t)co * C + cin) * 5 + r) * 5 + q] : 0.f;
    u16_t h, l;
    split_f32(v, h, l, fmt);
    out[i] = h;
    out[n + i] = l;
  }
}

// dW in im2col column order [Cout][KC] -> reference layout [Cout][C][5][5]
__global__ void unpack_dw_im2col5_kernel(const float* __restrict__ dwc, float* __restrict__ dw, int Cout, int C, int GW, int KC) {
  const size_t n = (size_t)Cout * C * 25;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % 5), r = (int)((i / 5) % 5), cin = (int)((i / 25) % C), co = (int)(i / ((size_t)25 * C));
    dw[i] = dwc[(size_t)co * KC + r * GW + q * C + cin];
  }
}

}  // namespace vp

using namespace vp;

// VP_HALO=0 sends the narrow-channel layers back to the implicit-GEMM kernels (A/B runs)
// XCD-aware tile order (igemm16.h): valid when the row-tile count is a multiple of 8 and there are >= 2 column tiles
static int xcd_map_for(long M, long N, int gz) {
  const char* e = getenv("VP_XCD_MAP");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return 0;
  const Tile16 t = choose_tile16(M, N, gz);
  const long gx = (M + t.bm - 1) / t.bm, gy = (N + t.bn - 1) / t.bn;
  if (mode == 2) return (gx % 8 == 0 && gx >= 16) ? 2 : 0;      // band order (igemm16.h): any column-tile / z count
  return (gx % 8 == 0 && gy >= 2) ? 1 : 0;
}

static bool halo_enabled() {
  static const bool on = [] { const char* e = getenv("VP_HALO"); return !e || atoi(e) != 0; }();
  return on;
}

// layers with fewer output tiles than this split K in two (A/B knob VP_CONV_SPLIT_TILES)
static long conv_split_tiles() {
  const char* e = getenv("VP_CONV_SPLIT_TILES");
  return e ? atol(e) : 384;
}

// Kernel choice for a plain 5x5 VAE layer on split planes: the pipelined LDS-DMA kernel (igemm16p.h, bit-identical results) takes
// the shapes where its 256x256 eight-wave tile fills the chip -- N a multiple of 256 and at least one workgroup per CU, i.e. the
// N >= 256 layers from ~128 images per GPU on (+5-10 % there, profiles/r02_notes.md); everything else stays on igemm16_kernel.
// VP_IGEMM16P=0 disables it, VP_IGEMM16P_CFG=<n> forces configuration n of igemm16p.h wherever it applies (experiments).
struct Launch16 { int pcfg, bm, bn; };
static Launch16 plan16(long M, long N, int gz, int ctile, int nsplit, size_t plane_elems_a, size_t plane_elems_b, long kmin) {
  const Tile16 t = choose_tile16(M, N, gz);
  Launch16 l = {PCFG_NONE, t.bm, t.bn};
  static const int mode = [] { const char* e = getenv("VP_IGEMM16P"); return e ? atoi(e) : 1; }();
  static const int forced = [] { const char* e = getenv("VP_IGEMM16P_CFG"); return e ? atoi(e) : 0; }();
  if (!mode || nsplit != 1 || ctile % 32 != 0 || kmin / 32 < 4) return l;
  if (plane_elems_a >= ((size_t)1 << 29) || plane_elems_b >= ((size_t)1 << 29)) return l;      // 32-bit byte offsets of both planes
  int cfg = PCFG_NONE;
  if (forced > PCFG_NONE && forced < PCFG_COUNT) cfg = forced;
  else if (N % 256 == 0 && M >= 256 && ((M + 255) / 256) * (N / 256) * gz >= 256) cfg = PCFG_256x256_S2;
  if (cfg == PCFG_NONE) return l;
  int bm, bn;
  pcfg_tile(cfg, bm, bn);
  if (bn > N || bm > M) return l;
  l.pcfg = cfg; l.bm = bm; l.bn = bn;
  return l;
}
static int xcd_map_tile(long M, long N, int bm, int bn) {
  const char* e = getenv("VP_XCD_MAP");
  if (e && atoi(e) == 0) return 0;
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

struct BnBwdArgs { const float *x, *mean, *rstd, *gamma, *beta; float* slab; int act; };

template <class PF>
static int gather16_t(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws, int Hb,
                      int Wb, int Cbig, int Csmall, int ks, int stride, int act, bool plain5, vp_stream stream, float* stat = nullptr,
                      const BnBwdArgs* bb = nullptr, float alpha = 1.f) {
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
  if ((stat || bb) && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_gather_{stats,bnbwd}_bf16x3: this shape splits K");
  if (bb) { p.bx = bb->x; p.bmean = bb->mean; p.brstd = bb->rstd; p.bgamma = bb->gamma; p.bbeta = bb->beta; p.bsum = bb->slab; p.bact = bb->act; }
  p.k_per_split = p.nsplit == 2 ? ((p.K / 64 + 1) / 2) * 64 : p.K;
  if (p.nsplit == 2 && hipMemsetAsync(small_out, 0, (size_t)p.M * p.N * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(VP_ERR_LAUNCH, "vp_conv5_gather_bf16x3: memset failed");
  if constexpr (std::is_same<PF, ProbF16>::value) {
    const Launch16 l = plan16(p.M, p.N, p.nsplit, Cbig, p.nsplit, p.big_plane, p.w_plane, p.k_per_split);
    if (l.pcfg != PCFG_NONE) {
      PF16 q;
      static_cast<ProbF16&>(q) = p;
      q.xcd_map = xcd_map_tile(p.M, p.N, l.bm, l.bn);
      launch_igemm16p(q, l.pcfg, p.M, p.N, p.nsplit, (hipStream_t)stream, Cbig, true, false);
      return check_launch("vp_conv_gather_bf16x3(pipelined)");
    }
  }
  p.xcd_map = xcd_map_for(p.M, p.N, p.nsplit);
  launch_igemm16(p, p.M, p.N, p.nsplit, (hipStream_t)stream, Cbig);
  return check_launch("vp_conv_gather_bf16x3");
}

template <class PT>
static int scatter16_t(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb, int Csmall,
                       int Cbig, int ks, int stride, bool plain5, vp_stream stream, float* stat = nullptr, const BnBwdArgs* bb = nullptr,
                       float alpha = 1.f) {
  PT p;
  p.alpha = alpha;
  p.zero = vp_zero_page();
  p.g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  p.small = (const u16*)small_split; p.small_plane = (size_t)B * Hs * Ws * Csmall;
  p.w = (const u16*)w_p1_split; p.w_plane = (size_t)Csmall * Cbig * p.g.nt;
  p.out = big_out; p.M = B * Hs * Ws; p.N = Cbig;
  p.nsplit = scatter_nsplit(p.M, p.N, Csmall, stride, plain5);
  p.stat = stat;
  if ((stat || bb) && p.nsplit != 1) return fail(VP_ERR_ARG, "vp_conv5_scatter_{stats,bnbwd}_bf16x3: this shape splits K");
  if (bb) { p.bx = bb->x; p.bmean = bb->mean; p.brstd = bb->rstd; p.bgamma = bb->gamma; p.bbeta = bb->beta; p.bsum = bb->slab; p.bact = bb->act; }
  if (p.nsplit == 2 &&
      hipMemsetAsync(big_out, 0, (size_t)B * p.g.Hb * p.g.Wb * Cbig * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return fail(VP_ERR_LAUNCH, "vp_conv5_scatter_bf16x3: memset failed");
  if constexpr (std::is_same<PT, ProbT16>::value) {
    const Launch16 l = plan16(p.M, p.N, stride * stride * p.nsplit, Csmall, p.nsplit, p.small_plane, p.w_plane, (long)stride * stride * Csmall);
    if (l.pcfg != PCFG_NONE && stride == 2) {
      PT16 q;
      static_cast<ProbT16&>(q) = p;
      q.xcd_map = xcd_map_tile(p.M, p.N, l.bm, l.bn);
      launch_igemm16p(q, l.pcfg, p.M, p.N, stride * stride * p.nsplit, (hipStream_t)stream, Csmall, true, false);
      return check_launch("vp_conv_scatter_bf16x3(pipelined)");
    }
  }
  p.xcd_map = xcd_map_for(p.M, p.N, stride * stride * p.nsplit);
  launch_igemm16(p, p.M, p.N, stride * stride * p.nsplit, (hipStream_t)stream, Csmall);
  return check_launch("vp_conv_scatter_bf16x3");
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

extern "C" {

static bool split_fmt_ok(int fmt, float scale) { return (fmt == SPLIT_BF16 || fmt == SPLIT_F16) && scale > 0.f; }

int vp_split_fmt_f32(const float* x, void* out_split, size_t n, int fmt, float scale, vp_stream stream) {
  VP_REQUIRE(x && out_split && n > 0, "vp_split_fmt_f32: bad arguments");
  VP_REQUIRE(split_fmt_ok(fmt, scale), "vp_split_fmt_f32: format 0 (bf16 pair) or 1 (fp16 pair), scale > 0");
  hipLaunchKernelGGL(split_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, (u16_t*)out_split, n, fmt, scale);
  return check_launch("vp_split_fmt_f32");
}

int vp_split_f32(const float* x, void* out_split, size_t n, vp_stream stream) {
  return vp_split_fmt_f32(x, out_split, n, SPLIT_BF16, 1.f, stream);
}

int vp_split_pad_f32(const float* x, void* out_split, size_t npix, int C, int Cpad, vp_stream stream) {
  VP_REQUIRE(x && out_split && npix > 0 && C > 0 && Cpad >= C && Cpad % 8 == 0, "vp_split_pad_f32: Cpad must be a multiple of 8 and >= C");
  hipLaunchKernelGGL(split_pad_kernel, dim3(grid_for(npix * (size_t)Cpad / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (u16_t*)out_split, npix, C, Cpad);
  return check_launch("vp_split_pad_f32");
}

int vp_pack_w5_split(const float* w_ref, void* p0_split, void* p1_split, int Csmall, int Cbig, vp_stream stream) {
  VP_REQUIRE(w_ref && (p0_split || p1_split) && Csmall > 0 && Cbig > 0, "vp_pack_w5_split: bad arguments");
  VP_REQUIRE(Csmall <= 65535 && Cbig <= 65535, "vp_pack_w5_split: channel count too large");
  return pack_w5_launch<true>(w_ref, p0_split, p1_split, Csmall, Cbig, (hipStream_t)stream, "vp_pack_w5_split");
}

int vp_pack_w5_batch(const vp_pack_job* jobs, int njobs, vp_stream stream) {
  VP_REQUIRE(jobs && njobs > 0, "vp_pack_w5_batch: bad arguments");
  PackTable tab;
  tab.n = 0;
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    const vp_pack_job& q = jobs[i];
    VP_REQUIRE(q.w && (q.p0 || q.p1) && q.Csmall > 0 && q.Cbig > 0, "vp_pack_w5_batch: job %d: bad arguments", i);
    VP_REQUIRE(q.Csmall <= 65535 && q.Cbig <= 65535, "vp_pack_w5_batch: job %d: channel count too large", i);
    const int CsP = q.Csmall_pad > q.Csmall ? q.Csmall_pad : q.Csmall;
    VP_REQUIRE(CsP == q.Csmall || (q.split && CsP % 8 == 0), "vp_pack_w5_batch: job %d: padding needs split planes and a multiple of 8", i);
    for (int mode = 0; mode < 2; ++mode) {
      void* out = mode == 0 ? q.p0 : q.p1;
      if (!out) continue;
      VP_REQUIRE(tab.n < kMaxPackJobs, "vp_pack_w5_batch: more than %d layouts in one batch", kMaxPackJobs);
      PackJob& j = tab.j[tab.n++];
      j.w = q.w; j.out = out; j.Cs = q.Csmall; j.Cb = q.Cbig; j.CsP = mode == 1 ? CsP : q.Csmall; j.mode = mode;
      j.split = q.split == 2 ? 2 : (q.split ? 1 : 0); j.tile_begin = tiles;
      tiles += mode == 0 ? q.Csmall * ((q.Cbig + 63) / 64) : q.Cbig * ((CsP + 63) / 64);
    }
  }
  hipLaunchKernelGGL(pack_w5_batch_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, tab);
  return check_launch("vp_pack_w5_batch");
}

int vp_pack_w5_p1_split_padded(const float* w_ref, void* p1_split, int Csmall, int Cbig, int Csmall_pad, vp_stream stream) {
  VP_REQUIRE(w_ref && p1_split && Csmall > 0 && Cbig > 0 && Csmall_pad >= Csmall && Csmall_pad % 8 == 0,
             "vp_pack_w5_p1_split_padded: bad arguments");
  return pack_w5_launch<true>(w_ref, nullptr, p1_split, Csmall, Cbig, (hipStream_t)stream, "vp_pack_w5_p1_split_padded", Csmall_pad);
}

static int gather16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws, int Hb,
                    int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream, int f16 = 0, float alpha = 1.f) {
  VP_REQUIRE(big_split && w_p0_split && small_out, "vp_conv_gather_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig > 0 && Csmall > 0 && Cbig % 8 == 0, "vp_conv_gather_bf16x3: Cbig must be a multiple of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_gather_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_gather_bf16x3: kernel size must be 1, 3 or 5");
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_SIGMOID, "vp_conv_gather_bf16x3: epilogue supports none|sigmoid");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if (f16) {    // fp16-pair planes, f16 = products per fragment pair: always the implicit-GEMM kernels (the halo kernels read bf16 pairs)
    VP_REQUIRE((f16 == 2 || f16 == 3) && alpha > 0.f, "vp_conv_gather_f16: products must be 2 or 3, out_scale positive");
#define VP_G16(PF, P5) gather16_t<PF>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, P5, stream, nullptr, nullptr, alpha)
    if (f16 == 2) return plain5 ? VP_G16(ProbF16X, true) : VP_G16(ProbF16KX, false);
    return plain5 ? VP_G16(ProbF16H, true) : VP_G16(ProbF16KH, false);
#undef VP_G16
  }
  if (plain5 && halo_enabled())
    if (const int kind = halo_gather_kind(B, Hs, Ws, Cbig, Csmall, stride))
      return halo_gather_launch(kind, big_split, w_p0_split, bias, small_out, B, Hs, Ws, Cbig, Csmall, stride, act, (hipStream_t)stream);
  if (!plain5) return gather16_t<ProbF16K>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, false, stream);
  return gather16_t<ProbF16>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, true, stream);
}


int vp_conv5_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs,
                           int Ws, int Cbig, int Csmall, int stride, int act, vp_stream stream) {
  return gather16(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream);
}

int vp_conv_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                          int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream) {
  return gather16(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, stream);
}

static int scatter16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb, int Csmall,
                     int Cbig, int ks, int stride, vp_stream stream, int f16 = 0, float alpha = 1.f) {
  VP_REQUIRE(small_split && w_p1_split && big_out, "vp_conv_scatter_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig > 0 && Csmall > 0 && Csmall % 8 == 0, "vp_conv_scatter_bf16x3: Csmall must be a multiple of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_scatter_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_scatter_bf16x3: kernel size must be 1, 3 or 5");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if (f16) {
    VP_REQUIRE((f16 == 2 || f16 == 3) && alpha > 0.f, "vp_conv_scatter_f16: products must be 2 or 3, out_scale positive");
#define VP_S16(PT, P5) scatter16_t<PT>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, P5, stream, nullptr, nullptr, alpha)
    if (f16 == 2) return plain5 ? VP_S16(ProbT16X, true) : VP_S16(ProbT16KX, false);
    return plain5 ? VP_S16(ProbT16H, true) : VP_S16(ProbT16KH, false);
#undef VP_S16
  }
  if (plain5 && halo_enabled())
    if (const int kind = halo_scatter_kind(B, Hs, Ws, Csmall, Cbig, stride))
      return halo_scatter_launch(kind, small_split, w_p1_split, big_out, B, Hs, Ws, Csmall, Cbig, (hipStream_t)stream);
  if (!plain5) return scatter16_t<ProbT16K>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, false, stream);
  return scatter16_t<ProbT16>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, true, stream);
}


int vp_conv5_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                            int Cbig, int stride, vp_stream stream) {
  return scatter16(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream);
}

int vp_conv_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                           int Csmall, int Cbig, int ks, int stride, vp_stream stream) {
  return scatter16(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, stream);
}

int vp_pack_w_split(const float* w_ref, void* p0_split, void* p1_split, int Csmall, int Cbig, int ks, vp_stream stream) {
  VP_REQUIRE(w_ref && (p0_split || p1_split) && Csmall > 0 && Cbig > 0, "vp_pack_w_split: bad arguments");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_pack_w_split: kernel size must be 1, 3 or 5");
  VP_REQUIRE(Csmall <= 65535 && Cbig <= 65535, "vp_pack_w_split: channel count too large");
  return pack_w5_launch<true>(w_ref, p0_split, p1_split, Csmall, Cbig, (hipStream_t)stream, "vp_pack_w_split", 0, ks * ks);
}

size_t vp_conv5_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride);
  return wgrad_slab_floats(g, wgrad_nsplit(g)) * sizeof(float);
}

size_t vp_conv_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride) {
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  return wgrad_slab_floats(g, wgrad_nsplit(g)) * sizeof(float);
}

static int wgrad16(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                   int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream, int f16 = 0, float alpha = 1.f);

int vp_conv5_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                          int Csmall, int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream);
}

int vp_conv_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                         int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16(big_split, small_split, dw_ref, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, ws, ws_bytes, stream);
}

static int wgrad16(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                   int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream, int f16, float alpha) {
  VP_REQUIRE(big_split && small_split && dw_ref && ws, "vp_conv_wgrad_bf16x3: null pointer");
  VP_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Cbig % 8 == 0 && Csmall % 8 == 0 && Cbig > 0 && Csmall > 0,
             "vp_conv_wgrad_bf16x3: channel counts must be multiples of 8");
  VP_REQUIRE(stride == 1 || stride == 2, "vp_conv_wgrad_bf16x3: stride must be 1 or 2");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_conv_wgrad_bf16x3: kernel size must be 1, 3 or 5");
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  const int ns = wgrad_nsplit(g);
  if (ws_bytes < wgrad_slab_floats(g, ns) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv_wgrad_bf16x3: workspace too small");
  const bool plain5 = ks == 5 && Hb == Hs * stride && Wb == Ws * stride;
  if (f16) {
    VP_REQUIRE(f16 == 2 && alpha > 0.f, "vp_conv_wgrad_f16: products must be 2 (weight gradients have no 3-product fp16 form), out_scale positive");
    if (!plain5) return wgrad16_t<ProbW16KX>(big_split, small_split, dw_ref, g, ns, ws, stream, alpha);
    return wgrad16_t<ProbW16X>(big_split, small_split, dw_ref, g, ns, ws, stream, alpha);
  }
  if (!plain5) return wgrad16_t<ProbW16K>(big_split, small_split, dw_ref, g, ns, ws, stream);
  return wgrad16_t<ProbW16>(big_split, small_split, dw_ref, g, ns, ws, stream);
}

// ---- fp16-pair planes ("f16x2" plans): the same three families with two or three MFMAs per fragment pair (igemm16.h mfma_split) -------
// Operands are written by the *_fmt producers with format 1; `products` = 3: al*bh + ah*bl + ah*bh (forward layers: ~1e-6 relative),
// 2: (ah + al)*bh (backward layers: the weight operand of the gather / scatter families and the `big` operand of the weight gradient
// contribute their fp16 hi plane only, ~2e-4 relative per layer).  out_scale multiplies the accumulators (1 / the scale the producer
// of a gradient operand applied; 1 for activations and weights).
int vp_conv5_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs,
                        int Ws, int Cbig, int Csmall, int stride, int act, int products, float out_scale, vp_stream stream) {
  return gather16(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream, products, out_scale);
}
int vp_conv_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                       int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, int products, float out_scale, vp_stream stream) {
  return gather16(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, stream, products, out_scale);
}
int vp_conv5_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                         int Cbig, int stride, int products, float out_scale, vp_stream stream) {
  return scatter16(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream, products, out_scale);
}
int vp_conv_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                        int Csmall, int Cbig, int ks, int stride, int products, float out_scale, vp_stream stream) {
  return scatter16(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, stream, products, out_scale);
}
int vp_conv5_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                         int Csmall, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream, 2, out_scale);
}
int vp_conv_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                        int Csmall, int ks, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16(big_split, small_split, dw_ref, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, ws, ws_bytes, stream, 2, out_scale);
}
}

// ---- BatchNorm statistics from the convolution epilogue -----------------------------------------------------------------
// A 5x5 VAE layer on the split-bf16 kernels can emit {pivot, sum(x - pivot), sum((x - pivot)^2)} per (workgroup, output channel)
// from its accumulators (igemm16.h epilogue_stats32); one finaliser launch then produces mean / rstd / running statistics.
// This replaces bn_partial_kernel<0>, i.e. one full read of the activation per BatchNorm layer.
namespace {
struct StatPlan { int ok, bm, tiles_m, gz, N; long M, R; };

StatPlan stat_plan(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride, bool x2 = false) {
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
  const Launch16 l = family == 0 ? plan16(sp.M, sp.N, sp.gz, Cbig, 1, act_plane, (size_t)Csmall * Cbig * 25, 25L * Cbig)
                                 : plan16(sp.M, sp.N, sp.gz, Csmall, 1, act_plane, (size_t)Csmall * Cbig * 25, (long)stride * stride * Csmall);
  sp.bm = (!x2 && l.pcfg != PCFG_NONE && !(family == 1 && stride != 2)) ? l.bm : choose_tile16(sp.M, sp.N, sp.gz).bm;
  sp.tiles_m = (int)((sp.M + sp.bm - 1) / sp.bm);
  sp.ok = 1;
  return sp;
}
}  // namespace

namespace vp {
// one 256-thread workgroup per channel; groups are shifted to the pivot of group 0 and summed in fp64:
//   sum(x - P) = s_g + n_g d,  sum((x - P)^2) = q_g + 2 d s_g + n_g d^2,  d = p_g - P
__global__ void __launch_bounds__(256) bn_stats_slab_final_kernel(const float* __restrict__ slab, int G, int tiles_m, int BM, long M, long R,
                                                                  int C, float eps, float momentum, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, float* __restrict__ rm, float* __restrict__ rv) {
  __shared__ double shs[4], shq[4];
  const int c = blockIdx.x;
  const float* pv = slab + ((size_t)0 * C + c) * G;
  const float* sv = slab + ((size_t)1 * C + c) * G;
  const float* qv = slab + ((size_t)2 * C + c) * G;
  const double P = (double)pv[0];
  double S = 0.0, Q = 0.0;
  for (int g = threadIdx.x; g < G; g += 256) {
    const int tile = g % tiles_m;
    const long left = M - (long)tile * BM;
    const double n = (double)(left < BM ? left : BM);
    const double d = (double)pv[g] - P, s = (double)sv[g], q = (double)qv[g];
    S += s + n * d;
    Q += q + 2.0 * d * s + n * d * d;
  }
  S = wave_sum_d(S);
  Q = wave_sum_d(Q);
  if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = S; shq[threadIdx.x >> 6] = Q; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  S = (shs[0] + shs[1]) + (shs[2] + shs[3]);
  Q = (shq[0] + shq[1]) + (shq[2] + shq[3]);
  const double ms = S / (double)R;
  double var = Q / (double)R - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = P + ms;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
  if (rv) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}
}  // namespace vp

extern "C" {

size_t vp_conv5_stats_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  const StatPlan sp = stat_plan(family, B, Hs, Ws, Cbig, Csmall, stride);
  return sp.ok ? (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float) : 0;
}

static int stats_finish(const StatPlan& sp, const float* slab, float eps, float momentum, float* mean, float* rstd, float* rm, float* rv,
                        vp_stream stream) {
  hipLaunchKernelGGL(bn_stats_slab_final_kernel, dim3(sp.N), dim3(256), 0, (hipStream_t)stream, slab, sp.tiles_m * sp.gz, sp.tiles_m, sp.bm,
                     sp.M, sp.R, sp.N, eps, momentum, mean, rstd, rm, rv);
  return check_launch("vp_conv5_*_stats_bf16x3(final)");
}

int vp_conv5_gather_stats_bf16x3(const void* big_split, const void* w_p0_split, float* small_out, int B, int Hs, int Ws, int Cbig,
                                 int Csmall, int stride, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                 float* running_var, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(big_split && w_p0_split && small_out && mean && rstd && ws, "vp_conv5_gather_stats_bf16x3: null pointer");
  const StatPlan sp = stat_plan(0, B, Hs, Ws, Cbig, Csmall, stride);
  VP_REQUIRE(sp.ok, "vp_conv5_gather_stats_bf16x3: this shape cannot emit statistics (vp_conv5_stats_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_gather_stats_bf16x3: workspace too small");
  int rc = gather16_t<ProbF16>(big_split, w_p0_split, nullptr, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride,
                               VP_ACT_NONE, true, stream, (float*)ws);
  if (rc) return rc;
  return stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
}

int vp_conv5_scatter_stats_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                                  int Cbig, int stride, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                  float* running_var, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(small_split && w_p1_split && big_out && mean && rstd && ws, "vp_conv5_scatter_stats_bf16x3: null pointer");
  const StatPlan sp = stat_plan(1, B, Hs, Ws, Cbig, Csmall, stride);
  VP_REQUIRE(sp.ok, "vp_conv5_scatter_stats_bf16x3: this shape cannot emit statistics (vp_conv5_stats_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)3 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_scatter_stats_bf16x3: workspace too small");
  int rc = scatter16_t<ProbT16>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, true, stream,
                                (float*)ws);
  if (rc) return rc;
  return stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
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
  return stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
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
  return stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
}

}

extern "C" {

int vp_im2col5s2_cols(int C) { return im2col5_kc(C); }

int vp_im2col5s2_split_fmt_f32(const float* x, void* out_split, int B, int C, int Hb, int Wb, int nchw, int fmt, vp_stream stream) {
  VP_REQUIRE(x && out_split && B > 0 && Hb > 0 && Wb > 0 && Hb % 2 == 0 && Wb % 2 == 0, "vp_im2col5s2_split_f32: bad arguments");
  VP_REQUIRE(C == 1 || C == 3, "vp_im2col5s2_split_f32: 1 or 3 image channels");
  VP_REQUIRE(split_fmt_ok(fmt, 1.f), "vp_im2col5s2_split_fmt_f32: format 0 (bf16 pair) or 1 (fp16 pair)");
  const dim3 grid((unsigned)(B * (Hb / 2)));
  if (C == 3) hipLaunchKernelGGL((im2col5s2_split_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, x, (u16_t*)out_split, B, Hb, Wb, Hb / 2, Wb / 2, nchw, fmt);
  else hipLaunchKernelGGL((im2col5s2_split_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, (u16_t*)out_split, B, Hb, Wb, Hb / 2, Wb / 2, nchw, fmt);
  return check_launch("vp_im2col5s2_split_f32");
}

int vp_im2col5s2_split_f32(const float* x, void* out_split, int B, int C, int Hb, int Wb, int nchw, vp_stream stream) {
  return vp_im2col5s2_split_fmt_f32(x, out_split, B, C, Hb, Wb, nchw, SPLIT_BF16, stream);
}

int vp_pack_w_im2col5_split_fmt(const float* w_ref, void* out_split, int Cout, int C, int fmt, vp_stream stream) {
  VP_REQUIRE(w_ref && out_split && Cout > 0 && (C == 1 || C == 3), "vp_pack_w_im2col5_split: bad arguments");
  VP_REQUIRE(split_fmt_ok(fmt, 1.f), "vp_pack_w_im2col5_split_fmt: format 0 (bf16 pair) or 1 (fp16 pair)");
  const int KC = im2col5_kc(C), GW = im2col5_gw(C);
  hipLaunchKernelGGL(pack_w_im2col5_split_kernel, dim3(grid_for((size_t)Cout * KC, 256)), dim3(256), 0, (hipStream_t)stream, w_ref,
                     (u16_t*)out_split, Cout, C, GW, KC, fmt);
  return check_launch("vp_pack_w_im2col5_split");
}

int vp_pack_w_im2col5_split(const float* w_ref, void* out_split, int Cout, int C, vp_stream stream) {
  return vp_pack_w_im2col5_split_fmt(w_ref, out_split, Cout, C, SPLIT_BF16, stream);
}

int vp_unpack_dw_im2col5_f32(const float* dw_cols, float* dw_ref, int Cout, int C, vp_stream stream) {
  VP_REQUIRE(dw_cols && dw_ref && Cout > 0 && (C == 1 || C == 3), "vp_unpack_dw_im2col5_f32: bad arguments");
  const int KC = im2col5_kc(C), GW = im2col5_gw(C);
  hipLaunchKernelGGL(unpack_dw_im2col5_kernel, dim3(grid_for((size_t)Cout * C * 25, 256)), dim3(256), 0, (hipStream_t)stream, dw_cols, dw_ref,
                     Cout, C, GW, KC);
  return check_launch("vp_unpack_dw_im2col5_f32");
}


int vp_conv5_smallout_bf16x3(const float* big, const float* w_p0, const float* bias, float* small_out, int B, int H, int W, int Cbig,
                             int Csmall, int act, vp_stream stream) {
  VP_REQUIRE(big && w_p0 && small_out && B > 0 && H > 0 && W > 0, "vp_conv5_smallout_bf16x3: bad arguments");
  VP_REQUIRE(((uintptr_t)big & 15) == 0 && ((uintptr_t)w_p0 & 15) == 0, "vp_conv5_smallout_bf16x3: operands must be 16-byte aligned");
  const ConvGeom g = make_geom(B, H, W, Csmall, Cbig, 1);
  VP_REQUIRE(tapn_gather_applicable(g, act), "vp_conv5_smallout_bf16x3: needs 64 input channels, 1 or 3 outputs, act none|sigmoid");
  return tapn_gather_launch(big, w_p0, bias, small_out, g, act, (hipStream_t)stream);
}

}

// ---- BatchNorm-backward partial sums from the input-gradient convolution's epilogue --------------------------------------------
namespace vp {
// one 256-thread workgroup per channel: sum_g, sum_gx (fp64 over the groups) -> also the affine gradients dbeta, dgamma
__global__ void __launch_bounds__(256) bn_bwd_slab_final_kernel(const float* __restrict__ slab, int G, int C, float* __restrict__ sum_g,
                                                                float* __restrict__ sum_gx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  __shared__ double shs[4], shq[4];
  const int c = blockIdx.x;
  const float* sv = slab + ((size_t)0 * C + c) * G;
  const float* qv = slab + ((size_t)1 * C + c) * G;
  double S = 0.0, Q = 0.0;
  for (int g = threadIdx.x; g < G; g += 256) { S += (double)sv[g]; Q += (double)qv[g]; }
  S = wave_sum_d(S);
  Q = wave_sum_d(Q);
  if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = S; shq[threadIdx.x >> 6] = Q; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  S = (shs[0] + shs[1]) + (shs[2] + shs[3]);
  Q = (shq[0] + shq[1]) + (shq[2] + shq[3]);
  sum_g[c] = (float)S;
  sum_gx[c] = (float)Q;
  if (dbeta) dbeta[c] = (float)S;
  if (dgamma) dgamma[c] = (float)Q;
}
}  // namespace vp

extern "C" {

static int bnbwd_finish(const StatPlan& sp, const float* slab, float* sums, float* dgamma, float* dbeta, vp_stream stream) {
  hipLaunchKernelGGL(bn_bwd_slab_final_kernel, dim3(sp.N), dim3(256), 0, (hipStream_t)stream, slab, sp.tiles_m * sp.gz, sp.N, sums, sums + sp.N,
                     dgamma, dbeta);
  return check_launch("vp_conv5_*_bnbwd_bf16x3(final)");
}

int vp_conv5_gather_bnbwd_bf16x3(const void* big_split, const void* w_p0_split, float* small_out, int B, int Hs, int Ws, int Cbig, int Csmall,
                                 int stride, const float* bn_x, const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                                 const float* bn_beta, int act, float* sums, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                 vp_stream stream) {
  VP_REQUIRE(big_split && w_p0_split && small_out && bn_x && bn_mean && bn_rstd && sums && ws, "vp_conv5_gather_bnbwd_bf16x3: null pointer");
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_RELU, "vp_conv5_gather_bnbwd_bf16x3: activation none|relu");
  const StatPlan sp = stat_plan(0, B, Hs, Ws, Cbig, Csmall, stride);
  VP_REQUIRE(sp.ok, "vp_conv5_gather_bnbwd_bf16x3: this shape cannot emit epilogue sums (vp_conv5_stats_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)2 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_gather_bnbwd_bf16x3: workspace too small");
  const BnBwdArgs bb = {bn_x, bn_mean, bn_rstd, bn_gamma, bn_beta, (float*)ws, act};
  int rc = gather16_t<ProbF16>(big_split, w_p0_split, nullptr, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride,
                               VP_ACT_NONE, true, stream, nullptr, &bb);
  if (rc) return rc;
  return bnbwd_finish(sp, (const float*)ws, sums, dgamma, dbeta, stream);
}

int vp_conv5_scatter_bnbwd_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall, int Cbig,
                                  int stride, const float* bn_x, const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                                  const float* bn_beta, int act, float* sums, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                  vp_stream stream) {
  VP_REQUIRE(small_split && w_p1_split && big_out && bn_x && bn_mean && bn_rstd && sums && ws, "vp_conv5_scatter_bnbwd_bf16x3: null pointer");
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_RELU, "vp_conv5_scatter_bnbwd_bf16x3: activation none|relu");
  const StatPlan sp = stat_plan(1, B, Hs, Ws, Cbig, Csmall, stride);
  VP_REQUIRE(sp.ok, "vp_conv5_scatter_bnbwd_bf16x3: this shape cannot emit epilogue sums (vp_conv5_stats_workspace_bytes() == 0)");
  if (ws_bytes < (size_t)2 * sp.N * sp.tiles_m * sp.gz * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_scatter_bnbwd_bf16x3: workspace too small");
  const BnBwdArgs bb = {bn_x, bn_mean, bn_rstd, bn_gamma, bn_beta, (float*)ws, act};
  int rc = scatter16_t<ProbT16>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, true, stream,
                                nullptr, &bb);
  if (rc) return rc;
  return bnbwd_finish(sp, (const float*)ws, sums, dgamma, dbeta, stream);
}

}
