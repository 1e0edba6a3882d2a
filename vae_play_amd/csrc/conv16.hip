// Split-bf16 ("bf16x3") convolution families: C ABI + the producers of split planes.
#define VP_PCFG_LIBRARY 1
#define VP_IGEMM16_M16_ON 1
#include "conv16_impl.h"

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

// the same im2col as fp32 [pixel][KC] (exact-f32 plan: both 1x1 layers then run on the fp32-MFMA kernels)
template <int C>
__global__ void __launch_bounds__(256) im2col5s2_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int Hb, int Wb,
                                                            int Hs, int Ws, int nchw) {
  constexpr int GW = C == 3 ? 16 : 8, KC = ((5 * GW + 31) / 32) * 32, G = KC / 4;
  const int row = blockIdx.x, b = row / Hs, hs = row - b * Hs;
  for (int i = threadIdx.x; i < Ws * G; i += blockDim.x) {
    const int ws = i / G, c0 = (i - ws * G) * 4;
    vp_f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + j, r = col / GW, rem = col - r * GW;
      const int q = rem / C, cin = rem - q * C;
      const int h = 2 * hs - 2 + r, w_ = 2 * ws - 2 + q;
      const bool ok = r < 5 && q < 5 && h >= 0 && h < Hb && w_ >= 0 && w_ < Wb;
      v[j] = ok ? (nchw ? x[(((size_t)b * C + cin) * Hb + h) * Wb + w_] : x[(((size_t)b * Hb + h) * Wb + w_) * C + cin]) : 0.f;
    }
    *reinterpret_cast<vp_f32x4*>(out + ((size_t)row * Ws + ws) * KC + c0) = v;
  }
}

__global__ void pack_w_im2col5_f32_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int C, int GW, int KC) {
  const size_t n = (size_t)Cout * KC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / KC), col = (int)(i - (size_t)co * KC);
    const int r = col / GW, rem = col - r * GW, q = rem / C, cin = rem - q * C;
    out[i] = (r < 5 && q < 5) ? w[(((size_t)co * C + cin) * 5 + r) * 5 + q] : 0.f;
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



int vp_conv5_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs,
                           int Ws, int Cbig, int Csmall, int stride, int act, vp_stream stream) {
  return gather16<0>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, act, stream);
}

int vp_conv_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                          int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream) {
  return gather16<0>(big_split, w_p0_split, bias, small_out, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, act, stream);
}



int vp_conv5_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                            int Cbig, int stride, vp_stream stream) {
  return scatter16<0>(small_split, w_p1_split, big_out, B, Hs, Ws, Hs * stride, Ws * stride, Csmall, Cbig, 5, stride, stream);
}

int vp_conv_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                           int Csmall, int Cbig, int ks, int stride, vp_stream stream) {
  return scatter16<0>(small_split, w_p1_split, big_out, B, Hs, Ws, Hb, Wb, Csmall, Cbig, ks, stride, stream);
}

int vp_pack_w_split(const float* w_ref, void* p0_split, void* p1_split, int Csmall, int Cbig, int ks, vp_stream stream) {
  VP_REQUIRE(w_ref && (p0_split || p1_split) && Csmall > 0 && Cbig > 0, "vp_pack_w_split: bad arguments");
  VP_REQUIRE(ks == 1 || ks == 3 || ks == 5, "vp_pack_w_split: kernel size must be 1, 3 or 5");
  VP_REQUIRE(Csmall <= 65535 && Cbig <= 65535, "vp_pack_w_split: channel count too large");
  return pack_w5_launch<true>(w_ref, p0_split, p1_split, Csmall, Cbig, (hipStream_t)stream, "vp_pack_w_split", 0, ks * ks);
}

size_t vp_conv5_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  return vp_conv_wgrad_bf16x3_workspace_bytes(B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride);
}

size_t vp_conv_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride) {
  ConvGeom g = make_geom(B, Hs, Ws, Csmall, Cbig, stride, ks, Hb, Wb);
  int ns = wgrad_nsplit(g);
  if (wgrad_pair_applicable(g)) ns = wgrad_pair_nsplit(g, ns);   // tap pairs split twice as deep
  size_t n = wgrad_slab_floats(g, ns);
  if (const int bn = wgrad5_kind(g)) {                           // rows of taps: one slab set per CU-filling round
    int kper = 0;
    const size_t n5 = wgrad5_slab_floats(g, bn, wgrad5_nsplit(g, bn, &kper));
    if (n5 > n) n = n5;
  }
  return n * sizeof(float);
}

int vp_conv5_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                          int Csmall, int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<0>(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream);
}

int vp_conv5_wgrad_bf16x3_cus(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                              int Csmall, int stride, int max_cus, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<0>(big_split, small_split, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cbig, Csmall, 5, stride, ws, ws_bytes, stream, 1.f, max_cus);
}

int vp_conv_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                         int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream) {
  return wgrad16<0>(big_split, small_split, dw_ref, B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride, ws, ws_bytes, stream);
}



}


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

}
namespace vp {
// the same finaliser for callers that do not see StatPlan (conv32.hip: the exact-f32 mode's statistics epilogue)
int stats_slab_finish(const float* slab, int groups, int tiles_m, int bm, long M, long R, int N, float eps, float momentum, float* mean,
                      float* rstd, float* rm, float* rv, hipStream_t stream) {
  hipLaunchKernelGGL(bn_stats_slab_final_kernel, dim3(N), dim3(256), 0, stream, slab, groups, tiles_m, bm, M, R, N, eps, momentum, mean, rstd, rm, rv);
  return check_launch("vp_conv5_*_stats_f32(final)");
}
}  // namespace vp
int vp16_stats_finish(const StatPlan& sp, const float* slab, float eps, float momentum, float* mean, float* rstd, float* rm, float* rv,
                        vp_stream stream) {
  hipLaunchKernelGGL(bn_stats_slab_final_kernel, dim3(sp.N), dim3(256), 0, (hipStream_t)stream, slab, sp.tiles_m * sp.gz, sp.tiles_m, sp.bm,
                     sp.M, sp.R, sp.N, eps, momentum, mean, rstd, rm, rv);
  return check_launch("vp_conv5_*_stats_bf16x3(final)");
}
extern "C" {

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
  return vp16_stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
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
  return vp16_stats_finish(sp, (const float*)ws, eps, momentum, mean, rstd, running_mean, running_var, stream);
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

int vp_im2col5s2_f32(const float* x, float* out, int B, int C, int Hb, int Wb, int nchw, vp_stream stream) {
  VP_REQUIRE(x && out && B > 0 && Hb > 0 && Wb > 0 && Hb % 2 == 0 && Wb % 2 == 0, "vp_im2col5s2_f32: bad arguments");
  VP_REQUIRE(C == 1 || C == 3, "vp_im2col5s2_f32: 1 or 3 image channels");
  VP_REQUIRE(((uintptr_t)out & 15) == 0, "vp_im2col5s2_f32: the output must be 16-byte aligned");
  const dim3 grid((unsigned)(B * (Hb / 2)));
  if (C == 3) hipLaunchKernelGGL((im2col5s2_f32_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, x, out, B, Hb, Wb, Hb / 2, Wb / 2, nchw);
  else hipLaunchKernelGGL((im2col5s2_f32_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, out, B, Hb, Wb, Hb / 2, Wb / 2, nchw);
  return check_launch("vp_im2col5s2_f32");
}

int vp_pack_w_im2col5_f32(const float* w_ref, float* out, int Cout, int C, vp_stream stream) {
  VP_REQUIRE(w_ref && out && Cout > 0 && (C == 1 || C == 3), "vp_pack_w_im2col5_f32: bad arguments");
  const int KC = im2col5_kc(C), GW = im2col5_gw(C);
  hipLaunchKernelGGL(pack_w_im2col5_f32_kernel, dim3(grid_for((size_t)Cout * KC, 256)), dim3(256), 0, (hipStream_t)stream, w_ref, out, Cout, C,
                     GW, KC);
  return check_launch("vp_pack_w_im2col5_f32");
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
  VP_REQUIRE(tapn_gather_applicable(g, act), "vp_conv5_smallout_bf16x3: needs 64 or 32 input channels, 1 or 3 outputs, act none|sigmoid");
  return tapn_gather_launch(big, w_p0, bias, small_out, g, act, (hipStream_t)stream);
}

}

