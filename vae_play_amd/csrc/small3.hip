// 3x3 / stride 1 / padding 1 convolutions with a HANDFUL of channels on both sides: the mask / edge heads of models/networks_BE.py:39-66
// (34 -> 8, 8 -> 8, 10 -> 4, 4 -> 4, 4 -> 8, 8 -> 4, 4 -> 1 channels at 128 x 128 and 256 x 256).  On the implicit-GEMM kernels these
// shapes fill 1.5 - 8 % of a 64 x 64 MFMA tile (channel counts padded to 8): their weight gradients were 35 % of the heads' training
// step (profiles/r02_notes.md).  Here they are what they are -- reductions over a million pixels into a few hundred numbers --
// on the vector ALUs in exact fp32:
//
//   dW[co][ci][tap] = sum_p dy[p][co] * x[p + tap][ci]
//
//   * a workgroup stages a TH x 32 pixel tile of dy (TH = 8 | 4 | 2 by LDS budget) and its halo tile of x in LDS (channels padded to a multiple of 4 there,
//     never in HBM) and walks tiles grid-stride, accumulating in registers;
//   * the 9 * ceil(Co/4) * ceil(Ci/4) output blocks of 4 x 4 are dealt to thread groups; the lanes of a group split the tile's pixels:
//     two 16-B LDS reads feed 16 FMAs;
//   * lanes are combined through LDS in a fixed order, every workgroup writes one slab [tap][co][ci], slab_reduce_deep_kernel sums the
//     slabs in a fixed order into the reference layout: bit-reproducible, no atomics.
#include <stdint.h>
#include "common.h"
#include "problems.h"
#include "narrow.h"

namespace vp {

// tile = TH x 32 output pixels, TH = 8 | 4 | 2: the largest whose halo tile (+ weights / dy tile) fits 64 KB of LDS
constexpr int S3_TW = 32;
// Measured limits: 64 -> 1 channels at 256 x 256 x 64 images (the font nets' last predictor) is SLOWER here than on the padded MFMA
// tiles (font-256 iteration 486 -> 500 ms), so the kernels keep to the networks_BE heads' range
constexpr int S3_MAX_CI = 36, S3_MAX_CO = 8, S3_MAX_CICO = 288;
constexpr size_t S3_LDS_BYTES = 64 * 1024;

// halo tile (TH + 2) x (TW + 2) pixels of image b with origin (h0 - 1, w0 - 1) -> LDS [pixel][CP] (zero outside the image; the
// padding channels [C, CP) are zeroed once by the caller); 16-B loads when C % 4 == 0
__device__ __forceinline__ void stage_halo(float* __restrict__ xs, const float* __restrict__ x, int b, int h0, int w0, int H, int W, int C,
                                           int CP, int tid, int TH) {
  if ((C & 3) == 0) {
    const int q = C >> 2, rowq = (S3_TW + 2) * q;
    for (int idx = tid; idx < (TH + 2) * rowq; idx += 256) {
      const int r = idx / rowq, j = idx - r * rowq, c = j / q, ch = (j - c * q) * 4;
      const int h = h0 - 1 + r, w = w0 - 1 + c;
      vp_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const vp_f32x4*>(x + ((size_t)(b * H + h) * W + w) * C + ch);
      *reinterpret_cast<vp_f32x4*>(xs + (r * (S3_TW + 2) + c) * CP + ch) = v;
    }
  } else {
    const int rowlen = (S3_TW + 2) * C;
    for (int r = 0; r < TH + 2; ++r) {
      const int h = h0 - 1 + r;
      const bool hok = h >= 0 && h < H;
      const float* src = x + ((size_t)(b * H + (hok ? h : 0)) * W) * C;
      for (int j = tid; j < rowlen; j += 256) {
        const int c = j / C, ch = j - c * C, w = w0 - 1 + c;
        xs[(r * (S3_TW + 2) + c) * CP + ch] = (hok && w >= 0 && w < W) ? src[(size_t)w * C + ch] : 0.f;
      }
    }
  }
}

__global__ void __launch_bounds__(256) conv3_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ slab, int B, int H, int W, int Ci, int Co,
                                                                int tiles_h, int tiles_w, int ntiles, int TH) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int IQ = (Ci + 3) >> 2, CQ = (Co + 3) >> 2, CiP = IQ * 4, CoP = CQ * 4;
  float* xs = lds;                                                   // [(TH + 2) * (TW + 2)][CiP]
  const int P = TH * S3_TW;
  float* dys = xs + (TH + 2) * (S3_TW + 2) * CiP;                    // [P][CoP]
  const int lds_floats = (TH + 2) * (S3_TW + 2) * CiP + P * CoP;
  const int nblk = 9 * CQ * IQ;
  const int lpb = nblk >= 256 ? 1 : 256 / nblk;                      // lanes per output block
  const int tid = threadIdx.x;
  const int blk = tid / lpb, lane = tid - blk * lpb;
  const bool active = blk < nblk;
  int iq = 0, cq = 0, tap = 0;
  if (active) { iq = blk % IQ; cq = (blk / IQ) % CQ; tap = blk / (IQ * CQ); }
  const int dr = tap / 3, dc = tap - dr * 3;                         // x pixel of output pixel (r, c): halo coordinates (r + dr, c + dc)
  for (int i = tid; i < lds_floats; i += 256) lds[i] = 0.f;          // padding channels stay zero for the whole kernel
  __syncthreads();
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / (tiles_h * tiles_w), rem = t - b * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * TH, w0 = tw * S3_TW;
    stage_halo(xs, x, b, h0, w0, H, W, Ci, CiP, tid, TH);
    if ((Co & 3) == 0) {                                             // 16-B loads when the pixel rows are 16-B aligned
      const int q = Co >> 2;
      for (int idx = tid; idx < P * q; idx += 256) {
        const int pix = idx / q, ch = (idx - pix * q) * 4;
        const int r = pix / S3_TW, c = pix - r * S3_TW, h = h0 + r, w = w0 + c;
        vp_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h < H && w < W) v = *reinterpret_cast<const vp_f32x4*>(dy + ((size_t)(b * H + h) * W + w) * Co + ch);
        *reinterpret_cast<vp_f32x4*>(dys + pix * CoP + ch) = v;
      }
    } else {
      for (int idx = tid; idx < P * Co; idx += 256) {
        const int pix = idx / Co, ch = idx - pix * Co;
        const int r = pix / S3_TW, c = pix - r * S3_TW, h = h0 + r, w = w0 + c;
        dys[pix * CoP + ch] = (h < H && w < W) ? dy[((size_t)(b * H + h) * W + w) * Co + ch] : 0.f;
      }
    }
    __syncthreads();
    if (active) {
      const float* xb = xs + (dr * (S3_TW + 2) + dc) * CiP + 4 * iq;
      const float* db = dys + 4 * cq;
      for (int p = lane; p < P; p += lpb) {
        const int r = p / S3_TW, c = p - r * S3_TW;
        const vp_f32x4 d = *reinterpret_cast<const vp_f32x4*>(db + p * CoP);
        const vp_f32x4 xv = *reinterpret_cast<const vp_f32x4*>(xb + (r * (S3_TW + 2) + c) * CiP);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(d[i], xv[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
  // combine the lanes of every block in a fixed order (LDS is free again: 256 x 16 floats)
  float* red = lds;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[tid * 16 + i * 4 + j] = acc[i][j];
  __syncthreads();
  if (active && lane == 0) {
    float* out = slab + (size_t)blockIdx.x * 9 * Co * Ci;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
        for (int l = 0; l < lpb; ++l) s += red[(blk * lpb + l) * 16 + i * 4 + j];
        const int co = 4 * cq + i, ci = 4 * iq + j;
        if (co < Co && ci < Ci) out[((size_t)tap * Co + co) * Ci + ci] = s;
      }
  }
}

// Forward and input gradient of the same layers: one output pixel per thread, all N outputs of the pixel in registers.
//   forward (DGRAD = false): out[p][n = co] = bias[co] + sum_{tap, k = ci} x[p + tap - 1][ci] * w[co][ci][tap]
//   input gradient (true):   out[p][n = ci] =            sum_{tap, k = co} dy[p + tap - 1][co] * w[co][ci][8 - tap]   (flipped taps)
// The input tile with its halo sits in LDS (K padded to a multiple of 4 there), the weights in LDS as [tap][k][NP]: every lane of a
// wave reads the same weight address (broadcast), so the LDS traffic that matters is one 16-B read of x per four k.
template <int NP, bool DGRAD>
__global__ void __launch_bounds__(256) conv3_small_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W,
                                                          int K, int N, int tiles_h, int tiles_w, int ntiles, int TH) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int KP = (K + 3) & ~3;
  float* xs = lds;                                              // [(TH + 2) * (TW + 2)][KP]
  float* wl = xs + (TH + 2) * (S3_TW + 2) * KP;                 // [9][K][NP]
  const int tid = threadIdx.x;
  for (int i = tid; i < (TH + 2) * (S3_TW + 2) * KP; i += 256) xs[i] = 0.f;
  for (int i = tid; i < 9 * K * NP; i += 256) {
    const int n = i % NP, k = (i / NP) % K, tap = i / (NP * K);
    float v = 0.f;
    if (n < N) v = DGRAD ? w[((size_t)k * N + n) * 9 + (8 - tap)] : w[((size_t)n * K + k) * 9 + tap];
    wl[i] = v;
  }
  float b0[NP];
#pragma unroll
  for (int n = 0; n < NP; ++n) b0[n] = (!DGRAD && bias && n < N) ? bias[n] : 0.f;
  __syncthreads();
  const int r = tid / S3_TW, c = tid - r * S3_TW;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / (tiles_h * tiles_w), rem = t - b * (tiles_h * tiles_w);
    const int th = rem / tiles_w, tw = rem - th * tiles_w;
    const int h0 = th * TH, w0 = tw * S3_TW;
    stage_halo(xs, in, b, h0, w0, H, W, K, KP, tid, TH);
    __syncthreads();
    float acc[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[n] = b0[n];
    if (r < TH)                     // (TH < 8: the upper thread rows only help staging)
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap - dr * 3;
      const float* xp = xs + ((r + dr) * (S3_TW + 2) + c + dc) * KP;
      const float* wp = wl + tap * K * NP;
      for (int k4 = 0; k4 < KP; k4 += 4) {
        const vp_f32x4 xv = *reinterpret_cast<const vp_f32x4*>(xp + k4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k4 + u < K) {
            const float* wk = wp + (k4 + u) * NP;
#pragma unroll
            for (int n = 0; n < NP; n += 4) {
              const vp_f32x4 wv = *reinterpret_cast<const vp_f32x4*>(wk + n);
              acc[n] = __builtin_fmaf(xv[u], wv[0], acc[n]);
              acc[n + 1] = __builtin_fmaf(xv[u], wv[1], acc[n + 1]);
              acc[n + 2] = __builtin_fmaf(xv[u], wv[2], acc[n + 2]);
              acc[n + 3] = __builtin_fmaf(xv[u], wv[3], acc[n + 3]);
            }
          }
        }
      }
    }
    const int h = h0 + r, ww = w0 + c;
    if (r < TH && h < H && ww < W) {
      float* dst = out + ((size_t)(b * H + h) * W + ww) * N;
      if ((N & 3) == 0) {
#pragma unroll
        for (int n = 0; n < NP; n += 4)
          if (n < N) *reinterpret_cast<vp_f32x4*>(dst + n) = vp_f32x4{acc[n], acc[n + 1], acc[n + 2], acc[n + 3]};
      } else {
#pragma unroll
        for (int n = 0; n < NP; ++n)
          if (n < N) dst[n] = acc[n];
      }
    }
    __syncthreads();
  }
}

static int np_for(int N) { return N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 12 ? 12 : (N <= 16 ? 16 : (N <= 40 ? 40 : 64)))); }
// tile heights (0 = does not fit): forward / input gradient keep the halo tile and [9][K][NP] weights, the weight gradient the halo
// tile and the dy tile (and needs 256 x 16 floats for its lane reduction)
static int conv_th(int K, int N) {
  const int KP = (K + 3) / 4 * 4;
  for (int th = 8; th >= 2; th >>= 1)
    if (((size_t)(th + 2) * (S3_TW + 2) * KP + (size_t)9 * K * np_for(N)) * sizeof(float) <= S3_LDS_BYTES) return th;
  return 0;
}
static int wgrad_th(int Ci, int Co) {
  const int CiP = (Ci + 3) / 4 * 4, CoP = (Co + 3) / 4 * 4;
  for (int th = 8; th >= 2; th >>= 1)
    if (((size_t)(th + 2) * (S3_TW + 2) * CiP + (size_t)th * S3_TW * CoP) * sizeof(float) <= S3_LDS_BYTES) return th;
  return 0;
}

template <bool DGRAD>
static int conv3_small_launch(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int K, int N,
                              hipStream_t s, const char* what) {
  const int TH = conv_th(K, N);
  if (!TH) return fail(VP_ERR_ARG, "%s: tile does not fit the LDS budget", what);
  const int th = (H + TH - 1) / TH, tw = (W + S3_TW - 1) / S3_TW;
  const int ntiles = B * th * tw;
  const int grid = ntiles < 2048 ? ntiles : 2048;
  const int KP = (K + 3) / 4 * 4;
  const int NP = np_for(N);
  const size_t lds = ((size_t)(TH + 2) * (S3_TW + 2) * KP + (size_t)9 * K * NP) * sizeof(float);
#define VP_S3(NPV) hipLaunchKernelGGL((conv3_small_kernel<NPV, DGRAD>), dim3(grid), dim3(256), lds, s, in, w, bias, out, B, H, W, K, N, th, tw, ntiles, TH)
  switch (NP) {
    case 4: VP_S3(4); break;
    case 8: VP_S3(8); break;
    case 12: VP_S3(12); break;
    case 16: VP_S3(16); break;
    case 40: VP_S3(40); break;
    default: VP_S3(64); break;
  }
#undef VP_S3
  return check_launch(what);
}

static bool small3_ok(int B, int H, int W, int Ci, int Co) {
  return B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && Ci <= S3_MAX_CI && Co <= S3_MAX_CO && Ci * Co <= S3_MAX_CICO &&
         conv_th(Ci, Co) && conv_th(Co, Ci) && wgrad_th(Ci, Co);
}
static int small3_tiles(int B, int H, int W, int TH, int& th, int& tw) {
  th = (H + TH - 1) / TH; tw = (W + S3_TW - 1) / S3_TW;
  return B * th * tw;
}
static int small3_grid(int ntiles) { return ntiles < 1024 ? ntiles : 1024; }

}  // namespace vp

using namespace vp;

extern "C" {

size_t vp_conv3_small_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (!small3_ok(B, H, W, Cin, Cout)) return 0;
  int th, tw;
  const int ntiles = small3_tiles(B, H, W, wgrad_th(Cin, Cout), th, tw);
  return (size_t)small3_grid(ntiles) * 9 * Cout * Cin * sizeof(float);
}

int vp_conv3_small_wgrad_f32(const float* x, const float* dy, float* dw_ref, int B, int H, int W, int Cin, int Cout, void* ws,
                             size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && dy && dw_ref && ws, "vp_conv3_small_wgrad_f32: null pointer");
  VP_REQUIRE(small3_ok(B, H, W, Cin, Cout), "vp_conv3_small_wgrad_f32: needs Cin <= 36, Cout <= 8, Cin * Cout <= 288");
  const size_t need = vp_conv3_small_wgrad_workspace_bytes(B, H, W, Cin, Cout);
  if (ws_bytes < need) return fail(VP_ERR_WORKSPACE, "vp_conv3_small_wgrad_f32: workspace too small");
  const int TH = wgrad_th(Cin, Cout);
  int th, tw;
  const int ntiles = small3_tiles(B, H, W, TH, th, tw);
  const int grid = small3_grid(ntiles);
  const int CiP = (Cin + 3) / 4 * 4, CoP = (Cout + 3) / 4 * 4;
  size_t lds_floats = (size_t)(TH + 2) * (S3_TW + 2) * CiP + (size_t)TH * S3_TW * CoP;
  if (lds_floats < 256 * 16) lds_floats = 256 * 16;                  // the lane reduction reuses the buffer
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3_small_wgrad_kernel, dim3(grid), dim3(256), lds_floats * sizeof(float), s, x, dy, (float*)ws, B, H, W, Cin, Cout,
                     th, tw, ntiles, TH);
  int rc = check_launch("vp_conv3_small_wgrad_f32");
  if (rc) return rc;
  return slab_reduce_launch((const float*)ws, dw_ref, Cout, Cin, grid, s, 9);
}

/* forward / input gradient of the same layers (exact fp32, reference weight layout (Cout, Cin, 3, 3), no packing) */
int vp_conv3_small_fwd_f32(const float* x, const float* w_ref, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                           vp_stream stream) {
  VP_REQUIRE(x && w_ref && y, "vp_conv3_small_fwd_f32: null pointer");
  VP_REQUIRE(small3_ok(B, H, W, Cin, Cout), "vp_conv3_small_fwd_f32: needs Cin <= 36, Cout <= 8, Cin * Cout <= 288");
  return conv3_small_launch<false>(x, w_ref, bias, y, B, H, W, Cin, Cout, (hipStream_t)stream, "vp_conv3_small_fwd_f32");
}

int vp_conv3_small_dgrad_f32(const float* dy, const float* w_ref, float* dx, int B, int H, int W, int Cin, int Cout, vp_stream stream) {
  VP_REQUIRE(dy && w_ref && dx, "vp_conv3_small_dgrad_f32: null pointer");
  VP_REQUIRE(small3_ok(B, H, W, Cin, Cout), "vp_conv3_small_dgrad_f32: needs Cin <= 36, Cout <= 8, Cin * Cout <= 288");
  return conv3_small_launch<true>(dy, w_ref, nullptr, dx, B, H, W, Cout, Cin, (hipStream_t)stream, "vp_conv3_small_dgrad_f32");
}

}  // extern "C"
