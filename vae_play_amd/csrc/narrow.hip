// Edge layers whose small side has 1 or 3 channels (the image side of the first / last conv).
// An MFMA tile would be >= 90 % padding there, so these run on the vector ALU instead:
//   * conv5s1_smallout : out[.,n<=3] = act(bias + conv5x5(in[.,C]))  -- final conv + sigmoid
//     (models/networks.py:100-103): 16x16 output pixels per workgroup, 20x20x16-channel halo
//     tiles staged through LDS (80-B pixel stride -> conflict-free ds_read_b128), weights read
//     through wave-uniform (scalar) loads;
//   * wgrad_narrow_small : dW[n<=3][C][25] of that layer: channel on the lane, a 5-wide sliding
//     register window along the image row (1 coalesced 256-B load per 15 FMAs);
//   * wgrad_narrow_big   : dW[C][n<=3][25] of the first encoder conv (stride 2,
//     models/networks.py:14 with channel_in = 1 or 3): dy on the lane, x through uniform loads.
// Partial sums go to per-workgroup slabs (no atomics -> bit-reproducible) and are combined by
// slab_reduce_kernel into the reference weight layout.
#include "common.h"
#include "problems.h"
#include "narrow.h"

namespace vp {

template <int NOUT>
__global__ void __launch_bounds__(256) conv5s1_smallout_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int H, int W, int C, int act) {
  constexpr int T = 16, HALO = T + 4, CC = 16, PS = 20;
  __shared__ __attribute__((aligned(16))) float tile[HALO * HALO * PS];
  const int b = blockIdx.z, h0 = blockIdx.y * T, w0 = blockIdx.x * T;
  const int tx = threadIdx.x % T, ty = threadIdx.x / T;
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    for (int i = threadIdx.x; i < HALO * HALO * 4; i += 256) {
      const int pix = i >> 2, v = i & 3;
      const int py = pix / HALO, px = pix - py * HALO;
      const int h = h0 + py - 2, ww = w0 + px - 2;
      vp_f32x4 val = zero4();
      if (h >= 0 && h < H && ww >= 0 && ww < W) val = ld4(in + ((size_t)(b * H + h) * W + ww) * C + c0 + v * 4);
      *reinterpret_cast<vp_f32x4*>(&tile[pix * PS + v * 4]) = val;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const float* tp = &tile[((ty + r) * HALO + tx + q) * PS];
        const float* wp = w + (size_t)(r * 5 + q) * C + c0;   // wave-uniform address
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const vp_f32x4 xv = *reinterpret_cast<const vp_f32x4*>(tp + v * 4);
#pragma unroll
          for (int n = 0; n < NOUT; ++n) {
            const float* wn = wp + (size_t)n * kTaps * C + v * 4;
            acc[n] = fmaf(xv[0], wn[0], acc[n]);
            acc[n] = fmaf(xv[1], wn[1], acc[n]);
            acc[n] = fmaf(xv[2], wn[2], acc[n]);
            acc[n] = fmaf(xv[3], wn[3], acc[n]);
          }
        }
      }
    }
  }
  const int h = h0 + ty, ww = w0 + tx;
  if (h < H && ww < W) {
    float* o = out + ((size_t)(b * H + h) * W + ww) * NOUT;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
      float v = acc[n] + (bias ? bias[n] : 0.f);
      if (act == ACT_SIGMOID) v = 1.f / (1.f + __builtin_expf(-v));
      o[n] = v;
    }
  }
}

// slab[blk][tap][n][c] += sum over this workgroup's image rows; stride 1; c on the lane.
template <int NS>
__global__ void __launch_bounds__(256) wgrad_narrow_small_kernel(const float* __restrict__ big, const float* __restrict__ small,
                                                                 float* __restrict__ slab, int B, int H, int W, int Cb,
                                                                 int rows_per_block) {
  __shared__ float red[4][5 * NS][64];
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.y * 64 + lane;
  float acc[kTaps][NS];
#pragma unroll
  for (int t = 0; t < kTaps; ++t)
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[t][n] = 0.f;
  const int nrows = B * H;
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(nrows, row0 + rows_per_block);
  for (int row = row0 + g; row < row1; row += 4) {
    const int b = row / H, h = row - b * H;
    const float* sp = small + (size_t)row * W * NS;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int hb = h + r - 2;
      if (hb < 0 || hb >= H) continue;   // wave-uniform
      const float* bp = big + ((size_t)(b * H + hb) * W) * Cb + c;
      // sliding window over the row: win[j] = big[hb][x0 + j - 2]; four pixels per iteration so that four
      // independent 256-B loads are in flight per wave (the one-pixel loop was latency-bound: 12 TFLOP/s)
      float win[8];
      win[0] = 0.f; win[1] = 0.f;
      win[2] = bp[0];
      win[3] = W > 1 ? bp[Cb] : 0.f;
      for (int x0 = 0; x0 < W; x0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) win[4 + j] = (x0 + 2 + j < W) ? bp[(size_t)(x0 + 2 + j) * Cb] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (x0 + j < W) {
            float sv[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) sv[n] = sp[(x0 + j) * NS + n];
#pragma unroll
            for (int n = 0; n < NS; ++n) {
              acc[r * 5 + 0][n] = fmaf(win[j + 0], sv[n], acc[r * 5 + 0][n]);
              acc[r * 5 + 1][n] = fmaf(win[j + 1], sv[n], acc[r * 5 + 1][n]);
              acc[r * 5 + 2][n] = fmaf(win[j + 2], sv[n], acc[r * 5 + 2][n]);
              acc[r * 5 + 3][n] = fmaf(win[j + 3], sv[n], acc[r * 5 + 3][n]);
              acc[r * 5 + 4][n] = fmaf(win[j + 4], sv[n], acc[r * 5 + 4][n]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) win[j] = win[4 + j];
      }
    }
  }
  // combine the 4 waves through LDS, one tap-row (5 taps) at a time, and write the slab
  float* dst = slab + (size_t)blockIdx.x * kTaps * NS * Cb;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int n = 0; n < NS; ++n) red[g][q * NS + n][lane] = acc[r * 5 + q][n];
    __syncthreads();
    for (int i = g; i < 5 * NS; i += 4) {
      const float v = (red[0][i][lane] + red[1][i][lane]) + (red[2][i][lane] + red[3][i][lane]);
      const int q = i / NS, n = i - q * NS;
      dst[((size_t)(r * 5 + q) * NS + n) * Cb + c] = v;
    }
  }
}

// slab[blk][tap][c][n] ; stride 2; dy channel c on the lane, x (NB channels) via uniform loads.
template <int NB>
__global__ void __launch_bounds__(256) wgrad_narrow_big_kernel(const float* __restrict__ big, const float* __restrict__ small,
                                                               float* __restrict__ slab, int B, int H, int W, int Cs,
                                                               int stride, int rows_per_block) {
  __shared__ float red[4][5 * NB][64];
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.y * 64 + lane;
  const int Hb = H * stride, Wb = W * stride;
  float acc[kTaps][NB];
#pragma unroll
  for (int t = 0; t < kTaps; ++t)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[t][n] = 0.f;
  const int nrows = B * H;
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(nrows, row0 + rows_per_block);
  for (int row = row0 + g; row < row1; row += 4) {
    const int b = row / H, h = row - b * H;
    const float* sp = small + (size_t)row * W * Cs + c;
    for (int x = 0; x < W; ++x) {
      const float dyv = sp[(size_t)x * Cs];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int hb = stride * h + r - 2;
        if (hb < 0 || hb >= Hb) continue;
        const float* brow = big + ((size_t)(b * Hb + hb) * Wb) * NB;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const int wb = stride * x + q - 2;
          if (wb < 0 || wb >= Wb) continue;
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[r * 5 + q][n] = fmaf(dyv, brow[(size_t)wb * NB + n], acc[r * 5 + q][n]);
        }
      }
    }
  }
  float* dst = slab + (size_t)blockIdx.x * kTaps * Cs * NB;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int n = 0; n < NB; ++n) red[g][q * NB + n][lane] = acc[r * 5 + q][n];
    __syncthreads();
    for (int i = g; i < 5 * NB; i += 4) {
      const float v = (red[0][i][lane] + red[1][i][lane]) + (red[2][i][lane] + red[3][i][lane]);
      const int q = i / NB, n = i - q * NB;
      dst[((size_t)(r * 5 + q) * Cs + c) * NB + n] = v;
    }
  }
}

// dw_ref[cs][cb][tap] = sum_split slab[split][tap][cs][cb]; 64 outputs x 4 split-lanes per workgroup.
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cs, int Cb,
                                                          int nsplit) {
  __shared__ float red[4][64];
  const size_t per = (size_t)kTaps * Cs * Cb;
  const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  float s = 0.f;
  if (i < per)
    for (int sp = sl; sp < nsplit; sp += 4) s += slab[(size_t)sp * per + i];
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && i < per) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int cb = (int)(i % Cb);
    const size_t r = i / Cb;
    const int cs = (int)(r % Cs), t = (int)(r / Cs);
    dw[((size_t)cs * Cb + cb) * kTaps + t] = v;
  }
}

// ---- host dispatch ------------------------------------------------------------------------------
bool narrow_gather_applicable(const ConvGeom& g, int act) {
  return g.stride == 1 && (g.Cs == 1 || g.Cs == 3) && g.Cb % 16 == 0 && (act == ACT_NONE || act == ACT_SIGMOID);
}

int narrow_gather_launch(const float* big, const float* w_p0, const float* bias, float* out, const ConvGeom& g, int act,
                         hipStream_t s) {
  dim3 grid((g.Ws + 15) / 16, (g.Hs + 15) / 16, g.B);
  if (g.Cs == 3)
    hipLaunchKernelGGL((conv5s1_smallout_kernel<3>), grid, dim3(256), 0, s, big, w_p0, bias, out, g.Hs, g.Ws, g.Cb, act);
  else
    hipLaunchKernelGGL((conv5s1_smallout_kernel<1>), grid, dim3(256), 0, s, big, w_p0, bias, out, g.Hs, g.Ws, g.Cb, act);
  return check_launch("conv5s1_smallout");
}

static int narrow_rows_per_block(const ConvGeom& g) {
  const int rows = g.B * g.Hs;
  int blocks = 1024;
  int rpb = (rows + blocks - 1) / blocks;
  if (rpb < 4) rpb = 4;
  return rpb;
}

int narrow_wgrad_kind(const ConvGeom& g) {
  if (g.stride == 1 && (g.Cs == 1 || g.Cs == 3) && g.Cb % 64 == 0) return 1;   // narrow small side
  if ((g.Cb == 1 || g.Cb == 3) && g.Cs % 64 == 0) return 2;                     // narrow big side
  return 0;
}

size_t narrow_wgrad_ws_floats(const ConvGeom& g) {
  const int rpb = narrow_rows_per_block(g);
  const int nblk = (g.B * g.Hs + rpb - 1) / rpb;
  return (size_t)nblk * kTaps * g.Cs * g.Cb;
}

int narrow_wgrad_launch(const float* big, const float* small, float* dw_ref, const ConvGeom& g, float* ws, hipStream_t s) {
  const int kind = narrow_wgrad_kind(g);
  const int rpb = narrow_rows_per_block(g);
  const int nblk = (g.B * g.Hs + rpb - 1) / rpb;
  if (kind == 1) {
    dim3 grid(nblk, g.Cb / 64);
    if (g.Cs == 3)
      hipLaunchKernelGGL((wgrad_narrow_small_kernel<3>), grid, dim3(256), 0, s, big, small, ws, g.B, g.Hs, g.Ws, g.Cb, rpb);
    else
      hipLaunchKernelGGL((wgrad_narrow_small_kernel<1>), grid, dim3(256), 0, s, big, small, ws, g.B, g.Hs, g.Ws, g.Cb, rpb);
  } else {
    dim3 grid(nblk, g.Cs / 64);
    if (g.Cb == 3)
      hipLaunchKernelGGL((wgrad_narrow_big_kernel<3>), grid, dim3(256), 0, s, big, small, ws, g.B, g.Hs, g.Ws, g.Cs, g.stride, rpb);
    else
      hipLaunchKernelGGL((wgrad_narrow_big_kernel<1>), grid, dim3(256), 0, s, big, small, ws, g.B, g.Hs, g.Ws, g.Cs, g.stride, rpb);
  }
  int rc = check_launch("wgrad_narrow");
  if (rc) return rc;
  const size_t per = (size_t)kTaps * g.Cs * g.Cb;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, s, (const float*)ws, dw_ref, g.Cs, g.Cb, nblk);
  return check_launch("slab_reduce");
}

int slab_reduce_launch(const float* slab, float* dw_ref, int Cs, int Cb, int nsplit, hipStream_t s) {
  const size_t per = (size_t)kTaps * Cs * Cb;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), 0, s, slab, dw_ref, Cs, Cb, nsplit);
  return check_launch("slab_reduce");
}

}  // namespace vp
