// BatchNorm (+ activation) over an [R][C] NHWC view.  HBM-bound: every pass streams the
// activation once with 16-B loads; reductions are per-thread fp32 partials -> LDS tree ->
// per-chunk slab -> fp64 finalize (deterministic, no atomics).
//   R = B*H*W for BatchNorm2d, R = B for BatchNorm1d (models/networks.py:16,40,66,89).
#include <stdlib.h>
#include "common.h"
#include "problems.h"
#include "split.h"

namespace vp {

constexpr int BN_TX = 16;   // float4 lanes across channels  (64 channels per block)
constexpr int BN_TY = 16;   // row lanes
constexpr int BN_CH = BN_TX * 4;

struct BnGrid { int chunks_r, chunks_c, rows_per_chunk; };

inline BnGrid bn_grid(int R, int C) {
  BnGrid g;
  g.chunks_c = (C + BN_CH - 1) / BN_CH;
  int total = 1024;
  int want = total / g.chunks_c;
  if (want < 1) want = 1;
  int maxr = (R + 63) / 64;   // at least ~64 rows per chunk
  if (maxr < 1) maxr = 1;
  g.chunks_r = want < maxr ? want : maxr;
  g.rows_per_chunk = (R + g.chunks_r - 1) / g.chunks_r;
  g.chunks_r = (R + g.rows_per_chunk - 1) / g.rows_per_chunk;
  return g;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
// pre-activation of the normalised value: gamma * ((x - mean) * rstd) + beta, written ONCE so that forward and backward
// round it identically (the backward kernels rebuild the activation mask from it).  The hoisted form x * scale + shift
// with shift = beta - mean * scale cancels two terms of size |mean| / sigma: for Linear -> BatchNorm1d over a few similar
// samples (|mean| / sigma ~ 1e3) that cost three digits of y and made the VAE-GAN gradients 10-50 x noisier than torch's.
__device__ __forceinline__ float bn_pre(float x, float mu, float rs, float ga, float be) {
  return fmaf(ga, (x - mu) * rs, be);
}
// derivative of act at pre-activation value u
__device__ __forceinline__ float act_grad_pre(float u, int act, float slope) {
  switch (act) {
    case ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case ACT_LRELU: return u > 0.f ? 1.f : slope;
    case ACT_TANH: { float t = tanhf(u); return 1.f - t * t; }
    case ACT_SIGMOID: { float s = 1.f / (1.f + __expf(-u)); return s * (1.f - s); }
    default: return 1.f;
  }
}

// MODE 0: (sum x, sum x^2).  MODE 1: (sum g, sum g*xhat) with g = dy * act'(gamma*xhat+beta).
template <int MODE>
__global__ void __launch_bounds__(256) bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ part, int R, int C, int rows_per_chunk,
                                                         int act, float slope, size_t bxs = 0, size_t bps = 0) {
  // blockIdx.z = image index of a batched (InstanceNorm) call: bxs elements per image, bps partial floats per image
  x += blockIdx.z * bxs; part += blockIdx.z * bps;
  if (MODE == 1) { dy += blockIdx.z * bxs; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C; }
  __shared__ float sh[2][BN_TY][BN_CH + 4];
  const int tx = threadIdx.x % BN_TX, ty = threadIdx.x / BN_TX;
  const int c = blockIdx.y * BN_CH + tx * 4;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (C % 4 == 0);
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = c + j < C;
    mu[j] = (MODE == 1 && ok) ? mean[c + j] : 0.f;
    rs[j] = (MODE == 1 && ok) ? rstd[c + j] : 0.f;
    ga[j] = (MODE == 1 && ok && gamma) ? gamma[c + j] : 1.f;
    be[j] = (MODE == 1 && ok && beta) ? beta[c + j] : 0.f;
    // MODE 0 sums (x - pivot) and (x - pivot)^2 with pivot = the channel's value in row 0 (bn_stats_final adds it back):
    // E[x^2] - E[x]^2 on raw values loses log2(mean^2 / var) bits, which for a Linear -> BatchNorm1d over a batch of 4
    // similar images (mean^2 / var ~ 1e3-1e4) turned 2e-7 of input rounding into 1e-3 of sigma
    if (MODE == 0) mu[j] = ok ? x[c + j] : 0.f;
  }
  if (c < C) {
    auto accum = [&](const float (&xv)[4], const float (&gv)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 0) {
          const float d = xv[j] - mu[j];
          s0[j] += d;
          s1[j] = fmaf(d, d, s1[j]);           // (explicit fma here and in bn_small_*: the two paths must round identically)
        } else {
          const float xh = (xv[j] - mu[j]) * rs[j];
          const float g = gv[j] * act_grad_pre(fmaf(ga[j], xh, be[j]), act, slope);
          s0[j] += g;
          s1[j] = fmaf(g, xh, s1[j]);
        }
      }
    };
    int r = r0 + ty;
    if (vec) {
      // four rows per trip: 4 (MODE 0) or 8 (MODE 1) independent 16-B loads in flight; the sums are still taken in
      // row order, so the result does not depend on the unrolling
      for (; r + 3 * BN_TY < r1; r += 4 * BN_TY) {
        vp_f32x4 tx4[4], td4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t off = (size_t)(r + u * BN_TY) * C + c;
          tx4[u] = *reinterpret_cast<const vp_f32x4*>(x + off);
          if (MODE == 1) td4[u] = *reinterpret_cast<const vp_f32x4*>(dy + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float xv[4] = {tx4[u][0], tx4[u][1], tx4[u][2], tx4[u][3]}, gv[4] = {0.f, 0.f, 0.f, 0.f};
          if (MODE == 1) { gv[0] = td4[u][0]; gv[1] = td4[u][1]; gv[2] = td4[u][2]; gv[3] = td4[u][3]; }
          accum(xv, gv);
        }
      }
    }
    for (; r < r1; r += BN_TY) {
      float xv[4], gv[4] = {0.f, 0.f, 0.f, 0.f};
      const size_t off = (size_t)r * C + c;
      if (vec) {
        vp_f32x4 t = *reinterpret_cast<const vp_f32x4*>(x + off);
        xv[0] = t[0]; xv[1] = t[1]; xv[2] = t[2]; xv[3] = t[3];
        if (MODE == 1) {
          vp_f32x4 d = *reinterpret_cast<const vp_f32x4*>(dy + off);
          gv[0] = d[0]; gv[1] = d[1]; gv[2] = d[2]; gv[3] = d[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xv[j] = (c + j < C) ? x[off + j] : 0.f;
          if (MODE == 1) gv[j] = (c + j < C) ? dy[off + j] : 0.f;
        }
      }
      accum(xv, gv);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[0][ty][tx * 4 + j] = s0[j];
    sh[1][ty][tx * 4 + j] = s1[j];
  }
  __syncthreads();
  if (threadIdx.x < 2 * BN_CH) {
    const int which = threadIdx.x / BN_CH, cc = threadIdx.x % BN_CH;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BN_TY; ++k) s += sh[which][k][cc];
    const int cg = blockIdx.y * BN_CH + cc;
    if (cg < C) part[((size_t)which * gridDim.x + blockIdx.x) * C + cg] = s;
  }
}

// Finalize: 4 channels per 256-thread block, one wavefront (64 lanes) per channel; each lane sums every
// 64th chunk partial in fp64 with two independent chains, then a shuffle reduction.  (The first version,
// one thread per channel looping over up to 2048 chunks, cost 185 us per call; 16 lanes per channel
// still 9.5 us.)
constexpr int FIN_C = 4;

__device__ __forceinline__ void bn_final_reduce(const float* __restrict__ part, int nchunk, int C, int c, int lane,
                                                double& s, double& q) {
  double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
  if (c < C) {
    int k = lane;
    for (; k + 64 < nchunk; k += 128) {
      s0 += (double)part[(size_t)k * C + c];
      s1 += (double)part[(size_t)(k + 64) * C + c];
      q0 += (double)part[((size_t)nchunk + k) * C + c];
      q1 += (double)part[((size_t)nchunk + k + 64) * C + c];
    }
    if (k < nchunk) {
      s0 += (double)part[(size_t)k * C + c];
      q0 += (double)part[((size_t)nchunk + k) * C + c];
    }
  }
  s = wave_sum_d(s0 + s1);
  q = wave_sum_d(q0 + q1);
}

__global__ void __launch_bounds__(256) bn_stats_final_kernel(const float* __restrict__ part, int nchunk, int R, int C, float eps,
                                                             float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ rm, float* __restrict__ rv, const float* __restrict__ x,
                                                             size_t bps = 0, size_t bxs = 0) {
  part += blockIdx.z * bps; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C; x += blockIdx.z * bxs;
  const int c = blockIdx.x * FIN_C + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double s, q;
  bn_final_reduce(part, nchunk, C, c, lane, s, q);
  if (lane != 0 || c >= C) return;
  const double ms = s / R;                 // mean of (x - pivot), pivot = x[row 0][c] as in bn_partial_kernel<0>
  double var = q / R - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = (double)x[c] + ms;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
  if (rv) {
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

__global__ void __launch_bounds__(256) bn_bwd_final_kernel(const float* __restrict__ part, int nchunk, int C,
                                                           float* __restrict__ sum_g, float* __restrict__ sum_gx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, size_t bps = 0) {
  part += blockIdx.z * bps; sum_g += blockIdx.z * (size_t)C; sum_gx += blockIdx.z * (size_t)C;
  const int c = blockIdx.x * FIN_C + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double s, q;
  bn_final_reduce(part, nchunk, C, c, lane, s, q);
  if (lane != 0 || c >= C) return;
  sum_g[c] = (float)s;
  sum_gx[c] = (float)q;
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)q;
}

// y = act(gamma * (x - mean) * rstd + beta)
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, size_t n,
                                                         int C, int act, float slope, u16_t* __restrict__ y_split,
                                                         size_t bxs = 0) {
  x += blockIdx.z * bxs; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C;
  if (y) y += blockIdx.z * bxs;
  const bool vec = (C % 4 == 0);
  if (vec) {
    const size_t n4 = n / 4;
    const int C4 = C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % C4) * 4;
      vp_f32x4 v = *reinterpret_cast<const vp_f32x4*>(x + i * 4);
      vp_f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = act_apply(bn_pre(v[j], mean[c + j], rstd[c + j], gamma ? gamma[c + j] : 1.f, beta ? beta[c + j] : 0.f), act, slope);
      }
      if (y) *reinterpret_cast<vp_f32x4*>(y + i * 4) = o;
      if (y_split) store_split4(y_split, n, i * 4, o[0], o[1], o[2], o[3]);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % C);
      y[i] = act_apply(bn_pre(x[i], mean[c], rstd[c], gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f), act, slope);
    }
  }
}

// dx = gamma*rstd*(g - [batch_stats]*(sum_g + xhat*sum_gx)/R)
__global__ void __launch_bounds__(256) bn_act_bwd_kernel(const float* __restrict__ x, const float* dy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                                         float* dx, size_t n, int C, float invR, int act,
                                                         float slope, u16_t* __restrict__ dx_split, size_t bxs = 0) {
  x += blockIdx.z * bxs; dy += blockIdx.z * bxs; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C;
  sum_g += blockIdx.z * (size_t)C; sum_gx += blockIdx.z * (size_t)C;
  if (dx) dx += blockIdx.z * bxs;
  const bool vec = (C % 4 == 0);
  const size_t cnt = vec ? n / 4 : n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    if (vec) {
      const int c = (int)(i % (C / 4)) * 4;
      vp_f32x4 xv = *reinterpret_cast<const vp_f32x4*>(x + i * 4);
      vp_f32x4 dv = *reinterpret_cast<const vp_f32x4*>(dy + i * 4);
      vp_f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ga = gamma ? gamma[c + j] : 1.f, be = beta ? beta[c + j] : 0.f;
        const float xh = (xv[j] - mean[c + j]) * rstd[c + j];
        const float g = dv[j] * act_grad_pre(fmaf(ga, xh, be), act, slope);
        o[j] = ga * rstd[c + j] * (g - (sum_g[c + j] + xh * sum_gx[c + j]) * invR);
      }
      if (dx) *reinterpret_cast<vp_f32x4*>(dx + i * 4) = o;
      if (dx_split) store_split4(dx_split, n, i * 4, o[0], o[1], o[2], o[3]);
    } else {
      const int c = (int)(i % C);
      const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
      const float xh = (x[i] - mean[c]) * rstd[c];
      const float g = dy[i] * act_grad_pre(fmaf(ga, xh, be), act, slope);
      dx[i] = ga * rstd[c] * (g - (sum_g[c] + xh * sum_gx[c]) * invR);
    }
  }
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = act_apply(x[i], act, slope);
}
__global__ void act_bwd_from_y_kernel(const float* __restrict__ y, const float* dy, float* dx,
                                      size_t n, int act, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = y[i];
    float d;
    switch (act) {
      case ACT_RELU: d = v > 0.f ? 1.f : 0.f; break;
      case ACT_LRELU: d = v > 0.f ? 1.f : slope; break;
      case ACT_TANH: d = 1.f - v * v; break;
      case ACT_SIGMOID: d = v * (1.f - v); break;
      default: d = 1.f;
    }
    dx[i] = dy[i] * d;
  }
}

// ---- fast paths for C % 4 == 0: same (16 float4 columns x 16 rows) thread map as bn_partial_kernel, so the
// per-channel constants are computed once per thread instead of once per element, and every 16-lane
// group streams 256 contiguous bytes per row.
__global__ void __launch_bounds__(256) bn_act_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               u16_t* __restrict__ y_split, int R, int C, int rows_per_chunk,
                                                               int act, float slope, size_t bxs = 0, int sfmt = SPLIT_BF16) {
  x += blockIdx.z * bxs; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C;
  if (y) y += blockIdx.z * bxs;
  const int tx = threadIdx.x % BN_TX, ty = threadIdx.x / BN_TX;
  const int c = blockIdx.y * BN_CH + tx * 4;
  if (c >= C) return;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[c + j]; rs[j] = rstd[c + j];
    ga[j] = gamma ? gamma[c + j] : 1.f; be[j] = beta ? beta[c + j] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  // split planes span all images of a batched (InstanceNorm) launch: plane = gridDim.z * R * C elements, image z at z * bxs
  const size_t n = (size_t)R * C * gridDim.z;
  if (y_split) y_split += blockIdx.z * bxs;
  // four rows per trip: four independent 16-B loads in flight before the first use (one load per trip left the
  // small layers at 2-3 TB/s)
  int r = r0 + ty;
  for (; r + 3 * BN_TY < r1; r += 4 * BN_TY) {
    vp_f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vp_f32x4*>(x + (size_t)(r + u * BN_TY) * C + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = (size_t)(r + u * BN_TY) * C + c;
      vp_f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = act_apply(bn_pre(v[u][j], mu[j], rs[j], ga[j], be[j]), act, slope);
      if (y) *reinterpret_cast<vp_f32x4*>(y + off) = o;
      if (y_split) store_split4(y_split, n, off, o[0], o[1], o[2], o[3], sfmt);
    }
  }
  for (; r < r1; r += BN_TY) {
    const size_t off = (size_t)r * C + c;
    const vp_f32x4 v = *reinterpret_cast<const vp_f32x4*>(x + off);
    vp_f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_apply(bn_pre(v[j], mu[j], rs[j], ga[j], be[j]), act, slope);
    if (y) *reinterpret_cast<vp_f32x4*>(y + off) = o;
    if (y_split) store_split4(y_split, n, off, o[0], o[1], o[2], o[3], sfmt);
  }
}

__global__ void __launch_bounds__(256) bn_act_bwd_tiled_kernel(const float* __restrict__ x, const float* dy,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ sum_g, const float* __restrict__ sum_gx,
                                                               float* dx, u16_t* __restrict__ dx_split, int R, int C,
                                                               int rows_per_chunk, float invR, int act, float slope,
                                                               size_t bxs = 0, int sfmt = SPLIT_BF16, float sscale = 1.f,
                                                               int* __restrict__ sat = nullptr) {
  x += blockIdx.z * bxs; dy += blockIdx.z * bxs; mean += blockIdx.z * (size_t)C; rstd += blockIdx.z * (size_t)C;
  sum_g += blockIdx.z * (size_t)C; sum_gx += blockIdx.z * (size_t)C;
  if (dx) dx += blockIdx.z * bxs;
  const int tx = threadIdx.x % BN_TX, ty = threadIdx.x / BN_TX;
  const int c = blockIdx.y * BN_CH + tx * 4;
  if (c >= C) return;
  float mu[4], rs[4], ga[4], be[4], k0[4], k1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[c + j]; rs[j] = rstd[c + j];
    ga[j] = gamma ? gamma[c + j] : 1.f; be[j] = beta ? beta[c + j] : 0.f;
    k0[j] = sum_g[c + j] * invR; k1[j] = sum_gx[c + j] * invR;
  }
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  const size_t n = (size_t)R * C * gridDim.z;          // (batched launch: see bn_act_fwd_tiled_kernel)
  if (dx_split) dx_split += blockIdx.z * bxs;
  auto emit = [&](size_t off, const vp_f32x4& xv, const vp_f32x4& dv) {
    vp_f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xv[j] - mu[j]) * rs[j];
      const float g = dv[j] * act_grad_pre(fmaf(ga[j], xh, be[j]), act, slope);
      o[j] = ga[j] * rs[j] * (g - fmaf(xh, k1[j], k0[j]));       // (explicit fma: bn_small_bwd_kernel must round identically)
    }
    if (dx) *reinterpret_cast<vp_f32x4*>(dx + off) = o;
    if (dx_split) store_split4(dx_split, n, off, o[0] * sscale, o[1] * sscale, o[2] * sscale, o[3] * sscale, sfmt);
    // fp16 pairs clamp at +-65504 (split.h): tell the caller that a scaled gradient left fp16's range (a sticky flag, any writer wins)
    if (sat && sfmt == SPLIT_F16 &&
        fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))) * sscale > 65504.f) *sat = 1;
  };
  int r = r0 + ty;
  for (; r + 3 * BN_TY < r1; r += 4 * BN_TY) {          // eight independent 16-B loads in flight
    vp_f32x4 xv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = (size_t)(r + u * BN_TY) * C + c;
      xv[u] = *reinterpret_cast<const vp_f32x4*>(x + off);
      dv[u] = *reinterpret_cast<const vp_f32x4*>(dy + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) emit((size_t)(r + u * BN_TY) * C + c, xv[u], dv[u]);
  }
  for (; r < r1; r += BN_TY) {
    const size_t off = (size_t)r * C + c;
    emit(off, *reinterpret_cast<const vp_f32x4*>(x + off), *reinterpret_cast<const vp_f32x4*>(dy + off));
  }
}

// elementwise passes want more, smaller chunks than the reductions (no partial slabs to combine)
inline BnGrid bn_apply_grid(int R, int C) {
  BnGrid g;
  g.chunks_c = (C + BN_CH - 1) / BN_CH;
  int total = 4096;
  int rows_min = 4;                                                     // trips of 16 rows per workgroup, at least
  int want = total / g.chunks_c;
  if (want < 1) want = 1;
  int maxr = (R + BN_TY * rows_min - 1) / (BN_TY * rows_min);
  g.chunks_r = want < maxr ? want : maxr;
  if (g.chunks_r < 1) g.chunks_r = 1;
  g.rows_per_chunk = (R + g.chunks_r - 1) / g.chunks_r;
  g.chunks_r = (R + g.rows_per_chunk - 1) / g.rows_per_chunk;
  return g;
}


// ---- BatchNorm over a handful of rows (the dense layers: R = batch <= 64) in ONE launch --------------------------------------------
// nn.BatchNorm1d + F.relu behind nn.Linear (models/networks.py:66-67,89-90) normalises over the batch only: the three-launch form
// (partial sums, finaliser, apply) is pure launch latency there (3 x ~6 us for 32 rows).  One workgroup owns 64 channels and all
// rows; a thread keeps its <= 4 rows of 4 channels in registers.  The arithmetic is the three-launch path's, step for step (pivoted
// fp32 sums in the same order, fp64 finalisation, bn_pre for the normalised value), so results are bit-identical to it.
constexpr int BN_SMALL_R = 4 * BN_TY;       // 64 rows

__global__ void __launch_bounds__(256) bn_small_fwd_kernel(const float* __restrict__ x, int R, int C, float eps, float momentum,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rm,
                                                           float* __restrict__ rv, float* __restrict__ y, int act, float slope) {
  __shared__ float sh[2][BN_TY][BN_CH + 4];
  __shared__ float fin[2][BN_CH];
  const int tx = threadIdx.x % BN_TX, ty = threadIdx.x / BN_TX;
  const int c = blockIdx.x * BN_CH + tx * 4;
  const bool cok = c < C;                     // C % 4 == 0: a quad is inside or outside
  vp_f32x4 xv[4], pv = {0.f, 0.f, 0.f, 0.f};
  if (cok) pv = *reinterpret_cast<const vp_f32x4*>(x + c);           // pivot = row 0
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + BN_TY * k;
    if (cok && r < R) {
      xv[k] = *reinterpret_cast<const vp_f32x4*>(x + (size_t)r * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = xv[k][j] - pv[j]; s0[j] += d; s1[j] = fmaf(d, d, s1[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh[0][ty][tx * 4 + j] = s0[j]; sh[1][ty][tx * 4 + j] = s1[j]; }
  __syncthreads();
  if (threadIdx.x < 2 * BN_CH) {
    const int which = threadIdx.x / BN_CH, cc = threadIdx.x % BN_CH;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BN_TY; ++k) s += sh[which][k][cc];
    fin[which][cc] = s;
  }
  __syncthreads();
  if (threadIdx.x < BN_CH) {
    const int cc = threadIdx.x, cg = blockIdx.x * BN_CH + cc;
    if (cg < C) {
      const double ms = (double)fin[0][cc] / R;
      double var = (double)fin[1][cc] / R - ms * ms;
      if (var < 0.0) var = 0.0;
      const double m = (double)x[cg] + ms;
      const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
      mean[cg] = mf;
      rstd[cg] = rf;
      if (rm) rm[cg] = (1.f - momentum) * rm[cg] + momentum * mf;
      if (rv) {
        const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
        rv[cg] = (1.f - momentum) * rv[cg] + momentum * (float)unb;
      }
      fin[0][cc] = mf;
      fin[1][cc] = rf;
    }
  }
  __syncthreads();
  if (!cok || !y) return;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = fin[0][tx * 4 + j]; rs[j] = fin[1][tx * 4 + j];
    ga[j] = gamma ? gamma[c + j] : 1.f; be[j] = beta ? beta[c + j] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + BN_TY * k;
    if (r < R) {
      vp_f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = act_apply(bn_pre(xv[k][j], mu[j], rs[j], ga[j], be[j]), act, slope);
      *reinterpret_cast<vp_f32x4*>(y + (size_t)r * C + c) = o;
    }
  }
}

__global__ void __launch_bounds__(256) bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int R, int C, float invR, int act, float slope) {
  __shared__ float sh[2][BN_TY][BN_CH + 4];
  __shared__ float fin[2][BN_CH];
  const int tx = threadIdx.x % BN_TX, ty = threadIdx.x / BN_TX;
  const int c = blockIdx.x * BN_CH + tx * 4;
  const bool cok = c < C;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = cok ? mean[c + j] : 0.f; rs[j] = cok ? rstd[c + j] : 0.f;
    ga[j] = (cok && gamma) ? gamma[c + j] : 1.f; be[j] = (cok && beta) ? beta[c + j] : 0.f;
  }
  float gq[4][4], xh[4][4];
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + BN_TY * k;
    if (cok && r < R) {
      const vp_f32x4 xv = *reinterpret_cast<const vp_f32x4*>(x + (size_t)r * C + c);
      const vp_f32x4 dv = *reinterpret_cast<const vp_f32x4*>(dy + (size_t)r * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[k][j] = (xv[j] - mu[j]) * rs[j];
        gq[k][j] = dv[j] * act_grad_pre(fmaf(ga[j], xh[k][j], be[j]), act, slope);
        s0[j] += gq[k][j];
        s1[j] = fmaf(gq[k][j], xh[k][j], s1[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh[0][ty][tx * 4 + j] = s0[j]; sh[1][ty][tx * 4 + j] = s1[j]; }
  __syncthreads();
  if (threadIdx.x < 2 * BN_CH) {
    const int which = threadIdx.x / BN_CH, cc = threadIdx.x % BN_CH;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BN_TY; ++k) s += sh[which][k][cc];
    fin[which][cc] = s;
    const int cg = blockIdx.x * BN_CH + cc;
    if (cg < C) {
      if (which == 0 && dbeta) dbeta[cg] = s;
      if (which == 1 && dgamma) dgamma[cg] = s;
    }
  }
  __syncthreads();
  if (!cok) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = ty + BN_TY * k;
    if (r < R) {
      vp_f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k0 = fin[0][tx * 4 + j] * invR, k1 = fin[1][tx * 4 + j] * invR;
        o[j] = ga[j] * rs[j] * (gq[k][j] - fmaf(xh[k][j], k1, k0));
      }
      *reinterpret_cast<vp_f32x4*>(dx + (size_t)r * C + c) = o;
    }
  }
}

inline size_t bn_ws_floats(int R, int C) {
  BnGrid g = bn_grid(R, C);
  return (size_t)2 * g.chunks_r * C + (size_t)2 * C;   // partial slabs + (sum_g, sum_gx)
}

}  // namespace vp

using namespace vp;

extern "C" {

size_t vp_bn_workspace_bytes(int R, int C) { return bn_ws_floats(R, C) * sizeof(float); }

int vp_bn_stats_f32(const float* x, int R, int C, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                    float* running_var, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && mean && rstd && ws && R > 0 && C > 0, "vp_bn_stats_f32: bad arguments");
  if (ws_bytes < vp_bn_workspace_bytes(R, C)) return fail(VP_ERR_WORKSPACE, "vp_bn_stats_f32: workspace too small");
  BnGrid g = bn_grid(R, C);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  hipLaunchKernelGGL((bn_partial_kernel<0>), dim3(g.chunks_r, g.chunks_c), dim3(256), 0, s, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, R, C,
                     g.rows_per_chunk, 0, 0.f);
  int rc = check_launch("vp_bn_stats_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, s, (const float*)part, g.chunks_r, R, C, eps,
                     momentum, mean, rstd, running_mean, running_var, x);
  return check_launch("vp_bn_stats_f32(final)");
}

static int bn_act_fwd_impl(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                           void* y_split, int R, int C, int act, float slope, vp_stream stream, int fmt = SPLIT_BF16) {
  VP_REQUIRE(fmt == SPLIT_BF16 || fmt == SPLIT_F16, "vp_bn_act_fwd_split_fmt_f32: format 0 (bf16 pair) or 1 (fp16 pair)");
  VP_REQUIRE(x && mean && rstd && (y || y_split) && R > 0 && C > 0, "vp_bn_act_fwd: bad arguments");
  VP_REQUIRE(!y_split || C % 4 == 0, "vp_bn_act_fwd_split_f32: C must be a multiple of 4");
  const size_t n = (size_t)R * C;
  if (C % 4 == 0) {
    BnGrid g = bn_apply_grid(R, C);
    hipLaunchKernelGGL(bn_act_fwd_tiled_kernel, dim3(g.chunks_r, g.chunks_c), dim3(256), 0, (hipStream_t)stream, x, mean, rstd,
                       gamma, beta, y, (u16_t*)y_split, R, C, g.rows_per_chunk, act, slope, (size_t)0, fmt);
  } else {
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma,
                       beta, y, n, C, act, slope, (u16_t*)y_split);
  }
  return check_launch("vp_bn_act_fwd");
}

int vp_bn_act_fwd_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                      int R, int C, int act, float slope, vp_stream stream) {
  VP_REQUIRE(y, "vp_bn_act_fwd_f32: y is null");
  return bn_act_fwd_impl(x, mean, rstd, gamma, beta, y, nullptr, R, C, act, slope, stream);
}

int vp_bn_act_fwd_split_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* y, void* y_split, int R, int C, int act, float slope, vp_stream stream) {
  return bn_act_fwd_impl(x, mean, rstd, gamma, beta, y, y_split, R, C, act, slope, stream);
}

int vp_bn_act_fwd_split_fmt_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                float* y, void* y_split, int R, int C, int act, float slope, int fmt, vp_stream stream) {
  return bn_act_fwd_impl(x, mean, rstd, gamma, beta, y, y_split, R, C, act, slope, stream, fmt);
}

static int bn_act_bwd_impl(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta, int R, int C, int act,
                           float slope, int batch_stats, void* ws, size_t ws_bytes, vp_stream stream, int fmt = SPLIT_BF16,
                           float scale = 1.f, int* sat = nullptr) {
  VP_REQUIRE(x && dy && mean && rstd && (dx || dx_split) && ws && R > 0 && C > 0, "vp_bn_act_bwd_f32: bad arguments");
  VP_REQUIRE((fmt == SPLIT_BF16 || fmt == SPLIT_F16) && scale > 0.f, "vp_bn_act_bwd_split_fmt_f32: format 0 | 1, scale > 0");
  VP_REQUIRE(!dx_split || C % 4 == 0, "vp_bn_act_bwd_split_f32: C must be a multiple of 4");
  if (ws_bytes < vp_bn_workspace_bytes(R, C)) return fail(VP_ERR_WORKSPACE, "vp_bn_act_bwd_f32: workspace too small");
  BnGrid g = bn_grid(R, C);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  float* sum_g = part + (size_t)2 * g.chunks_r * C;
  float* sum_gx = sum_g + C;
  hipLaunchKernelGGL((bn_partial_kernel<1>), dim3(g.chunks_r, g.chunks_c), dim3(256), 0, s, x, dy, mean, rstd, gamma, beta, part,
                     R, C, g.rows_per_chunk, act, slope);
  int rc = check_launch("vp_bn_act_bwd_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, s, (const float*)part, g.chunks_r, C, sum_g, sum_gx,
                     dgamma, dbeta);
  rc = check_launch("vp_bn_act_bwd_f32(final)");
  if (rc) return rc;
  const size_t n = (size_t)R * C;
  const float invR = batch_stats ? 1.f / (float)R : 0.f;
  if (C % 4 == 0) {
    BnGrid ga = bn_apply_grid(R, C);
    hipLaunchKernelGGL(bn_act_bwd_tiled_kernel, dim3(ga.chunks_r, ga.chunks_c), dim3(256), 0, s, x, dy, mean, rstd, gamma, beta,
                       (const float*)sum_g, (const float*)sum_gx, dx, (u16_t*)dx_split, R, C, ga.rows_per_chunk, invR, act, slope,
                       (size_t)0, fmt, scale, sat);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, s, x, dy, mean, rstd, gamma, beta,
                       (const float*)sum_g, (const float*)sum_gx, dx, n, C, invR, act, slope, (u16_t*)dx_split);
  }
  return check_launch("vp_bn_act_bwd_f32(apply)");
}

int vp_bn_act_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, float* dx, float* dgamma, float* dbeta, int R, int C, int act, float slope,
                      int batch_stats, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(dx, "vp_bn_act_bwd_f32: dx is null");
  return bn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, dx, nullptr, dgamma, dbeta, R, C, act, slope, batch_stats, ws, ws_bytes, stream);
}

int vp_bn_act_bwd_split_fmt_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta, int R, int C, int act,
                                float slope, int batch_stats, int fmt, float scale, void* ws, size_t ws_bytes, vp_stream stream) {
  return bn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, dx, dx_split, dgamma, dbeta, R, C, act, slope, batch_stats, ws, ws_bytes, stream,
                         fmt, scale);
}

int vp_bn_act_bwd_split_fmt_sat_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta, int R, int C, int act,
                                    float slope, int batch_stats, int fmt, float scale, int* saturated, void* ws, size_t ws_bytes,
                                    vp_stream stream) {
  return bn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, dx, dx_split, dgamma, dbeta, R, C, act, slope, batch_stats, ws, ws_bytes, stream,
                         fmt, scale, saturated);
}

int vp_bn_act_bwd_split_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta, int R, int C, int act,
                            float slope, int batch_stats, void* ws, size_t ws_bytes, vp_stream stream) {
  return bn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, dx, dx_split, dgamma, dbeta, R, C, act, slope, batch_stats, ws, ws_bytes, stream);
}

// ---- nn.InstanceNorm2d(affine=False) + activation: the same kernels with blockIdx.z = image (models/blocks.py:22) --------
size_t vp_instnorm_workspace_bytes(int B, int R, int C) { return (size_t)B * bn_ws_floats(R, C) * sizeof(float); }

static int instnorm_act_fwd_impl(const float* x, float* y, void* y_split, float* mean, float* rstd, int B, int R, int C, float eps, int act,
                                 float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && y && mean && rstd && ws && B > 0 && R > 0 && C > 0, "vp_instnorm_act_fwd_f32: bad arguments");
  VP_REQUIRE(!y_split || C % 4 == 0, "vp_instnorm_act_fwd_split_f32: C must be a multiple of 4");
  if (ws_bytes < vp_instnorm_workspace_bytes(B, R, C)) return fail(VP_ERR_WORKSPACE, "vp_instnorm_act_fwd_f32: workspace too small");
  const BnGrid g = bn_grid(R, C);
  const size_t bxs = (size_t)R * C, bps = (size_t)2 * g.chunks_r * C;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  hipLaunchKernelGGL((bn_partial_kernel<0>), dim3(g.chunks_r, g.chunks_c, B), dim3(256), 0, s, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, R, C,
                     g.rows_per_chunk, 0, 0.f, bxs, bps);
  int rc = check_launch("vp_instnorm_act_fwd_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FIN_C - 1) / FIN_C, 1, B), dim3(256), 0, s, (const float*)part, g.chunks_r, R, C,
                     eps, 0.f, mean, rstd, (float*)nullptr, (float*)nullptr, x, bps, bxs);
  rc = check_launch("vp_instnorm_act_fwd_f32(final)");
  if (rc) return rc;
  if (C % 4 == 0) {
    const BnGrid ga = bn_apply_grid(R, C);
    hipLaunchKernelGGL(bn_act_fwd_tiled_kernel, dim3(ga.chunks_r, ga.chunks_c, B), dim3(256), 0, s, x, (const float*)mean,
                       (const float*)rstd, (const float*)nullptr, (const float*)nullptr, y, (u16_t*)y_split, R, C, ga.rows_per_chunk,
                       act, slope, bxs);
  } else {
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(bxs / 4 + 1, 256), 1, B), dim3(256), 0, s, x, (const float*)mean,
                       (const float*)rstd, (const float*)nullptr, (const float*)nullptr, y, bxs, C, act, slope, (u16_t*)nullptr, bxs);
  }
  return check_launch("vp_instnorm_act_fwd_f32(apply)");
}

int vp_instnorm_act_fwd_f32(const float* x, float* y, float* mean, float* rstd, int B, int R, int C, float eps, int act,
                            float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  return instnorm_act_fwd_impl(x, y, nullptr, mean, rstd, B, R, C, eps, act, slope, ws, ws_bytes, stream);
}

int vp_instnorm_act_fwd_split_f32(const float* x, float* y, void* y_split, float* mean, float* rstd, int B, int R, int C, float eps,
                                  int act, float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(y_split, "vp_instnorm_act_fwd_split_f32: null split output");
  return instnorm_act_fwd_impl(x, y, y_split, mean, rstd, B, R, C, eps, act, slope, ws, ws_bytes, stream);
}

static int instnorm_act_bwd_impl(const float* x, const float* dy, const float* mean, const float* rstd, float* dx, void* dx_split, int B,
                                 int R, int C, int act, float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && dy && mean && rstd && dx && ws && B > 0 && R > 0 && C > 0, "vp_instnorm_act_bwd_f32: bad arguments");
  VP_REQUIRE(!dx_split || C % 4 == 0, "vp_instnorm_act_bwd_split_f32: C must be a multiple of 4");
  if (ws_bytes < vp_instnorm_workspace_bytes(B, R, C)) return fail(VP_ERR_WORKSPACE, "vp_instnorm_act_bwd_f32: workspace too small");
  const BnGrid g = bn_grid(R, C);
  const size_t bxs = (size_t)R * C, bps = (size_t)2 * g.chunks_r * C;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  float* sum_g = part + (size_t)B * bps;          // [B][C], then sum_gx [B][C]
  float* sum_gx = sum_g + (size_t)B * C;
  hipLaunchKernelGGL((bn_partial_kernel<1>), dim3(g.chunks_r, g.chunks_c, B), dim3(256), 0, s, x, dy, mean, rstd, (const float*)nullptr,
                     (const float*)nullptr, part, R, C, g.rows_per_chunk, act, slope, bxs, bps);
  int rc = check_launch("vp_instnorm_act_bwd_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FIN_C - 1) / FIN_C, 1, B), dim3(256), 0, s, (const float*)part, g.chunks_r, C, sum_g,
                     sum_gx, (float*)nullptr, (float*)nullptr, bps);
  rc = check_launch("vp_instnorm_act_bwd_f32(final)");
  if (rc) return rc;
  const float invR = 1.f / (float)R;
  if (C % 4 == 0) {
    const BnGrid ga = bn_apply_grid(R, C);
    hipLaunchKernelGGL(bn_act_bwd_tiled_kernel, dim3(ga.chunks_r, ga.chunks_c, B), dim3(256), 0, s, x, dy, mean, rstd,
                       (const float*)nullptr, (const float*)nullptr, (const float*)sum_g, (const float*)sum_gx, dx, (u16_t*)dx_split, R, C,
                       ga.rows_per_chunk, invR, act, slope, bxs);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(grid_for(bxs / 4 + 1, 256), 1, B), dim3(256), 0, s, x, dy, mean, rstd, (const float*)nullptr,
                       (const float*)nullptr, (const float*)sum_g, (const float*)sum_gx, dx, bxs, C, invR, act, slope, (u16_t*)nullptr, bxs);
  }
  return check_launch("vp_instnorm_act_bwd_f32(apply)");
}

int vp_instnorm_act_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd, float* dx, int B, int R, int C,
                            int act, float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  return instnorm_act_bwd_impl(x, dy, mean, rstd, dx, nullptr, B, R, C, act, slope, ws, ws_bytes, stream);
}

int vp_instnorm_act_bwd_split_f32(const float* x, const float* dy, const float* mean, const float* rstd, float* dx, void* dx_split, int B,
                                  int R, int C, int act, float slope, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(dx_split, "vp_instnorm_act_bwd_split_f32: null split output");
  return instnorm_act_bwd_impl(x, dy, mean, rstd, dx, dx_split, B, R, C, act, slope, ws, ws_bytes, stream);
}

int vp_bn_small_fwd_f32(const float* x, int R, int C, float eps, float momentum, const float* gamma, const float* beta, float* mean,
                        float* rstd, float* running_mean, float* running_var, float* y, int act, float slope, vp_stream stream) {
  VP_REQUIRE(x && mean && rstd && R > 0 && C > 0, "vp_bn_small_fwd_f32: bad arguments");
  VP_REQUIRE(R <= BN_SMALL_R && C % 4 == 0, "vp_bn_small_fwd_f32: at most 64 rows, C a multiple of 4 (use vp_bn_stats_f32 + vp_bn_act_fwd_f32)");
  hipLaunchKernelGGL(bn_small_fwd_kernel, dim3((C + BN_CH - 1) / BN_CH), dim3(256), 0, (hipStream_t)stream, x, R, C, eps, momentum, gamma, beta,
                     mean, rstd, running_mean, running_var, y, act, slope);
  return check_launch("vp_bn_small_fwd_f32");
}

int vp_bn_small_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        float* dx, float* dgamma, float* dbeta, int R, int C, int act, float slope, int batch_stats, vp_stream stream) {
  VP_REQUIRE(x && dy && mean && rstd && dx && R > 0 && C > 0, "vp_bn_small_bwd_f32: bad arguments");
  VP_REQUIRE(R <= BN_SMALL_R && C % 4 == 0, "vp_bn_small_bwd_f32: at most 64 rows, C a multiple of 4 (use vp_bn_act_bwd_f32)");
  hipLaunchKernelGGL(bn_small_bwd_kernel, dim3((C + BN_CH - 1) / BN_CH), dim3(256), 0, (hipStream_t)stream, x, dy, mean, rstd, gamma, beta, dx,
                     dgamma, dbeta, R, C, batch_stats ? 1.f / (float)R : 0.f, act, slope);
  return check_launch("vp_bn_small_bwd_f32");
}

int vp_act_fwd_f32(const float* x, float* y, size_t n, int act, float slope, vp_stream stream) {
  VP_REQUIRE(x && y && n > 0, "vp_act_fwd_f32: bad arguments");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, act, slope);
  return check_launch("vp_act_fwd_f32");
}

int vp_act_bwd_from_y_f32(const float* y, const float* dy, float* dx, size_t n, int act, float slope, vp_stream stream) {
  VP_REQUIRE(y && dy && dx && n > 0, "vp_act_bwd_from_y_f32: bad arguments");
  hipLaunchKernelGGL(act_bwd_from_y_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n, act, slope);
  return check_launch("vp_act_bwd_from_y_f32");
}
}
