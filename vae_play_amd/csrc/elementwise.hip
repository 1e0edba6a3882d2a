// HBM-bound glue kernels: layout transposes, latent (reparameterise + KL), per-pixel BCE with
// wavefront reductions, and the fused flat-arena optimiser steps.
#include <stdlib.h>
#include "common.h"
#include "problems.h"
#include "split.h"

namespace vp {

// 512 zero bytes in global memory (a padded row's chunks are read at offsets up to 128 B): the "zero page" that out-of-range gathers of the implicit-GEMM kernels
// read (problems.h).  A __device__ symbol of this code object, not an allocation.
__device__ __attribute__((aligned(64))) unsigned int vp_zero_page_storage[128] = {0};

const void* vp_zero_page() {
  static const void* ptr = [] {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vp_zero_page_storage)) != hipSuccess) p = nullptr;
    return (const void*)p;
  }();
  return ptr;
}

// out[b][c][r] = in[b][r][c]  (32x32 LDS tile, +1 pad: conflict-free for ds_read_b32 columns)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols,
                                 u16_t* __restrict__ out_split, size_t plane, int fmt) {
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = in[base + (size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) {
      const float v = tile[threadIdx.x][j];
      const size_t o = base + (size_t)c * rows + r;
      if (out) out[o] = v;
      if (out_split) {
        u16_t h, l;
        split_f32(v, h, l, fmt);
        out_split[o] = h;
        out_split[plane + o] = l;
      }
    }
  }
}

inline int launch_transpose(const float* in, float* out, int B, int rows, int cols, hipStream_t s, const char* what,
                            void* out_split = nullptr, int fmt = SPLIT_BF16) {
  VP_REQUIRE(in && (out || out_split) && B > 0 && rows > 0 && cols > 0, "%s: bad arguments", what);
  VP_REQUIRE(fmt == SPLIT_BF16 || fmt == SPLIT_F16, "%s: format 0 (bf16 pair) or 1 (fp16 pair)", what);
  VP_REQUIRE(B <= 65535, "%s: batch > 65535", what);
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, B);
  VP_REQUIRE(grid.y <= 65535, "%s: too many row tiles", what);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s, in, out, rows, cols, (u16_t*)out_split, (size_t)B * rows * cols, fmt);
  return check_launch(what);
}

// one wavefront per sample: z = eps*exp(0.5*lv)+mu ; kl[b] = -0.5*sum(1 + lv - mu^2 - exp(lv))
__global__ void latent_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                  float* __restrict__ z, float* __restrict__ kl, int Z) {
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int j = threadIdx.x; j < Z; j += 64) {
    const size_t i = (size_t)b * Z + j;
    const float m = mu[i], l = lv[i];
    z[i] = eps[i] * expf(0.5f * l) + m;
    acc += -expf(l) - m * m + l + 1.f;
  }
  acc = wave_sum(acc);
  if (kl && threadIdx.x == 0) kl[b] = -0.5f * acc;
}

__global__ void latent_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                  const float* __restrict__ dz, const float* __restrict__ gkl, float gkl_scalar,
                                  float* __restrict__ dmu, float* __restrict__ dlv, int B, int Z) {
  const size_t n = (size_t)B * Z;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Z);
    const float g = (gkl ? gkl[b] : 0.f) + gkl_scalar;
    const float m = mu[i], l = lv[i];
    const float d = dz ? dz[i] : 0.f;
    dmu[i] = d + g * m;
    dlv[i] = d * eps[i] * 0.5f * expf(0.5f * l) + g * 0.5f * (expf(l) - 1.f);
  }
}

// MODE 0: BCE term  MODE 1: plain sum
template <int MODE>
__global__ void __launch_bounds__(256) reduce_partial_kernel(const float* __restrict__ p, const float* __restrict__ t, size_t n,
                                                             float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    vp_f32x4 pv = *reinterpret_cast<const vp_f32x4*>(p + i * 4);
    if (MODE == 0) {
      vp_f32x4 tv = *reinterpret_cast<const vp_f32x4*>(t + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lp = fmaxf(logf(pv[j]), -100.f), lq = fmaxf(logf(1.f - pv[j]), -100.f);
        acc -= tv[j] * lp + (1.f - tv[j]) * lq;
      }
    } else {
      acc += (pv[0] + pv[1]) + (pv[2] + pv[3]);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
    const size_t i = n4 * 4 + threadIdx.x;
    if (MODE == 0) {
      const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(logf(1.f - p[i]), -100.f);
      acc -= t[i] * lp + (1.f - t[i]) * lq;
    } else {
      acc += p[i];
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void reduce_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) acc += (double)part[i];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) out[0] = (float)acc;
}


// loss tail of the VAE step in one launch: recon = sum of the BCE partials, kl_sum = sum_b kl[b], loss = recon + kl_sum
// (replaces two reduce_final launches, a partial launch over B values and an ATen add on the benchmarked path)
__global__ void vae_loss_final_kernel(const float* __restrict__ part, int nblocks, const float* __restrict__ kl, int B,
                                      float* __restrict__ recon, float* __restrict__ kl_sum, float* __restrict__ loss, float loss_scale) {
  double a = 0.0, k = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) a += (double)part[i];
  for (int i = threadIdx.x; i < B; i += 64) k += (double)kl[i];
  a = wave_sum_d(a);
  k = wave_sum_d(k);
  if (threadIdx.x == 0) {
    recon[0] = (float)a;
    kl_sum[0] = (float)k;
    loss[0] = ((float)a + (float)k) * loss_scale;   // fp32 add of the two rounded sums, then the caller's 1/B (an exact scaling
                                                    // for power-of-two batches; torch's (recon + kl) / B otherwise differs by <= 1 ulp)
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

// ---- 0.5*(a-b)^2: per element (VaeGan.loss "nle", models/networks.py:267) and summed per row (":273", the
// feature-matching term between discriminator layers) ------------------------------------------------------
__global__ void half_sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    out[i] = 0.5f * d * d;
  }
}

// one workgroup per row: 256 lanes stride the row, fp32 lane partials, fp64 cross-lane tree (bit-reproducible)
__global__ void __launch_bounds__(256) half_sqdiff_rowsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out, int n) {
  __shared__ double sh[4];
  const size_t base = (size_t)blockIdx.x * n;
  float s0 = 0.f, s1 = 0.f;
  int j = threadIdx.x;
  for (; j + 256 < n; j += 512) {
    const float d0 = a[base + j] - b[base + j], d1 = a[base + j + 256] - b[base + j + 256];
    s0 += 0.5f * d0 * d0;
    s1 += 0.5f * d1 * d1;
  }
  if (j < n) { const float d0 = a[base + j] - b[base + j]; s0 += 0.5f * d0 * d0; }
  const double w = wave_sum_d((double)s0 + (double)s1);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// da = g * (a - b), db = -da with g per element (per_row = 0) or per row (per_row = 1)
__global__ void half_sqdiff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                       float* __restrict__ da, float* __restrict__ db, size_t total, int n, int per_row) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float gv = per_row ? g[i / (size_t)n] : g[i];
    const float d = gv * (a[i] - b[i]);
    if (da) da[i] = d;
    if (db) db[i] = -d;
  }
}

// ---- VAE-GAN loss heads (VaeGan.loss, models/networks.py:275-279; train.py:63-66) ----
// GAN head over the (3B) discriminator logits of (original | reconstructed | sampled): p = sigmoid(logit),
// bce_original = -log(p + 1e-3), bce_predicted / bce_sampled = -log(1 - p + 1e-3); sums[0..2] = their sums over the B rows
// of each group, dlogit = coef * d(sum of the three)/dlogit.  One workgroup (3B is a few dozen rows).
__global__ void __launch_bounds__(256) gan_head_kernel(const float* __restrict__ logit, int B, float coef, float* __restrict__ p_out,
                                                       float* __restrict__ sums, float* __restrict__ dlogit) {
  __shared__ float sh[3][4];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 3 * B; i += 256) {
    const float y = 1.f / (1.f + __expf(-logit[i]));
    const int grp = i / B;
    float u, dldy;
    if (grp == 0) { u = y + 1e-3f; dldy = -1.f / u; }
    else          { u = (1.f - y) + 1e-3f; dldy = 1.f / u; }
    acc[grp] += -__logf(u);
    if (p_out) p_out[i] = y;
    if (dlogit) dlogit[i] = coef * dldy * y * (1.f - y);
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float w = wave_sum(acc[g]);
    if ((threadIdx.x & 63) == 0) sh[g][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if (threadIdx.x < 3 && sums) sums[threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// F.smooth_l1_loss(targets, cat(a, b), reduction="sum") * scale (models/networks.py:279: beta = 1) over B rows of n1 + n2 columns;
// a [B][n1] and b [B][n2] are the two heads of DirectDecoder before the cat; da / db = d loss / d a, b.  One workgroup.
__global__ void __launch_bounds__(256) smooth_l1_cat_kernel(const float* __restrict__ t, const float* __restrict__ a, const float* __restrict__ b,
                                                            int B, int n1, int n2, float scale, float* __restrict__ loss,
                                                            float* __restrict__ da, float* __restrict__ db) {
  __shared__ float sh[4];
  const int n = n1 + n2;
  float acc = 0.f;
  for (int i = threadIdx.x; i < B * n; i += 256) {
    const int r = i / n, c = i - r * n;
    const float pv = c < n1 ? a[r * n1 + c] : b[r * n2 + (c - n1)];
    const float d = t[i] - pv, ad = fabsf(d);
    acc += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
    const float g = -scale * fminf(fmaxf(d, -1.f), 1.f);
    if (c < n1) { if (da) da[r * n1 + c] = g; }
    else if (db) db[r * n2 + (c - n1)] = g;
  }
  const float w = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0 && loss) loss[0] = scale * ((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// ---- segmentation loss of train_BE.py:58-59: w * BCEWithLogits(x, t) (mean) + dice(sigmoid(x), t) (tools/ops.py:12-19)
// sums[b] = { sum bce_i, sum p_i t_i, sum p_i, sum t_i } over the n elements of sample b, p = sigmoid(x).
// Two stages (chunk partials in fp32 lanes -> fp64 tree, then a fixed-order fp64 sum over chunks): bit-reproducible.
constexpr int BE_CHUNK = 4096;

// PROBS: x already holds probabilities (plain dice loss, tools/ops.py:178-185): no sigmoid, no BCE term
template <bool PROBS>
__global__ void __launch_bounds__(256) be_loss_partial_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                              double* __restrict__ part, int n, int nchunk) {
  __shared__ double sh[4][4];
  const int b = blockIdx.y, c = blockIdx.x;
  const size_t base = (size_t)b * n;
  const int i0 = c * BE_CHUNK, i1 = min(n, i0 + BE_CHUNK);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    const float xv = x[base + i], tv = t[base + i];
    float p = xv;
    if constexpr (!PROBS) {
      const float e = __expf(-fabsf(xv));
      p = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      s[0] += fmaxf(xv, 0.f) - xv * tv + log1pf(e);
    }
    s[1] += p * tv;
    s[2] += p;
    s[3] += tv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double w = wave_sum_d((double)s[k]);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    part[((size_t)b * nchunk + c) * 4 + k] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
  }
}

__global__ void be_loss_final_kernel(const double* __restrict__ part, float* __restrict__ sums, float* __restrict__ loss, int B,
                                     int n, int nchunk, float bce_weight, float smooth) {
  __shared__ double acc[2];
  if (threadIdx.x == 0) { acc[0] = 0.0; acc[1] = 0.0; }
  __syncthreads();
  if (threadIdx.x == 0) {                      // B is a minibatch size: a serial fixed-order sum is both cheap and exact
    double bce = 0.0, dice = 0.0;
    for (int b = 0; b < B; ++b) {
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c = 0; c < nchunk; ++c)
        for (int k = 0; k < 4; ++k) v[k] += part[((size_t)b * nchunk + c) * 4 + k];
      for (int k = 0; k < 4; ++k) sums[b * 4 + k] = (float)v[k];
      bce += v[0];
      dice += (2.0 * v[1] + smooth) / (v[2] + v[3] + smooth);
    }
    loss[0] = (float)(bce_weight * bce / ((double)B * n) + 1.0 - dice / B);
  }
}

// dx_i = g * [ w/(B n) (p_i - t_i) + (a_b t_i + b_b) p_i (1 - p_i) ],  D = P + T + s, a_b = -2/(B D), b_b = (2 I + s)/(B D^2)
template <bool PROBS>
__global__ void be_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ sums,
                                   const float* __restrict__ gptr, float* __restrict__ dx, int B, int n, float bce_weight,
                                   float smooth) {
  const int b = blockIdx.y;
  const float g = gptr ? gptr[0] : 1.f;
  const float I = sums[b * 4 + 1], D = sums[b * 4 + 2] + sums[b * 4 + 3] + smooth;
  const float cb = g * bce_weight / ((float)B * (float)n);
  const float ab = g * (-2.f / ((float)B * D)), bb = g * ((2.f * I + smooth) / ((float)B * D * D));
  const size_t base = (size_t)b * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float xv = x[base + i], tv = t[base + i];
    if constexpr (PROBS) {
      dx[base + i] = ab * tv + bb;
    } else {
      const float e = __expf(-fabsf(xv));
      const float p = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      dx[base + i] = cb * (p - tv) + (ab * tv + bb) * p * (1.f - p);
    }
  }
}

// ---- nn.AdaptiveAvgPool2d((1,1)) on NHWC (models/networks_BE_font.py:61): out[b][c] = mean over the HW pixels ------
// one workgroup per (image, 64-channel group): 4 pixel-lanes x 64 channels, fixed-order tree -> bit-reproducible
__global__ void __launch_bounds__(256) global_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C) {
  __shared__ float sh[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int p = g; p < HW; p += 4) s += x[((size_t)b * HW + p) * C + c];
  sh[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < C) out[(size_t)b * C + c] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x])) / (float)HW;
}

__global__ void global_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t total, int HW, int C) {
  const float inv = 1.f / (float)HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)HW * C);
    dx[i] = dy[b * C + (i % C)] * inv;
  }
}

// ---- nn.Softmax(dim=-1) over rows of n (SelfAttentionBlock, models/blocks.py:75,90): one wavefront per row --------
__global__ void __launch_bounds__(256) softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (size_t)row * n;
  float m = -3.4e38f;
  for (int j = lane; j < n; j += 64) m = fmaxf(m, xr[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += __expf(xr[j] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int j = lane; j < n; j += 64) y[(size_t)row * n + j] = __expf(xr[j] - m) * inv;
}

// dx = y * (dy - sum_j dy_j y_j)
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int R, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const size_t base = (size_t)row * n;
  float d = 0.f;
  for (int j = lane; j < n; j += 64) d += dy[base + j] * y[base + j];
  d = wave_sum(d);
  for (int j = lane; j < n; j += 64) dx[base + j] = y[base + j] * (dy[base + j] - d);
}

// ---- F.l1_loss(a, b) (mean) of train_BE_font.py:158: |a - b| partial sums, then sign(a - b) * g / n ----------------
__global__ void __launch_bounds__(256) l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         double* __restrict__ part, size_t n) {
  __shared__ double sh[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += fabsf(a[i] - b[i]);
  const double w = wave_sum_d((double)s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void l1_final_kernel(const double* __restrict__ part, int nb, size_t n, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    out[0] = (float)(s / (double)n);
  }
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gptr,
                              float* __restrict__ da, float* __restrict__ db, size_t n) {
  const float g = (gptr ? gptr[0] : 1.f) / (float)n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    const float v = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    if (da) da[i] = v;
    if (db) db[i] = -v;
  }
}

inline unsigned reduce_blocks(size_t n) { return grid_for(n / 4 + 1, 256, 1024); }

__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ gptr,
                               float gscale, float* __restrict__ dp, size_t n) {
  const float g = (gptr ? gptr[0] : 1.f) * gscale;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float pv = p[i];
    dp[i] = g * (pv - t[i]) / fmaxf((1.f - pv) * pv, 1e-12f);
  }
}

__global__ void bce_sigmoid_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, float gscale,
                                       float* __restrict__ dl, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    vp_f32x4 pv = *reinterpret_cast<const vp_f32x4*>(p + i * 4);
    vp_f32x4 tv = *reinterpret_cast<const vp_f32x4*>(t + i * 4);
    vp_f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gscale * (pv[j] - tv[j]);
    *reinterpret_cast<vp_f32x4*>(dl + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    dl[i] = gscale * (p[i] - t[i]);
  }
}

// dlogit = gscale*(p - t) as fp32 [npix][C] and, zero-padded to CP channels, as split bf16 planes
// [npix][CP] (so that the 3-channel final-conv dgrad can run on the bf16x3 scatter kernel with K = 25*CP)
__global__ void bce_sigmoid_bwd_pad_kernel(const float* __restrict__ p, const float* __restrict__ t, float gscale,
                                           float* __restrict__ dl, u16_t* __restrict__ dl_split, size_t npix, int C, int CP) {
  const size_t n = npix * CP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / CP;
    const int c = (int)(i - pix * CP);
    float v = 0.f;
    if (c < C) {
      v = gscale * (p[pix * C + c] - t[pix * C + c]);
      dl[pix * C + c] = v;
    }
    u16_t h, l;
    split_f32(v, h, l);
    dl_split[i] = h;
    dl_split[n + i] = l;
  }
}

// ---- models/blocks.py helpers (NHWC) -----------------------------------------------------------------------
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): src = max((o + 0.5)/2 - 0.5, 0)
__device__ __forceinline__ void bilin_src(int o, int in_size, int& i0, int& i1, float& lam) {
  float src = (o + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
  lam = src - (float)i0;
}

__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t n = (size_t)B * Ho * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho);
    const int b = (int)(r / Ho);
    int h0, h1, w0, w1; float lh, lw;
    bilin_src(oh, H, h0, h1, lh);
    bilin_src(ow, W, w0, w1, lw);
    const float* xb = x + (size_t)b * H * W * C + c;
    const float v00 = xb[((size_t)h0 * W + w0) * C], v01 = xb[((size_t)h0 * W + w1) * C];
    const float v10 = xb[((size_t)h1 * W + w0) * C], v11 = xb[((size_t)h1 * W + w1) * C];
    y[i] = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
  }
}

// gather form of the adjoint (no atomics): dx[h][w] = sum over the <= 5x5 output pixels whose stencil touches (h, w)
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t n = (size_t)B * H * W * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    const float* db = dy + (size_t)b * Ho * Wo * C + c;
    float acc = 0.f;
    for (int oh = max(0, 2 * h - 2); oh <= min(Ho - 1, 2 * h + 2); ++oh) {
      int h0, h1; float lh;
      bilin_src(oh, H, h0, h1, lh);
      const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh == 0.f) continue;
      for (int ow = max(0, 2 * w - 2); ow <= min(Wo - 1, 2 * w + 2); ++ow) {
        int w0, w1; float lw;
        bilin_src(ow, W, w0, w1, lw);
        const float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
        if (ww != 0.f) acc += wh * ww * db[((size_t)oh * Wo + ow) * C];
      }
    }
    dx[i] = acc;
  }
}

// AddCoords (models/blocks.py:97-112): out[..., :C] = x, out[..., C] = column index, out[..., C+1] = row index
__global__ void add_coords_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C, int normalize) {
  const int Co = C + 2;
  const size_t n = (size_t)B * H * W * Co;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Co);
    const size_t pix = i / Co;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    float v;
    if (c < C) v = x[pix * C + c];
    else if (c == C) v = normalize ? ((float)w / (float)W - 0.5f) / 0.5f : (float)w;
    else v = normalize ? ((float)h / (float)H - 0.5f) / 0.5f : (float)h;
    out[i] = v;
  }
}

// out[p][c] = in[p][c] for c < Cout (the gradient of AddCoords / a channel slice)
__global__ void slice_channels_kernel(const float* __restrict__ in, float* __restrict__ out, size_t npix, int Cin, int Cout) {
  const size_t n = npix * Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / Cout;
    out[i] = in[pix * Cin + (i - pix * Cout)];
  }
}

// torch.optim.Adam (single-tensor form): m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float one_minus_b1, float b2,
                                                   float one_minus_b2, float eps, float step_size, float bc2_sqrt,
                                                   float grad_scale) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto upd = [&](vp_f32x4& pv, const vp_f32x4& gv, vp_f32x4& mv, vp_f32x4& vv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gv[j] * grad_scale;
      mv[j] = mv[j] + one_minus_b1 * (gr - mv[j]);
      vv[j] = vv[j] * b2 + one_minus_b2 * gr * gr;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pv[j] = pv[j] - step_size * (mv[j] / denom);
    }
  };
  // two independent 16-B quads per thread and iteration: 8 loads in flight before the first use; the moments
  // are streamed with non-temporal accesses (touched once per step), p and g stay cacheable for their consumers
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const size_t a = i * 4, b = (i + stride) * 4;
    vp_f32x4 pa = *reinterpret_cast<vp_f32x4*>(p + a), pb = *reinterpret_cast<vp_f32x4*>(p + b);
    vp_f32x4 ga = *reinterpret_cast<const vp_f32x4*>(g + a), gb = *reinterpret_cast<const vp_f32x4*>(g + b);
    vp_f32x4 ma = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(m + a));
    vp_f32x4 mb = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(m + b));
    vp_f32x4 va = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(v + a));
    vp_f32x4 vb = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(v + b));
    upd(pa, ga, ma, va);
    upd(pb, gb, mb, vb);
    *reinterpret_cast<vp_f32x4*>(p + a) = pa;
    *reinterpret_cast<vp_f32x4*>(p + b) = pb;
    __builtin_nontemporal_store(ma, reinterpret_cast<vp_f32x4*>(m + a));
    __builtin_nontemporal_store(mb, reinterpret_cast<vp_f32x4*>(m + b));
    __builtin_nontemporal_store(va, reinterpret_cast<vp_f32x4*>(v + a));
    __builtin_nontemporal_store(vb, reinterpret_cast<vp_f32x4*>(v + b));
  }
  if (i < n4) {
    const size_t a = i * 4;
    vp_f32x4 pa = *reinterpret_cast<vp_f32x4*>(p + a);
    vp_f32x4 ga = *reinterpret_cast<const vp_f32x4*>(g + a);
    vp_f32x4 ma = *reinterpret_cast<vp_f32x4*>(m + a), va = *reinterpret_cast<vp_f32x4*>(v + a);
    upd(pa, ga, ma, va);
    *reinterpret_cast<vp_f32x4*>(p + a) = pa;
    *reinterpret_cast<vp_f32x4*>(m + a) = ma;
    *reinterpret_cast<vp_f32x4*>(v + a) = va;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float gr = g[i] * grad_scale;
    const float mm = m[i] + one_minus_b1 * (gr - m[i]);
    const float vv = v[i] * b2 + one_minus_b2 * gr * gr;
    m[i] = mm; v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

// Adam update of a weight matrix p[R][Cn] whose gradient is a sum of K outer products, g = grad_scale * A^T B with A [K][R] and
// B [K][Cn] (a Linear layer's weight gradient: A = the output gradients, B = the inputs of the K samples).  The gradient is
// contracted in registers and consumed at once: for the encoder's first dense layer (1024 x 32768, 63 % of the model's parameters,
// K = 32 images) that saves writing and re-reading a 134-MB gradient matrix per step.  One workgroup = 16 rows x 1024 columns,
// one thread = 16 rows x 4 columns (64 accumulators); per k it loads one 16-B quad of B and 16 wave-uniform values of A.
template <int AO_TI>
__global__ void __launch_bounds__(256) adam_outer_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ A, const float* __restrict__ Bm, int K, int R, int Cn,
                                                         float one_minus_b1, float b2, float one_minus_b2, float eps, float step_size,
                                                         float bc2_sqrt, float grad_scale) {
  const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int i0 = blockIdx.y * AO_TI;
  if (j >= Cn) return;
  float acc[AO_TI][4];
#pragma unroll
  for (int i = 0; i < AO_TI; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
  const bool full = i0 + AO_TI <= R;
  auto kstep = [&](int k) {
    const vp_f32x4 b4 = *reinterpret_cast<const vp_f32x4*>(Bm + (size_t)k * Cn + j);
    const float* ak = A + (size_t)k * R + i0;
    float a[AO_TI];
    if (full) {
#pragma unroll
      for (int q = 0; q < AO_TI / 4; ++q) {
        const vp_f32x4 t = *reinterpret_cast<const vp_f32x4*>(ak + 4 * q);
        a[4 * q] = t[0]; a[4 * q + 1] = t[1]; a[4 * q + 2] = t[2]; a[4 * q + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < AO_TI; ++i) a[i] = i0 + i < R ? ak[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < AO_TI; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = fmaf(a[i], b4[c], acc[i][c]);
  };
  int k = 0;
  for (; k + 4 <= K; k += 4) {      // four k per trip: their loads are issued together
    kstep(k); kstep(k + 1); kstep(k + 2); kstep(k + 3);
  }
  for (; k < K; ++k) kstep(k);
  auto upd = [&](vp_f32x4& pv, vp_f32x4& mv, vp_f32x4& vv, const float (&g4)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gr = g4[c] * grad_scale;
      mv[c] = mv[c] + one_minus_b1 * (gr - mv[c]);
      vv[c] = vv[c] * b2 + one_minus_b2 * gr * gr;
      const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
      pv[c] = pv[c] - step_size * (mv[c] / denom);
    }
  };
  // rows in groups of four: 12 independent 16-B loads in flight before the first use (every row index is a compile-time
  // constant: a runtime index would send the accumulators to scratch)
#pragma unroll
  for (int i4 = 0; i4 < AO_TI; i4 += 4) {
    vp_f32x4 pv[4], mv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = i0 + i4 + u;
      const size_t o = (size_t)(r < R ? r : R - 1) * Cn + j;         // (rows past the end: clamped load, no store)
      pv[u] = *reinterpret_cast<vp_f32x4*>(p + o);
      mv[u] = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(m + o));
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<vp_f32x4*>(v + o));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = i0 + i4 + u;
      if (r >= R) continue;
      const size_t o = (size_t)r * Cn + j;
      upd(pv[u], mv[u], vv[u], acc[i4 + u]);
      *reinterpret_cast<vp_f32x4*>(p + o) = pv[u];
      __builtin_nontemporal_store(mv[u], reinterpret_cast<vp_f32x4*>(m + o));
      __builtin_nontemporal_store(vv[u], reinterpret_cast<vp_f32x4*>(v + o));
    }
  }
}


// torch.optim.RMSprop: sq = alpha*sq + (1-alpha)*g*g; p -= lr * g / (sqrt(sq) + eps)
__global__ void __launch_bounds__(256) rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                      size_t n, float lr, float alpha, float one_minus_alpha, float eps,
                                                      float grad_scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gr = g[i] * grad_scale;
    const float s = sq[i] * alpha + one_minus_alpha * gr * gr;
    sq[i] = s;
    p[i] = p[i] - lr * (gr / (sqrtf(s) + eps));
  }
}

}  // namespace vp

using namespace vp;

extern "C" {

int vp_abi_version(void) { return 1; }
const char* vp_last_error(void) { return err_buf(); }

int vp_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, vp_stream stream) {
  return launch_transpose(in, out, B, C, H * W, (hipStream_t)stream, "vp_nchw_to_nhwc_f32");
}
int vp_nchw_to_nhwc_split_f32(const float* in, float* out, void* out_split, int B, int C, int H, int W, vp_stream stream) {
  return launch_transpose(in, out, B, C, H * W, (hipStream_t)stream, "vp_nchw_to_nhwc_split_f32", out_split);
}
int vp_nchw_to_nhwc_split_fmt_f32(const float* in, float* out, void* out_split, int B, int C, int H, int W, int fmt, vp_stream stream) {
  return launch_transpose(in, out, B, C, H * W, (hipStream_t)stream, "vp_nchw_to_nhwc_split_fmt_f32", out_split, fmt);
}
int vp_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, vp_stream stream) {
  return launch_transpose(in, out, B, H * W, C, (hipStream_t)stream, "vp_nhwc_to_nchw_f32");
}

int vp_latent_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, float* kl, int B, int Z, vp_stream stream) {
  VP_REQUIRE(mu && logvar && eps && z && B > 0 && Z > 0, "vp_latent_fwd_f32: bad arguments");
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mu, logvar, eps, z, kl, Z);
  return check_launch("vp_latent_fwd_f32");
}

int vp_latent_bwd_f32(const float* mu, const float* logvar, const float* eps, const float* dz, const float* gkl,
                      float gkl_scalar, float* dmu, float* dlogvar, int B, int Z, vp_stream stream) {
  VP_REQUIRE(mu && logvar && eps && dmu && dlogvar && B > 0 && Z > 0, "vp_latent_bwd_f32: bad arguments");
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(grid_for((size_t)B * Z, 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, dz,
                     gkl, gkl_scalar, dmu, dlogvar, B, Z);
  return check_launch("vp_latent_bwd_f32");
}

size_t vp_reduce_workspace_bytes(size_t n) { return (size_t)reduce_blocks(n) * sizeof(float); }

int vp_bce_sum_f32(const float* p, const float* t, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(p && t && out && ws && n > 0, "vp_bce_sum_f32: bad arguments");
  const unsigned nb = reduce_blocks(n);
  if (ws_bytes < nb * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_bce_sum_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((reduce_partial_kernel<0>), dim3(nb), dim3(256), 0, s, p, t, n, (float*)ws);
  int rc = check_launch("vp_bce_sum_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, (int)nb, out);
  return check_launch("vp_bce_sum_f32(final)");
}

int vp_vae_loss_f32(const float* x_tilde, const float* x, size_t n, const float* kl, int B, float* recon, float* kl_sum, float* loss,
                    float loss_scale, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x_tilde && x && kl && recon && kl_sum && loss && ws && n > 0 && B > 0, "vp_vae_loss_f32: bad arguments");
  const unsigned nb = reduce_blocks(n);
  if (ws_bytes < nb * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_vae_loss_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((reduce_partial_kernel<0>), dim3(nb), dim3(256), 0, s, x_tilde, x, n, (float*)ws);
  int rc = check_launch("vp_vae_loss_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(vae_loss_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, (int)nb, kl, B, recon, kl_sum, loss, loss_scale);
  return check_launch("vp_vae_loss_f32(final)");
}

int vp_add_f32(const float* a, const float* b, float* out, size_t n, vp_stream stream) {
  VP_REQUIRE(a && b && out && n > 0, "vp_add_f32: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return check_launch("vp_add_f32");
}

int vp_sum_f32(const float* x, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && out && ws && n > 0, "vp_sum_f32: bad arguments");
  const unsigned nb = reduce_blocks(n);
  if (ws_bytes < nb * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_sum_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((reduce_partial_kernel<1>), dim3(nb), dim3(256), 0, s, x, (const float*)nullptr, n, (float*)ws);
  int rc = check_launch("vp_sum_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, (int)nb, out);
  return check_launch("vp_sum_f32(final)");
}

int vp_global_avgpool_fwd_f32(const float* x_nhwc, float* out, int B, int HW, int C, vp_stream stream) {
  VP_REQUIRE(x_nhwc && out && B > 0 && HW > 0 && C > 0, "vp_global_avgpool_fwd_f32: bad arguments");
  hipLaunchKernelGGL(global_avgpool_fwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x_nhwc, out, HW, C);
  return check_launch("vp_global_avgpool_fwd_f32");
}

int vp_global_avgpool_bwd_f32(const float* dy, float* dx_nhwc, int B, int HW, int C, vp_stream stream) {
  VP_REQUIRE(dy && dx_nhwc && B > 0 && HW > 0 && C > 0, "vp_global_avgpool_bwd_f32: bad arguments");
  const size_t total = (size_t)B * HW * C;
  hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx_nhwc, total, HW, C);
  return check_launch("vp_global_avgpool_bwd_f32");
}

int vp_softmax_rows_fwd_f32(const float* x, float* y, int R, int n, vp_stream stream) {
  VP_REQUIRE(x && y && R > 0 && n > 0, "vp_softmax_rows_fwd_f32: bad arguments");
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, R, n);
  return check_launch("vp_softmax_rows_fwd_f32");
}

int vp_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, int R, int n, vp_stream stream) {
  VP_REQUIRE(y && dy && dx && R > 0 && n > 0, "vp_softmax_rows_bwd_f32: bad arguments");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, dy, dx, R, n);
  return check_launch("vp_softmax_rows_bwd_f32");
}

int vp_l1_mean_f32(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(a && b && out && ws && n > 0, "vp_l1_mean_f32: bad arguments");
  const unsigned nb = reduce_blocks(n);
  if (ws_bytes < nb * sizeof(double)) return fail(VP_ERR_WORKSPACE, "vp_l1_mean_f32: workspace too small (use 2 * vp_reduce_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, (double*)ws, n);
  int rc = check_launch("vp_l1_mean_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, n, out);
  return check_launch("vp_l1_mean_f32(final)");
}

int vp_l1_mean_bwd_f32(const float* a, const float* b, const float* gptr, float* da, float* db, size_t n, vp_stream stream) {
  VP_REQUIRE(a && b && (da || db) && n > 0, "vp_l1_mean_bwd_f32: bad arguments");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, gptr, da, db, n);
  return check_launch("vp_l1_mean_bwd_f32");
}

size_t vp_be_loss_workspace_bytes(int B, int n) {
  const int nchunk = (n + BE_CHUNK - 1) / BE_CHUNK;
  return (size_t)B * nchunk * 4 * sizeof(double);
}

int vp_be_loss_fwd_f32(const float* logits, const float* targets, float* loss, float* sums, int B, int n, float bce_weight,
                       float smooth, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(logits && targets && loss && sums && ws && B > 0 && n > 0, "vp_be_loss_fwd_f32: bad arguments");
  if (ws_bytes < vp_be_loss_workspace_bytes(B, n)) return fail(VP_ERR_WORKSPACE, "vp_be_loss_fwd_f32: workspace too small");
  const int nchunk = (n + BE_CHUNK - 1) / BE_CHUNK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((be_loss_partial_kernel<false>), dim3(nchunk, B), dim3(256), 0, s, logits, targets, (double*)ws, n, nchunk);
  int rc = check_launch("vp_be_loss_fwd_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(be_loss_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, sums, loss, B, n, nchunk, bce_weight, smooth);
  return check_launch("vp_be_loss_fwd_f32(final)");
}

int vp_be_loss_bwd_f32(const float* logits, const float* targets, const float* sums, const float* gptr, float* dlogits, int B,
                       int n, float bce_weight, float smooth, vp_stream stream) {
  VP_REQUIRE(logits && targets && sums && dlogits && B > 0 && n > 0, "vp_be_loss_bwd_f32: bad arguments");
  hipLaunchKernelGGL((be_loss_bwd_kernel<false>), dim3(grid_for((size_t)n, 256, 256), B), dim3(256), 0, (hipStream_t)stream, logits,
                     targets, sums, gptr, dlogits, B, n, bce_weight, smooth);
  return check_launch("vp_be_loss_bwd_f32");
}

int vp_dice_loss_fwd_f32(const float* probs, const float* targets, float* loss, float* sums, int B, int n, float smooth, void* ws,
                         size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(probs && targets && loss && sums && ws && B > 0 && n > 0, "vp_dice_loss_fwd_f32: bad arguments");
  if (ws_bytes < vp_be_loss_workspace_bytes(B, n)) return fail(VP_ERR_WORKSPACE, "vp_dice_loss_fwd_f32: workspace too small");
  const int nchunk = (n + BE_CHUNK - 1) / BE_CHUNK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((be_loss_partial_kernel<true>), dim3(nchunk, B), dim3(256), 0, s, probs, targets, (double*)ws, n, nchunk);
  int rc = check_launch("vp_dice_loss_fwd_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(be_loss_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, sums, loss, B, n, nchunk, 0.f, smooth);
  return check_launch("vp_dice_loss_fwd_f32(final)");
}

int vp_dice_loss_bwd_f32(const float* probs, const float* targets, const float* sums, const float* gptr, float* dprobs, int B, int n,
                         float smooth, vp_stream stream) {
  VP_REQUIRE(probs && targets && sums && dprobs && B > 0 && n > 0, "vp_dice_loss_bwd_f32: bad arguments");
  hipLaunchKernelGGL((be_loss_bwd_kernel<true>), dim3(grid_for((size_t)n, 256, 256), B), dim3(256), 0, (hipStream_t)stream, probs,
                     targets, sums, gptr, dprobs, B, n, 0.f, smooth);
  return check_launch("vp_dice_loss_bwd_f32");
}

int vp_half_sqdiff_f32(const float* a, const float* b, float* out, size_t n, vp_stream stream) {
  VP_REQUIRE(a && b && out && n > 0, "vp_half_sqdiff_f32: bad arguments");
  hipLaunchKernelGGL(half_sqdiff_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return check_launch("vp_half_sqdiff_f32");
}

int vp_half_sqdiff_rowsum_f32(const float* a, const float* b, float* out, int R, int n_per_row, vp_stream stream) {
  VP_REQUIRE(a && b && out && R > 0 && n_per_row > 0, "vp_half_sqdiff_rowsum_f32: bad arguments");
  hipLaunchKernelGGL(half_sqdiff_rowsum_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, a, b, out, n_per_row);
  return check_launch("vp_half_sqdiff_rowsum_f32");
}

int vp_half_sqdiff_bwd_f32(const float* a, const float* b, const float* g, float* da, float* db, int R, int n_per_row,
                           int g_per_row, vp_stream stream) {
  VP_REQUIRE(a && b && g && (da || db) && R > 0 && n_per_row > 0, "vp_half_sqdiff_bwd_f32: bad arguments");
  const size_t total = (size_t)R * n_per_row;
  hipLaunchKernelGGL(half_sqdiff_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, b, g, da, db, total,
                     n_per_row, g_per_row ? 1 : 0);
  return check_launch("vp_half_sqdiff_bwd_f32");
}

int vp_gan_head_f32(const float* logit, int B, float coef, float* p_out, float* sums, float* dlogit, vp_stream stream) {
  VP_REQUIRE(logit && B > 0 && (p_out || sums || dlogit), "vp_gan_head_f32: bad arguments");
  hipLaunchKernelGGL(gan_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logit, B, coef, p_out, sums, dlogit);
  return check_launch("vp_gan_head_f32");
}

int vp_smooth_l1_cat_f32(const float* targets, const float* a, const float* b, int B, int n1, int n2, float scale, float* loss,
                         float* da, float* db, vp_stream stream) {
  VP_REQUIRE(targets && a && B > 0 && n1 > 0 && n2 >= 0 && (b || n2 == 0) && (loss || da || db), "vp_smooth_l1_cat_f32: bad arguments");
  hipLaunchKernelGGL(smooth_l1_cat_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, targets, a, b, B, n1, n2, scale, loss, da, db);
  return check_launch("vp_smooth_l1_cat_f32");
}

int vp_bce_bwd_f32(const float* p, const float* t, const float* gptr, float gscale, float* dp, size_t n, vp_stream stream) {
  VP_REQUIRE(p && t && dp && n > 0, "vp_bce_bwd_f32: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, t, gptr, gscale, dp, n);
  return check_launch("vp_bce_bwd_f32");
}

int vp_bce_sigmoid_bwd_f32(const float* p, const float* t, float gscale, float* dlogit, size_t n, vp_stream stream) {
  VP_REQUIRE(p && t && dlogit && n > 0, "vp_bce_sigmoid_bwd_f32: bad arguments");
  hipLaunchKernelGGL(bce_sigmoid_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, t, gscale, dlogit, n);
  return check_launch("vp_bce_sigmoid_bwd_f32");
}

int vp_bce_sigmoid_bwd_pad_split_f32(const float* p, const float* t, float gscale, float* dlogit, void* dlogit_split,
                                     size_t npix, int C, int Cpad, vp_stream stream) {
  VP_REQUIRE(p && t && dlogit && dlogit_split && npix > 0 && C > 0 && Cpad >= C && Cpad % 8 == 0,
             "vp_bce_sigmoid_bwd_pad_split_f32: bad arguments");
  hipLaunchKernelGGL(bce_sigmoid_bwd_pad_kernel, dim3(grid_for(npix * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, p, t, gscale,
                     dlogit, (u16_t*)dlogit_split, npix, C, Cpad);
  return check_launch("vp_bce_sigmoid_bwd_pad_split_f32");
}

int vp_upsample2x_bilinear_fwd_f32(const float* x, float* y, int B, int H, int W, int C, vp_stream stream) {
  VP_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "vp_upsample2x_bilinear_fwd_f32: bad arguments");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((size_t)B * 4 * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
  return check_launch("vp_upsample2x_bilinear_fwd_f32");
}

int vp_upsample2x_bilinear_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, vp_stream stream) {
  VP_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "vp_upsample2x_bilinear_bwd_f32: bad arguments");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((size_t)B * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C);
  return check_launch("vp_upsample2x_bilinear_bwd_f32");
}

int vp_add_coords_f32(const float* x, float* out, int B, int H, int W, int C, int normalize, vp_stream stream) {
  VP_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0, "vp_add_coords_f32: bad arguments");
  hipLaunchKernelGGL(add_coords_kernel, dim3(grid_for((size_t)B * H * W * (C + 2), 256)), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, normalize);
  return check_launch("vp_add_coords_f32");
}

int vp_slice_channels_f32(const float* in, float* out, size_t npix, int Cin, int Cout, vp_stream stream) {
  VP_REQUIRE(in && out && npix > 0 && Cout > 0 && Cout <= Cin, "vp_slice_channels_f32: bad arguments");
  hipLaunchKernelGGL(slice_channels_kernel, dim3(grid_for(npix * Cout, 256)), dim3(256), 0, (hipStream_t)stream, in, out, npix, Cin, Cout);
  return check_launch("vp_slice_channels_f32");
}

int vp_adam_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                float grad_scale, vp_stream stream) {
  VP_REQUIRE(p && g && m && v && n > 0 && step >= 1, "vp_adam_f32: bad arguments");
  VP_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "vp_adam_f32: arena must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int cap = 256 * 64;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256, cap)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     1.f - beta1, beta2, 1.f - beta2, eps, step_size, bc2_sqrt, grad_scale);
  return check_launch("vp_adam_f32");
}

int vp_adam_outer_f32(float* p, float* m, float* v, const float* A, const float* Bm, int K, int R, int Cn, float lr, float beta1,
                      float beta2, float eps, int step, float grad_scale, vp_stream stream) {
  VP_REQUIRE(p && m && v && A && Bm && K > 0 && R > 0 && Cn > 0 && step >= 1, "vp_adam_outer_f32: bad arguments");
  VP_REQUIRE(Cn % 4 == 0 && R % 4 == 0, "vp_adam_outer_f32: R and Cn must be multiples of 4");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int ti = 8;      // rows per thread: 8 (142 us on the 1024 x 32768 layer) | 16 (150 us)
  const int TI = ti == 8 ? 8 : 16;
  const dim3 grid((unsigned)((Cn / 4 + 255) / 256), (unsigned)((R + TI - 1) / TI));
  VP_REQUIRE(grid.y <= 65535, "vp_adam_outer_f32: too many rows");
  if (TI == 8)
    hipLaunchKernelGGL(adam_outer_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p, m, v, A, Bm, K, R, Cn, 1.f - beta1, beta2, 1.f - beta2,
                       eps, step_size, bc2_sqrt, grad_scale);
  else
    hipLaunchKernelGGL(adam_outer_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, p, m, v, A, Bm, K, R, Cn, 1.f - beta1, beta2, 1.f - beta2,
                       eps, step_size, bc2_sqrt, grad_scale);
  return check_launch("vp_adam_outer_f32");
}

int vp_rmsprop_f32(float* p, const float* g, float* sq, size_t n, float lr, float alpha, float eps, float grad_scale,
                   vp_stream stream) {
  VP_REQUIRE(p && g && sq && n > 0, "vp_rmsprop_f32: bad arguments");
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, p, g, sq, n, lr, alpha,
                     1.f - alpha, eps, grad_scale);
  return check_launch("vp_rmsprop_f32");
}
}

// ---- F.cross_entropy(logits, labels) (mean over rows): train_BE_GAN.py:135,159, train_BE_font.py:109 ---------------------------
// R rows of n logits, labels int64.  One workgroup; a thread owns whole rows (R = batch size, n = a handful of classes), the row
// losses are summed in a fixed order: bit-reproducible.  prob[r][j] = softmax (kept for the backward pass).
namespace vp {
__global__ void __launch_bounds__(256) cross_entropy_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ labels,
                                                                float* __restrict__ loss, float* __restrict__ prob, int R, int n) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float* xr = x + (size_t)r * n;
    float m = xr[0];
    for (int j = 1; j < n; ++j) m = fmaxf(m, xr[j]);
    float ssum = 0.f;
    for (int j = 0; j < n; ++j) ssum += __builtin_expf(xr[j] - m);
    const float inv = 1.f / ssum;
    for (int j = 0; j < n; ++j) prob[(size_t)r * n + j] = __builtin_expf(xr[j] - m) * inv;
    const long long lb = labels[r];
    const float xl = (lb >= 0 && lb < n) ? xr[lb] : 0.f;          // (an out-of-range label contributes lse only; the host checks)
    acc += (double)((m + __builtin_logf(ssum)) - xl);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += part[i];
    loss[0] = (float)(t / (double)R);
  }
}
__global__ void cross_entropy_bwd_kernel(const float* __restrict__ prob, const long long* __restrict__ labels, const float* __restrict__ g,
                                         float* __restrict__ dx, int R, int n) {
  const float gs = g[0] / (float)R;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), j = (int)(i - (size_t)r * n);
    dx[i] = gs * (prob[i] - (labels[r] == j ? 1.f : 0.f));
  }
}
}  // namespace vp

extern "C" {
int vp_cross_entropy_fwd_f32(const float* logits, const long long* labels, float* loss, float* prob, int R, int n, vp_stream stream) {
  VP_REQUIRE(logits && labels && loss && prob && R > 0 && n > 0, "vp_cross_entropy_fwd_f32: bad arguments");
  hipLaunchKernelGGL(vp::cross_entropy_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, loss, prob, R, n);
  return vp::check_launch("vp_cross_entropy_fwd_f32");
}
int vp_cross_entropy_bwd_f32(const float* prob, const long long* labels, const float* gptr, float* dlogits, int R, int n, vp_stream stream) {
  VP_REQUIRE(prob && labels && gptr && dlogits && R > 0 && n > 0, "vp_cross_entropy_bwd_f32: bad arguments");
  hipLaunchKernelGGL(vp::cross_entropy_bwd_kernel, dim3(vp::grid_for((size_t)R * n, 256)), dim3(256), 0, (hipStream_t)stream, prob, labels, gptr,
                     dlogits, R, n);
  return vp::check_launch("vp_cross_entropy_bwd_f32");
}
}
