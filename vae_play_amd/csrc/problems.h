// Implicit-GEMM problem descriptors for the 5x5 convolution families and the plain GEMMs.
//
// Every hot contraction of the VAE step is expressed as  C[m][n] = sum_k A(m,k) * B(n,k)
// where A/B are *accessors* over NHWC activations and packed weights -- no im2col buffer is
// ever materialised in HBM.  The same accessor code is compiled for the device (igemm.h, MFMA
// kernel) and for the host (tests/host_emul, a naive triple loop) so that the index math is
// validated on CPU against torch before it ever runs on a GPU.
//
// Vocabulary ("big"/"small"): a 5x5, pad-2 convolution with stride s relates a big image
// (Hb = s*Hs) and a small image (Hs) by  big[s*h - 2 + r] <-> small[h],  r in 0..4.
//   reference weight tensor  W[small_ch][big_ch][r][q]   (Conv2d: (Cout,Cin,5,5);
//                                                          ConvTranspose2d: (Cin,Cout,5,5))
//   F family ("gather")  : small = f(big)   Conv2d fwd, ConvTranspose2d dgrad
//   T family ("scatter") : big   = f(small) ConvTranspose2d fwd, Conv2d dgrad (phase-decomposed,
//                                           no zero insertion: 9/6/6/4 taps at stride 2)
//   W family ("wgrad")   : dW    = f(big, small)
// Reference: models/networks.py:14 (Conv2d k5 s2 p2), :38 (ConvTranspose2d k5 s2 p2 op1),
//            :100-103 (Conv2d k5 s1 p2 + bias + Sigmoid).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "env.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define VP_HD __host__ __device__ __forceinline__
#else
#define VP_HD inline
#endif

typedef float vp_f32x4 __attribute__((ext_vector_type(4)));

namespace vp {

constexpr int kTaps = 25;

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };

VP_HD vp_f32x4 zero4() { vp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// Zero page: out-of-range gathers (padding taps, rows past M, k past K) load 16 zero bytes from a small
// GLOBAL buffer (`zero` member of every descriptor, see vp_zero_page()) instead of being predicated or
// zero-selected after the load.  A select right after the load makes the compiler wait for the prefetch
// *before* the MFMAs of the current tile (the K-loop then pays the full memory latency every tile); with
// the zero page the first use of the loaded registers is the LDS write after the MFMAs.  The pointer
// must be a plain global pointer: selecting against a constant-address-space symbol turns the gathers
// into flat_load, whose lgkmcnt accounting stalls every LDS wait behind the prefetch.
// 16-byte load.  Callers guarantee 16-B alignment whenever they take the vector path (the
// host sets the vec flags from strides/channel counts), so the device gets one dwordx4 load.
VP_HD vp_f32x4 ld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const vp_f32x4*>(p);
#else
  vp_f32x4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
#endif
}

// Division by a loop-invariant divisor (Granlund-Montgomery): q = (t + ((n - t) >> sh1)) >> sh2 with
// t = mulhi(m, n).  gfx950 has no integer divider (a runtime `/` is ~40 VALU instructions); the
// loaders decompose k -> (tap, channel) and pixel -> (b, h, w) for every 16-byte gather, which cost
// more VALU issue time than the bf16 MFMAs of a K-tile before this was introduced.
struct FastDiv {
  uint32_t d, m, sh1, sh2;
  VP_HD uint32_t div(uint32_t n) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(m, n);
#else
    const uint32_t t = (uint32_t)(((uint64_t)m * n) >> 32);
#endif
    return (t + ((n - t) >> sh1)) >> sh2;
  }
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t L = 0;
  while ((1ull << L) < d) ++L;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
  f.sh1 = L < 1 ? L : 1;
  f.sh2 = L > 0 ? L - 1 : 0;
  return f;
}

// t / tw for the tap-column counts that occur (1..5): branch-free select of constant-divisor results
VP_HD int div_small(int t, int tw) {
  const int q3 = (int)(((unsigned)t * 43691u) >> 17);    // t / 3 for t < 98304
  const int q5 = (int)(((unsigned)t * 52429u) >> 18);    // t / 5 for t < 81920
  int q = t;
  q = tw == 2 ? (t >> 1) : q;
  q = tw == 3 ? q3 : q;
  q = tw == 4 ? (t >> 2) : q;
  q = tw == 5 ? q5 : q;
  return q;
}

struct ConvGeom {
  int B, Hs, Ws, Hb, Wb, Cs, Cb, stride;
  int ks, pad, nt;             // kernel size (1, 3 or 5), padding (ks-1)/2, taps ks*ks
  FastDiv dHW, dW, dCs, dCb;   // divisors Hs*Ws, Ws, Cs, Cb
};

// --------------------------------------------------------------------------------------------
// F family: out_small[b,hs,ws,n] = bias[n] + sum_{r,q,c} big[b, s*hs-2+r, s*ws-2+q, c] * wp0[n][r*5+q][c]
//   M = B*Hs*Ws, N = Cs, K = 25*Cb.  Both operands are "MK" (k contiguous).
// --------------------------------------------------------------------------------------------
struct ProbF {
  static constexpr bool A_KM = false, B_KM = false;
  const float* big;
  const float* w;     // packed P0: [Cs][25][Cb]
  const float* bias;  // nullable
  float* out;         // [B,Hs,Ws,Cs]
  const void* zero;   // >= 16 zero bytes in global memory
  ConvGeom g;
  int act;
  int M, N, K;
  int vec;  // Cb % 4 == 0

  struct ZCtx { int k_begin, k_end; };
  struct ARow { int pix_base, h0, w0, valid; };
  struct BRow { int off, valid; };

  VP_HD void z_setup(int, ZCtx& z) const { z.k_begin = 0; z.k_end = K; }

  VP_HD ARow a_row(int m, const ZCtx&) const {
    ARow r;
    r.valid = m < M;
    int mm = r.valid ? m : 0;
    int b = (int)g.dHW.div((uint32_t)mm);
    int rem = mm - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem);
    int ws = rem - hs * g.Ws;
    r.pix_base = b * g.Hb * g.Wb;
    r.h0 = g.stride * hs - g.pad;
    r.w0 = g.stride * ws - g.pad;
    return r;
  }
  VP_HD float a_elem(const ARow& r, int k) const {
    if (!r.valid || k >= K) return 0.f;
    int tap = (int)g.dCb.div((uint32_t)k), c = k - tap * g.Cb;
    int rr = div_small(tap, g.ks), qq = tap - rr * g.ks;
    int h = r.h0 + rr, w = r.w0 + qq;
    if (h < 0 || h >= g.Hb || w < 0 || w >= g.Wb) return 0.f;
    return big[(size_t)(r.pix_base + h * g.Wb + w) * g.Cb + c];
  }
  VP_HD vp_f32x4 a_load(const ARow& r, int k, const ZCtx&) const {
    if (vec) {   // branch-free: always load (address clamped to element 0), then select
      int tap = (int)g.dCb.div((uint32_t)k), c = k - tap * g.Cb;
      int rr = div_small(tap, g.ks), qq = tap - rr * g.ks;
      int h = r.h0 + rr, w = r.w0 + qq;
      const bool ok = r.valid && k < K && h >= 0 && h < g.Hb && w >= 0 && w < g.Wb;
      return ld4(ok ? big + (size_t)(r.pix_base + h * g.Wb + w) * g.Cb + c : reinterpret_cast<const float*>(zero));
    }
    vp_f32x4 v = {a_elem(r, k), a_elem(r, k + 1), a_elem(r, k + 2), a_elem(r, k + 3)};
    return v;
  }
  VP_HD BRow b_row(int n, const ZCtx&) const {
    BRow r;
    r.valid = n < N;
    r.off = (r.valid ? n : 0) * K;
    return r;
  }
  VP_HD vp_f32x4 b_load(const BRow& r, int k, const ZCtx&) const {
    if (vec) {
      const bool ok = r.valid && k + 3 < K;
      return ld4(ok ? w + (size_t)r.off + k : reinterpret_cast<const float*>(zero));
    }
    if (!r.valid) return zero4();
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (k + j < K) v[j] = w[(size_t)r.off + k + j];
    return v;
  }
  VP_HD void store(int m, int n, float v, const ZCtx&) const {
    if (m >= M || n >= N) return;
    if (bias) v += bias[n];
    if (act == ACT_SIGMOID) v = 1.f / (1.f + __builtin_expf(-v));
    out[(size_t)m * N + n] = v;
  }
};

// --------------------------------------------------------------------------------------------
// T family: out_big[b, s*q+ph, s*p+pw, n] = sum_{r',q',c} small[b, q+d0-r', p+d0-q', c] * wp1[n][(ph+s*r')*5+(pw+s*q')][c]
//   one GEMM per output phase z = ph*s+pw; taps_h = ceil((5-ph)/s); d0 = 2/s.
//   M = B*Hs*Ws, N = Cb, K = taps_h*taps_w*Cs.
// --------------------------------------------------------------------------------------------
struct ProbT {
  static constexpr bool A_KM = false, B_KM = false;
  const float* small;
  const float* w;  // packed P1: [Cb][25][Cs]
  float* out;      // [B,Hb,Wb,Cb]
  const void* zero;
  ConvGeom g;
  int M, N;
  int vec;  // Cs % 4 == 0

  // phase (ph, pw) of the output: taps r = r0 + s*r' (r0 = (ph+pad) mod s), source row = q + bh - r'
  struct ZCtx { int k_begin, k_end, ph, pw, th, tw, r0h, r0w, bh, bw; };
  struct ARow { int pix_base, q, p, valid; };
  struct BRow { int off, valid; };

  VP_HD void z_setup(int zi, ZCtx& z) const {
    int s = g.stride;
    z.ph = zi / s;
    z.pw = zi - z.ph * s;
    z.r0h = (z.ph + g.pad) % s;
    z.r0w = (z.pw + g.pad) % s;
    z.th = z.r0h < g.ks ? (g.ks - z.r0h + s - 1) / s : 0;
    z.tw = z.r0w < g.ks ? (g.ks - z.r0w + s - 1) / s : 0;
    z.bh = (z.ph + g.pad - z.r0h) / s;
    z.bw = (z.pw + g.pad - z.r0w) / s;
    z.k_begin = 0;
    z.k_end = z.th * z.tw * g.Cs;
  }
  VP_HD ARow a_row(int m, const ZCtx&) const {
    ARow r;
    r.valid = m < M;
    int mm = r.valid ? m : 0;
    int b = (int)g.dHW.div((uint32_t)mm);
    int rem = mm - b * (g.Hs * g.Ws);
    r.q = (int)g.dW.div((uint32_t)rem);
    r.p = rem - r.q * g.Ws;
    r.pix_base = b * g.Hs * g.Ws;
    return r;
  }
  VP_HD float a_elem(const ARow& r, int k, const ZCtx& z) const {
    if (!r.valid || k >= z.k_end) return 0.f;
    int t = (int)g.dCs.div((uint32_t)k), c = k - t * g.Cs;
    int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    int h = r.q + z.bh - rp, w_ = r.p + z.bw - qp;
    if (h < 0 || h >= g.Hs || w_ < 0 || w_ >= g.Ws) return 0.f;
    return small[(size_t)(r.pix_base + h * g.Ws + w_) * g.Cs + c];
  }
  VP_HD vp_f32x4 a_load(const ARow& r, int k, const ZCtx& z) const {
    if (vec) {
      int t = (int)g.dCs.div((uint32_t)k), c = k - t * g.Cs;
      int rp = div_small(t, z.tw), qp = t - rp * z.tw;
      int h = r.q + z.bh - rp, w_ = r.p + z.bw - qp;
      const bool ok = r.valid && k < z.k_end && h >= 0 && h < g.Hs && w_ >= 0 && w_ < g.Ws;
      return ld4(ok ? small + (size_t)(r.pix_base + h * g.Ws + w_) * g.Cs + c : reinterpret_cast<const float*>(zero));
    }
    vp_f32x4 v = {a_elem(r, k, z), a_elem(r, k + 1, z), a_elem(r, k + 2, z), a_elem(r, k + 3, z)};
    return v;
  }
  VP_HD BRow b_row(int n, const ZCtx&) const {
    BRow r;
    r.valid = n < N;
    r.off = (r.valid ? n : 0) * g.nt * g.Cs;
    return r;
  }
  VP_HD float b_elem(const BRow& r, int k, const ZCtx& z) const {
    if (!r.valid || k >= z.k_end) return 0.f;
    int t = (int)g.dCs.div((uint32_t)k), c = k - t * g.Cs;
    int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    int tap = (z.r0h + g.stride * rp) * g.ks + (z.r0w + g.stride * qp);
    return w[(size_t)r.off + tap * g.Cs + c];
  }
  VP_HD vp_f32x4 b_load(const BRow& r, int k, const ZCtx& z) const {
    if (vec) {
      int t = (int)g.dCs.div((uint32_t)k), c = k - t * g.Cs;
      int rp = div_small(t, z.tw), qp = t - rp * z.tw;
      int tap = (z.r0h + g.stride * rp) * g.ks + (z.r0w + g.stride * qp);
      const bool ok = r.valid && k < z.k_end;
      return ld4(ok ? w + (size_t)r.off + tap * g.Cs + c : reinterpret_cast<const float*>(zero));
    }
    vp_f32x4 v = {b_elem(r, k, z), b_elem(r, k + 1, z), b_elem(r, k + 2, z), b_elem(r, k + 3, z)};
    return v;
  }
  VP_HD void store(int m, int n, float v, const ZCtx& z) const {
    if (m >= M || n >= N) return;
    int b = (int)g.dHW.div((uint32_t)m);
    int rem = m - b * (g.Hs * g.Ws);
    int q = (int)g.dW.div((uint32_t)rem), p = rem - q * g.Ws;
    int oh = g.stride * q + z.ph, ow = g.stride * p + z.pw;
    if (oh >= g.Hb || ow >= g.Wb) return;   // odd big sizes: the last phase row/column does not exist
    out[((size_t)(b * g.Hb + oh) * g.Wb + ow) * g.Cb + n] = v;
  }
};

// --------------------------------------------------------------------------------------------
// W family: slab[split][tap][m=cs][n=cb] = sum_{pixels in split} small[pix][cs] * big[shift_tap(pix)][cb]
//   z = tap*nsplit + split.  M = Cs, N = Cb, K = B*Hs*Ws.  Both operands "KM" (k = pixel is the slow index).
//   A second pass (wgrad_reduce) sums the splits and writes the reference layout dW[cs][cb][tap].
// --------------------------------------------------------------------------------------------
struct ProbW {
  static constexpr bool A_KM = true, B_KM = true;
  const float* big;
  const float* small;
  float* slab;  // [nsplit][25][Cs][Cb]
  ConvGeom g;
  int M, N, K;
  int nsplit, k_per_split;
  int vec_a, vec_b;  // Cs % 4 == 0, Cb % 4 == 0

  struct ZCtx { int k_begin, k_end, rr, qq, tap, split; };

  VP_HD void z_setup(int zi, ZCtx& z) const {
    z.tap = zi / nsplit;
    z.split = zi - z.tap * nsplit;
    z.rr = div_small(z.tap, g.ks);
    z.qq = z.tap - z.rr * g.ks;
    z.k_begin = z.split * k_per_split;
    int e = z.k_begin + k_per_split;
    z.k_end = e < K ? e : K;
  }
  // A(k = pixel, m .. m+3) = small[pixel][m..m+3]
  VP_HD vp_f32x4 a_load_km(int k, int m, const ZCtx& z) const {
    if (k >= z.k_end) return zero4();
    const float* p = small + (size_t)k * g.Cs;
    if (vec_a && m + 3 < M) return ld4(p + m);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (m + j < M) v[j] = p[m + j];
    return v;
  }
  // B(k = pixel, n .. n+3) = big[b, s*hs-2+r, s*ws-2+q][n..n+3]
  VP_HD vp_f32x4 b_load_km(int k, int n, const ZCtx& z) const {
    if (k >= z.k_end) return zero4();
    int b = (int)g.dHW.div((uint32_t)k);
    int rem = k - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    int h = g.stride * hs - g.pad + z.rr, w_ = g.stride * ws - g.pad + z.qq;
    if (h < 0 || h >= g.Hb || w_ < 0 || w_ >= g.Wb) return zero4();
    const float* p = big + ((size_t)(b * g.Hb + h) * g.Wb + w_) * g.Cb;
    if (vec_b && n + 3 < N) return ld4(p + n);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (n + j < N) v[j] = p[n + j];
    return v;
  }
  VP_HD void store(int m, int n, float v, const ZCtx& z) const {
    if (m >= M || n >= N) return;
    slab[(((size_t)z.split * g.nt + z.tap) * M + m) * N + n] = v;
  }
};

// --------------------------------------------------------------------------------------------
// Plain GEMM  C[m][n] = sum_k A(m,k) B(n,k) (+bias[n]) with arbitrary element strides.
//   template flags pick the LDS image per operand: MK (k contiguous in memory) or KM (m contiguous).
//   Linear fwd   y = x W^T      : A = x  (MK), B = W  (MK)
//   Linear dgrad dx = dy W      : A = dy (MK), B = W  (KM: B(n,k) = W[k][n])
//   Linear wgrad dW = dy^T x    : A = dy (KM: A(m,k) = dy[k][m]), B = x (KM)
//   Reference: nn.Linear at models/networks.py:65,69-70,88.
//   Split-K: z = split; partials go to slab[split][M][N], summed (+bias) by gemm_reduce.
// --------------------------------------------------------------------------------------------
template <bool AKM, bool BKM>
struct ProbG {
  static constexpr bool A_KM = AKM, B_KM = BKM;
  const float* A;
  const float* Bm;
  float* C;           // direct output (nsplit == 1) or slab base
  const float* bias;  // only applied when nsplit == 1
  int M, N, K;
  long sam, sak, sbn, sbk;  // element strides
  int ldc;
  int nsplit, k_per_split;
  int vec_a, vec_b;  // contiguous-dim stride == 1, other stride % 4 == 0, 16B-aligned base

  struct ZCtx { int k_begin, k_end, split; };
  struct ARow { long off; int valid; };
  struct BRow { long off; int valid; };

  VP_HD void z_setup(int zi, ZCtx& z) const {
    z.split = zi;
    z.k_begin = zi * k_per_split;
    int e = z.k_begin + k_per_split;
    z.k_end = e < K ? e : K;
  }
  VP_HD ARow a_row(int m, const ZCtx&) const { ARow r; r.valid = m < M; r.off = (long)(r.valid ? m : 0) * sam; return r; }
  VP_HD BRow b_row(int n, const ZCtx&) const { BRow r; r.valid = n < N; r.off = (long)(r.valid ? n : 0) * sbn; return r; }
  VP_HD vp_f32x4 a_load(const ARow& r, int k, const ZCtx& z) const {
    if (!r.valid || k >= z.k_end) return zero4();
    if (vec_a && k + 3 < z.k_end) return ld4(A + r.off + k);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (k + j < z.k_end) v[j] = A[r.off + (long)(k + j) * sak];
    return v;
  }
  VP_HD vp_f32x4 b_load(const BRow& r, int k, const ZCtx& z) const {
    if (!r.valid || k >= z.k_end) return zero4();
    if (vec_b && k + 3 < z.k_end) return ld4(Bm + r.off + k);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (k + j < z.k_end) v[j] = Bm[r.off + (long)(k + j) * sbk];
    return v;
  }
  VP_HD vp_f32x4 a_load_km(int k, int m, const ZCtx& z) const {
    if (k >= z.k_end) return zero4();
    const float* p = A + (long)k * sak;
    if (vec_a && m + 3 < M) return ld4(p + m);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (m + j < M) v[j] = p[(long)(m + j) * sam];
    return v;
  }
  VP_HD vp_f32x4 b_load_km(int k, int n, const ZCtx& z) const {
    if (k >= z.k_end) return zero4();
    const float* p = Bm + (long)k * sbk;
    if (vec_b && n + 3 < N) return ld4(p + n);
    vp_f32x4 v = zero4();
    for (int j = 0; j < 4; ++j)
      if (n + j < N) v[j] = p[(long)(n + j) * sbn];
    return v;
  }
  VP_HD void store(int m, int n, float v, const ZCtx& z) const {
    if (m >= M || n >= N) return;
    if (nsplit == 1) {
      if (bias) v += bias[n];
      C[(size_t)m * ldc + n] = v;
    } else {
      C[((size_t)z.split * M + m) * N + n] = v;
    }
  }
};

}  // namespace vp

// ---------------------------------------------------------------------------------------------
// host-side builders (shared by the C ABI in conv.hip / gemm.hip and by tests/host_emul)
// ---------------------------------------------------------------------------------------------
namespace vp {

// big side defaults to stride*small (the VAE's layers); Conv2d with odd inputs passes Hb/Wb explicitly
// (Hs = floor((Hb + 2*pad - ks)/stride) + 1).
inline ConvGeom make_geom(int B, int Hs, int Ws, int Cs, int Cb, int stride, int ks = 5, int Hb = 0, int Wb = 0) {
  ConvGeom g;
  g.B = B; g.Hs = Hs; g.Ws = Ws; g.Hb = Hb > 0 ? Hb : Hs * stride; g.Wb = Wb > 0 ? Wb : Ws * stride;
  g.Cs = Cs; g.Cb = Cb; g.stride = stride;
  g.ks = ks; g.pad = (ks - 1) / 2; g.nt = ks * ks;
  g.dHW = make_fastdiv((uint32_t)(Hs * Ws)); g.dW = make_fastdiv((uint32_t)Ws);
  g.dCs = make_fastdiv((uint32_t)Cs); g.dCb = make_fastdiv((uint32_t)Cb);
  return g;
}

const void* vp_zero_page();   // device build: elementwise.hip; host emulation: tests/host_emul/emul.cpp

inline ProbF make_probF(const float* big, const float* wp0, const float* bias, float* out, const ConvGeom& g, int act) {
  ProbF p;
  p.zero = vp_zero_page();
  p.big = big; p.w = wp0; p.bias = bias; p.out = out; p.g = g; p.act = act;
  p.M = g.B * g.Hs * g.Ws; p.N = g.Cs; p.K = g.nt * g.Cb;
  p.vec = (g.Cb % 4 == 0);
  return p;
}

inline ProbT make_probT(const float* small, const float* wp1, float* out, const ConvGeom& g) {
  ProbT p;
  p.zero = vp_zero_page();
  p.small = small; p.w = wp1; p.out = out; p.g = g;
  p.M = g.B * g.Hs * g.Ws; p.N = g.Cb;
  p.vec = (g.Cs % 4 == 0);
  return p;
}

// K (= pixels) is split so that tiles * 25 taps * nsplit gives every CU one to two workgroups.
inline int wgrad_nsplit(const ConvGeom& g) {
  long K = (long)g.B * g.Hs * g.Ws;
  long tiles = ((g.Cs + 127) / 128) * (long)((g.Cb + 127) / 128) * g.nt;
  // ~1.5 workgroups per CU: measured optimum with the weight gradients on the side stream (tools/ab_build.py:
  // 256 -> 4.61, 320 -> 4.56, 384 -> 4.35, 448 -> 4.37, 512 -> 4.44, 768 -> 4.47 ms/step); fewer splits also mean
  // smaller slabs for the reduction kernel
  long target = 384;
  // per-shape search on the final schedule (round 2, profiles/r02_notes.md section 9): the single-tile layers (25 workgroups per split) keep
  // 16 splits, the 2- and 8-tile layers gain 0.6 % / 0.35 % of the step with 12 and 3 splits instead of 8 and 2
  if (tiles >= 50) target = 600;
  // tiny weights (<= 1024 entries per tap: the 1-8-channel predictor convolutions of the segmentation heads, run over
  // 256x256 images): one 32-row tile per tap does all the work of a million-pixel contraction, so split far deeper
  // (the slabs stay small); measured on tools/bench_be_heads.py
  const bool tiny = (long)g.Cs * g.Cb <= 1024;
  if (tiny) target = 4096;
  long want = (target + tiles - 1) / tiles;
  long maxs = (K + 511) / 512;  // keep >= 16 K-tiles of 32 per split
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  // a 1x1 layer whose whole weight is one or two tiles (the first encoder conv on its im2col: 64 x 96, 131 072 pixels at
  // B = 32) offers no tap parallelism at all: split the pixels up to 256 ways
  const long cap = tiny ? 512 : (tiles <= 2 ? 256 : 64);
  if (s > cap) s = cap;
  return (int)s;
}

inline ProbW make_probW(const float* big, const float* small, float* slab, const ConvGeom& g, int nsplit) {
  ProbW p;
  p.big = big; p.small = small; p.slab = slab; p.g = g;
  p.M = g.Cs; p.N = g.Cb; p.K = g.B * g.Hs * g.Ws;
  p.nsplit = nsplit;
  int per = (p.K + nsplit - 1) / nsplit;
  p.k_per_split = ((per + 31) / 32) * 32;
  p.vec_a = (g.Cs % 4 == 0);
  p.vec_b = (g.Cb % 4 == 0);
  return p;
}

inline size_t wgrad_slab_floats(const ConvGeom& g, int nsplit) {
  return (size_t)nsplit * g.nt * g.Cs * g.Cb;
}

inline int gemm_nsplit(long M, long N, long K) {
  long bm = M <= 32 ? 32 : 128, bn = M <= 32 ? 128 : (N <= 32 ? 32 : 128);
  long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  if (tiles >= 128) return 1;
  long target = 384;
  long want = (target + tiles - 1) / tiles;
  long maxs = (K + 255) / 256;
  long s = want < maxs ? want : maxs;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

template <bool AKM, bool BKM>
inline ProbG<AKM, BKM> make_probG(const float* A, long sam, long sak, const float* B, long sbn, long sbk,
                                  float* C, int ldc, const float* bias, int M, int N, int K, int nsplit) {
  ProbG<AKM, BKM> p;
  p.A = A; p.Bm = B; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K;
  p.sam = sam; p.sak = sak; p.sbn = sbn; p.sbk = sbk; p.ldc = ldc;
  p.nsplit = nsplit;
  int per = (K + nsplit - 1) / nsplit;
  p.k_per_split = ((per + 31) / 32) * 32;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  p.vec_a = AKM ? (sam == 1 && sak % 4 == 0 && al16(A)) : (sak == 1 && sam % 4 == 0 && al16(A));
  p.vec_b = BKM ? (sbn == 1 && sbk % 4 == 0 && al16(B)) : (sbk == 1 && sbn % 4 == 0 && al16(B));
  return p;
}

}  // namespace vp
