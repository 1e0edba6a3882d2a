// Split-bf16 ("bf16x3") implicit-GEMM kernel for gfx950.
//
// fp32 operands are stored as two bf16 planes, x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (16 significant bits, fp32 exponent range).  A product is contracted as
//      a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi          (the 2^-18 lo*lo term is dropped)
// with three v_mfma_f32_32x32x16_bf16 per fragment pair and fp32 accumulation: ~5e-6 relative
// error per contraction (plain bf16: 2e-3, outside the 1e-3 parity bar) at a 5.3x higher MFMA
// ceiling than v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s / 3 vs 157 TFLOP/s).
//
//  * planes are written by the producers (BN/activation, transposes, weight packing), never by
//    this kernel: the GEMM only moves 16-B chunks of 8 bf16 and issues MFMAs;
//  * 4 wavefronts per workgroup, K-tile of 32 elements (two 16-deep MFMA steps);
//  * "MK" operands (k contiguous in HBM: activations gathered per tap, packed weights):
//    LDS image [row][32 bf16 + 16 B pad] (80-B stride: a 16-lane ds_read_b128 group covers all 64
//    banks), fragment = one ds_read_b128 per tile and step;
//  * "KM" operands (the weight-gradient GEMM contracts over pixels, which are the SLOW index of
//    an NHWC tensor): LDS image [pixel][channels] with the row stride == 16 dwords (mod 64) and the
//    fragment fetched with the hardware transposing read ds_read_b64_tr_b16 (two per tile and
//    step), so no transpose pass through HBM or VALU is needed;
//  * accumulator layout and store path identical to igemm.h (lane = output channel).
#pragma once
#include <stdio.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "problems.h"

namespace vp {

#ifndef VP_IGEMM16_SWZ_ON
#define VP_IGEMM16_SWZ_ON 1      // swizzled, unpadded LDS rows for the 128x64 tile (see igemm16_kernel); 0: the padded rows everywhere
#endif
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

// K-tile depth BKT (elements) is a template parameter: 32 or 64.  MK rows are padded by 16 B
// (80-B / 144-B stride: every 16-lane ds_read_b128 group covers all 64 banks).
template <int BKT>
struct MkStride { static constexpr int bytes = BKT * 2 + 16; };

VP_HD u32x4_t zero_u4() { u32x4_t z = {0u, 0u, 0u, 0u}; return z; }
VP_HD u32x4_t ld16(const u16* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const u32x4_t*>(p);
#else
  u32x4_t v; __builtin_memcpy(&v, p, 16); return v;
#endif
}

// row stride (bytes) of a KM image with BR channel columns: data + pad, == 64 B (mod 256 B)
template <int BR>
struct KmStride {
  static constexpr int bytes = (BR == 128) ? 320 : (BR == 64 ? 192 : (BR == 32 ? 64 : -1));
};
// The lo plane starts 64 B past a multiple of 128 B after the hi plane: an 8-lane ds_write_b128 group
// that stores 4 hi + 4 lo chunks of one row (BKT = 32) then covers 32 distinct banks (it was a 2-way
// conflict -- 1/3 of all LDS cycles in the first PMC profile -- when both planes started on bank 0).
template <int BR, bool KM, int BKT>
struct Lds16 {
  static constexpr int raw = KM ? BKT * KmStride<BR>::bytes : BR * MkStride<BKT>::bytes;
  static constexpr int plane_bytes = ((raw + 127) / 128) * 128 + 64;
};

// ---- problem descriptors (vector path only: channel counts are multiples of 8) -----------------
// F family on split planes: A(m=(b,hs,ws), k=(tap,c)) = big[...]; B(n, k) = wp0[n][tap][c]
template <bool K5, int MODE_ = 0>
struct ProbF16T {
  static constexpr int MODE = MODE_;  // 0: bf16 pairs, 3 MFMAs | 1: fp16 pairs, 2 MFMAs | 2: fp16 pairs, 3 MFMAs (see mfma_split) | 3: fp32 (see F32)
  static constexpr bool X2 = MODE_ == 1, F16 = MODE_ == 1 || MODE_ == 2;
  // MODE 3, exact fp32 through this kernel's data path: ONE plane per operand holding fp32 values, addressed in 16-bit units (the
  // host doubles the gathered channel count: [pix][C] fp32 = [pix][2C] u16), a 16-B chunk = 4 floats, a 64-deep K-tile = 32 floats;
  // every fragment pair is contracted by four v_mfma_f32_32x32x2_f32.  Gather / scatter families only.
  static constexpr bool F32 = MODE_ == 3;
  static constexpr bool IS_K5 = K5;
  float alpha = 1.f;                  // F16: the accumulators are multiplied by alpha before the epilogue (undoes a producer's scale)
  // K5: the 5x5 / padding-2 / Hb = stride*Hs case of the VAE layers with its constants folded in (the generic form costs
  // the hot path ~1 %); !K5: kernel size, padding and the big image size come from the geometry (models/blocks.py)
  VP_HD int ks() const { return K5 ? 5 : g.ks; }
  VP_HD int pad() const { return K5 ? 2 : g.pad; }
  VP_HD int nt() const { return K5 ? 25 : g.nt; }
  static constexpr bool A_KM = false, B_KM = false;
  const u16* big; size_t big_plane;     // hi plane at big, lo plane at big + big_plane (elements)
  const u16* w; size_t w_plane;         // packed P0 planes [Cs][25][Cb]
  const float* bias; float* out; const void* zero;
  ConvGeom g; int act; int M, N, K;
  int nsplit, k_per_split;   // nsplit == 2: both halves of K are atomically added onto a zeroed output
  int xcd_map;               // remap (blockIdx.x, blockIdx.y) so that column tiles of one row tile share an XCD
  // BatchNorm statistics from the epilogue (nsplit == 1 only): per workgroup and output channel {pivot, sum(x - pivot),
  // sum((x - pivot)^2)} over the workgroup's valid rows, slab layout [3][N][groups], group = blockIdx.z * row tiles + row tile
  static constexpr bool HAS_STAT = true;
  float* stat = nullptr;
  struct ZCtx { int k_begin, k_end; };
  struct ARow { int pix_base, h0, w0, valid; };
  VP_HD bool out_index(int m, int n, const ZCtx&, size_t& idx) const { idx = (size_t)m * N + n; return m < M && n < N; }
  struct BRow { int off, valid; };
  VP_HD void z_setup(int zi, ZCtx& z) const {
    z.k_begin = zi * k_per_split;
    const int e = z.k_begin + k_per_split;
    z.k_end = e < K ? e : K;
  }
  VP_HD ARow a_row(int m, const ZCtx&) const {
    ARow r; r.valid = m < M; int mm = r.valid ? m : 0;
    int b = (int)g.dHW.div((uint32_t)mm); int rem = mm - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    r.pix_base = b * g.Hb * g.Wb; r.h0 = g.stride * hs - pad(); r.w0 = g.stride * ws - pad();
    return r;
  }
  // Every gather is branch-free: out-of-range chunks read the zero page, so a K-tile's loads are issued
  // back to back.  FAST (channel count a multiple of the K-tile depth): the whole tile [k0, k0+BKT) lies
  // inside ONE tap, so tap -> (rr, qq) is computed from the workgroup-uniform k0 on the scalar unit; only
  // the bounds test and the address remain per-lane work (VALU per MFMA fell from 5-13 to ~3, PMC).
  template <bool FAST>
  VP_HD u32x4_t a_load(const ARow& r, int k0, int k8, int plane, const ZCtx& z) const {
    const int kt = FAST ? k0 : k0 + k8;
    const int tap = (int)g.dCb.div((uint32_t)kt);
    const int c = k0 + k8 - tap * g.Cb;
    const int rr = div_small(tap, ks()), qq = tap - rr * ks();
    const int h = r.h0 + rr, w_ = r.w0 + qq;
    const bool ok = r.valid && kt < z.k_end && h >= 0 && h < g.Hb && w_ >= 0 && w_ < g.Wb;
    return ld16(ok ? big + plane * big_plane + (size_t)(r.pix_base + h * g.Wb + w_) * g.Cb + c : reinterpret_cast<const u16*>(zero));
  }
  VP_HD BRow b_row(int n, const ZCtx&) const { BRow r; r.valid = n < N; r.off = (r.valid ? n : 0) * K; return r; }
  template <bool FAST>
  VP_HD u32x4_t b_load(const BRow& r, int k0, int k8, int plane, const ZCtx& z) const {
    const bool ok = r.valid && (FAST ? k0 : k0 + k8) < z.k_end;
    return ld16(ok ? w + plane * w_plane + (size_t)r.off + k0 + k8 : reinterpret_cast<const u16*>(zero));
  }
  // FAST row-level form: one bounds test + one address per (row, K-tile); the caller adds plane / chunk offsets
  // FAST K order is CHANNEL-CHUNK major, tap minor: K-tile index kt -> (cc = kt / 25, tap = kt % 25).  The 25
  // taps of one 64-channel chunk gather overlapping pixels back to back, so the re-reads hit the per-XCD L2
  // (tap-major order re-reads a pixel only after a full channel sweep -- far more than 4 MB per XCD -- and
  // every conv kernel then streamed its operands at the same ~10 TB/s regardless of tile shape).
  template <int SH = 6>
  VP_HD void fast_tile(int k0, int ntap, int& tap, int& c0) const {
    const int kt = k0 >> SH;
    const int cc = kt / ntap;            // scalar: k0 and ntap are workgroup-uniform
    tap = kt - cc * ntap;
    c0 = cc << SH;
  }
  template <int SH = 6>
  VP_HD bool a_base(const ARow& r, int k0, const ZCtx& z, size_t& off) const {
    int tap, c0;
    fast_tile<SH>(k0, nt(), tap, c0);
    const int rr = div_small(tap, ks()), qq = tap - rr * ks();
    const int h = r.h0 + rr, w_ = r.w0 + qq;
    off = (size_t)(r.pix_base + h * g.Wb + w_) * g.Cb + c0;
    return r.valid && k0 < z.k_end && h >= 0 && h < g.Hb && w_ >= 0 && w_ < g.Wb;
  }
  template <int SH = 6>
  VP_HD bool b_base(const BRow& r, int k0, const ZCtx& z, size_t& off) const {
    int tap, c0;
    fast_tile<SH>(k0, nt(), tap, c0);
    off = (size_t)r.off + tap * g.Cb + c0;
    return r.valid && k0 < z.k_end;
  }
  VP_HD const u16* a_ptr() const { return big; }
  VP_HD size_t a_plane() const { return big_plane; }
  VP_HD const u16* b_ptr() const { return w; }
  VP_HD size_t b_plane() const { return w_plane; }
  VP_HD void store(int m, int n, float v, const ZCtx&) const {
    if (m >= M || n >= N) return;
    if (F16) v *= alpha;
#if defined(__HIP_DEVICE_COMPILE__)
    if (nsplit > 1) { atomicAdd(out + (size_t)m * N + n, v); return; }   // 2 addends onto 0: order-independent
#endif
    if (bias) v += bias[n];
    if (act == ACT_SIGMOID) v = 1.f / (1.f + __builtin_expf(-v));
    out[(size_t)m * N + n] = v;
  }
};
using ProbF16 = ProbF16T<true>;
using ProbF16K = ProbF16T<false>;
using ProbF16X = ProbF16T<true, 1>;
using ProbF16KX = ProbF16T<false, 1>;
using ProbF16H = ProbF16T<true, 2>;
using ProbF16KH = ProbF16T<false, 2>;

// T family on split planes (phase-decomposed transposed conv)
template <bool K5, int MODE_ = 0>
struct ProbT16T {
  static constexpr int MODE = MODE_;  // 0: bf16 pairs, 3 MFMAs | 1: fp16 pairs, 2 MFMAs | 2: fp16 pairs, 3 MFMAs (see mfma_split) | 3: fp32 (see F32)
  static constexpr bool X2 = MODE_ == 1, F16 = MODE_ == 1 || MODE_ == 2;
  // MODE 3, exact fp32 through this kernel's data path: ONE plane per operand holding fp32 values, addressed in 16-bit units (the
  // host doubles the gathered channel count: [pix][C] fp32 = [pix][2C] u16), a 16-B chunk = 4 floats, a 64-deep K-tile = 32 floats;
  // every fragment pair is contracted by four v_mfma_f32_32x32x2_f32.  Gather / scatter families only.
  static constexpr bool F32 = MODE_ == 3;
  static constexpr bool IS_K5 = K5;
  float alpha = 1.f;                  // F16: the accumulators are multiplied by alpha before the epilogue (undoes a producer's scale)
  // K5: the 5x5 / padding-2 / Hb = stride*Hs case of the VAE layers with its constants folded in (the generic form costs
  // the hot path ~1 %); !K5: kernel size, padding and the big image size come from the geometry (models/blocks.py)
  VP_HD int ks() const { return K5 ? 5 : g.ks; }
  VP_HD int pad() const { return K5 ? 2 : g.pad; }
  VP_HD int nt() const { return K5 ? 25 : g.nt; }
  static constexpr bool A_KM = false, B_KM = false;
  const u16* small; size_t small_plane;
  const u16* w; size_t w_plane;          // packed P1 planes [Cb][25][Cs]
  float* out; const void* zero; ConvGeom g; int M, N;
  int nsplit;                // 1, or 2: z = phase*2 + half, halves atomically added onto a zeroed output
  int xcd_map;
  int pair_phases = 0;       // see z_setup
  static constexpr bool HAS_STAT = true;
  float* stat = nullptr;     // see ProbF16T
  struct ZCtx { int k_begin, k_end, ph, pw, th, tw, r0h, r0w, bh, bw; };   // phase geometry as in problems.h ProbT
  VP_HD bool out_index(int m, int n, const ZCtx& z, size_t& idx) const {
    if (m >= M || n >= N) { idx = 0; return false; }
    int b = (int)g.dHW.div((uint32_t)m); int rem = m - b * (g.Hs * g.Ws);
    int q = (int)g.dW.div((uint32_t)rem), p = rem - q * g.Ws;
    int oh = g.stride * q + z.ph, ow = g.stride * p + z.pw;
    idx = ((size_t)(b * g.Hb + oh) * g.Wb + ow) * g.Cb + n;
    return K5 || (oh < g.Hb && ow < g.Wb);
  }
  struct ARow { int pix_base, q, p, valid; };
  struct BRow { int off, valid; };
  VP_HD void z_setup(int zi, ZCtx& z) const {
    int phase = zi / nsplit;
    const int half = zi - phase * nsplit;
    // stride 2: the four phases contract 9 / 6 / 6 / 4 taps.  Workgroups are dealt to the CUs round-robin in z-major order, so when a
    // launch is at most two workgroups per CU (`pair_phases`, set by the host: dec1.fwd at 32 images has 512) CU c runs z = 0 beside
    // z = 2 and z = 1 beside z = 3: issuing the 4-tap phase third pairs 9 + 4 and 6 + 6 instead of 9 + 6 and 6 + 4 (dec1.fwd 187 ->
    // 180 us, 522 -> 472 us in fp32).  Launches of several rounds keep longest-first (there the swap costs 3 - 20 %).
    if (pair_phases && g.stride == 2 && phase >= 2) phase = 5 - phase;
    int s = g.stride; z.ph = phase / s; z.pw = phase - z.ph * s;
    z.r0h = (z.ph + pad()) % s; z.r0w = (z.pw + pad()) % s;
    z.th = z.r0h < ks() ? (ks() - z.r0h + s - 1) / s : 0; z.tw = z.r0w < ks() ? (ks() - z.r0w + s - 1) / s : 0;
    z.bh = (z.ph + pad() - z.r0h) / s; z.bw = (z.pw + pad() - z.r0w) / s;
    const int kall = z.th * z.tw * g.Cs;
    // split at a multiple of 64 so that FAST tiles stay inside one tap / chunk
    const int kh = nsplit > 1 ? (((kall / 64) + 1) / 2) * 64 : kall;
    z.k_begin = half * kh;
    z.k_end = half + 1 < nsplit ? kh : kall;
  }
  VP_HD ARow a_row(int m, const ZCtx&) const {
    ARow r; r.valid = m < M; int mm = r.valid ? m : 0;
    int b = (int)g.dHW.div((uint32_t)mm); int rem = mm - b * (g.Hs * g.Ws);
    r.q = (int)g.dW.div((uint32_t)rem); r.p = rem - r.q * g.Ws; r.pix_base = b * g.Hs * g.Ws;
    return r;
  }
  template <bool FAST>
  VP_HD u32x4_t a_load(const ARow& r, int k0, int k8, int plane, const ZCtx& z) const {
    const int kt = FAST ? k0 : k0 + k8;
    const int t = (int)g.dCs.div((uint32_t)kt);
    const int c = k0 + k8 - t * g.Cs;
    const int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    const int h = r.q + z.bh - rp, w_ = r.p + z.bw - qp;
    const bool ok = r.valid && kt < z.k_end && h >= 0 && h < g.Hs && w_ >= 0 && w_ < g.Ws;
    return ld16(ok ? small + plane * small_plane + (size_t)(r.pix_base + h * g.Ws + w_) * g.Cs + c : reinterpret_cast<const u16*>(zero));
  }
  VP_HD BRow b_row(int n, const ZCtx&) const { BRow r; r.valid = n < N; r.off = (r.valid ? n : 0) * nt() * g.Cs; return r; }
  template <bool FAST>
  VP_HD u32x4_t b_load(const BRow& r, int k0, int k8, int plane, const ZCtx& z) const {
    const int kt = FAST ? k0 : k0 + k8;
    const int t = (int)g.dCs.div((uint32_t)kt);
    const int c = k0 + k8 - t * g.Cs;
    const int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    const int tap = (z.r0h + g.stride * rp) * ks() + (z.r0w + g.stride * qp);
    const bool ok = r.valid && kt < z.k_end;
    return ld16(ok ? w + plane * w_plane + (size_t)r.off + tap * g.Cs + c : reinterpret_cast<const u16*>(zero));
  }
  template <int SH = 6>
  VP_HD void fast_tile(int k0, int ntap, int& t, int& c0) const {   // channel-chunk major, tap minor (see ProbF16)
    const int kt = k0 >> SH;
    const int cc = kt / ntap;
    t = kt - cc * ntap;
    c0 = cc << SH;
  }
  template <int SH = 6>
  VP_HD bool a_base(const ARow& r, int k0, const ZCtx& z, size_t& off) const {
    int t, c0;
    fast_tile<SH>(k0, z.th * z.tw, t, c0);
    const int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    const int h = r.q + z.bh - rp, w_ = r.p + z.bw - qp;
    off = (size_t)(r.pix_base + h * g.Ws + w_) * g.Cs + c0;
    return r.valid && k0 < z.k_end && h >= 0 && h < g.Hs && w_ >= 0 && w_ < g.Ws;
  }
  template <int SH = 6>
  VP_HD bool b_base(const BRow& r, int k0, const ZCtx& z, size_t& off) const {
    int t, c0;
    fast_tile<SH>(k0, z.th * z.tw, t, c0);
    const int rp = div_small(t, z.tw), qp = t - rp * z.tw;
    const int tap = (z.r0h + g.stride * rp) * ks() + (z.r0w + g.stride * qp);
    off = (size_t)r.off + tap * g.Cs + c0;
    return r.valid && k0 < z.k_end;
  }
  VP_HD const u16* a_ptr() const { return small; }
  VP_HD size_t a_plane() const { return small_plane; }
  VP_HD const u16* b_ptr() const { return w; }
  VP_HD size_t b_plane() const { return w_plane; }
  VP_HD void store(int m, int n, float v, const ZCtx& z) const {
    if (m >= M || n >= N) return;
    int b = (int)g.dHW.div((uint32_t)m); int rem = m - b * (g.Hs * g.Ws);
    int q = (int)g.dW.div((uint32_t)rem), p = rem - q * g.Ws;
    int oh = g.stride * q + z.ph, ow = g.stride * p + z.pw;
    if (!K5 && (oh >= g.Hb || ow >= g.Wb)) return;   // odd big sizes: the last phase row/column does not exist
    float* dst = out + ((size_t)(b * g.Hb + oh) * g.Wb + ow) * g.Cb + n;
    if (F16) v *= alpha;
#if defined(__HIP_DEVICE_COMPILE__)
    if (nsplit > 1) { atomicAdd(dst, v); return; }
#endif
    *dst = v;
  }
};
using ProbT16 = ProbT16T<true>;
using ProbT16K = ProbT16T<false>;
using ProbT16X = ProbT16T<true, 1>;
using ProbT16KX = ProbT16T<false, 1>;
using ProbT16H = ProbT16T<true, 2>;
using ProbT16KH = ProbT16T<false, 2>;

// W family on split planes: slab[split][tap][cs][cb]; both operands pixel-major (KM).
// PAIR (narrow big side, Cb = 32 | 64): one workgroup contracts TWO taps, the columns [0, Cb) of its virtual N = 2*Cb tile belong to
// tap 2*pair and [Cb, 2*Cb) to tap 2*pair + 1 (same `small` rows, two shifted `big` pixels per staged k row): a 32-channel layer
// fills a 64-column tile, a 64-channel layer the 128-column one, instead of leaving half of the MFMA columns (or of the A-fragment
// reuse) unused.  13 pairs cover the 25 taps; the odd half of the last pair is masked.
template <bool K5, int MODE_ = 0, bool PAIR_ = false>
struct ProbW16T {
  static constexpr bool PAIR = PAIR_;
  static constexpr int MODE = MODE_;  // 0: bf16 pairs, 3 MFMAs | 1: fp16 pairs, 2 MFMAs | 2: fp16 pairs, 3 MFMAs (see mfma_split) | 3: fp32 (see F32)
  static constexpr bool X2 = MODE_ == 1, F16 = MODE_ == 1 || MODE_ == 2;
  // MODE 3, exact fp32 through this kernel's data path: ONE plane per operand holding fp32 values, addressed in 16-bit units (the
  // host doubles the gathered channel count: [pix][C] fp32 = [pix][2C] u16), a 16-B chunk = 4 floats, a 64-deep K-tile = 32 floats;
  // every fragment pair is contracted by four v_mfma_f32_32x32x2_f32.  Gather / scatter families only.
  static constexpr bool F32 = MODE_ == 3;
  static constexpr bool IS_K5 = K5;
  float alpha = 1.f;                  // F16: the accumulators are multiplied by alpha before the epilogue (undoes a producer's scale)
  // K5: the 5x5 / padding-2 / Hb = stride*Hs case of the VAE layers with its constants folded in (the generic form costs
  // the hot path ~1 %); !K5: kernel size, padding and the big image size come from the geometry (models/blocks.py)
  VP_HD int ks() const { return K5 ? 5 : g.ks; }
  VP_HD int pad() const { return K5 ? 2 : g.pad; }
  VP_HD int nt() const { return K5 ? 25 : g.nt; }
  static constexpr bool A_KM = true, B_KM = true;
  static constexpr bool HAS_STAT = false;
  const u16* big; size_t big_plane;
  const u16* small; size_t small_plane;
  float* slab; const void* zero; ConvGeom g; int M, N, K; int nsplit, k_per_split;
  struct ZCtx { int k_begin, k_end, rr, qq, tap, split, rr2, qq2; };
  VP_HD void z_setup(int zi, ZCtx& z) const {
    z.tap = zi / nsplit; z.split = zi - z.tap * nsplit;
    if (PAIR) {                      // zi / nsplit = the pair: taps 2*pair and 2*pair + 1 (the latter may be 25: masked)
      const int t2 = 2 * z.tap + 1;
      z.tap = 2 * z.tap;
      z.rr2 = t2 / ks(); z.qq2 = t2 - z.rr2 * ks();
    } else { z.rr2 = z.qq2 = 0; }
    z.rr = z.tap / ks(); z.qq = z.tap - z.rr * ks();
    z.k_begin = z.split * k_per_split;
    int e = z.k_begin + k_per_split; z.k_end = e < K ? e : K;
  }
  VP_HD u32x4_t a_load_km(int k, int m, int plane, const ZCtx& z) const {   // small[pixel k][m..m+7]
    const bool ok = k < z.k_end && m < M;
    return ld16(ok ? small + plane * small_plane + (size_t)k * g.Cs + m : reinterpret_cast<const u16*>(zero));
  }
  VP_HD u32x4_t b_load_km(int k, int n, int plane, const ZCtx& z) const {   // big[shifted pixel][n..n+7]
    int b = (int)g.dHW.div((uint32_t)k); int rem = k - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    int rr = z.rr, qq = z.qq;
    bool tap_ok = true;
    if (PAIR && n >= g.Cb) { rr = z.rr2; qq = z.qq2; n -= g.Cb; tap_ok = z.tap + 1 < nt(); }
    int h = g.stride * hs - pad() + rr, w_ = g.stride * ws - pad() + qq;
    const bool ok = tap_ok && k < z.k_end && n < (PAIR ? g.Cb : N) && h >= 0 && h < g.Hb && w_ >= 0 && w_ < g.Wb;
    return ld16(ok ? big + plane * big_plane + ((size_t)(b * g.Hb + h) * g.Wb + w_) * g.Cb + n : reinterpret_cast<const u16*>(zero));
  }
  // FAST row-level form (tile fully inside M x N): element offset of channel 0 of pixel k
  VP_HD bool a_base_km(int k, const ZCtx& z, size_t& off) const {
    off = (size_t)k * g.Cs;
    return k < z.k_end;
  }
  VP_HD bool b_base_km(int k, const ZCtx& z, size_t& off) const {
    int b = (int)g.dHW.div((uint32_t)k); int rem = k - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    int h = g.stride * hs - pad() + z.rr, w_ = g.stride * ws - pad() + z.qq;
    off = ((size_t)(b * g.Hb + h) * g.Wb + w_) * g.Cb;
    return k < z.k_end && h >= 0 && h < g.Hb && w_ >= 0 && w_ < g.Wb;
  }
  // PAIR: the second tap's pixel of the same k row
  VP_HD bool b_base_km2(int k, const ZCtx& z, size_t& off) const {
    int b = (int)g.dHW.div((uint32_t)k); int rem = k - b * (g.Hs * g.Ws);
    int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    int h = g.stride * hs - pad() + z.rr2, w_ = g.stride * ws - pad() + z.qq2;
    off = ((size_t)(b * g.Hb + h) * g.Wb + w_) * g.Cb;
    return z.tap + 1 < nt() && k < z.k_end && h >= 0 && h < g.Hb && w_ >= 0 && w_ < g.Wb;
  }
  VP_HD const u16* a_ptr() const { return small; }
  VP_HD size_t a_plane() const { return small_plane; }
  VP_HD const u16* b_ptr() const { return big; }
  VP_HD size_t b_plane() const { return big_plane; }
  VP_HD void store(int m, int n, float v, const ZCtx& z) const {
    if (m >= M || n >= N) return;
    if (PAIR) {                      // N = 2*Cb virtual columns -> (tap, real column); the slab keeps its [split][tap][cs][cb] layout
      const int second = n >= g.Cb ? 1 : 0;
      if (z.tap + second >= nt()) return;
      slab[(((size_t)z.split * nt() + z.tap + second) * M + m) * g.Cb + (n - second * g.Cb)] = F16 ? v * alpha : v;
      return;
    }
    slab[(((size_t)z.split * nt() + z.tap) * M + m) * N + n] = F16 ? v * alpha : v;
  }
};
using ProbW16 = ProbW16T<true>;
using ProbW16K = ProbW16T<false>;
using ProbW16P = ProbW16T<true, 0, true>;     // tap pairs (plain 5x5 layers with 32 or 64 big channels)
using ProbW16X = ProbW16T<true, 1>;
using ProbW16KX = ProbW16T<false, 1>;

#if defined(__HIPCC__)

// One fragment pair of the split contraction, fp32 accumulate.
//   MODE 0 (bf16 pairs): a*b ~= al*bh + ah*bl + ah*bh, three v_mfma_f32_32x32x16_bf16 (16 significant bits per operand);
//   MODE 2 (fp16 pairs): the same three products with v_mfma_f32_32x32x16_f16 (up to 22 significant bits per operand);
//   MODE 1 (fp16 pairs): a*b ~= al*bh + ah*bh = a*bh, two MFMAs: operand A keeps hi + lo, operand B only its fp16 hi plane
//                        (11 bits: 2^-12 rms relative rounding per B element); B's lo plane is not read.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__device__ __forceinline__ f32x16_t mfma_split(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x16_t c) {
  if constexpr (MODE != 0) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
    if constexpr (MODE == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  }
  return c;
}

// MODE 3: both fragments are 4 fp32 values (k = 4 lh + j of an 8-float block, the same permutation for A and B): four
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate)
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16_t mfma_f32x4(const bf16x8_t& a, const bf16x8_t& b, f32x16_t c) {
  const f32x4v_t af = __builtin_bit_cast(f32x4v_t, a), bf = __builtin_bit_cast(f32x4v_t, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], c, 0, 0, 0);
  return c;
}

template <class P, int NR, bool KM> struct Rows16A;
template <class P, int NR> struct Rows16A<P, NR, false> { typename P::ARow r[NR]; };
template <class P, int NR> struct Rows16A<P, NR, true> {};
template <class P, int NR, bool KM> struct Rows16B;
template <class P, int NR> struct Rows16B<P, NR, false> { typename P::BRow r[NR]; };
template <class P, int NR> struct Rows16B<P, NR, true> {};

// fragment fetch: 8 bf16 (k = 8*lh .. 8*lh+7 of MFMA step s) of tile row/column `row`
// SWZ (k-contiguous rows of 64 elements only): 128-B rows without padding, the 16-B chunk index XOR-ed with (row >> 1) & 7 -- sixteen
// consecutive rows then cover all 64 banks with one chunk each (8 * (row & 1) + chunk ^ swizzle takes 16 distinct values)
template <int BR, bool KM, int BKT, bool SWZ = false>
__device__ __forceinline__ bf16x8_t frag16(const unsigned char* plane, int row, int s, int li, int lh, int lane) {
  if constexpr (!KM && SWZ) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(plane + row * 128 + (((s * 2 + lh) ^ ((row >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8_t, v);
  } else if constexpr (!KM) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(plane + row * MkStride<BKT>::bytes + s * 32 + lh * 16);
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    // transposing read: within a 16-lane group, lane i supplies &T[k0 + (i>>2)][col0 + 4*(i&3)] and
    // receives column i of the 4 rows.  col0 = row - (li & 15): the group's first column.
    constexpr int S = KmStride<BR>::bytes;
    const int i = lane & 15;
    const int col0 = row - i;
    const int k0 = s * 16 + lh * 8;
    const unsigned char* p0 = plane + (k0 + (i >> 2)) * S + (col0 + 4 * (i & 3)) * 2;
    typedef bf16x4_t __attribute__((address_space(3))) * lds_v4;
    const bf16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(p0));
    const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(p0 + 4 * S));
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// BatchNorm statistics of the workgroup's output tile, taken from the accumulators (32x32 MFMA layout: lane = column,
// rows in the 16 registers): replaces one full read of the activation by bn_partial_kernel<0>.  Sums are pivoted on the
// tile's first row (see bn.hip: raw sum(x^2) loses log2(mean^2/var) bits), combined across the wave rows through LDS in a
// fixed order and written as {pivot, sum(x - pivot), sum((x - pivot)^2)} per (channel, group).  Call with every wave past its
// last LDS read of the main loop (a barrier).  `group` = blockIdx.z * (row tiles) + row tile, `groups` = their total.
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void epilogue_stats32(float* __restrict__ stat, int groups, int group, int M, int N, const f32x16_t (&acc)[TM][TN],
                                                 unsigned char* lds, int m0, int n0, int wm, int wn, int li, int lh, int tid) {
  float* pv = reinterpret_cast<float*>(lds);      // [BN] pivots
  float* red = pv + BN;                            // [WM][BN][2]
  if (wm == 0 && lh == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) pv[wn * (BN / WN) + 32 * j + li] = acc[0][j][0];   // row m0 (always < M)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / WN) + 32 * j + li;
    const float pvt = pv[col];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float d = m < M ? acc[i][j][r] - pvt : 0.f;
        s += d;
        q += d * d;
      }
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    if (lh == 0) { red[(wm * BN + col) * 2] = s; red[(wm * BN + col) * 2 + 1] = q; }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
    const size_t n = (size_t)(n0 + tid);
    stat[(0 * (size_t)N + n) * groups + group] = pv[tid];
    stat[(1 * (size_t)N + n) * groups + group] = s;
    stat[(2 * (size_t)N + n) * groups + group] = q;
  }
}

// The same statistics from the accumulators of the 16x16x32 form (lane = column lane % 16 of a 16-column block, rows 4 * (lane / 16) + r
// in the 4 registers): the four lane groups are combined by shuffles, the wave rows through LDS, in a fixed order.
typedef float f32x4_t16 __attribute__((ext_vector_type(4)));
template <int BM, int BN, int WM, int WN, int TM16, int TN16>
__device__ __forceinline__ void epilogue_stats16(float* __restrict__ stat, int groups, int group, int M, int N, const f32x4_t16 (&acc)[TM16][TN16],
                                                 unsigned char* lds, int m0, int n0, int wm, int wn, int lane, int tid) {
  float* pv = reinterpret_cast<float*>(lds);      // [BN] pivots
  float* red = pv + BN;                            // [WM][BN][2]
  const int lc = lane & 15, lg = lane >> 4;
  if (wm == 0 && lg == 0) {
#pragma unroll
    for (int j = 0; j < TN16; ++j) pv[wn * (BN / WN) + 16 * j + lc] = acc[0][j][0];   // row m0 (always < M)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN16; ++j) {
    const int col = wn * (BN / WN) + 16 * j + lc;
    const float pvt = pv[col];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (BM / WM) + 16 * i + 4 * lg + r;
        const float d = m < M ? acc[i][j][r] - pvt : 0.f;
        s += d;
        q += d * d;
      }
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    if (lg == 0) { red[(wm * BN + col) * 2] = s; red[(wm * BN + col) * 2 + 1] = q; }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
    const size_t n = (size_t)(n0 + tid);
    stat[(0 * (size_t)N + n) * groups + group] = pv[tid];
    stat[(1 * (size_t)N + n) * groups + group] = s;
    stat[(2 * (size_t)N + n) * groups + group] = q;
  }
}

// M16: contract with v_mfma_f32_16x16x32_bf16 (one MFMA = a 32-deep k-step of a 16 x 16 block; same FLOPs per cycle as the 32x32x16
// form, lower power per product: tools/kbench measured +5 - 8 % in-kernel clock).  Gather / scatter families on bf16 pairs only
// (k-contiguous operands: a fragment is the same 16-B read at row lane % 16, k chunk lane / 16 -- the 144-B row pitch covers all 64
// banks per 16-lane group as it does for the 32x32 form).  Results equal the 32x32x16 form's to rounding (another summation order).
// (the 128x128 tile of this form needs ~200 registers if left alone -- one workgroup per CU, 40 % slower: its launch bounds ask for two)
template <class P, int BM, int BN, int WM, int WN, int BKT, bool FAST, bool M16 = false>
__global__ void __launch_bounds__(256, (M16 && BM * BN >= 128 * 128) ? 2 : 1) igemm16_kernel(const P p) {
  static_assert(!M16 || (!P::A_KM && !P::B_KM && P::MODE == 0 && BKT % 32 == 0), "16x16x32 form: k-contiguous bf16-pair operands");
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BKT == 32 || BKT == 64, "K-tile depth");
  constexpr int SH = BKT == 64 ? 6 : 5;       // FAST gather/scatter kernels decompose k0 in BKT-deep channel chunks
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
  constexpr int KC = BKT / 8;                 // 16-B chunks per row and plane
  constexpr int NA_KM = BM * KC / 128, NB_KM = BN * KC / 128;   // 16-B chunks per thread per K-tile (both planes)
  // SWZ: the 128x64 tile on 64-deep k-contiguous planes stores unpadded, swizzled rows (frag16): 49 KB instead of 55 KB of LDS = THREE
  // workgroups per CU instead of two
  constexpr bool SWZ = VP_IGEMM16_SWZ_ON && !P::A_KM && !P::B_KM && !P::F32 && BM == 128 && BN == 64 && BKT == 64;
  constexpr int A_PLANE = SWZ ? BM * 128 + 64 : Lds16<BM, P::A_KM, BKT>::plane_bytes;
  constexpr int B_PLANE = SWZ ? BN * 128 + 64 : Lds16<BN, P::B_KM, BKT>::plane_bytes;
  constexpr int MKS = SWZ ? 128 : MkStride<BKT>::bytes;
  auto swz = [](int row, int chunk) { return SWZ ? (chunk ^ ((row >> 1) & 7)) : chunk; };
  // MK staging map: 4 threads per row; a thread stages, for BM/64 (BN/64) rows, CPT chunks of the row:
  // both planes x the k-chunks {l4*CPH .. l4*CPH+CPH-1}.  One bounds test / address per row serves all
  // its chunks (FAST path), a 4-lane group reads 64 contiguous bytes per plane, and an 8-lane
  // ds_write_b128 group (2 rows x 4 lanes) covers 32 distinct banks.
  constexpr int TPR = 4, RPP = 256 / TPR;     // 64 rows per pass
  constexpr int CPT = 2 * KC / TPR, CPH = CPT / 2;   // chunks per thread per row / per plane
  constexpr int NRA = (BM + RPP - 1) / RPP, NRB = (BN + RPP - 1) / RPP;   // a 32-row operand uses half the threads
  constexpr bool A_PART = BM < RPP, B_PART = BN < RPP;
  constexpr int NA = P::A_KM ? NA_KM : NRA * CPT, NB = P::B_KM ? NB_KM : NRB * CPT;
  static_assert(NA >= 1 && NB >= 1, "staging map");
  // (the two-product fp16 mode never stages B's lo plane: its LDS image is one plane -- a third workgroup per CU on the 128x64 tiles)
  constexpr bool A_LO = !P::F32, B_LO = !P::X2 && !P::F32;      // is the operand's lo plane staged and read?
  static_assert(!P::F32 || (!P::A_KM && !P::B_KM), "fp32 mode: k-contiguous operands only");
  __shared__ __attribute__((aligned(16))) unsigned char lds[(A_LO ? 2 : 1) * A_PLANE + (B_LO ? 2 : 1) * B_PLANE];
  unsigned char* As = lds;                   // [plane][...]
  unsigned char* Bs = lds + (A_LO ? 2 : 1) * A_PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int tx = blockIdx.x, ty = blockIdx.y, tz = blockIdx.z;
  if constexpr (!P::A_KM) {
    if (p.xcd_map == 1) {
      // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs in launch order; give the gridDim.y
      // column tiles that read the same activation rows hardware ids that are equal modulo 8 (same XCD, same L2)
      const int hid = blockIdx.x + gridDim.x * blockIdx.y;
      const int r = hid & 7, q = hid >> 3;
      tx = r + 8 * (q / (int)gridDim.y);
      ty = q % (int)gridDim.y;
    }
  }
  const int m0 = tx * BM, n0 = ty * BN;

  typename P::ZCtx z;
  p.z_setup(tz, z);

  // staging map.  MK: chunk = (row = tid/(2*KC) + RPP*i, sub = tid%(2*KC) -> plane = sub/KC, k8 = sub%KC)
  //               KM: chunk = (krow, col8, plane) with V = BR/8 chunks per k-row and plane
  Rows16A<P, (P::A_KM ? 1 : NRA), P::A_KM> ra;
  Rows16B<P, (P::B_KM ? 1 : NRB), P::B_KM> rb;
  const int l4 = tid % TPR, srow = tid / TPR;
  const bool a_on = !A_PART || srow < BM, b_on = !B_PART || srow < BN;
  if constexpr (!P::A_KM) {
#pragma unroll
    for (int i = 0; i < NRA; ++i) ra.r[i] = p.a_row(m0 + (a_on ? srow : 0) + RPP * i, z);
  }
  if constexpr (!P::B_KM) {
#pragma unroll
    for (int i = 0; i < NRB; ++i) rb.r[i] = p.b_row(n0 + (b_on ? srow : 0) + RPP * i, z);
  }
  u32x4_t sa[NA], sb[NB];

  auto stage_load = [&](int k0) {
    if constexpr (!P::A_KM) {
      if (a_on)
#pragma unroll
      for (int i = 0; i < NRA; ++i) {
        if constexpr (FAST) {
          size_t off;
          const bool ok = p.template a_base<SH>(ra.r[i], k0, z, off);
          const u16* b0 = ok ? p.a_ptr() + off : reinterpret_cast<const u16*>(p.zero);
          const u16* b1 = ok ? p.a_ptr() + p.a_plane() + off : reinterpret_cast<const u16*>(p.zero);
#pragma unroll
          for (int j = 0; j < CPH; ++j) {
            sa[i * CPT + j] = ld16(b0 + (l4 * CPH + j) * 8);
            if constexpr (A_LO) sa[i * CPT + CPH + j] = ld16(b1 + (l4 * CPH + j) * 8);
          }
        } else {
#pragma unroll
          for (int j = 0; j < CPH; ++j) {
            sa[i * CPT + j] = p.template a_load<false>(ra.r[i], k0, (l4 * CPH + j) * 8, 0, z);
            if constexpr (A_LO) sa[i * CPT + CPH + j] = p.template a_load<false>(ra.r[i], k0, (l4 * CPH + j) * 8, 1, z);
          }
        }
      }
    } else {
      // one pixel (k-row) per thread and K-tile: TP threads share a pixel and each stages NA of its
      // 16-B chunks, so the pixel -> (b, h, w) decomposition is done once, not once per chunk
      constexpr int V = BM / 8, TP = 256 / BKT;
      const int kr = tid / TP, c0 = tid % TP;
      if constexpr (FAST) {
        size_t off;
        const bool ok = p.a_base_km(k0 + kr, z, off);
        const u16* b0 = ok ? p.a_ptr() + off + m0 : reinterpret_cast<const u16*>(p.zero);
        const u16* b1 = ok ? p.a_ptr() + p.a_plane() + off + m0 : reinterpret_cast<const u16*>(p.zero);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int ch = c0 + TP * i;
          sa[i] = ld16((ch / V ? b1 : b0) + (ch % V) * 8);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int ch = c0 + TP * i;             // chunk index inside the pixel: [plane][V]
          sa[i] = p.a_load_km(k0 + kr, m0 + (ch % V) * 8, ch / V, z);
        }
      }
    }
    if constexpr (!P::B_KM) {
      if (b_on)
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        if constexpr (FAST) {
          size_t off;
          const bool ok = p.template b_base<SH>(rb.r[i], k0, z, off);
          const u16* b0 = ok ? p.b_ptr() + off : reinterpret_cast<const u16*>(p.zero);
          const u16* b1 = ok ? p.b_ptr() + p.b_plane() + off : reinterpret_cast<const u16*>(p.zero);
#pragma unroll
          for (int j = 0; j < CPH; ++j) {
            sb[i * CPT + j] = ld16(b0 + (l4 * CPH + j) * 8);
            if constexpr (B_LO) sb[i * CPT + CPH + j] = ld16(b1 + (l4 * CPH + j) * 8);      // X2 / F32: B's lo plane is never read
          }
        } else {
#pragma unroll
          for (int j = 0; j < CPH; ++j) {
            sb[i * CPT + j] = p.template b_load<false>(rb.r[i], k0, (l4 * CPH + j) * 8, 0, z);
            if constexpr (B_LO) sb[i * CPT + CPH + j] = p.template b_load<false>(rb.r[i], k0, (l4 * CPH + j) * 8, 1, z);
          }
        }
      }
    } else {
      constexpr int V = BN / 8, TP = 256 / BKT;
      const int kr = tid / TP, c0 = tid % TP;
      if constexpr (FAST && P::PAIR) {
        // two taps per workgroup: chunk columns [0, V/2) of a plane come from tap A's pixel, [V/2, V) from tap B's (n0 = 0: one tile)
        size_t offa, offb;
        const bool oka = p.b_base_km(k0 + kr, z, offa), okb = p.b_base_km2(k0 + kr, z, offb);
        const u16* const zp = reinterpret_cast<const u16*>(p.zero);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (P::X2 && TP * i >= V) continue;
          const int ch = c0 + TP * i, pl = ch / V, cv = ch % V;
          const bool second = cv >= V / 2;
          const bool ok = second ? okb : oka;
          const u16* src = ok ? p.b_ptr() + (pl ? p.b_plane() : 0) + (second ? offb : offa) + (cv - (second ? V / 2 : 0)) * 8 : zp;
          sb[i] = ld16(src);
        }
      } else if constexpr (FAST) {
        size_t off;
        const bool ok = p.b_base_km(k0 + kr, z, off);
        const u16* b0 = ok ? p.b_ptr() + off + n0 : reinterpret_cast<const u16*>(p.zero);
        const u16* b1 = ok ? p.b_ptr() + p.b_plane() + off + n0 : reinterpret_cast<const u16*>(p.zero);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (P::X2 && TP * i >= V) continue;     // (compile-time after unrolling) chunks of the lo plane: not read by X2
          const int ch = c0 + TP * i;
          sb[i] = ld16((ch / V ? b1 : b0) + (ch % V) * 8);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (P::X2 && TP * i >= V) continue;
          const int ch = c0 + TP * i;
          sb[i] = p.b_load_km(k0 + kr, n0 + (ch % V) * 8, ch / V, z);
        }
      }
    }
  };
  auto stage_write = [&]() {
    if constexpr (!P::A_KM) {
      if (a_on)
#pragma unroll
      for (int i = 0; i < NRA; ++i)
#pragma unroll
        for (int j = 0; j < CPH; ++j) {
          unsigned char* dst = As + (srow + RPP * i) * MKS + swz(srow + RPP * i, l4 * CPH + j) * 16;
          *reinterpret_cast<u32x4_t*>(dst) = sa[i * CPT + j];
          if constexpr (A_LO) *reinterpret_cast<u32x4_t*>(dst + A_PLANE) = sa[i * CPT + CPH + j];
        }
    } else {
      constexpr int V = BM / 8, TP = 256 / BKT, S = KmStride<BM>::bytes;
      const int kr = tid / TP, c0 = tid % TP;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int ch = c0 + TP * i;
        *reinterpret_cast<u32x4_t*>(As + (ch / V) * A_PLANE + kr * S + (ch % V) * 16) = sa[i];
      }
    }
    if constexpr (!P::B_KM) {
      if (b_on)
#pragma unroll
      for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int j = 0; j < CPH; ++j) {
          unsigned char* dst = Bs + (srow + RPP * i) * MKS + swz(srow + RPP * i, l4 * CPH + j) * 16;
          *reinterpret_cast<u32x4_t*>(dst) = sb[i * CPT + j];
          if constexpr (B_LO) *reinterpret_cast<u32x4_t*>(dst + B_PLANE) = sb[i * CPT + CPH + j];
        }
    } else {
      constexpr int V = BN / 8, TP = 256 / BKT, S = KmStride<BN>::bytes;
      const int kr = tid / TP, c0 = tid % TP;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (P::X2 && TP * i >= V) continue;
        const int ch = c0 + TP * i;
        *reinterpret_cast<u32x4_t*>(Bs + (ch / V) * B_PLANE + kr * S + (ch % V) * 16) = sb[i];
      }
    }
  };

  const int klen = z.k_end - z.k_begin;
  const int nk = klen > 0 ? (klen + BKT - 1) / BKT : 0;

  if constexpr (M16) {
    constexpr int TM16 = BM / WM / 16, TN16 = BN / WN / 16;
    f32x4_t16 acc16[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
    const int lc = lane & 15, lg = lane >> 4;
    // (row blocks are 16 rows apart: (row >> 1) & 7 -- the swizzle -- is the same for every block of a lane)
    const int arow = wm * (BM / WM) + lc, brow = wn * (BN / WN) + lc;
    const unsigned char* const a0 = As + arow * MKS;
    const unsigned char* const b0 = Bs + brow * MKS;
    if (nk > 0) {
      stage_load(z.k_begin);
      stage_write();
      __syncthreads();
    }
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) stage_load(z.k_begin + (kt + 1) * BKT);
#pragma unroll
      for (int s = 0; s < BKT / 32; ++s) {
        // (the wave's row blocks in halves: 16 + 32 fragment registers live instead of 64 -- two workgroups per CU on the 128x128 tile)
        constexpr int TMH = TM16 >= 2 ? TM16 / 2 : TM16;
        bf16x8_t bh[TN16], bl[TN16];
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
          bh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(b0 + 16 * j * MKS + swz(brow, 4 * s + lg) * 16));
          bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(b0 + B_PLANE + 16 * j * MKS + swz(brow, 4 * s + lg) * 16));
        }
#pragma unroll
        for (int h = 0; h < TM16 / TMH; ++h) {
          bf16x8_t ah[TMH], al[TMH];
#pragma unroll
          for (int i = 0; i < TMH; ++i) {
            ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(a0 + 16 * (h * TMH + i) * MKS + swz(arow, 4 * s + lg) * 16));
            al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(a0 + A_PLANE + 16 * (h * TMH + i) * MKS + swz(arow, 4 * s + lg) * 16));
          }
#pragma unroll
          for (int i = 0; i < TMH; ++i)
#pragma unroll
            for (int j = 0; j < TN16; ++j) {
              f32x4_t16 c = acc16[h * TMH + i][j];
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
              acc16[h * TMH + i][j] = c;
            }
        }
      }
      __syncthreads();
      if (more) {
        stage_write();
        __syncthreads();
      }
    }
    if constexpr (P::HAS_STAT) {
      if (p.stat)
        epilogue_stats16<BM, BN, WM, WN, TM16, TN16>(p.stat, (int)(gridDim.x * gridDim.z), (int)(tz * gridDim.x) + tx, p.M, p.N, acc16, lds, m0,
                                                     n0, wm, wn, lane, tid);
    }
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          p.store(m0 + wm * (BM / WM) + 16 * i + 4 * lg + r, n0 + wn * (BN / WN) + 16 * j + lc, acc16[i][j][r], z);
    return;
  }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow0 = wm * (BM / WM) + li;
  const int brow0 = wn * (BN / WN) + li;

  if (nk > 0) {
    stage_load(z.k_begin);
    stage_write();
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) stage_load(z.k_begin + (kt + 1) * BKT);
#pragma unroll
    for (int s = 0; s < BKT / 16; ++s) {
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag16<BM, P::A_KM, BKT, SWZ>(As, arow0 + 32 * i, s, li, lh, lane);
        if constexpr (A_LO) al[i] = frag16<BM, P::A_KM, BKT, SWZ>(As + A_PLANE, arow0 + 32 * i, s, li, lh, lane);
        else al[i] = ah[i];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = frag16<BN, P::B_KM, BKT, SWZ>(Bs, brow0 + 32 * j, s, li, lh, lane);
        if constexpr (!B_LO) bl[j] = bh[j];      // not read by the two-product / fp32 forms
        else bl[j] = frag16<BN, P::B_KM, BKT, SWZ>(Bs + B_PLANE, brow0 + 32 * j, s, li, lh, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (P::F32) acc[i][j] = mfma_f32x4(ah[i], bh[j], acc[i][j]);
          else acc[i][j] = mfma_split<P::MODE>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
        }
    }
    __syncthreads();
    if (more) {
      stage_write();
      __syncthreads();
    }
  }

  if constexpr (P::HAS_STAT) {
    if (p.stat)      // (workgroup-uniform) every wave is past its last LDS read: the loop ends with a barrier
      epilogue_stats32<BM, BN, WM, WN, TM, TN>(p.stat, (int)(gridDim.x * gridDim.z), (int)(tz * gridDim.x) + tx, p.M, p.N, acc, lds, m0, n0,
                                               wm, wn, li, lh, tid);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + wm * (BM / WM) + 32 * i + row;
        const int n = n0 + wn * (BN / WN) + 32 * j + li;
        p.store(m, n, acc[i][j][r], z);
      }
}

#ifndef VP_IGEMM16_M16_ON
#define VP_IGEMM16_M16_ON 0      // translation units that dispatch the 16x16x32 form of igemm16_kernel define it to 1 (conv16.hip)
#endif
struct Tile16 { int bm, bn; };   // wave grid: 2 x 2, or 4 x 1 for the 32-column tiles

inline Tile16 choose_tile16(long M, long N, int gz, bool /*pixel_major*/ = false, int ctile = 0) {
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long)gz; };
  const long MINB = 384;      // (per-shape tile searches on the VAE and VAE-GAN steps found nothing beyond noise: profiles/r02_notes.md section 7)
  // a 32-column operand (the 32-channel end of the last decoder block): 64-column tiles would idle half the MFMAs.
  // Unless k runs over 64-channel chunks (ctile = the gathered / scattered side's channel count) and the launch is large: the
  // 32-column tiles exist with 32-deep K-tiles only, and the 64x64 tile on 64-deep FAST K-tiles is 1.4-2.3x faster there even
  // with half of its MFMA columns idle (profiles/r02_g_narrow_n_microbench.log: gather 5x5 s2 64 -> 32 channels 238 -> 106 us, the VAE-GAN
  // discriminator's 64 -> 32 input gradient 252 -> 175 us; 32- and 40-channel chunks keep the narrow tiles: 134 vs 173 us).
  const bool deep = ctile > 0 && ctile % 64 == 0 && (gz == 1 || M * gz >= (1L << 19));
  if (N <= 32 && M >= 128 && !deep) return (M >= 256 && blocks(256, 32) >= MINB) ? Tile16{256, 32} : Tile16{128, 32};
  if (M >= 128 && N >= 128 && blocks(128, 128) >= MINB) return {128, 128};
  if (M >= 128 && N >= 64 && blocks(128, 64) >= MINB) return {128, 64};
  return {64, 64};
}

// K-tile depth: 64 for the gather/scatter families (half the barriers per MFMA), 32 for the weight
// gradient (its [pixel][channel] LDS images at depth 64 leave one workgroup per CU).  Measured on
// MI355X, profiles/.
inline int igemm16_bk(bool km, bool x2 = false) {
  // the two-product fp16 mode stages three planes instead of four: the weight gradient's 64-deep K-tile then leaves two
  // workgroups per CU (61 KB each) and halves its barriers (measured -20 us per step, f16x2)
  return km ? (x2 ? 64 : 32) : 64;
}

template <class P, int BKT, bool FAST>
inline void launch_igemm16_bk(const P& p, long M, long N, int gz, hipStream_t stream, int ctile = 0, bool m16 = false) {
  Tile16 t = choose_tile16(M, N, gz, P::A_KM, ctile);
  dim3 block(256);
  auto grid = [&](int bm, int bn) { return dim3((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)gz); };
  if constexpr (VP_IGEMM16_M16_ON && !P::A_KM && !P::B_KM && P::MODE == 0 && P::IS_K5 && BKT == 64 && FAST) {
    if (m16 && t.bm == 128 && (t.bn == 128 || t.bn == 64)) {      // the 16x16x32 form of the same tiles (the host decides where: igemm16_m16_rule)
      if (t.bn == 128) hipLaunchKernelGGL((igemm16_kernel<P, 128, 128, 2, 2, BKT, FAST, true>), grid(128, 128), block, 0, stream, p);
      else hipLaunchKernelGGL((igemm16_kernel<P, 128, 64, 2, 2, BKT, FAST, true>), grid(128, 64), block, 0, stream, p);
      return;
    }
  }
  if (t.bm == 128 && t.bn == 128) {
    hipLaunchKernelGGL((igemm16_kernel<P, 128, 128, 2, 2, BKT, FAST>), grid(128, 128), block, 0, stream, p);
  } else if (t.bm == 128 && t.bn == 64) {
    hipLaunchKernelGGL((igemm16_kernel<P, 128, 64, 2, 2, BKT, FAST>), grid(128, 64), block, 0, stream, p);
  } else if (t.bn == 32) {
    // 32-column tiles run 32-deep K-tiles (LDS budget: 2-3 workgroups per CU)
    if constexpr (BKT == 32) {
      if (t.bm == 256) hipLaunchKernelGGL((igemm16_kernel<P, 256, 32, 4, 1, 32, FAST>), grid(256, 32), block, 0, stream, p);
      else hipLaunchKernelGGL((igemm16_kernel<P, 128, 32, 4, 1, 32, FAST>), grid(128, 32), block, 0, stream, p);
    }
  } else {
    hipLaunchKernelGGL((igemm16_kernel<P, 64, 64, 2, 2, BKT, FAST>), grid(64, 64), block, 0, stream, p);
  }
}

// ctile = the channel count that k is decomposed by (0 for the pixel-major wgrad family): FAST kernels
// are used when it is a multiple of the K-tile depth.
template <class P>
inline void launch_igemm16(const P& p, long M, long N, int gz, hipStream_t stream, int ctile = 0, bool m16 = false) {
  const int bk = igemm16_bk(P::A_KM, P::X2);
  if constexpr (P::A_KM) {
    // pixel-major (wgrad) family: FAST when the chosen tile lies fully inside M x N (no channel tails)
    Tile16 t = choose_tile16(M, N, gz, true);
    const bool fast = (M % t.bm == 0) && (N % t.bn == 0);
    if (bk == 32) {
      if (fast) launch_igemm16_bk<P, 32, true>(p, M, N, gz, stream);
      else launch_igemm16_bk<P, 32, false>(p, M, N, gz, stream);
    } else if (t.bn == 32) {
      if (fast) launch_igemm16_bk<P, 32, true>(p, M, N, gz, stream);
      else launch_igemm16_bk<P, 32, false>(p, M, N, gz, stream);
    } else {
      if constexpr (P::X2) {
        if (fast) { launch_igemm16_bk<P, 64, true>(p, M, N, gz, stream); return; }
      }
      launch_igemm16_bk<P, 64, false>(p, M, N, gz, stream);
    }
  } else {
    const Tile16 t0 = choose_tile16(M, N, gz, false, ctile);
    const bool narrow = t0.bn == 32;   // tiles that exist with 32-deep K-tiles only
    if (ctile > 0 && ctile % 64 == 0 && bk == 64 && !narrow) launch_igemm16_bk<P, 64, true>(p, M, N, gz, stream, ctile, m16);
    else if (ctile > 0 && ctile % 32 == 0) launch_igemm16_bk<P, 32, true>(p, M, N, gz, stream, ctile);   // 32-channel chunks
    else if (bk == 32 || narrow) launch_igemm16_bk<P, 32, false>(p, M, N, gz, stream, ctile);
    else launch_igemm16_bk<P, 64, false>(p, M, N, gz, stream, ctile);
  }
}
#endif  // __HIPCC__

}  // namespace vp
