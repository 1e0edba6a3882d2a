// Weight gradient of the 5x5 / stride-2 / padding-2 layers on split planes, ONE KERNEL ROW OF TAPS per workgroup (round 3).
//
//   dW[cs][cb][r][q] = sum_{b,hs,ws} small[b,hs,ws,cs] * big[b, 2hs-2+r, 2ws-2+q, cb]
//   replaces: the weight gradient cuDNN computes behind nn.Conv2d / nn.ConvTranspose2d (k5 s2 p2), models/networks.py:14,38.
//
// igemm16_kernel<ProbW16T> gives every tap its own workgroup set (grid z = 25 taps x splits): each of the 25 taps re-stages BOTH
// operands of its pixel range (3.7x the algorithmic traffic, 34.6 % MFMA-busy: profiles/r02_g_*).  Here a workgroup owns
//   128 small channels x BN big channels x the 5 taps of one kernel row r x one pixel range (split):
//   * per K-tile of 32 small pixels the `small` tile [32 px][128 ch] is staged ONCE for five taps, and the `big` operand as the
//     halo patch of the tile -- NR image rows x (2*Wt + 4) columns of big row 2*hs - 2 + r -- in its natural pixel order;
//     a tap q is then a per-lane LDS address offset (q rows of the patch) of the transposing fragment read, the stride-2 walk
//     over pixels a row step of two: the row pitch is == 32 B (mod 128 B) so that the four rows of a ds_read_b64_tr_b16 group
//     (two pitches apart) cover all 64 banks;
//   * the `small` fragments of a k-step stay in registers for the five taps (LDS reads per MFMA: 1.33 -> 0.93 transposing reads);
//   * eight waves (two per SIMD), wave tile 64 x 32 per tap = 32 accumulator registers x 5 taps; BN = 64 (the layers with 64 big
//     channels) splits the K-tile's two 16-pixel steps over two wave groups, which emit separate slabs;
//   * LDS double-buffered, ONE barrier per K-tile; the next tile's global loads are issued in two halves so that at most four
//     16-B staging registers are live beside the 160 accumulators.
// L2 -> LDS bytes per MFMA fall 3.1x (50 KB per 480 MFMAs against 160 KB).  Slabs keep the layout [split][tap][cs][cb] and are
// summed by the existing fixed-order reduction (bit-reproducible).
#pragma once
#include "igemm16.h"

namespace vp {

struct ProbW5 {
  const u16* big; size_t big_plane;        // [B][Hb][Wb][Cb]: hi plane, lo plane at + big_plane (elements)
  const u16* small; size_t small_plane;    // [B][Hs][Ws][Cs]
  float* slab;                             // [nsplit * KG][25][Cs][Cb]
  const void* zero;
  int Hs, Ws, Hb, Wb, Cs, Cb;
  int K;                                   // B * Hs * Ws small pixels (a multiple of 32)
  int nsplit, k_per_split;                 // k_per_split a multiple of 32, every split non-empty
  int tiles_n, inner, total, g8;           // inner = tiles_m * tiles_n * 5 work items per split; total = inner * nsplit; g8 = ceil(total / 8)
  int lgWt, HW, HP;                        // K-tile: 32 / Wt image rows of Wt = min(Ws, 32) pixels; halo row = HW = 2*Wt + 4 big pixels
  FastDiv dImg, dW;                        // divisors Hs*Ws, Ws
  float alpha;
  unsigned long long* dbg;                 // diagnostics only (tools/kbench): per-workgroup clock stamps
};

#if defined(__HIPCC__)

typedef bf16x4_t __attribute__((address_space(3))) * lds_v4_t;

// 8 bf16 (k = 8 consecutive pixels) of one channel column: two transposing reads, `step4` bytes apart (4 pixels further)
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p, int step4) {
  const bf16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(p));
  const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(p + step4));
  return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// BN = 128: wave grid 2 x 4, both k-steps per wave.  BN = 64: two K groups (k-step = group) x wave grid 2 x 2.
// MODE as in mfma_split (0: bf16 pairs, three products; 1: fp16 pairs, two products -- big's lo plane is neither staged nor read).
template <int BN, int MODE>
__global__ void __launch_bounds__(512) wgrad5_kernel(const ProbW5 p) {
  static_assert(BN == 128 || BN == 64, "big-channel tile");
  constexpr bool X2 = MODE == 1;
  constexpr int NT = 512, BM = 128;
  constexpr int WN = BN / 32;                  // wave columns
  constexpr int KG = 8 / (2 * WN);             // K groups: 1 | 2
  constexpr int NS = 2 / KG;                   // k-steps of 16 pixels per wave and K-tile
  constexpr int SA = 320, SB = BN * 2 + 32;    // row pitches (bytes): A == 64 (mod 256), B == 32 (mod 128)
  constexpr int HPA = 96;                      // halo rows reserved per stage (HP <= 80)
  constexpr int A_PLANE = 32 * SA, B_PLANE = HPA * SB;
  constexpr int STAGE = 2 * A_PLANE + (X2 ? 1 : 2) * B_PLANE;
  constexpr int B_CH = BN / 8;                 // 16-B chunks per halo pixel and plane
  constexpr int PPB = NT / B_CH;               // halo pixels per staging pass: 32 | 64
  constexpr int NPB = (HPA + PPB - 1) / PPB;   // passes: 3 | 2
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (2 * WN), wr = wave % (2 * WN), wm = wr / WN, wn = wr % WN;
  const int li = lane & 31, lh = lane >> 5;

  // work item: XCD x (= hardware id mod 8) owns the contiguous range [x * g8, (x + 1) * g8) of the split-major item order, so that the
  // tiles and tap rows of one pixel range -- which read the same operand bytes -- share an L2
  const int item = (int)(blockIdx.x & 7) * p.g8 + (int)(blockIdx.x >> 3);
  if (item >= p.total) return;
  const int split = item / p.inner, in = item - split * p.inner;
  const int t = in / 5, r5 = in - 5 * t;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * p.k_per_split;
  const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
  const int nk = (k_end - k_begin) >> 5;

  unsigned long long t0 = 0, r0 = 0;
  if (p.dbg && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- staging maps ---------------------------------------------------------------------------------------------------
  // A: pixel = tid / 16, chunk = tid % 16, both planes.  B pass P: halo pixel hp = P * PPB + tid / B_CH, chunk = tid % B_CH.
  const int a_px = tid >> 4, a_c8 = tid & 15;
  const size_t a_off0 = (size_t)a_px * p.Cs + m0 + a_c8 * 8;
  const int a_dst = a_px * SA + a_c8 * 16;
  const int b_c8 = tid % B_CH;
  int b_h[NPB], b_w[NPB], b_dst[NPB];          // hb = 2 * hs0 + b_h, wb = 2 * ws0 + b_w; b_h = -2^20 for rows past the patch
#pragma unroll
  for (int P = 0; P < NPB; ++P) {
    const int hp = P * PPB + tid / B_CH;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) j += hp >= q * p.HW;
    const int c = hp - j * p.HW;
    b_h[P] = hp < p.HP ? 2 * j - 2 + r5 : -(1 << 20);
    b_w[P] = c - 2;
    b_dst[P] = hp * SB + b_c8 * 16;
  }
  const u16* const zp = reinterpret_cast<const u16*>(p.zero);

  u32x4_t sa[2], sb[NPB][2];
  int tb = 0, ths = 0, tws = 0;                // tile base (image, row, column) of the tile being loaded: workgroup-uniform
  auto tile_base = [&](int k0) {
    tb = (int)p.dImg.div((uint32_t)k0);
    const int rem = k0 - tb * (p.Hs * p.Ws);
    ths = (int)p.dW.div((uint32_t)rem);
    tws = rem - ths * p.Ws;
  };
  auto load_a = [&](int k0) {
    const u16* s = p.small + (size_t)k0 * p.Cs + a_off0;
    sa[0] = ld16(s);
    sa[1] = ld16(s + p.small_plane);
  };
  auto load_b = [&](int P) {
    const int hb = 2 * ths + b_h[P], wb = 2 * tws + b_w[P];
    const bool ok = (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
    const u16* s = ok ? p.big + ((size_t)(tb * p.Hb + hb) * p.Wb + wb) * p.Cb + n0 + b_c8 * 8 : zp;
    sb[P][0] = ld16(s);
    if constexpr (!X2) sb[P][1] = ld16(ok ? s + p.big_plane : zp);
  };
  auto write_a = [&](unsigned char* st) {
    *reinterpret_cast<u32x4_t*>(st + a_dst) = sa[0];
    *reinterpret_cast<u32x4_t*>(st + A_PLANE + a_dst) = sa[1];
  };
  auto write_b = [&](unsigned char* st, int P) {
    if (PPB * (P + 1) <= HPA || P * PPB + tid / B_CH < HPA) {
      *reinterpret_cast<u32x4_t*>(st + 2 * A_PLANE + b_dst[P]) = sb[P][0];
      if constexpr (!X2) *reinterpret_cast<u32x4_t*>(st + 2 * A_PLANE + B_PLANE + b_dst[P]) = sb[P][1];
    }
  };

  // ---- fragment read offsets (per lane) --------------------------------------------------------------------------------
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int a_frag = (lh * 8 + (i16 >> 2)) * SA + (wm * 64 + 16 * g16 + 4 * (i16 & 3)) * 2;
  int b_frag[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kkb = (KG == 1 ? s : kg) * 16 + lh * 8;           // first of the lane's 8 pixels inside the K-tile
    const int j = kkb >> p.lgWt, w = kkb - (j << p.lgWt);
    b_frag[s] = (j * p.HW + 2 * w + 2 * (i16 >> 2)) * SB + (wn * 32 + 16 * g16 + 4 * (i16 & 3)) * 2;
  }

  f32x16_t acc[5][2];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

  auto compute = [&](const unsigned char* st, int s) {
    const int sk = KG == 1 ? s : kg;
    const unsigned char* ap = st + a_frag + sk * 16 * SA;
    bf16x8_t ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = tr_frag(ap + i * 64, 4 * SA);
      al[i] = tr_frag(ap + A_PLANE + i * 64, 4 * SA);
    }
    const unsigned char* bp = st + 2 * A_PLANE + b_frag[KG == 1 ? s : 0];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const bf16x8_t bh = tr_frag(bp + q * SB, 8 * SB);
      bf16x8_t bl = bh;
      if constexpr (!X2) bl = tr_frag(bp + B_PLANE + q * SB, 8 * SB);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[q][i] = mfma_split<MODE>(ah[i], al[i], bh, bl, acc[q][i]);
    }
  };

  if (nk > 0) {
    tile_base(k_begin);
    load_a(k_begin);
#pragma unroll
    for (int P = 0; P < NPB; ++P) load_b(P);
    write_a(lds);
#pragma unroll
    for (int P = 0; P < NPB; ++P) write_b(lds, P);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    unsigned char* nxt = lds + ((kt & 1) ^ 1) * STAGE;
    const bool more = kt + 1 < nk;
    const int k1 = k_begin + (kt + 1) * 32;
    if (more) {
      tile_base(k1);
      load_a(k1);
      load_b(0);
    }
    compute(cur, 0);
    if (more) {
      write_a(nxt);
      write_b(nxt, 0);
#pragma unroll
      for (int P = 1; P < NPB; ++P) load_b(P);
    }
    if constexpr (NS == 2) compute(cur, 1);
    if (more) {
#pragma unroll
      for (int P = 1; P < NPB; ++P) write_b(nxt, P);
    }
    __syncthreads();
  }

  // ---- slabs [split * KG + kg][tap = 5 * r5 + q][cs][cb] ------------------------------------------------------------------
  float* const sl = p.slab + ((size_t)(split * KG + kg) * 25 + 5 * r5) * p.Cs * p.Cb;
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = n0 + wn * 32 + li;
        float v = acc[q][i][r];
        if constexpr (MODE != 0) v *= p.alpha;
        sl[((size_t)q * p.Cs + m) * p.Cb + n] = v;
      }
  if (p.dbg && tid == 0) {
    unsigned long long* d = p.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t0; d[1] = r0; d[2] = __builtin_amdgcn_s_memtime(); d[3] = __builtin_amdgcn_s_memrealtime();
  }
}


#if defined(VP_WGRAD5_Q)      // prototype, compiled by tools/kbench/wbench only (not dispatched: +2.5-4.5 % on two layers, slower on dec0)
// ---- the same workgroup on v_mfma_f32_16x16x32_bf16 ("q" form, BN = 128 only) --------------------------------------------------
// Equal FLOPs per cycle, but the chip holds a higher clock on this MFMA shape under load (MI355X_MICROARCH.md, DVFS give-back 7;
// profiles/r02_notes.md section 1: +3-7 % on the gather / scatter kernels).  One MFMA covers the whole 32-pixel K-tile; the wave
// tile 64 x 32 is 4 x 2 tiles of 16 x 16.  Lane group g = lane / 16 contracts the pixels {4g .. 4g+3} and {16+4g .. 16+4g+3} of
// the K-tile (any bijection works as long as both operands use it), so that the two 16-lane groups of a 32-lane LDS access read
// EIGHT consecutive image rows: with a row pitch == 32 B (mod 256 B) -- 16 B (mod 128 B) for the stride-2 walk over the halo
// patch -- they cover all 64 banks.  Results equal the 32x32x16 form to rounding (different summation order inside an MFMA).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ f32x4_t mfma_split16(const bf16x8_t& ah, const bf16x8_t& al, const bf16x8_t& bh, const bf16x8_t& bl, f32x4_t c) {
  if constexpr (MODE != 0) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  }
  return c;
}

template <int MODE>
__global__ void __launch_bounds__(512) wgrad5q_kernel(const ProbW5 p) {
  constexpr bool X2 = MODE == 1;
  constexpr int NT = 512, BM = 128, BN = 128;
  constexpr int SA = 288, SB = 272;            // row pitches (bytes): A == 32 (mod 256), B == 16 (mod 128)
  constexpr int HPA = 96;
  constexpr int A_PLANE = 32 * SA, B_PLANE = HPA * SB;
  constexpr int STAGE = 2 * A_PLANE + (X2 ? 1 : 2) * B_PLANE;
  constexpr int B_CH = BN / 8, PPB = NT / B_CH, NPB = 3;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int item = (int)(blockIdx.x & 7) * p.g8 + (int)(blockIdx.x >> 3);
  if (item >= p.total) return;
  const int split = item / p.inner, in = item - split * p.inner;
  const int t = in / 5, r5 = in - 5 * t;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * p.k_per_split;
  const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
  const int nk = (k_end - k_begin) >> 5;

  unsigned long long t0 = 0, r0 = 0;
  if (p.dbg && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

  const int a_px = tid >> 4, a_c8 = tid & 15;
  const size_t a_off0 = (size_t)a_px * p.Cs + m0 + a_c8 * 8;
  const int a_dst = a_px * SA + a_c8 * 16;
  const int b_c8 = tid % B_CH;
  int b_hw[NPB];                               // (2j - 2 + r5) << 16 | (c - 2) & 0xffff; row -4096 = past the patch (registers are scarce here)
  const int b_dst0 = (tid / B_CH) * SB + b_c8 * 16;
#pragma unroll
  for (int P = 0; P < NPB; ++P) {
    const int hp = P * PPB + tid / B_CH;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) j += hp >= q * p.HW;
    const int c = hp - j * p.HW;
    b_hw[P] = (hp < p.HP ? 2 * j - 2 + r5 : -4096) * 65536 + ((c - 2) & 0xffff);
  }
  const u16* const zp = reinterpret_cast<const u16*>(p.zero);

  u32x4_t sa[2], sb[NPB][2];
  int tb = 0, ths = 0, tws = 0;
  auto tile_base = [&](int k0) {
    tb = (int)p.dImg.div((uint32_t)k0);
    const int rem = k0 - tb * (p.Hs * p.Ws);
    ths = (int)p.dW.div((uint32_t)rem);
    tws = rem - ths * p.Ws;
  };
  auto load_a = [&](int k0) {
    const u16* s = p.small + (size_t)k0 * p.Cs + a_off0;
    sa[0] = ld16(s);
    sa[1] = ld16(s + p.small_plane);
  };
  auto load_b = [&](int P) {
    const int hb = 2 * ths + (b_hw[P] >> 16), wb = 2 * tws + (int)(short)(b_hw[P] & 0xffff);
    const bool ok = (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
    const u16* s = ok ? p.big + ((size_t)(tb * p.Hb + hb) * p.Wb + wb) * p.Cb + n0 + b_c8 * 8 : zp;
    sb[P][0] = ld16(s);
    if constexpr (!X2) sb[P][1] = ld16(ok ? s + p.big_plane : zp);
  };
  auto write_a = [&](unsigned char* st) {
    *reinterpret_cast<u32x4_t*>(st + a_dst) = sa[0];
    *reinterpret_cast<u32x4_t*>(st + A_PLANE + a_dst) = sa[1];
  };
  auto write_b = [&](unsigned char* st, int P) {
    *reinterpret_cast<u32x4_t*>(st + 2 * A_PLANE + P * PPB * SB + b_dst0) = sb[P][0];
    if constexpr (!X2) *reinterpret_cast<u32x4_t*>(st + 2 * A_PLANE + B_PLANE + P * PPB * SB + b_dst0) = sb[P][1];
  };

  // fragment offsets: lane group g contracts pixels kk = 4g + (i >> 2) and kk + 16 (i = lane % 16)
  const int i16 = lane & 15, g4 = lane >> 4;
  const int kk0 = 4 * g4 + (i16 >> 2);
  const int a_frag = kk0 * SA + (wm * 64 + 4 * (i16 & 3)) * 2;
  int b_frag[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kk = kk0 + 16 * h;
    const int j = kk >> p.lgWt, w = kk - (j << p.lgWt);
    b_frag[h] = (j * p.HW + 2 * w) * SB + (wn * 32 + 4 * (i16 & 3)) * 2;
  }

  f32x4_t acc[5][4][2];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][i][j][r] = 0.f;

  bf16x8_t ah[4], al[4];
  auto load_afrags = [&](const unsigned char* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = tr_frag(st + a_frag + i * 32, 16 * SA);
      al[i] = tr_frag(st + A_PLANE + a_frag + i * 32, 16 * SA);
    }
  };
  auto tr_frag2 = [&](const unsigned char* p0, const unsigned char* p1) {
    const bf16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(p0));
    const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(p1));
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto tap = [&](const unsigned char* st, int q) {
    const unsigned char* b0 = st + 2 * A_PLANE + b_frag[0] + q * SB;
    const unsigned char* b1 = st + 2 * A_PLANE + b_frag[1] + q * SB;
    bf16x8_t bh[2], bl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = tr_frag2(b0 + j * 32, b1 + j * 32);
      if constexpr (!X2) bl[j] = tr_frag2(b0 + B_PLANE + j * 32, b1 + B_PLANE + j * 32);
      else bl[j] = bh[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[q][i][j] = mfma_split16<MODE>(ah[i], al[i], bh[j], bl[j], acc[q][i][j]);
  };

  if (nk > 0) {
    tile_base(k_begin);
    load_a(k_begin);
#pragma unroll
    for (int P = 0; P < NPB; ++P) load_b(P);
    write_a(lds);
#pragma unroll
    for (int P = 0; P < NPB; ++P) write_b(lds, P);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    unsigned char* nxt = lds + ((kt & 1) ^ 1) * STAGE;
    const bool more = kt + 1 < nk;
    const int k1 = k_begin + (kt + 1) * 32;
    // the next tile is staged in four phases of two 16-B loads (eight staging registers live beside 160 accumulators and the
    // 48 fragment registers), each written to the other LDS stage one tap later
    if (more) { tile_base(k1); load_a(k1); }
    load_afrags(cur);
    tap(cur, 0);
    if (more) { write_a(nxt); load_b(0); }
    tap(cur, 1);
    if (more) { write_b(nxt, 0); load_b(1); }
    tap(cur, 2);
    if (more) { write_b(nxt, 1); load_b(2); }
    tap(cur, 3);
    if (more) write_b(nxt, 2);
    tap(cur, 4);
    __syncthreads();
  }

  // 16 x 16 accumulator tile: lane -> column lane % 16, rows 4 * (lane / 16) + r
  float* const sl = p.slab + ((size_t)split * 25 + 5 * r5) * p.Cs * p.Cb;
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + 16 * i + 4 * g4 + r;
          const int n = n0 + wn * 32 + 16 * j + i16;
          float v = acc[q][i][j][r];
          if constexpr (MODE != 0) v *= p.alpha;
          sl[((size_t)q * p.Cs + m) * p.Cb + n] = v;
        }
  if (p.dbg && tid == 0) {
    unsigned long long* d = p.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t0; d[1] = r0; d[2] = __builtin_amdgcn_s_memtime(); d[3] = __builtin_amdgcn_s_memrealtime();
  }
}


#endif  // VP_WGRAD5_Q

// ---- exact fp32: the same workgroup on v_mfma_f32_32x32x2_f32 ---------------------------------------------------------------------
// The exact-f32 plan's weight gradients ran igemm_kernel<ProbW> (one tap per workgroup, 45-55 % of the fp32-MFMA peak).  Same
// row-of-taps structure on fp32 operands: `small` tile [32 px][128 ch] (16 KB) and the halo patch [<= 96 px][BN ch] of `big`, pixel
// major; an MFMA step contracts two pixels (lane half = pixel), its operands are single ds_read_b32 -- 32 consecutive channels per
// lane group, conflict-free at any pitch -- 112 reads against 160 MFMAs of 64 cycles per K-tile and wave.  LGWT = log2 of the K-tile's
// row width Wt (8 | 16 | 32) is a template parameter so that every fragment offset is an instruction immediate.
struct ProbW5F {
  const float* big; const float* small; float* slab;
  int Hs, Ws, Hb, Wb, Cs, Cb, K, nsplit, k_per_split, tiles_n, inner, total, g8;
  FastDiv dImg, dW;
};

template <int BN, int LGWT>
__global__ void __launch_bounds__(512) wgrad5f_kernel(const ProbW5F p) {
  static_assert(BN == 128 || BN == 64, "big-channel tile");
  constexpr int NT = 512, BM = 128;
  constexpr int WN = BN / 32, KG = 8 / (2 * WN);          // wave columns; K groups (BN = 64: two groups of eight pixel pairs)
  constexpr int Wt = 1 << LGWT, HW = 2 * Wt + 4, HP = (32 / Wt) * HW;
  constexpr int SA = BM * 4, SB = BN * 4;                 // row pitches (bytes)
  constexpr int HPA = 96;
  constexpr int A_BYTES = 32 * SA, B_BYTES = HPA * SB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_CH = BM / 4, B_CH = BN / 4;             // 16-B chunks per pixel
  constexpr int PPA = NT / A_CH, NPA = 32 / PPA;          // A: 16 pixels per pass, 2 passes
  constexpr int PPB = NT / B_CH, NPB = HPA / PPB;         // B: 16 | 32 halo pixels per pass, 6 | 3 passes
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (2 * WN), wr = wave % (2 * WN), wm = wr / WN, wn = wr % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int item = (int)(blockIdx.x & 7) * p.g8 + (int)(blockIdx.x >> 3);
  if (item >= p.total) return;
  const int split = item / p.inner, in = item - split * p.inner;
  const int t = in / 5, r5 = in - 5 * t;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * p.k_per_split;
  const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
  const int nk = (k_end - k_begin) >> 5;

  const int a_px = tid / A_CH, a_c4 = tid % A_CH;
  const size_t a_off0 = (size_t)a_px * p.Cs + m0 + a_c4 * 4;
  const int a_dst = a_px * SA + a_c4 * 16;
  const int b_c4 = tid % B_CH;
  int b_hw[NPB];                               // (2j - 2 + r5) << 16 | (c - 2) & 0xffff; row -4096 = past the patch
  const int b_dst0 = (tid / B_CH) * SB + b_c4 * 16;
#pragma unroll
  for (int P = 0; P < NPB; ++P) {
    const int hp = P * PPB + tid / B_CH;
    const int j = hp / HW, c = hp - j * HW;    // HW is a compile-time constant
    b_hw[P] = (hp < HP ? 2 * j - 2 + r5 : -4096) * 65536 + ((c - 2) & 0xffff);
  }

  vp_f32x4 sa[NPA], sb[NPB];
  int tb = 0, ths = 0, tws = 0;
  auto tile_base = [&](int k0) {
    tb = (int)p.dImg.div((uint32_t)k0);
    const int rem = k0 - tb * (p.Hs * p.Ws);
    ths = (int)p.dW.div((uint32_t)rem);
    tws = rem - ths * p.Ws;
  };
  auto load_a = [&](int k0) {
#pragma unroll
    for (int P = 0; P < NPA; ++P) sa[P] = ld4(p.small + (size_t)(k0 + P * PPA) * p.Cs + a_off0);
  };
  auto load_b = [&](int P) {
    const int hb = 2 * ths + (b_hw[P] >> 16), wb = 2 * tws + (int)(short)(b_hw[P] & 0xffff);
    const bool ok = (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
    const vp_f32x4 v = ld4(p.big + ((size_t)(tb * p.Hb + (ok ? hb : 0)) * p.Wb + (ok ? wb : 0)) * p.Cb + n0 + b_c4 * 4);
    sb[P] = ok ? v : zero4();                  // (the select is consumed by the LDS write, a phase after the load was issued)
  };
  auto write_a = [&](unsigned char* st) {
#pragma unroll
    for (int P = 0; P < NPA; ++P) *reinterpret_cast<vp_f32x4*>(st + P * PPA * SA + a_dst) = sa[P];
  };
  auto write_b = [&](unsigned char* st, int P) { *reinterpret_cast<vp_f32x4*>(st + A_BYTES + P * PPB * SB + b_dst0) = sb[P]; };

  // fragments: MFMA step kp contracts the pixels 2 kp + lh of the K-tile; lane = channel
  const int a_frag = lh * SA + (wm * 64 + li) * 4;
  const int b_frag = 2 * lh * SB + (wn * 32 + li) * 4;          // halo column of pixel lh (w = lh: Wt >= 8) + tap 0

  f32x16_t acc[5][2];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;

  // BN = 64: K group kg contracts the pixels [16 kg, 16 kg + 16) of the tile; their fragment offsets are the first group's plus a
  // constant (16 pixels further in `small`; in the halo patch 32 columns, one image row or two, by the tile's row width)
  constexpr int KG_A = 16 * SA, KG_B = (Wt == 32 ? 32 : (16 / Wt) * HW) * SB;
  const int a_frag_g = a_frag + (KG == 1 ? 0 : kg * KG_A), b_frag_g = b_frag + (KG == 1 ? 0 : kg * KG_B);
  auto compute = [&](const unsigned char* st, int kp0, int kp1) {
#pragma unroll
    for (int kp = kp0; kp < kp1; ++kp) {
      const int kk = 2 * kp, j = kk >> LGWT, w = kk & (Wt - 1);      // compile-time after unrolling
      const unsigned char* ap = st + a_frag_g + kk * SA;
      const float a0 = *reinterpret_cast<const float*>(ap), a1 = *reinterpret_cast<const float*>(ap + 128);
      const unsigned char* bp = st + A_BYTES + b_frag_g + (j * HW + 2 * w) * SB;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const float b = *reinterpret_cast<const float*>(bp + q * SB);
        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[q][1], 0, 0, 0);
      }
    }
  };
  constexpr int NKP = 16 / KG;                 // pixel pairs per wave and K-tile

  if (nk > 0) {
    tile_base(k_begin);
    load_a(k_begin);
#pragma unroll
    for (int P = 0; P < NPB; ++P) load_b(P);
    write_a(lds);
#pragma unroll
    for (int P = 0; P < NPB; ++P) write_b(lds, P);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
    unsigned char* nxt = lds + ((kt & 1) ^ 1) * STAGE;
    const bool more = kt + 1 < nk;
    const int k1 = k_begin + (kt + 1) * 32;
    constexpr int H1 = NPB / 2;                 // halo passes staged in the first phase
    if (more) {
      tile_base(k1);
      load_a(k1);
#pragma unroll
      for (int P = 0; P < H1; ++P) load_b(P);
    }
    compute(cur, 0, NKP / 2);
    if (more) {
      write_a(nxt);
#pragma unroll
      for (int P = 0; P < H1; ++P) write_b(nxt, P);
#pragma unroll
      for (int P = H1; P < NPB; ++P) load_b(P);
    }
    compute(cur, NKP / 2, NKP);
    if (more) {
#pragma unroll
      for (int P = H1; P < NPB; ++P) write_b(nxt, P);
    }
    __syncthreads();
  }
  float* const sl = p.slab + ((size_t)(split * KG + kg) * 25 + 5 * r5) * p.Cs * p.Cb;
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = n0 + wn * 32 + li;
        sl[((size_t)q * p.Cs + m) * p.Cb + n] = acc[q][i][r];
      }
}

#endif  // __HIPCC__

// ---- host side ------------------------------------------------------------------------------------------------------------
// Shapes the row-of-taps kernel takes: plain 5x5 stride-2 layers (Hb = 2*Hs), 128 | Cs, Cb a multiple of 128 or exactly 64,
// K-tiles of 32 pixels that are whole image rows (Ws in {8, 16}) or row segments (32 | Ws), 32-bit element offsets.
inline int wgrad5_bn(const ConvGeom& g) {
  if (g.ks != 5 || g.stride != 2 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return 0;
  if (g.Cs % 128 != 0) return 0;
  const int bn = g.Cb % 128 == 0 ? 128 : (g.Cb == 64 ? 64 : 0);
  if (!bn) return 0;
  const bool wok = g.Ws % 32 == 0 || g.Ws == 16 || g.Ws == 8;
  if (!wok || ((long)g.Hs * g.Ws) % 32 != 0) return 0;
  if ((size_t)g.B * g.Hb * g.Wb * g.Cb >= ((size_t)1 << 31) || (size_t)g.B * g.Hs * g.Ws * g.Cs >= ((size_t)1 << 31)) return 0;
  return bn;
}

// pixel ranges: one workgroup per CU (the kernel holds 160 accumulators per lane at two waves per SIMD), i.e. ~256 work items
// `items` = work items (= CUs, the kernel owns one per work item) the launch may take; <= 0: the whole chip.  A caller that runs the
// weight gradient beside other kernels leaves part of the chip to them: on the side stream of the fused step ~5/8 of the chip is
// the measured optimum (step 3.588 ms with the one-tap kernels; rows of taps with 256 work items 3.52, 176: 3.479, 160: 3.455,
// 144: 3.471, 128: 3.484, 96: 3.67 -- profiles/r03_notes.md): the main stream's kernels keep the rest instead of queueing behind
// one-workgroup-per-CU launches, and the slabs shrink with the count.
inline int wgrad5_nsplit(const ConvGeom& g, int bn, int* k_per_split, int items = 0) {
  const long K = (long)g.B * g.Hs * g.Ws;
  const long inner = (long)(g.Cs / 128) * (g.Cb / bn) * 5;
  long target = items > 0 ? items : 256;
  if (const char* e = VP_GETENV("VP_WGRAD5_BLOCKS")) target = atol(e);       // A/B knob: overrides the caller's count
  long ns = target / inner;
  if (ns < 1) ns = 1;
  const long maxs = K / 128 > 0 ? K / 128 : 1;                               // at least four K-tiles per split
  if (ns > maxs) ns = maxs;
  long per = ((K + ns - 1) / ns + 31) / 32 * 32;
  ns = (K + per - 1) / per;                                                  // no empty split
  *k_per_split = (int)per;
  return (int)ns;
}

inline size_t wgrad5_slab_floats(const ConvGeom& g, int bn, int ns) {
  return (size_t)ns * (bn == 64 ? 2 : 1) * 25 * g.Cs * g.Cb;
}

#if defined(__HIPCC__)
// exact fp32 (wgrad5f_kernel): same shapes, same split rule; 16-B aligned operands
inline void wgrad5f_launch(const float* big, const float* small, float* slab, const ConvGeom& g, int bn, int ns, int k_per_split,
                           hipStream_t stream, int* slab_splits) {
  ProbW5F p;
  p.big = big; p.small = small; p.slab = slab;
  p.Hs = g.Hs; p.Ws = g.Ws; p.Hb = g.Hb; p.Wb = g.Wb; p.Cs = g.Cs; p.Cb = g.Cb;
  p.K = g.B * g.Hs * g.Ws; p.nsplit = ns; p.k_per_split = k_per_split;
  p.tiles_n = g.Cb / bn; p.inner = (g.Cs / 128) * p.tiles_n * 5; p.total = p.inner * ns; p.g8 = (p.total + 7) / 8;
  p.dImg = g.dHW; p.dW = g.dW;
  const int Wt = g.Ws < 32 ? g.Ws : 32;
  const dim3 grid((unsigned)(8 * p.g8));
#define VP_W5F(BN_, LG_) hipLaunchKernelGGL((wgrad5f_kernel<BN_, LG_>), grid, dim3(512), 0, stream, p)
  if (bn == 128) { if (Wt == 32) VP_W5F(128, 5); else if (Wt == 16) VP_W5F(128, 4); else VP_W5F(128, 3); }
  else { if (Wt == 32) VP_W5F(64, 5); else if (Wt == 16) VP_W5F(64, 4); else VP_W5F(64, 3); }
#undef VP_W5F
  *slab_splits = ns * (bn == 64 ? 2 : 1);
}

// fills the descriptor and launches; the caller reduces `slab_splits` slabs
template <int MODE>
inline void wgrad5_launch(const void* big_split, const void* small_split, float* slab, const ConvGeom& g, int bn, int ns, int k_per_split,
                          float alpha, hipStream_t stream, int* slab_splits, unsigned long long* dbg = nullptr, bool m16 = false) {
  ProbW5 p;
  p.big = (const u16*)big_split; p.big_plane = (size_t)g.B * g.Hb * g.Wb * g.Cb;
  p.small = (const u16*)small_split; p.small_plane = (size_t)g.B * g.Hs * g.Ws * g.Cs;
  p.slab = slab; p.zero = vp_zero_page();
  p.Hs = g.Hs; p.Ws = g.Ws; p.Hb = g.Hb; p.Wb = g.Wb; p.Cs = g.Cs; p.Cb = g.Cb;
  p.K = g.B * g.Hs * g.Ws; p.nsplit = ns; p.k_per_split = k_per_split;
  p.tiles_n = g.Cb / bn; p.inner = (g.Cs / 128) * p.tiles_n * 5; p.total = p.inner * ns; p.g8 = (p.total + 7) / 8;
  const int Wt = g.Ws < 32 ? g.Ws : 32;
  p.lgWt = Wt == 32 ? 5 : (Wt == 16 ? 4 : 3);
  p.HW = 2 * Wt + 4; p.HP = (32 / Wt) * p.HW;
  p.dImg = g.dHW; p.dW = g.dW;
  p.alpha = alpha; p.dbg = dbg;
  const dim3 grid((unsigned)(8 * p.g8));
#if defined(VP_WGRAD5_Q)
  if (bn == 128 && m16) { hipLaunchKernelGGL((wgrad5q_kernel<MODE>), grid, dim3(512), 0, stream, p); *slab_splits = ns; return; }
#endif
  (void)m16;
  if (bn == 128) hipLaunchKernelGGL((wgrad5_kernel<128, MODE>), grid, dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((wgrad5_kernel<64, MODE>), grid, dim3(512), 0, stream, p);
  *slab_splits = ns * (bn == 64 ? 2 : 1);
}
#endif

}  // namespace vp
