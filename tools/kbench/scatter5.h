// PROTOTYPE (tools/kbench only, not in the library): forward of the 5x5 / stride-2 / padding-2 transposed convolution on split planes,
// TWO PHASES of an output tile per workgroup from ONE halo patch of the small image (round 3).  Measured against the library kernel in
// sbench (random data) and in the step (profiles/r03_notes.md section 9): 1.04 - 1.25x faster stand-alone, NOT faster in the step, where
// the library kernel -- power-limited -- gains 20 % from the ReLU-sparse activations and this one -- bound by its barriers and LDS -- does not.
//
//   big[b, 2q + ph, 2p + pw, n] = sum_{rp, qp, c} small[b, q + 1 - rp, p + 1 - qp, c] * W[c][n][ph + 2 rp][pw + 2 qp]
//   (nn.ConvTranspose2d(k5, s2, p2, output_padding 1) forward, models/networks.py:38)
//
// igemm16_kernel<ProbT16T> makes every phase its own workgroup set and every tap a K-tile that re-stages its 128 pixels x 64 channels from
// L2, although the 9 + 6 + 6 + 4 taps of the four phases read the same 3 x 3 neighbourhood of the small image.  Here a workgroup owns
// 8 x 16 small pixels x 64 output channels x one PHASE GROUP -- A = phases (0,0) + (1,1), 9 + 4 taps; B = (0,1) + (1,0), 6 + 6 taps:
//   * per 64-channel chunk the (8 + 2) x (16 + 2) halo patch is staged ONCE (52 KB, both planes) in pixel-major rows of 144 B; a tap is a
//     wave-uniform byte offset -(rp * 18 + qp) * 144 on the lane's fragment address;
//   * the weights of TWO taps of one phase are staged per step, double-buffered, one barrier per step (7 | 6 steps per chunk);
//   * eight waves = 2 pixel halves x 4 K quarters (2 taps x 2 halves of the 64-channel chunk); a wave keeps 2 phases x 64 pixels x 64
//     channels = 128 accumulator registers and reads 0.67 KB of fragments per MFMA like the 64 x 64 wave tiles of igemm16_kernel; the K
//     quarters are added through LDS in a fixed order (bit-reproducible), then stored; optional BatchNorm statistics partials;
//   * the phase group is a property of the WORKGROUP (two branched bodies inside one workgroup made the compiler spill ~400 registers
//     at every merge of the 128 accumulators).
// L2 -> LDS bytes per MFMA fall 2.5x (944 KB per tile against 2.4 MB for the same products); 254 VGPRs, no spill, 126 KB of LDS.
#pragma once
#include <type_traits>
#include "../../vae_play_amd/csrc/igemm16.h"

namespace vp {

struct ProbS5 {
  const u16* small; size_t small_plane;    // [B][Hs][Ws][Cs]: hi plane, lo plane at + small_plane (elements)
  const u16* w; size_t w_plane;            // packed P1 planes [Cb][25][Cs]
  float* out;                              // [B][2 Hs][2 Ws][Cb]
  const void* zero;
  int Hs, Ws, Cs, Cb;
  int tiles_p, tiles_img, tiles_n;         // Ws / 16, (Hs / 8) * (Ws / 16), Cb / 64
  int total, g8;                           // work items (tile, column block, phase group); g8 = ceil(total / 8)
  float* stat;                             // BatchNorm statistics partials [3][Cb][2 * tiles] (pivot, sum, sum of squares) or null
  unsigned long long* dbg;                 // diagnostics only (tools/kbench)
};

#if defined(__HIPCC__)

struct S5Tap { int tap, off, slot, on; };   // packed-weight tap index, patch offset (pixels, subtracted), accumulator slot, 0 = idle

// GROUP 0: phase (0,0) 3 x 3 taps (slot 0), then (1,1) 2 x 2 (slot 1); GROUP 1: (0,1) 3 x 2 (slot 0), then (1,0) 2 x 3 (slot 1).
// Step `st` stages the taps (st, 0) and (st, 1) of ONE phase; group 0's fifth step has a single tap.
template <int GROUP>
__device__ __forceinline__ constexpr S5Tap s5_tap(int st, int half) {
  if (GROUP == 0) {
    if (st < 5) {
      const int t = 2 * st + half;
      if (t > 8) return {0, 0, 0, 0};
      const int rp = t / 3, qp = t % 3;
      return {(2 * rp) * 5 + 2 * qp, rp * 18 + qp, 0, 1};
    }
    const int t = 2 * (st - 5) + half, rp = t / 2, qp = t % 2;
    return {(1 + 2 * rp) * 5 + 1 + 2 * qp, rp * 18 + qp, 1, 1};
  }
  if (st < 3) { const int t = 2 * st + half, rp = t / 2, qp = t % 2; return {(2 * rp) * 5 + 1 + 2 * qp, rp * 18 + qp, 0, 1}; }
  const int t = 2 * (st - 3) + half, rp = t / 3, qp = t % 3;
  return {(1 + 2 * rp) * 5 + 2 * qp, rp * 18 + qp, 1, 1};
}

template <int MODE, int GROUP>
__device__ __forceinline__ void scatter5_body(const ProbS5& p, unsigned char* lds, int tile, int tn) {
  constexpr int NT = 512, PW = 18, NPIX = 10 * PW, SR = MkStride<64>::bytes;          // 144-B pixel / weight rows
  constexpr int A_PLANE = ((NPIX * SR + 127) / 128) * 128 + 64, A_BYTES = 2 * A_PLANE;
  constexpr int B_PLANE = 64 * SR + 64, B_STAGE = 4 * B_PLANE;                        // [tap of the step][plane][64 columns]
  constexpr int NPA = (NPIX * 8 + NT - 1) / NT;                                       // staging passes of the patch: 3
  constexpr int NST = GROUP == 0 ? 7 : 6;                                             // steps per channel chunk
  static_assert(A_BYTES + 2 * B_STAGE <= 131072, "LDS budget");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mh = wave >> 2, th = (wave >> 1) & 1, kh = wave & 1;                      // pixel half, tap of the step, half of the chunk
  const int li = lane & 31, lh = lane >> 5;
  const int b = tile / p.tiles_img, rem = tile - b * p.tiles_img;
  const int tq = rem / p.tiles_p, tp = rem - tq * p.tiles_p;
  const int q0 = tq * 8, p0 = tp * 16, n0 = tn * 64;

  // ---- staging maps ---------------------------------------------------------------------------------------------------------
  const u16* const zp = reinterpret_cast<const u16*>(p.zero);
  int a_src[NPA], a_dst[NPA];                 // element offset into `small` (-1: outside the image or past the patch), LDS byte offset
#pragma unroll
  for (int P = 0; P < NPA; ++P) {
    const int idx = P * NT + tid, hp = idx >> 3, c8 = idx & 7;
    const int pr = hp / PW, pc = hp - pr * PW;
    const int h = q0 - 1 + pr, w_ = p0 - 1 + pc;
    const bool ok = hp < NPIX && (unsigned)h < (unsigned)p.Hs && (unsigned)w_ < (unsigned)p.Ws;
    a_src[P] = ok ? ((b * p.Hs + h) * p.Ws + w_) * p.Cs + c8 * 8 : -1;
    a_dst[P] = hp < NPIX ? hp * SR + c8 * 16 : -1;
  }
  const int b_n = tid >> 3, b_c8 = tid & 7;
  const size_t b_row = (size_t)(n0 + b_n) * 25 * p.Cs + b_c8 * 8;
  const int b_dst = b_n * SR + b_c8 * 16;

  u32x4_t sa[NPA][2], sb[4];
  auto load_a = [&](int c0) {
#pragma unroll
    for (int P = 0; P < NPA; ++P) {
      const u16* s = a_src[P] >= 0 ? p.small + (size_t)a_src[P] + c0 : zp;
      sa[P][0] = ld16(s);
      sa[P][1] = ld16(a_src[P] >= 0 ? s + p.small_plane : zp);
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int P = 0; P < NPA; ++P)
      if (a_dst[P] >= 0) {
        *reinterpret_cast<u32x4_t*>(lds + a_dst[P]) = sa[P][0];
        *reinterpret_cast<u32x4_t*>(lds + A_PLANE + a_dst[P]) = sa[P][1];
      }
  };
  auto load_b = [&](int c0, int tap0, int tap1) {             // tap1 < 0: the step has one tap, the second slot is staged as zeros
    const u16* s0 = p.w + b_row + (size_t)tap0 * p.Cs + c0;
    sb[0] = ld16(s0); sb[1] = ld16(s0 + p.w_plane);
    if (tap1 >= 0) {
      const u16* s1 = p.w + b_row + (size_t)tap1 * p.Cs + c0;
      sb[2] = ld16(s1); sb[3] = ld16(s1 + p.w_plane);
    } else {
      sb[2] = zero_u4(); sb[3] = zero_u4();
    }
  };
  auto tap_of = [](const S5Tap& t) { return t.on ? t.tap : -1; };
  auto write_b = [&](unsigned char* st) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4_t*>(st + u * B_PLANE + b_dst) = sb[u];
  };

  f32x16_t acc[2][2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

  // fragment bases: row block i of the wave = tile rows mh*4 + 2i + (li >> 4), column li & 15; patch pixel = (+2, +2) - tap offset
  const int a_lane = (((mh * 4 + (li >> 4)) + 2) * PW + (li & 15) + 2) * SR + kh * 64 + lh * 16;
  const int b_lane = A_BYTES + (th * 2) * B_PLANE + li * SR + kh * 64 + lh * 16;

  auto mma = [&](auto slot_c, int off, const unsigned char* bst) {
    constexpr int SLOT = decltype(slot_c)::value;
    const unsigned char* ap = lds + a_lane - off * SR;
    const unsigned char* bp = bst + b_lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(ap + i * 2 * PW * SR + s * 32));
        al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(ap + A_PLANE + i * 2 * PW * SR + s * 32));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(bp + j * 32 * SR + s * 32));
        bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(bp + B_PLANE + j * 32 * SR + s * 32));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[SLOT][i][j] = mfma_split<MODE>(ah[i], al[i], bh[j], bl[j], acc[SLOT][i][j]);
    }
  };

  const int nch = p.Cs >> 6;
  load_a(0);
  load_b(0, s5_tap<GROUP>(0, 0).tap, tap_of(s5_tap<GROUP>(0, 1)));
  write_a();
  write_b(lds + A_BYTES);
  __syncthreads();
  int cur = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const int c0 = ch << 6;
    const bool more_ch = ch + 1 < nch;
    auto step = [&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr bool last = st == NST - 1;
      if constexpr (!last) load_b(c0, s5_tap<GROUP>(st + 1, 0).tap, tap_of(s5_tap<GROUP>(st + 1, 1)));
      else if (more_ch) load_b(c0 + 64, s5_tap<GROUP>(0, 0).tap, tap_of(s5_tap<GROUP>(0, 1)));
      const unsigned char* bst = lds + cur * B_STAGE;
      constexpr S5Tap t0 = s5_tap<GROUP>(st, 0), t1 = s5_tap<GROUP>(st, 1);
      static_assert(t0.slot == t1.slot || !t1.on, "both taps of a step belong to one phase");
      mma(std::integral_constant<int, t0.slot>(), th == 0 ? t0.off : t1.off, bst);     // (a missing second tap contracts zero weights)
      if (!last || more_ch) write_b(lds + A_BYTES + (cur ^ 1) * B_STAGE);
      __syncthreads();
      cur ^= 1;
      if (last && more_ch) {               // every wave is past its last read of this chunk's patch: the next chunk's replaces it
        load_a(c0 + 64);
        write_a();
        __syncthreads();
      }
    };
#define VP_S5_STEP(k) if constexpr (k < NST) step(std::integral_constant<int, k>());
    VP_S5_STEP(0) VP_S5_STEP(1) VP_S5_STEP(2) VP_S5_STEP(3) VP_S5_STEP(4) VP_S5_STEP(5) VP_S5_STEP(6)
#undef VP_S5_STEP
  }

  // ---- the four K quarters, added in a fixed order through LDS ((0 += 2, 1 += 3), then 0 += 1); then the stores --------------------
  float* const red = reinterpret_cast<float*>(lds);
  const int kq = wave & 3;                      // th * 2 + kh
  auto park = [&](int slot4) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((slot4 * 8 + s * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[s][i][j][r];
  };
  auto take = [&](int slot4) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[s][i][j][r] += red[((slot4 * 8 + s * 4 + i * 2 + j) * 16 + r) * 64 + lane];
  };
  if (kq >= 2) park(mh * 2 + (kq - 2));         // four waves x 32 KB
  __syncthreads();
  if (kq < 2) take(mh * 2 + kq);
  __syncthreads();
  if (kq == 1) park(mh);
  __syncthreads();
  if (kq == 0) take(mh);
  if (p.stat) {
    // BatchNorm statistics of the workgroup's 256 output pixels (2 phases x 128) per channel, from the final accumulators of the two
    // kq == 0 waves: {pivot = the tile's first pixel, sum(x - pivot), sum((x - pivot)^2)}, wave halves by shuffle, the two waves through
    // LDS in a fixed order; group index = tile * 2 + GROUP, every group holds exactly 256 rows (bn_stats_slab_final_kernel)
    float* const pv = red;                      // [64] pivots
    float* const sq = red + 64;                 // [2 waves][64][2]
    __syncthreads();                            // (the reduction scratch is free again)
    if (kq == 0 && mh == 0 && lh == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) pv[32 * j + li] = acc[0][0][j][0];
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float pvt = pv[32 * j + li];
        float sm = 0.f, qs = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[s][i][j][r] - pvt; sm += d; qs += d * d; }
        sm += __shfl_xor(sm, 32, 64);
        qs += __shfl_xor(qs, 32, 64);
        if (lh == 0) { sq[(mh * 64 + 32 * j + li) * 2] = sm; sq[(mh * 64 + 32 * j + li) * 2 + 1] = qs; }
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int G = p.total / p.tiles_n, gidx = tile * 2 + GROUP;        // groups per channel: (tile, phase group)
      const size_t n = (size_t)(n0 + tid);
      p.stat[(0 * (size_t)p.Cb + n) * G + gidx] = pv[tid];
      p.stat[(1 * (size_t)p.Cb + n) * G + gidx] = sq[tid * 2] + sq[(64 + tid) * 2];
      p.stat[(2 * (size_t)p.Cb + n) * G + gidx] = sq[tid * 2 + 1] + sq[(64 + tid) * 2 + 1];
    }
  }
  if (kq == 0) {
    const int Hb = 2 * p.Hs, Wb = 2 * p.Ws;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ph = s, pw = GROUP == 0 ? s : 1 - s;                   // group 0: (0,0), (1,1); group 1: (0,1), (1,0)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int trow = mh * 4 + 2 * i + (rr >> 4), tcol = rr & 15;
          const int oh = 2 * (q0 + trow) + ph, ow = 2 * (p0 + tcol) + pw;
          float* dst = p.out + ((size_t)(b * Hb + oh) * Wb + ow) * p.Cb + n0 + li;
#pragma unroll
          for (int j = 0; j < 2; ++j) dst[32 * j] = acc[s][i][j][r];
        }
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(512) scatter5_kernel(const ProbS5 p) {
  static_assert(MODE == 0, "bf16 pairs, three products");
  constexpr int LDS_BYTES = 131072;             // patch + two weight stages (126.3 KB); == the K-quarter reduction scratch
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int item = (int)(blockIdx.x & 7) * p.g8 + (int)(blockIdx.x >> 3);             // XCD x owns a contiguous range of items
  if (item >= p.total) return;
  unsigned long long t0 = 0, r0 = 0;
  if (p.dbg && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // item = ((tile * tiles_n) + tn) * 2 + group: the two groups and the column blocks of a tile read the same patch -- neighbours in L2
  const int grp = item & 1, it2 = item >> 1;
  const int tile = it2 / p.tiles_n, tn = it2 - tile * p.tiles_n;
  if (grp == 0) scatter5_body<MODE, 0>(p, lds, tile, tn);
  else scatter5_body<MODE, 1>(p, lds, tile, tn);
  if (p.dbg && threadIdx.x == 0) {
    unsigned long long* d = p.dbg + 4 * (size_t)blockIdx.x;
    d[0] = t0; d[1] = r0; d[2] = __builtin_amdgcn_s_memtime(); d[3] = __builtin_amdgcn_s_memrealtime();
  }
}

#endif  // __HIPCC__

// ---- host side -----------------------------------------------------------------------------------------------------------------
// taken: plain 5x5 stride-2 layers, small image a multiple of 8 x 16 pixels, small channels a multiple of 64, big channels of 64
inline bool scatter5_ok(int B, int Hs, int Ws, int Cs, int Cb) {
  return B > 0 && Hs % 8 == 0 && Ws % 16 == 0 && Cs % 64 == 0 && Cb % 64 == 0 &&
         (size_t)B * Hs * Ws * Cs < ((size_t)1 << 31) && (size_t)Cb * 25 * Cs < ((size_t)1 << 31);
}
inline int scatter5_items(int B, int Hs, int Ws, int Cb) { return B * (Hs / 8) * (Ws / 16) * (Cb / 64) * 2; }

// statistics partials: groups per channel and rows per group (every group is full)
inline int scatter5_stat_groups(int B, int Hs, int Ws) { return B * (Hs / 8) * (Ws / 16) * 2; }
constexpr int kScatter5StatRows = 256;

#if defined(__HIPCC__)

inline void scatter5_launch(const void* small_split, const void* w_p1_split, float* out, int B, int Hs, int Ws, int Cs, int Cb,
                            hipStream_t stream, float* stat = nullptr, unsigned long long* dbg = nullptr) {
  ProbS5 p;
  p.small = (const u16*)small_split; p.small_plane = (size_t)B * Hs * Ws * Cs;
  p.w = (const u16*)w_p1_split; p.w_plane = (size_t)Cs * Cb * 25;
  p.out = out; p.zero = vp_zero_page();
  p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.Cb = Cb;
  p.tiles_p = Ws / 16; p.tiles_img = (Hs / 8) * p.tiles_p; p.tiles_n = Cb / 64;
  p.total = scatter5_items(B, Hs, Ws, Cb);
  p.g8 = (p.total + 7) / 8;
  p.stat = stat;
  p.dbg = dbg;
  hipLaunchKernelGGL((scatter5_kernel<0>), dim3((unsigned)(8 * p.g8)), dim3(512), 0, stream, p);
}
#endif

}  // namespace vp
