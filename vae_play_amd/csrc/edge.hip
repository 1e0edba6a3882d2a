// Weight gradient of the final conv (nn.Conv2d(64 -> C, k5, s1, p2), C = 1 or 3 image channels, models/networks.py:100-103)
// on the matrix cores, split-bf16 arithmetic:
//     dW[n][c][tap] = sum_{b,h,w} dlogit[b,h,w,n] * u[b, h+r-2, w+q-2, c]
// An implicit GEMM with the 3 output channels as a matrix dimension is >= 90 % padding (the VALU kernel of narrow.hip took
// 151 us at B = 32, 128x128).  Here the 25 taps x C gradient channels ARE the row dimension ("taps in M"):
//     dW[(tap, n)][c] = sum_p  D[(tap, n)][p] * u[p][c],      D[(tap, n)][p] = dlogit[p - shift(tap)][n]
// one GEMM with M = 25*C <= 96 rows, N = 64 channels and K = every pixel p of u; D is never materialised:
//   * a workgroup owns a band of TH image rows of one image; the band's dlogit rows (+2 halo rows either side, 2 zero columns
//     left and right) are split to bf16 hi/lo once and parked in LDS, planar per channel; an A fragment (8 consecutive
//     pixels of row (tap, n)) is 8 x ds_read_u16 at a per-lane constant offset (the tap shift) + a wave-uniform offset;
//   * u streams through LDS 64 pixels at a time: coalesced fp32 loads (256 B per pixel), split to bf16 hi/lo in registers,
//     [pixel][channel] image with the 192-B row pitch of igemm16.h's pixel-major operands, B fragments by the transposing
//     read ds_read_b64_tr_b16; double-buffered, one barrier per 64 pixels;
//   * waves 0..2 own one 32-row block of (tap, n) each (two 32x32 accumulators: channels 0-31 / 32-63), wave 3 only stages;
//   * every workgroup writes its [25*C][64] partial to a slab (layout [tap][n][c] = narrow.hip's), reduced in fixed order by
//     slab_reduce_deep_kernel: bit-reproducible, no atomics.
// The kernel is bound by streaming u once (134 MB at B = 32): ~30 us.
#include <stdlib.h>
#include <stdint.h>
#include "common.h"
#include "igemm16.h"
#include "narrow.h"
#include "split.h"

namespace vp {

template <int NOUT, int TH>
__global__ void __launch_bounds__(256, 2) wgrad_tapm_kernel(const float* __restrict__ u, const float* __restrict__ dlogit,
                                                            float* __restrict__ slab, int H, int W, int bands_per_img) {
  constexpr int C = 64, NCOL = 25 * NOUT, MT = (NCOL + 31) / 32;
  constexpr int SS = 64;                                   // pixels per staged super-step (4 MFMA k-steps)
  constexpr int US = KmStride<64>::bytes;                  // 192-B row pitch of the [pixel][channel] image
  constexpr int UPLANE = ((SS * US + 127) / 128) * 128 + 64;
  constexpr int UBUF = 2 * UPLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int PC = ((W + 4 + 7) / 8) * 8;                    // patch row pitch (elements)
  const int PLANE_E = (TH + 4) * PC;                       // elements per (channel) plane of the dlogit patch
  u16* const dhi = reinterpret_cast<u16*>(smem + 2 * UBUF);
  u16* const dlo = dhi + NOUT * PLANE_E;
  u16* const zeros = dlo + NOUT * PLANE_E;                 // 32 zero elements: fragments of the padding rows m >= 25*C
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / bands_per_img, h0 = (blockIdx.x - b * bands_per_img) * TH;

  // ---- dlogit patch: rows h0-2 .. h0+TH+1, columns -2 .. W+1 (zeros outside the image), split to bf16 hi / lo -----------
  for (int i = tid; i < NOUT * PLANE_E; i += 256) {
    const int n = i / PLANE_E, rem = i - n * PLANE_E;
    const int pr = rem / PC, pc = rem - pr * PC;
    const int h = h0 - 2 + pr, w = pc - 2;
    float v = 0.f;
    if (h >= 0 && h < H && w >= 0 && w < W) v = dlogit[((size_t)(b * H + h) * W + w) * NOUT + n];
    u16_t hh, ll;
    split_f32(v, hh, ll);
    dhi[i] = hh;
    dlo[i] = ll;
  }
  if (tid < 32) zeros[tid] = 0;

  // ---- per-lane constants of the A fragments: row m = 32*wave + li -> (tap, n) -> element offset of the tap shift ----------
  const int m = 32 * wave + li;
  const bool mrow = wave < MT && m < NCOL;
  const int tap = mrow ? m / NOUT : 0, n_ = mrow ? m - tap * NOUT : 0;
  const int r_ = tap / 5, q_ = tap - 5 * r_;
  const int a_off = n_ * PLANE_E + (4 - r_) * PC + (4 - q_) + 8 * lh;      // + (h' - h0) * PC + w0 per k-step

  // ---- u staging: 4 threads per pixel, 16 channels (64 B of fp32) each ---------------------------------------------------
  const int spx = tid >> 2, sch = (tid & 3) * 16;
  vp_f32x4 st[4];
  const int nss = TH * W / SS;                             // super-steps of this band (W % 64 == 0)
  auto load_ss = [&](int ss) {
    const int p0 = ss * SS + spx;                          // pixel index inside the band (row-major)
    const int hr = p0 / W, w = p0 - hr * W;
    const float* src = u + ((size_t)(b * H + h0 + hr) * W + w) * C + sch;
#pragma unroll
    for (int v = 0; v < 4; ++v) st[v] = ld4(src + 4 * v);
  };
  auto write_ss = [&](int buf) {
    unsigned char* base = smem + buf * UBUF + spx * US + sch * 2;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      u16x4_t h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { u16_t a, c; split_f32(st[v][j], a, c); h[j] = a; l[j] = c; }
      *reinterpret_cast<u16x4_t*>(base + 8 * v) = h;
      *reinterpret_cast<u16x4_t*>(base + UPLANE + 8 * v) = l;
    }
  };

  f32x16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  load_ss(0);
  write_ss(0);
  __syncthreads();
  for (int ss = 0; ss < nss; ++ss) {
    const bool more = ss + 1 < nss;
    if (more) load_ss(ss + 1);
    if (wave < MT) {
      const unsigned char* ub = smem + (ss & 1) * UBUF;
      const int pbase = ss * SS;                           // first pixel of the super-step inside the band
      const int hr = pbase / W, w0 = pbase - hr * W;      // SS divides W: one image row per super-step
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // A: 8 consecutive pixels of row (tap, n), both planes
        const int e = mrow ? a_off + hr * PC + w0 + 16 * s : 0;
        const u16* ph = mrow ? dhi + e : zeros;
        const u16* pl = mrow ? dlo + e : zeros;
        bf16x8_t ah, al;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ah[j] = __builtin_bit_cast(__bf16, ph[j]);
          al[j] = __builtin_bit_cast(__bf16, pl[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8_t bh = frag16<64, true, 64>(ub, 32 * j + li, s, li, lh, lane);
          const bf16x8_t bl = frag16<64, true, 64>(ub + UPLANE, 32 * j + li, s, li, lh, lane);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      }
    }
    if (more) write_ss((ss + 1) & 1);
    __syncthreads();
  }
  if (wave < MT) {
    float* out = slab + (size_t)blockIdx.x * NCOL * C;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < NCOL) out[(size_t)row * C + 32 * j + li] = acc[j][r];
      }
  }
}

// Exact-fp32 form of the taps-in-M weight gradient (round 3): the same band / patch / slab scheme on v_mfma_f32_32x32x2_f32.
//   * the dlogit patch sits in LDS as fp32; an A element of row (tap, n) and pixel p is ONE ds_read_b32 at the lane's tap shift;
//   * u streams through LDS 64 pixels at a time as fp32, [pixel][channel] with a 96-float pitch: an MFMA step contracts the pixel pair
//     (2j, 2j + 1), lane half lh takes pixel 2j + lh, and the two halves of a B read fall into disjoint bank halves;
//   * all four waves compute: each owns 16 of a super-step's 64 pixels and ALL 3 x 2 accumulator blocks (96 registers) -- the K split
//     keeps the four SIMDs equally loaded where "one 32-row block per wave" leaves one idle -- and the four partial sums are added in
//     a fixed order through LDS before the slab is written (bit-reproducible);
//   * 48 MFMAs of 64 cycles per wave and super-step: 41 us of matrix time at B = 32, 128 x 128 (the VALU kernel took 151 us).
template <int NOUT, int TH>
__global__ void __launch_bounds__(256, 2) wgrad_tapm_f32_kernel(const float* __restrict__ u, const float* __restrict__ dlogit,
                                                                float* __restrict__ slab, int H, int W, int bands_per_img) {
  constexpr int C = 64, NCOL = 25 * NOUT, MT = (NCOL + 31) / 32;
  constexpr int SS = 64, UP = 96;                          // pixels per super-step, u row pitch (floats)
  constexpr int UBUF = SS * UP;                            // floats per u buffer
  static_assert(MT <= 3, "row blocks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const ubuf = reinterpret_cast<float*>(smem);
  float* const patch = ubuf + 2 * UBUF;
  const int PC = ((W + 4 + 7) / 8) * 8;
  const int PLANE_E = (TH + 4) * PC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / bands_per_img, h0 = (blockIdx.x - b * bands_per_img) * TH;

  for (int i = tid; i < NOUT * PLANE_E; i += 256) {
    const int n = i / PLANE_E, rem = i - n * PLANE_E;
    const int pr = rem / PC, pc = rem - pr * PC;
    const int h = h0 - 2 + pr, w = pc - 2;
    patch[i] = (h >= 0 && h < H && w >= 0 && w < W) ? dlogit[((size_t)(b * H + h) * W + w) * NOUT + n] : 0.f;
  }

  // per-lane constants: rows m = 32*rb + li -> (tap, n) -> element offset of the tap shift (+ this wave's pixel quarter + lane half)
  int a_off[MT];
  bool mrow[MT];
#pragma unroll
  for (int rb = 0; rb < MT; ++rb) {
    const int m = 32 * rb + li;
    mrow[rb] = m < NCOL;
    const int tap = mrow[rb] ? m / NOUT : 0, n_ = mrow[rb] ? m - tap * NOUT : 0;
    const int r_ = tap / 5, q_ = tap - 5 * r_;
    a_off[rb] = n_ * PLANE_E + (4 - r_) * PC + (4 - q_) + 16 * wave + lh;
  }

  const int spx = tid >> 2, sch = (tid & 3) * 16;
  vp_f32x4 st[4];
  const int nss = TH * W / SS;
  auto load_ss = [&](int ss) {
    const int p0 = ss * SS + spx;
    const int hr = p0 / W, w = p0 - hr * W;
    const float* src = u + ((size_t)(b * H + h0 + hr) * W + w) * C + sch;
#pragma unroll
    for (int v = 0; v < 4; ++v) st[v] = ld4(src + 4 * v);
  };
  auto write_ss = [&](int buf) {
    float* base = ubuf + buf * UBUF + spx * UP + sch;
#pragma unroll
    for (int v = 0; v < 4; ++v) *reinterpret_cast<vp_f32x4*>(base + 4 * v) = st[v];
  };

  f32x16_t acc[MT][2];
#pragma unroll
  for (int rb = 0; rb < MT; ++rb)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][j][r] = 0.f;

  load_ss(0);
  write_ss(0);
  __syncthreads();
  for (int ss = 0; ss < nss; ++ss) {
    const bool more = ss + 1 < nss;
    if (more) load_ss(ss + 1);
    {
      const float* ub = ubuf + (ss & 1) * UBUF + (16 * wave + lh) * UP + li;
      const int pbase = ss * SS;
      const int hr = pbase / W, w0 = pbase - hr * W;
      const float* pa = patch + hr * PC + w0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float b0 = ub[2 * j * UP], b1 = ub[2 * j * UP + 32];
#pragma unroll
        for (int rb = 0; rb < MT; ++rb) {
          float a = pa[a_off[rb] + 2 * j];
          a = mrow[rb] ? a : 0.f;
          acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[rb][0], 0, 0, 0);
          acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[rb][1], 0, 0, 0);
        }
      }
    }
    if (more) write_ss((ss + 1) & 1);
    __syncthreads();
  }
  // the four waves' partial sums, added in a fixed order through the (now free) u buffers: (0 += 2, 1 += 3), then 0 += 1
  constexpr int WREGS = MT * 2 * 16;
  float* const red = ubuf;                                  // 2 x WREGS x 64 floats <= 2 * UBUF
  static_assert(2 * WREGS * 64 <= 2 * UBUF, "reduction scratch");
  auto park = [&](int slot) {
#pragma unroll
    for (int rb = 0; rb < MT; ++rb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((slot * WREGS) + (rb * 2 + j) * 16 + r) * 64 + lane] = acc[rb][j][r];
  };
  auto take = [&](int slot) {
#pragma unroll
    for (int rb = 0; rb < MT; ++rb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][j][r] += red[((slot * WREGS) + (rb * 2 + j) * 16 + r) * 64 + lane];
  };
  if (wave >= 2) park(wave - 2);
  __syncthreads();
  if (wave < 2) take(wave);
  __syncthreads();
  if (wave == 1) park(0);
  __syncthreads();
  if (wave == 0) {
    take(0);
    float* out = slab + (size_t)blockIdx.x * NCOL * C;
#pragma unroll
    for (int rb = 0; rb < MT; ++rb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < NCOL) out[(size_t)row * C + 32 * j + li] = acc[rb][j][r];
        }
  }
}

constexpr int TAPM_TH = 8;        // band height: 8 rows = two workgroups per CU (16 rows, one per CU, left too few loads in flight: 57 us)

bool tapm_wgrad_applicable(const ConvGeom& g) {
  const bool on = true;
  return on && g.ks == 5 && g.stride == 1 && g.Hb == g.Hs && g.Wb == g.Ws && (g.Cs == 1 || g.Cs == 3) && g.Cb == 64 &&
         g.Ws % 64 == 0 && g.Hs % TAPM_TH == 0 && g.Ws <= 512;
}

size_t tapm_wgrad_ws_floats(const ConvGeom& g) { return (size_t)g.B * (g.Hs / TAPM_TH) * kTaps * g.Cs * g.Cb; }

static size_t tapm_lds_bytes(const ConvGeom& g) {
  constexpr int US = KmStride<64>::bytes, UPLANE = ((64 * US + 127) / 128) * 128 + 64;
  const int PC = ((g.Ws + 4 + 7) / 8) * 8;
  return (size_t)2 * 2 * UPLANE + (size_t)2 * g.Cs * (TAPM_TH + 4) * PC * 2 + 64;
}

int tapm_wgrad_launch(const float* big_f32, const float* small_f32, float* dw_ref, const ConvGeom& g, float* ws, hipStream_t s) {
  const int bands = g.Hs / TAPM_TH, nblk = g.B * bands;
  const size_t lds = tapm_lds_bytes(g);
  if (lds > 160 * 1024) return fail(VP_ERR_ARG, "wgrad_tapm: image too wide for the LDS patch");
  if (g.Cs == 3) {
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute((const void*)wgrad_tapm_kernel<3, TAPM_TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr3 = true; }
    hipLaunchKernelGGL((wgrad_tapm_kernel<3, TAPM_TH>), dim3(nblk), dim3(256), lds, s, big_f32, small_f32, ws, g.Hs, g.Ws, bands);
  } else {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)wgrad_tapm_kernel<1, TAPM_TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL((wgrad_tapm_kernel<1, TAPM_TH>), dim3(nblk), dim3(256), lds, s, big_f32, small_f32, ws, g.Hs, g.Ws, bands);
  }
  int rc = check_launch("wgrad_tapm");
  if (rc) return rc;
  return slab_reduce_launch(ws, dw_ref, g.Cs, g.Cb, nblk, s, kTaps);      // few outputs, many slabs: the deep reduction kernel
}

static size_t tapm_f32_lds_bytes(const ConvGeom& g) {
  const int PC = ((g.Ws + 4 + 7) / 8) * 8;
  return (size_t)2 * 64 * 96 * 4 + (size_t)g.Cs * (TAPM_TH + 4) * PC * 4;
}

int tapm_wgrad_f32_launch(const float* big_f32, const float* small_f32, float* dw_ref, const ConvGeom& g, float* ws, hipStream_t s) {
  const int bands = g.Hs / TAPM_TH, nblk = g.B * bands;
  const size_t lds = tapm_f32_lds_bytes(g);
  if (lds > 160 * 1024) return fail(VP_ERR_ARG, "wgrad_tapm_f32: image too wide for the LDS patch");
  if (g.Cs == 3) {
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute((const void*)wgrad_tapm_f32_kernel<3, TAPM_TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr3 = true; }
    hipLaunchKernelGGL((wgrad_tapm_f32_kernel<3, TAPM_TH>), dim3(nblk), dim3(256), lds, s, big_f32, small_f32, ws, g.Hs, g.Ws, bands);
  } else {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)wgrad_tapm_f32_kernel<1, TAPM_TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL((wgrad_tapm_f32_kernel<1, TAPM_TH>), dim3(nblk), dim3(256), lds, s, big_f32, small_f32, ws, g.Hs, g.Ws, bands);
  }
  int rc = check_launch("wgrad_tapm_f32");
  if (rc) return rc;
  return slab_reduce_launch(ws, dw_ref, g.Cs, g.Cb, nblk, s, kTaps);
}


// ---- input gradient of the final conv: one kernel ROW of taps per MFMA k-step ("rows in K") ---------------------------------------
//     du[b,h,w,cf] = sum_{r,q,n} dlogit[b, h-r+2, w-q+2, n] * W[n][cf][r][q]          (nn.Conv2d(64 -> C, k5, s1, p2) backward, C = 1 | 3)
// The implicit GEMM on channel-padded planes (halo.hip: 3 -> 8 channels, K = 25*8 = 200) spends 2.7x the MFMAs the layer needs and
// is MFMA-bound (61 us at B = 32, 128x128) where the layer's floor is writing its 134-MB fp32 result.  In the NHWC gradient image the
// 5 taps of one kernel row are 5*C <= 15 CONTIGUOUS floats, (w-2 .. w+2) x C: they are one 16-deep MFMA k-step (k = 3*d + n, d = the
// column offset, tap q = 4 - d; k >= 5*C is zero), so K = 5 rows x 16 = 80.
//   * a workgroup (8 waves) owns 8 rows x 32 columns of output pixels; the 12 x 36 x C patch of dlogit is staged in LDS as fp32
//     (zeros outside the image: no masks later), a wave takes one output row = one 32-pixel MFMA row tile;
//   * A fragment of kernel row r: 8 consecutive fp32 of patch row (row + 4 - r), split to bf16 hi / lo in registers;
//   * B fragments (5 rows x 64 output channels x 16 k, from the reference weight layout) are gathered and split once per workgroup
//     and parked in LDS in fragment order; workgroups are persistent over the tile list, two per CU, and request the next tile's
//     patch before the current tile's MFMAs and stores;
//   * 30 MFMAs (5 k-steps x 2 column tiles x 3 products) per 32 pixels x 64 channels, then 8 KB of stores.
// FWD = false: the input gradient above (taps flipped, weights [NIN][CF][5][5]);  FWD = true: the FORWARD pass of a stride-1 conv with
// 1 | 3 input channels, big_out[b,h,w,cf] = act(bias[cf] + sum_{r,q,n} small[b, h+r-2, w+q-2, n] * W[cf][n][r][q]) (weights
// [CF][NIN][5][5], act none | relu): the VAE-GAN discriminator's first layer, models/networks.py:160-163.  CF = 64 | 32 wide channels.
template <int NIN, int CF, bool FWD>
__global__ void __launch_bounds__(512, 2) dgrad_rowk_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                                            float* __restrict__ out, int H, int W, int tiles_x, int tiles_per_img,
                                                            int ntiles, const float* __restrict__ bias = nullptr, int act = ACT_NONE) {
  static_assert(CF == 64 || CF == 32, "wide channel count");
  constexpr int NJT = CF / 32;
  constexpr int TR = 8, TC = 32, PR = TR + 4, PC = (TC + 4) * NIN + 8;     // patch: 12 rows x (36 pixels x NIN floats, + slack)
  constexpr int KROW = 5 * NIN;                                                     // useful k per kernel row (<= 15)
  __shared__ float patch[PR * PC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
  typedef float f32x16_l __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) bf16x8_l bfr[5 * NJT * 2 * 64];                              // weight fragments [r][jt][hi|lo][lane]
  auto split8 = [](const float (&x)[8], bf16x8_l& h, bf16x8_l& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hh = (__bf16)x[j];
      h[j] = hh;
      l[j] = (__bf16)(x[j] - (float)hh);
    }
  };
  // B[r][cf][k]: k = NIN*d + n  <->  W[n][cf][r][4 - d]; a lane's fragment: cf = 32*jt + li, k = 8*lh + 0..7.  Gathered and split
  // once per workgroup (wave r builds kernel row r) and parked in LDS in fragment order: registers stay free for two workgroups per CU
  if (wave < 5) {
    const int r = wave;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * lh + j, d = k / NIN, n = k - d * NIN, cf = 32 * jt + li;
        const size_t wi = FWD ? (((size_t)cf * NIN + n) * 5 + r) * 5 + d : (((size_t)n * CF + cf) * 5 + r) * 5 + (4 - d);
        x[j] = k < KROW ? w[wi] : 0.f;
      }
      bf16x8_l h8, l8;
      split8(x, h8, l8);
      bfr[((r * NJT + jt) * 2 + 0) * 64 + lane] = h8;
      bfr[((r * NJT + jt) * 2 + 1) * 64 + lane] = l8;
    }
  }
  // the patch of a tile: PR x (TC+4) x NIN floats, <= 3 per thread; the NEXT tile's values are requested before this tile's MFMAs and
  // stores and parked in registers (one workgroup per CU: nothing else would hide that latency)
  constexpr int NPE = PR * (TC + 4) * NIN, NPV = (NPE + 511) / 512;
  float pv[NPV];
  auto load_patch = [&](int tile) {
    const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
    const int h0 = (rem / tiles_x) * TR, w0 = (rem % tiles_x) * TC;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int i = tid + 512 * u;
      const int pr = i / ((TC + 4) * NIN), e = i - pr * ((TC + 4) * NIN);
      const int px = e / NIN, n = e - px * NIN;
      const int h = h0 - 2 + pr, ww = w0 - 2 + px;
      pv[u] = (i < NPE && h >= 0 && h < H && ww >= 0 && ww < W) ? dlogit[((size_t)(b * H + h) * W + ww) * NIN + n] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) load_patch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
    const int h0 = (rem / tiles_x) * TR, w0 = (rem % tiles_x) * TC;
    __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int i = tid + 512 * u;
      if (i < NPE) { const int pr = i / ((TC + 4) * NIN); patch[pr * PC + (i - pr * ((TC + 4) * NIN))] = pv[u]; }
    }
    if (tid < PR * 8) patch[(tid >> 3) * PC + (TC + 4) * NIN + (tid & 7)] = 0.f;      // the slack a k >= KROW read may touch
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_patch(tile + (int)gridDim.x);
    f32x16_l acc[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      // output pixel (row = wave, column li): gradient row h - r + 2 = patch row wave + 4 - r, columns w-2 .. w+2 = patch pixel li ..
      // (forward: input row h + r - 2 = patch row wave + r)
      const float* q = &patch[(FWD ? wave + r : wave + 4 - r) * PC + li * NIN + 8 * lh];
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (8 * lh + j < KROW) ? q[j] : 0.f;
      bf16x8_l ah, al;
      split8(x, ah, al);
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const bf16x8_l bh = bfr[((r * NJT + jt) * 2 + 0) * 64 + lane], bl = bfr[((r * NJT + jt) * 2 + 1) * 64 + lane];
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[jt], 0, 0, 0);
      }
    }
    const int h = h0 + wave;
    if (h < H) {
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const float bv = (FWD && bias) ? bias[32 * jt + li] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ww = w0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          float v = acc[jt][e] + bv;
          if (FWD && act == ACT_RELU) v = v > 0.f ? v : 0.f;
          if (ww < W) out[((size_t)(b * H + h) * W + ww) * CF + 32 * jt + li] = v;
        }
      }
    }
  }
}

// Exact-fp32 form of the input gradient above (round 3): the same patch, tile list and output mapping on v_mfma_f32_32x32x2_f32.  The
// k = 16 of a kernel row is permuted so that a lane keeps the bf16 kernel's loads: lane half lh holds k = 8*lh + j (j = 0..7) and MFMA
// step j contracts k in {j, 8 + j}; the weight fragments are parked in LDS in that order, as [row][column tile][j / 4][lane][j % 4] so that
// a lane reads them with two conflict-free 16-byte loads.  40 k-steps x CF/32 column tiles = 80 MFMAs of 64 cycles per 32 pixels x 64
// channels: 34 us of matrix time at B = 32, 128 x 128 (the layer's floor, writing its 134-MB result, is 27 us).
template <int NIN, int CF>
__global__ void __launch_bounds__(512, 2) dgrad_rowk_f32_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                                                float* __restrict__ out, int H, int W, int tiles_x, int tiles_per_img,
                                                                int ntiles) {
  static_assert(CF == 64 || CF == 32, "wide channel count");
  constexpr int NJT = CF / 32;
  constexpr int TR = 8, TC = 32, PR = TR + 4, PC = (TC + 4) * NIN + 8;
  constexpr int KROW = 5 * NIN;
  __shared__ float patch[PR * PC];
  typedef float f32x4_l __attribute__((ext_vector_type(4)));
  typedef float f32x16_l __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) f32x4_l bfr[5 * NJT * 2 * 64];                 // weight fragments [r][jt][j / 4][lane] x (j % 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  if (wave < 5) {
    const int r = wave;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        f32x4_l x;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int k = 8 * lh + 4 * jh + jj, d = k / NIN, n = k - d * NIN, cf = 32 * jt + li;
          x[jj] = k < KROW ? w[(((size_t)n * CF + cf) * 5 + r) * 5 + (4 - d)] : 0.f;
        }
        bfr[((r * NJT + jt) * 2 + jh) * 64 + lane] = x;
      }
  }
  constexpr int NPE = PR * (TC + 4) * NIN, NPV = (NPE + 511) / 512;
  float pv[NPV];
  auto load_patch = [&](int tile) {
    const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
    const int h0 = (rem / tiles_x) * TR, w0 = (rem % tiles_x) * TC;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int i = tid + 512 * u;
      const int pr = i / ((TC + 4) * NIN), e = i - pr * ((TC + 4) * NIN);
      const int px = e / NIN, n = e - px * NIN;
      const int h = h0 - 2 + pr, ww = w0 - 2 + px;
      pv[u] = (i < NPE && h >= 0 && h < H && ww >= 0 && ww < W) ? dlogit[((size_t)(b * H + h) * W + ww) * NIN + n] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) load_patch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
    const int h0 = (rem / tiles_x) * TR, w0 = (rem % tiles_x) * TC;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int i = tid + 512 * u;
      if (i < NPE) { const int pr = i / ((TC + 4) * NIN); patch[pr * PC + (i - pr * ((TC + 4) * NIN))] = pv[u]; }
    }
    if (tid < PR * 8) patch[(tid >> 3) * PC + (TC + 4) * NIN + (tid & 7)] = 0.f;
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_patch(tile + (int)gridDim.x);
    f32x16_l acc[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const float* q = &patch[(wave + 4 - r) * PC + li * NIN + 8 * lh];
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (8 * lh + j < KROW) ? q[j] : 0.f;
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const f32x4_l b0 = bfr[((r * NJT + jt) * 2 + 0) * 64 + lane], b1 = bfr[((r * NJT + jt) * 2 + 1) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[j], b0[j], acc[jt], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[4 + j], b1[j], acc[jt], 0, 0, 0);
      }
    }
    const int h = h0 + wave;
    if (h < H) {
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ww = w0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (ww < W) out[((size_t)(b * H + h) * W + ww) * CF + 32 * jt + li] = acc[jt][e];
        }
    }
  }
}

bool rowk_dgrad_applicable(int B, int H, int W, int Cbig, int Csmall) {
  const bool on = true;
  return on && B > 0 && H > 0 && W > 0 && Cbig == 64 && (Csmall == 1 || Csmall == 3);
}

}  // namespace vp

using namespace vp;

extern "C" {

size_t vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(int B, int H, int W, int Cbig, int Csmall) {
  const ConvGeom g = make_geom(B, H, W, Csmall, Cbig, 1);
  return tapm_wgrad_applicable(g) ? tapm_wgrad_ws_floats(g) * sizeof(float) : 0;
}

int vp_conv5_smallout_wgrad_bf16x3(const float* big, const float* small, float* dw_ref, int B, int H, int W, int Cbig, int Csmall,
                                   void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(big && small && dw_ref && ws && B > 0 && H > 0 && W > 0, "vp_conv5_smallout_wgrad_bf16x3: bad arguments");
  VP_REQUIRE(((uintptr_t)big & 15) == 0, "vp_conv5_smallout_wgrad_bf16x3: the activation must be 16-byte aligned");
  const ConvGeom g = make_geom(B, H, W, Csmall, Cbig, 1);
  VP_REQUIRE(tapm_wgrad_applicable(g), "vp_conv5_smallout_wgrad_bf16x3: needs 64 input channels, 1 or 3 outputs, width a multiple of 64, height of 8");
  if (ws_bytes < tapm_wgrad_ws_floats(g) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_smallout_wgrad_bf16x3: workspace too small");
  return tapm_wgrad_launch(big, small, dw_ref, g, (float*)ws, (hipStream_t)stream);
}

size_t vp_conv5_smallout_wgrad_f32_workspace_bytes(int B, int H, int W, int Cbig, int Csmall) {
  return vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(B, H, W, Cbig, Csmall);       // (the same bands, the same slabs)
}

int vp_conv5_smallout_wgrad_f32(const float* big, const float* small, float* dw_ref, int B, int H, int W, int Cbig, int Csmall,
                                void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(big && small && dw_ref && ws && B > 0 && H > 0 && W > 0, "vp_conv5_smallout_wgrad_f32: bad arguments");
  VP_REQUIRE(((uintptr_t)big & 15) == 0, "vp_conv5_smallout_wgrad_f32: the activation must be 16-byte aligned");
  const ConvGeom g = make_geom(B, H, W, Csmall, Cbig, 1);
  VP_REQUIRE(tapm_wgrad_applicable(g), "vp_conv5_smallout_wgrad_f32: needs 64 input channels, 1 or 3 outputs, width a multiple of 64, height of 8");
  if (ws_bytes < tapm_wgrad_ws_floats(g) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_smallout_wgrad_f32: workspace too small");
  return tapm_wgrad_f32_launch(big, small, dw_ref, g, (float*)ws, (hipStream_t)stream);
}

int vp_conv5_smallin_dgrad_bf16x3(const float* small, const float* w_ref, float* big_out, int B, int H, int W, int Csmall, int Cbig,
                                  vp_stream stream) {
  VP_REQUIRE(small && w_ref && big_out, "vp_conv5_smallin_dgrad_bf16x3: null pointer");
  VP_REQUIRE(rowk_dgrad_applicable(B, H, W, Cbig, Csmall), "vp_conv5_smallin_dgrad_bf16x3: needs 64 big channels and 1 or 3 small channels");
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8, ntiles = B * tiles_x * tiles_y;
  const int grid = ntiles < 512 ? ntiles : 512;                 // persistent: two workgroups (8 waves each) per CU
  if (Csmall == 3)
    hipLaunchKernelGGL((dgrad_rowk_kernel<3, 64, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, small, w_ref, big_out, H, W, tiles_x,
                       tiles_x * tiles_y, ntiles, (const float*)nullptr, 0);
  else
    hipLaunchKernelGGL((dgrad_rowk_kernel<1, 64, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, small, w_ref, big_out, H, W, tiles_x,
                       tiles_x * tiles_y, ntiles, (const float*)nullptr, 0);
  return check_launch("vp_conv5_smallin_dgrad_bf16x3");
}

int vp_conv5_smallin_dgrad_f32(const float* small, const float* w_ref, float* big_out, int B, int H, int W, int Csmall, int Cbig,
                               vp_stream stream) {
  VP_REQUIRE(small && w_ref && big_out, "vp_conv5_smallin_dgrad_f32: null pointer");
  VP_REQUIRE(rowk_dgrad_applicable(B, H, W, Cbig, Csmall), "vp_conv5_smallin_dgrad_f32: needs 64 big channels and 1 or 3 small channels");
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8, ntiles = B * tiles_x * tiles_y;
  const int grid = ntiles < 512 ? ntiles : 512;
  if (Csmall == 3)
    hipLaunchKernelGGL((dgrad_rowk_f32_kernel<3, 64>), dim3(grid), dim3(512), 0, (hipStream_t)stream, small, w_ref, big_out, H, W, tiles_x,
                       tiles_x * tiles_y, ntiles);
  else
    hipLaunchKernelGGL((dgrad_rowk_f32_kernel<1, 64>), dim3(grid), dim3(512), 0, (hipStream_t)stream, small, w_ref, big_out, H, W, tiles_x,
                       tiles_x * tiles_y, ntiles);
  return check_launch("vp_conv5_smallin_dgrad_f32");
}

int vp_conv5_smallin_fwd_bf16x3(const float* small, const float* w_ref, const float* bias, float* big_out, int B, int H, int W, int Csmall,
                                int Cbig, int act, vp_stream stream) {
  VP_REQUIRE(small && w_ref && big_out && B > 0 && H > 0 && W > 0, "vp_conv5_smallin_fwd_bf16x3: bad arguments");
  VP_REQUIRE((Cbig == 32 || Cbig == 64) && (Csmall == 1 || Csmall == 3), "vp_conv5_smallin_fwd_bf16x3: needs 1 or 3 input and 32 or 64 output channels");
  VP_REQUIRE(act == VP_ACT_NONE || act == VP_ACT_RELU, "vp_conv5_smallin_fwd_bf16x3: activation none|relu");
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8, ntiles = B * tiles_x * tiles_y;
  const int grid = ntiles < 512 ? ntiles : 512;
#define VP_ROWK_FWD(NI, CFF) hipLaunchKernelGGL((dgrad_rowk_kernel<NI, CFF, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, small, w_ref, big_out, \
                                                H, W, tiles_x, tiles_x * tiles_y, ntiles, bias, act)
  if (Cbig == 64) { if (Csmall == 3) VP_ROWK_FWD(3, 64); else VP_ROWK_FWD(1, 64); }
  else { if (Csmall == 3) VP_ROWK_FWD(3, 32); else VP_ROWK_FWD(1, 32); }
#undef VP_ROWK_FWD
  return check_launch("vp_conv5_smallin_fwd_bf16x3");
}

}
