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
__global__ void __launch_bounds__(256, 1) wgrad_tapm_kernel(const float* __restrict__ u, const float* __restrict__ dlogit,
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

constexpr int TAPM_TH = 16;

bool tapm_wgrad_applicable(const ConvGeom& g) {
  static const bool on = [] { const char* e = getenv("VP_TAPM"); return !e || atoi(e) != 0; }();
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
  VP_REQUIRE(tapm_wgrad_applicable(g), "vp_conv5_smallout_wgrad_bf16x3: needs 64 input channels, 1 or 3 outputs, width a multiple of 64, height of 16");
  if (ws_bytes < tapm_wgrad_ws_floats(g) * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_conv5_smallout_wgrad_bf16x3: workspace too small");
  return tapm_wgrad_launch(big, small, dw_ref, g, (float*)ws, (hipStream_t)stream);
}

}
