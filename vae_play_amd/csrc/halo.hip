// Halo-resident 5x5 convolution on MFMA (split-bf16 "bf16x3") for the layers with a narrow channel side.
//
// The implicit-GEMM kernels (igemm16.h) re-gather the activation operand once per tap: every A element
// is fetched from L2 and written to LDS 25 times.  That is amortised when both channel counts are >= 64,
// but with 32 (or 3, padded to 8) channels on the gathered side a K-tile is a single tap, the MFMA work
// per staged byte is 4-8x smaller and those layers ran at 15-20 % of the others' rate.  Here a workgroup
// stages the input pixels of its output tile ONCE, halo included, and serves all 25 taps from LDS: a tap
// is just a per-lane address offset of the A-fragment read.
//
//   out[pix][n] = sum_{tap=(r,q)} sum_{c<Cin} in[pixel(pix) + off(tap)][c] * w[n][tap][c]
//   (Cin is swept in chunks of CB channels; the halo tile of one chunk is staged once and serves all 25 taps)
//
//   * gather, stride S in {1, 2} (Conv2d forward, ConvTranspose2d input gradient):
//       in = the big tensor, pixel = S*(y, x) - 2, off = (r, q), w = P0 layout [small][25][big];
//   * mirrored gather (Conv2d stride-1 input gradient == scatter family at stride 1):
//       in = the small tensor, pixel = (y, x) - 2, off = (4-r, 4-q), w = P1 layout [big][25][small].
//
// LDS (dynamic, no statics -> the carve stays 16-B aligned):
//   A  [plane][halo row][(parity)][column][CB bf16 (+16 B pad when CB > 8)]
//        stride 2 keeps even and odd halo columns in separate runs so that the 32 consecutive output
//        pixels of a fragment read stay at the conflict-free 80-B stride;
//   B  CB = 32: one tap [plane][n][32 bf16 + pad], double-buffered, prefetched one tap ahead (1 barrier/tap);
//      CB = 32, at most 8 real weight rows (BRES): all 25 taps of those rows + one zero row resident (no barrier);
//      CB = 8 : all 25 taps resident [plane][n][26 x 16 B + pad] (tap 25 = 0: the odd half of the last step).
// Four waves, wave grid WM x WN, wave tile TM x TN fragments of 32 x 32, three v_mfma_f32_32x32x16_bf16 per
// fragment pair exactly as igemm16.h.  Output tiles are TH x 16 pixels (TH = WM*TM*2).
#include <hip/hip_runtime.h>
#include "common.h"
#include "problems.h"
#include "igemm16.h"
#include "halo.h"

namespace vp {

struct HaloArgs {
  const u16* in; size_t in_plane;   // [plane][B][Hi][Wi][Cin]
  const u16* w; size_t w_plane;     // [plane][Nw][25][Cin]
  const float* bias; float* out;    // [B][Ho][Wo][N]
  const void* zero;
  int B, Hi, Wi, Ho, Wo, N, Nw, act;
  int Cin;                          // channels per pixel in HBM: a multiple of CB, swept in chunks of CB
  int tiles_x, tiles_y;
};

template <int CB, int S, int TM, int TN, int WM, int WN, bool MIRROR, bool BRES = false>
struct HaloCfg {
  static constexpr int TW = 16, MB = WM * TM * 32, TH = MB / TW, NB = WN * TN * 32;
  static constexpr int HH = S * (TH - 1) + 5;
  static constexpr int HWE = S == 1 ? TW + 4 : TW + 2;      // entries per halo row (per parity at stride 2)
  static constexpr int RROW = S == 1 ? HWE : 2 * HWE;
  static constexpr int ROWB = CB == 8 ? 16 : CB * 2 + 16;
  static constexpr int CPE = CB / 8;                         // 16-B chunks per pixel and plane
  static constexpr int A_PLANE = ((HH * RROW * ROWB + 127) / 128) * 128 + 64;
  // BRES (CB = 32, at most 8 real weight rows): all 25 taps of the 8 rows + one zero row stay resident
  static constexpr int BROWB = CB == 8 ? 26 * 16 + 16 : (BRES ? kTaps * 64 + 16 : CB * 2 + 16);
  static constexpr int BROWS = BRES ? 9 : NB;
  static constexpr int B_PLANE = ((BROWS * BROWB + 127) / 128) * 128 + 64;
  static constexpr int NBUF = (CB == 8 || BRES) ? 1 : 2;
  static constexpr int LDS = 2 * A_PLANE + NBUF * 2 * B_PLANE;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(CB == 8 || CB == 32, "8 (3 padded) or 32 channels on the gathered side");
  static_assert(S == 1 || S == 2, "stride");
  static_assert(!MIRROR || S == 1, "the mirrored form is the stride-1 input gradient");
  static_assert(!BRES || CB == 32, "resident weight rows: 32-channel chunks");
};

template <int CB, int S, int TM, int TN, int WM, int WN, bool MIRROR, bool BRES>
__global__ void __launch_bounds__(256) halo_conv_kernel(const HaloArgs a) {
  using C = HaloCfg<CB, S, TM, TN, WM, WN, MIRROR, BRES>;
  constexpr bool STREAM = CB == 32 && !BRES;    // weights of one tap at a time, double-buffered
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * C::A_PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int n0 = blockIdx.y * C::NB;
  const int h_org = S * (ty * C::TH) - 2, w_org = S * (tx * C::TW) - 2;
  const u16* zero = reinterpret_cast<const u16*>(a.zero);

  // ---- weights of one tap (CB = 32) -> registers / LDS -------------------------------------------
  constexpr int BCH = STREAM ? (C::NB * 4 * 2 + 255) / 256 : 1;
  u32x4_t breg[BCH];
  auto b_fetch = [&](int tap, int c0) {
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int idx = tid + 256 * u;
      const int ch = idx & 3, n = (idx >> 2) % C::NB, plane = idx / (4 * C::NB);
      const bool ok = plane < 2 && n0 + n < a.Nw;
      breg[u] = ld16(ok ? a.w + plane * a.w_plane + ((size_t)(n0 + n) * kTaps + tap) * a.Cin + c0 + ch * 8 : zero);
    }
  };
  auto b_commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int idx = tid + 256 * u;
      const int ch = idx & 3, n = (idx >> 2) % C::NB, plane = idx / (4 * C::NB);
      if (plane < 2)
        *reinterpret_cast<u32x4_t*>(Bs + (buf * 2 + plane) * C::B_PLANE + n * C::BROWB + ch * 16) = breg[u];
    }
  };
  // ---- per-lane fragment bases ----------------------------------------------------------------------
  int abase[TM], bbase[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 32 + li;
    const int y = m >> 4, x = m & 15;
    abase[i] = ((S * y) * C::RROW + x) * C::ROWB;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = (wn * TN + j) * 32 + li;
    bbase[j] = (BRES ? (nl < 8 ? nl : 8) : nl) * C::BROWB;     // BRES: columns >= 8 read the zero row
  }
  auto tap_off = [&](int tap) {
    int r = div_small(tap, 5), q = tap - r * 5;
    if constexpr (MIRROR) { r = 4 - r; q = 4 - q; }
    if constexpr (S == 1) return (r * C::RROW + q) * C::ROWB;
    else return (r * C::RROW + (q & 1) * C::HWE + (q >> 1)) * C::ROWB;
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](const bf16x8_t (&ah)[TM], const bf16x8_t (&al)[TM], const bf16x8_t (&bh)[TN], const bf16x8_t (&bl)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };
  auto ldf = [&](const unsigned char* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); };

  // ---- channel chunks: the halo tile of CB channels is staged once per chunk and serves all 25 taps ----
#pragma unroll 1
  for (int c0 = 0; c0 < a.Cin; c0 += CB) {
  if constexpr (STREAM) b_fetch(0, c0);
  {
    constexpr int ENT = C::HH * C::RROW, TOT = ENT * C::CPE * 2, U = 8;
    for (int base = 0; base < TOT; base += 256 * U) {
      u32x4_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256 + tid;
        const int ch = idx % C::CPE, e2 = idx / C::CPE;
        const int plane = e2 / ENT, e = e2 - plane * ENT;
        const int hr = e / C::RROW, ri = e - hr * C::RROW;
        int hc = ri;
        if constexpr (S == 2) { const int par = ri / C::HWE; hc = 2 * (ri - par * C::HWE) + par; }
        const int h = h_org + hr, w_ = w_org + hc;
        const bool ok = idx < TOT && h >= 0 && h < a.Hi && w_ >= 0 && w_ < a.Wi;
        v[u] = ld16(ok ? a.in + plane * a.in_plane + ((size_t)(b * a.Hi + h) * a.Wi + w_) * a.Cin + c0 + ch * 8 : zero);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256 + tid;
        if (idx < TOT) {
          const int ch = idx % C::CPE, e2 = idx / C::CPE;
          const int plane = e2 / ENT, e = e2 - plane * ENT;
          *reinterpret_cast<u32x4_t*>(As + plane * C::A_PLANE + e * C::ROWB + ch * 16) = v[u];
        }
      }
    }
  }
  if constexpr (CB == 8) {
    // all taps of the NB weight rows stay resident; entry 25 of every row is zero
    constexpr int TOT = C::NB * 26 * 2;
    for (int idx = tid; idx < TOT; idx += 256) {
      const int tap = idx % 26, r2 = idx / 26;
      const int n = r2 % C::NB, plane = r2 / C::NB;
      const bool ok = tap < kTaps && n0 + n < a.Nw;
      const u32x4_t v = ld16(ok ? a.w + plane * a.w_plane + ((size_t)(n0 + n) * kTaps + tap) * a.Cin + c0 : zero);
      *reinterpret_cast<u32x4_t*>(Bs + plane * C::B_PLANE + n * C::BROWB + tap * 16) = v;
    }
  } else if constexpr (BRES) {
    constexpr int TOT = 9 * kTaps * 4 * 2;
    for (int idx = tid; idx < TOT; idx += 256) {
      const int ch = idx & 3, r2 = idx >> 2;
      const int tap = r2 % kTaps, r3 = r2 / kTaps;
      const int n = r3 % 9, plane = r3 / 9;
      const bool ok = n < 8 && n0 + n < a.Nw;
      const u32x4_t v = ld16(ok ? a.w + plane * a.w_plane + ((size_t)(n0 + n) * kTaps + tap) * a.Cin + c0 + ch * 8 : zero);
      *reinterpret_cast<u32x4_t*>(Bs + plane * C::B_PLANE + n * C::BROWB + tap * 64 + ch * 16) = v;
    }
  } else {
    b_commit(0);
  }
  __syncthreads();

  if constexpr (BRES) {
#pragma unroll 1
    for (int tap = 0; tap < kTaps; ++tap) {
      const int toff = tap_off(tap);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned char* p = As + abase[i] + toff + ks * 32 + lh * 16;
          ah[i] = ldf(p);
          al[i] = ldf(p + C::A_PLANE);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned char* p = Bs + bbase[j] + tap * 64 + ks * 32 + lh * 16;
          bh[j] = ldf(p);
          bl[j] = ldf(p + C::B_PLANE);
        }
        mma(ah, al, bh, bl);
      }
    }
    if (c0 + CB < a.Cin) __syncthreads();   // the resident tiles are about to be replaced
  } else if constexpr (CB == 32) {
    for (int tap = 0; tap < kTaps; ++tap) {
      const bool more = tap + 1 < kTaps;
      if (more) b_fetch(tap + 1, c0);
      const int toff = tap_off(tap);
      const unsigned char* Bt = Bs + (tap & 1) * 2 * C::B_PLANE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned char* p = As + abase[i] + toff + ks * 32 + lh * 16;
          ah[i] = ldf(p);
          al[i] = ldf(p + C::A_PLANE);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned char* p = Bt + bbase[j] + ks * 32 + lh * 16;
          bh[j] = ldf(p);
          bl[j] = ldf(p + C::B_PLANE);
        }
        mma(ah, al, bh, bl);
      }
      if (more) b_commit((tap + 1) & 1);   // the other buffer: its last readers passed the previous barrier
      __syncthreads();
    }
  } else {
    // one MFMA step = two taps of 8 channels: the lane's half (lh) picks the tap
#pragma unroll 1
    for (int kk = 0; kk < 13; ++kk) {
      const int tap = 2 * kk + lh;
      const int toff = tap_off(tap < kTaps ? tap : kTaps - 1);   // tap 25 multiplies zero weights
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* p = As + abase[i] + toff;
        ah[i] = ldf(p);
        al[i] = ldf(p + C::A_PLANE);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* p = Bs + bbase[j] + tap * 16;
        bh[j] = ldf(p);
        bl[j] = ldf(p + C::B_PLANE);
      }
      mma(ah, al, bh, bl);
    }
    if (c0 + CB < a.Cin) __syncthreads();   // the resident tile is about to be replaced
  }
  }   // channel chunks

  // ---- epilogue: lane = output channel ----------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + li;
    if (n >= a.N) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = (wm * TM + i) * 32 + row;
        const int y = ty * C::TH + (m >> 4), x = tx * C::TW + (m & 15);
        float v = acc[i][j][r] + bv;
        if (a.act == ACT_SIGMOID) v = 1.f / (1.f + __builtin_expf(-v));
        a.out[((size_t)(b * a.Ho + y) * a.Wo + x) * a.N + n] = v;
      }
  }
}

template <int CB, int S, int TM, int TN, int WM, int WN, bool MIRROR, bool BRES = false>
static int halo_launch(HaloArgs a, hipStream_t s, const char* what) {
  using C = HaloCfg<CB, S, TM, TN, WM, WN, MIRROR, BRES>;
  auto kern = halo_conv_kernel<CB, S, TM, TN, WM, WN, MIRROR, BRES>;
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  if (attr != hipSuccess) return fail(VP_ERR_LAUNCH, "%s: cannot reserve %d B of LDS: %s", what, C::LDS, hipGetErrorString(attr));
  a.tiles_x = a.Wo / C::TW;
  a.tiles_y = a.Ho / C::TH;
  a.zero = vp_zero_page();
  dim3 grid((unsigned)(a.B * a.tiles_x * a.tiles_y), (unsigned)((a.N + C::NB - 1) / C::NB));
  hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS, s, a);
  return check_launch(what);
}

// ---- host dispatch ---------------------------------------------------------------------------------
int halo_gather_kind(int B, int Hs, int Ws, int Cbig, int Csmall, int stride) {
  if (Ws % 16 != 0 || B <= 0) return 0;
  if (Cbig % 32 == 0 && Cbig <= 128 && stride == 1 && Csmall <= 32 && Hs % 16 == 0) return 1;
  if (Cbig == 32 && stride == 2 && Csmall >= 64 && Hs % 8 == 0) return 2;
  if (Cbig == 8 && stride == 2 && Hs % 8 == 0) return 3;
  return 0;
}

int halo_gather_launch(int kind, const void* big_split, const void* w_p0_split, const float* bias, float* out, int B, int Hs, int Ws,
                       int Cbig, int Csmall, int stride, int act, hipStream_t s) {
  HaloArgs a{};
  a.B = B; a.Ho = Hs; a.Wo = Ws; a.Hi = Hs * stride; a.Wi = Ws * stride; a.N = Csmall; a.Nw = Csmall; a.act = act; a.Cin = Cbig;
  a.in = (const u16*)big_split; a.in_plane = (size_t)B * a.Hi * a.Wi * Cbig;
  a.w = (const u16*)w_p0_split; a.w_plane = (size_t)Csmall * kTaps * Cbig;
  a.bias = bias; a.out = out;
  if (kind == 1) {
    if (Csmall <= 8) return halo_launch<32, 1, 2, 1, 4, 1, false, true>(a, s, "halo_gather<32,s1,resident>");
    return halo_launch<32, 1, 2, 1, 4, 1, false>(a, s, "halo_gather<32,s1>");
  }
  if (kind == 2) {
    if (Csmall % 128 == 0) return halo_launch<32, 2, 2, 2, 2, 2, false>(a, s, "halo_gather<32,s2,128>");
    return halo_launch<32, 2, 2, 1, 2, 2, false>(a, s, "halo_gather<32,s2,64>");
  }
  return halo_launch<8, 2, 2, 1, 2, 2, false>(a, s, "halo_gather<8,s2>");
}

int halo_scatter_kind(int B, int Hs, int Ws, int Csmall, int Cbig, int stride) {
  if (stride == 1 && Csmall == 8 && Ws % 16 == 0 && Hs % 16 == 0 && B > 0) return 1;
  return 0;
}

int halo_scatter_launch(int kind, const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                        int Cbig, hipStream_t s) {
  (void)kind;
  HaloArgs a{};
  a.B = B; a.Ho = Hs; a.Wo = Ws; a.Hi = Hs; a.Wi = Ws; a.N = Cbig; a.Nw = Cbig; a.act = ACT_NONE; a.Cin = Csmall;
  a.in = (const u16*)small_split; a.in_plane = (size_t)B * Hs * Ws * Csmall;
  a.w = (const u16*)w_p1_split; a.w_plane = (size_t)Cbig * kTaps * Csmall;
  a.bias = nullptr; a.out = big_out;
  return halo_launch<8, 1, 2, 1, 4, 1, true>(a, s, "halo_scatter<8,s1>");
}

}  // namespace vp
