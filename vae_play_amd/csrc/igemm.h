// fp32 implicit-GEMM kernel for gfx950 (CDNA4): C[m][n] = sum_k A(m,k) * B(n,k).
//
//  * contraction on the matrix cores with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, exact
//    fp32 -- bit-equal to an fmaf chain), 64-lane wavefronts, 4 waves per workgroup;
//  * A/B tiles are gathered from NHWC activations / packed weights by the accessor structs of
//    problems.h (im2col is implicit), register-staged one K-tile ahead of the MFMAs (issue
//    global loads -> compute current tile from LDS -> write staged registers), so HBM/L2
//    latency hides under the 64-cycle MFMAs;
//  * LDS images: "MK" operands [row][BK+4] (k contiguous, 16-B loads in / ds_read_b128 out,
//    row stride 36 dwords keeps each 16-lane b128 group on 64 distinct banks), "KM" operands
//    [k][rows] (row index contiguous, ds_read_b32 conflict-free across the 32 lanes of a half);
//  * k order inside an 8-deep block is permuted the same way for A and B (lane half h takes
//    k = 8*blk + 4*h + j at MFMA step j) so one ds_read_b128 feeds four MFMAs;
//  * the accumulator layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) puts the
//    output channel on the lane: every store instruction writes 128-B contiguous NHWC segments.
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"

namespace vp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;

template <int BR, bool KM>
struct LdsImage {
  static constexpr int floats = KM ? BK * BR : BR * LDK;
};

template <class P, int NR, bool KM>
struct RowsA;
template <class P, int NR>
struct RowsA<P, NR, false> { typename P::ARow r[NR]; };
template <class P, int NR>
struct RowsA<P, NR, true> {};
template <class P, int NR, bool KM>
struct RowsB;
template <class P, int NR>
struct RowsB<P, NR, false> { typename P::BRow r[NR]; };
template <class P, int NR>
struct RowsB<P, NR, true> {};

template <class P, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) igemm_kernel(const P p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
  constexpr int NA = BM / 32, NB = BN / 32;  // 16-B loads per thread per K-tile
  constexpr int A_FLOATS = LdsImage<BM, P::A_KM>::floats;
  constexpr int B_FLOATS = LdsImage<BN, P::B_KM>::floats;
  __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + B_FLOATS];
  float* As = lds;
  float* Bs = lds + A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  typename P::ZCtx z;
  p.z_setup(blockIdx.z, z);

  RowsA<P, NA, P::A_KM> ra;
  RowsB<P, NB, P::B_KM> rb;
  if constexpr (!P::A_KM) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra.r[i] = p.a_row(m0 + (tid >> 3) + 32 * i, z);
  }
  if constexpr (!P::B_KM) {
#pragma unroll
    for (int i = 0; i < NB; ++i) rb.r[i] = p.b_row(n0 + (tid >> 3) + 32 * i, z);
  }

  vp_f32x4 sa[NA], sb[NB];

  auto stage_load = [&](int k0) {
    if constexpr (!P::A_KM) {
      const int kc = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < NA; ++i) sa[i] = p.a_load(ra.r[i], k0 + kc, z);
    } else {
      constexpr int V = BM / 4, RP = 256 / V;
      const int mc = (tid % V) * 4, kr = tid / V;
#pragma unroll
      for (int i = 0; i < NA; ++i) sa[i] = p.a_load_km(k0 + kr + RP * i, m0 + mc, z);
    }
    if constexpr (!P::B_KM) {
      const int kc = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < NB; ++i) sb[i] = p.b_load(rb.r[i], k0 + kc, z);
    } else {
      constexpr int V = BN / 4, RP = 256 / V;
      const int nc = (tid % V) * 4, kr = tid / V;
#pragma unroll
      for (int i = 0; i < NB; ++i) sb[i] = p.b_load_km(k0 + kr + RP * i, n0 + nc, z);
    }
  };
  auto stage_write = [&]() {
    if constexpr (!P::A_KM) {
      const int kc = (tid & 7) * 4, r0 = tid >> 3;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<vp_f32x4*>(&As[(r0 + 32 * i) * LDK + kc]) = sa[i];
    } else {
      constexpr int V = BM / 4, RP = 256 / V;
      const int mc = (tid % V) * 4, kr = tid / V;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<vp_f32x4*>(&As[(kr + RP * i) * BM + mc]) = sa[i];
    }
    if constexpr (!P::B_KM) {
      const int kc = (tid & 7) * 4, r0 = tid >> 3;
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<vp_f32x4*>(&Bs[(r0 + 32 * i) * LDK + kc]) = sb[i];
    } else {
      constexpr int V = BN / 4, RP = 256 / V;
      const int nc = (tid % V) * 4, kr = tid / V;
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<vp_f32x4*>(&Bs[(kr + RP * i) * BN + nc]) = sb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int klen = z.k_end - z.k_begin;
  const int nk = klen > 0 ? (klen + BK - 1) / BK : 0;
  const int arow0 = wm * (BM / WM) + li;
  const int brow0 = wn * (BN / WN) + li;

  if (nk > 0) {
    stage_load(z.k_begin);
    stage_write();
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) stage_load(z.k_begin + (kt + 1) * BK);
#pragma unroll
    for (int blk = 0; blk < BK / 8; ++blk) {
      vp_f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (!P::A_KM) {
          a[i] = *reinterpret_cast<const vp_f32x4*>(&As[(arow0 + 32 * i) * LDK + blk * 8 + lh * 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[i][j] = As[(blk * 8 + lh * 4 + j) * BM + arow0 + 32 * i];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if constexpr (!P::B_KM) {
          b[i] = *reinterpret_cast<const vp_f32x4*>(&Bs[(brow0 + 32 * i) * LDK + blk * 8 + lh * 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[i][j] = Bs[(blk * 8 + lh * 4 + j) * BN + brow0 + 32 * i];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[jn][j], acc[i][jn], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stage_write();
      __syncthreads();
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + wm * (BM / WM) + 32 * i + row;
        const int n = n0 + wn * (BN / WN) + 32 * jn + li;
        p.store(m, n, acc[i][jn][r], z);
      }
}

// ---- host-side launcher: picks a tile shape from (M, N, #z) ---------------------------------
struct TileChoice { int bm, bn; };

inline TileChoice choose_tile(long M, long N, int gz) {
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long)gz; };
  if (M <= 32) return {32, 128};
  if (N <= 32) return {128, 32};
  if (M >= 128 && N >= 128 && blocks(128, 128) >= 192) return {128, 128};
  if (M >= 128 && N >= 64 && blocks(128, 64) >= 192) return {128, 64};
  return {64, 64};
}

template <class P>
inline void launch_igemm(const P& p, long M, long N, int gz, hipStream_t stream, TileChoice force = {0, 0}) {
  TileChoice t = force.bm ? force : choose_tile(M, N, gz);
  dim3 block(256);
  auto grid = [&](int bm, int bn) { return dim3((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)gz); };
  if (t.bm == 128 && t.bn == 128) {
    hipLaunchKernelGGL((igemm_kernel<P, 128, 128, 2, 2>), grid(128, 128), block, 0, stream, p);
  } else if (t.bm == 128 && t.bn == 64) {
    hipLaunchKernelGGL((igemm_kernel<P, 128, 64, 2, 2>), grid(128, 64), block, 0, stream, p);
  } else if (t.bm == 128 && t.bn == 32) {
    hipLaunchKernelGGL((igemm_kernel<P, 128, 32, 4, 1>), grid(128, 32), block, 0, stream, p);
  } else if (t.bm == 32 && t.bn == 128) {
    hipLaunchKernelGGL((igemm_kernel<P, 32, 128, 1, 4>), grid(32, 128), block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((igemm_kernel<P, 64, 64, 2, 2>), grid(64, 64), block, 0, stream, p);
  }
}

}  // namespace vp
