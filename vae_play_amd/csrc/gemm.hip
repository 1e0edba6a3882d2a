// Dense layers (nn.Linear fwd / dgrad / wgrad) on the MFMA implicit-GEMM kernel, split-K with a
// deterministic slab reduction, plus the column-sum used for bias gradients.
#include "common.h"
#include "igemm.h"

namespace vp {

__global__ void gemm_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, const float* __restrict__ bias,
                                   int M, int N, int ldc, int nsplit) {
  const size_t per = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const size_t m = i / N;
    float s = bias ? bias[n] : 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += slab[(size_t)sp * per + i];
    C[m * ldc + n] = s;
  }
}

// Few outputs, many slabs (a 32 x 128 input gradient contracted over 32768 features in 128 splits): one thread per output
// walked the slabs as 128 dependent loads (38 us for 4096 outputs).  Here 64 outputs x 4 split-lanes per workgroup, four
// loads in flight per lane, combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) gemm_reduce_deep_kernel(const float* __restrict__ slab, float* __restrict__ C,
                                                               const float* __restrict__ bias, int M, int N, int ldc, int nsplit) {
  __shared__ float red[4][64];
  const size_t per = (size_t)M * N;
  const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < per) {
    int sp = sl;
    for (; sp + 12 < nsplit; sp += 16) {
      s0 += slab[(size_t)sp * per + i];
      s1 += slab[(size_t)(sp + 4) * per + i];
      s2 += slab[(size_t)(sp + 8) * per + i];
      s3 += slab[(size_t)(sp + 12) * per + i];
    }
    for (; sp < nsplit; sp += 4) s0 += slab[(size_t)sp * per + i];
  }
  red[sl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < per) {
    const int n = (int)(i % N);
    const size_t m = i / N;
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    C[m * ldc + n] = v + (bias ? bias[n] : 0.f);
  }
}

// partial[chunk][c] = sum over the chunk's rows of x[r][c]; generic C (bias grads have C = 3).
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int R, int C, int rows_per_chunk) {
  const int chunk = blockIdx.x;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  // blockDim = (64 lanes over rows, 4 channel groups): one wavefront reduces one channel at a time
  for (int c = threadIdx.y; c < C; c += blockDim.y) {
    float s = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) s += x[(size_t)r * C + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) partial[(size_t)chunk * C + c] = s;
  }
}

// Narrow rows with C % 4 == 0 (conv bias gradients over B*H*W rows of 8 .. 60 channels, e.g. the VAE-GAN discriminator's first conv:
// 786 432 rows x 32): 16-B loads, C/4 lanes per row, 256 / (C/4) rows per pass -- every 128-B row is read once, coalesced.  The row-lane
// form above reads a row once per channel (0.9 TB/s on that layer: 116 us for 100 MB).
__global__ void __launch_bounds__(256) colsum_rows4_kernel(const float* __restrict__ x, float* __restrict__ partial, int R, int C,
                                                           int rows_per_chunk) {
  __shared__ vp_f32x4 sh[256];
  const int q = C / 4, rpp = 256 / q;                    // float4 columns, rows per pass
  const int rl = threadIdx.x / q, c4 = threadIdx.x - rl * q;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  vp_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (rl < rpp) {
    int r = r0 + rl;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {             // four independent loads in flight
      const vp_f32x4 a = *reinterpret_cast<const vp_f32x4*>(x + (size_t)r * C + 4 * c4);
      const vp_f32x4 b = *reinterpret_cast<const vp_f32x4*>(x + (size_t)(r + rpp) * C + 4 * c4);
      const vp_f32x4 c = *reinterpret_cast<const vp_f32x4*>(x + (size_t)(r + 2 * rpp) * C + 4 * c4);
      const vp_f32x4 d = *reinterpret_cast<const vp_f32x4*>(x + (size_t)(r + 3 * rpp) * C + 4 * c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (a[j] + b[j]) + (c[j] + d[j]);
    }
    for (; r < r1; r += rpp) {
      const vp_f32x4 a = *reinterpret_cast<const vp_f32x4*>(x + (size_t)r * C + 4 * c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += a[j];
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < q) {                                   // row lanes summed in order: deterministic
    vp_f32x4 t = sh[threadIdx.x];
    for (int k = 1; k < rpp; ++k) {
      const vp_f32x4 u = sh[k * q + threadIdx.x];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] += u[j];
    }
    *reinterpret_cast<vp_f32x4*>(partial + (size_t)blockIdx.x * C + 4 * threadIdx.x) = t;
  }
}

// Wide form (C >= 64: dense-layer bias gradients, R = batch): one thread per channel, coalesced across channels, rows in a
// loop -- the row-lane form above walks the channels one wavefront at a time (40 us for C = 256, R = 8).
__global__ void __launch_bounds__(256) colsum_wide_kernel(const float* __restrict__ x, float* __restrict__ partial, int R, int C,
                                                          int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += x[(size_t)r * C + c];
  partial[(size_t)blockIdx.y * C + c] = s;
}

// one wavefront per channel: the 64 lanes split the chunk partials, fp64 wave reduction
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int nchunk) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int k = threadIdx.x; k < nchunk; k += 64) s += (double)partial[(size_t)k * C + c];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[c] = (float)s;
}

// ---------------------------------------------------------------------------------------------------------
// Skinny dense layers: M <= 64 rows (the batch) against a weight matrix that is read exactly once -- HBM-bound.
// The general tile kernel stages BOTH operands through LDS with two barriers per K-tile and reached 2-3 TB/s on the
// 134 MB encoder.fc.0 matrix.  Here the weights go straight from global memory into the MFMA B operand (each element
// is used by one wave only, so LDS buys nothing), eight 16-B loads in flight per lane, and only the activation rows
// (a few MB, L2-resident) are shared.  v_mfma_f32_32x32x2_f32: exact fp32, the 32 batch rows are the MFMA's M.
// The k order inside a tile is permuted identically for A and B (mode 0: lane half h, load i, step j <-> k = 32 h + 4 i + j,
// so that a lane walks one 128-B line of its weight row per tile; mode 1: k = 8 q + 4 h + j).
// ---------------------------------------------------------------------------------------------------------
constexpr int SK_KT = 64;   // k per tile

// mode 0 (Linear forward): A[m][k] and B[n][k] both k-contiguous.  Workgroup = 4 waves x 32 columns; the A tile
// [32 TM][64] is shared through LDS (two buffers, one barrier per tile).  TM = 1 | 2 blocks of 32 batch rows: a weight
// fragment feeds TM MFMAs.  grid = (ceil(N/128), nsplit).
template <int TM>
__global__ void __launch_bounds__(256) skinny_mk_kernel(const float* __restrict__ A, long sam, const float* __restrict__ Bm, long sbn,
                                                        float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                        int K, int nsplit, int kps) {
  __shared__ __attribute__((aligned(16))) float xs[2][32 * TM][SK_KT + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x * 128 + wave * 32 + li;
  const float* wrow = Bm + (size_t)(n < N ? n : 0) * sbn;
  const int k_begin = blockIdx.y * kps;
  const int k_end = min(K, k_begin + kps);
  const int nk = (k_end - k_begin) / SK_KT;
  const int xrow = tid >> 3, xc = tid & 7;
  const float* arow[TM];
  bool xok[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    xok[b] = xrow + 32 * b < M;
    arow[b] = A + (size_t)(xok[b] ? xrow + 32 * b : 0) * sam;
  }
  vp_f32x4 xr[TM][2], w[8], wn[8];
  f32x16 acc[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  auto load_tile = [&](int kb) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      xr[b][0] = xok[b] ? ld4(arow[b] + kb + xc * 4) : zero4();
      xr[b][1] = xok[b] ? ld4(arow[b] + kb + 32 + xc * 4) : zero4();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) wn[i] = ld4(wrow + kb + lh * 32 + i * 4);   // one 128-B line per lane and tile
  };
  auto write_x = [&](int buf) {
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      *reinterpret_cast<vp_f32x4*>(&xs[buf][xrow + 32 * b][xc * 4]) = xr[b][0];
      *reinterpret_cast<vp_f32x4*>(&xs[buf][xrow + 32 * b][32 + xc * 4]) = xr[b][1];
    }
  };
  if (nk > 0) {
    load_tile(k_begin);
    write_x(0);
    __syncthreads();
  }
  for (int t = 0; t < nk; ++t) {
    const bool more = t + 1 < nk;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = wn[i];
    if (more) load_tile(k_begin + (t + 1) * SK_KT);
    const int cur = t & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      vp_f32x4 x4[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) x4[b] = *reinterpret_cast<const vp_f32x4*>(&xs[cur][li + 32 * b][lh * 32 + i * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[b][j], w[i][j], acc[b], 0, 0, 0);
    }
    if (more) write_x(cur ^ 1);      // last read in iteration t-1, which every wave left through the barrier below
    __syncthreads();
  }
  if (n >= N) return;
#pragma unroll
  for (int b = 0; b < TM; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= M) continue;
      if (nsplit == 1) C[(size_t)m * ldc + n] = acc[b][r] + (bias ? bias[n] : 0.f);
      else C[((size_t)blockIdx.y * M + m) * N + n] = acc[b][r];
    }
}

// mode 1 (Linear input gradient): A[m][k] k-contiguous, B[k][n] n-contiguous.  A lane's 16-B weight load covers 4 columns of
// one k row, so a wave owns 128 columns (4 accumulators per 32 batch rows) and reuses its A value four times; the 4 waves of a
// workgroup take interleaved 64-deep k tiles of the workgroup's K range and are summed through LDS in a fixed order at the
// end.  No barrier in the main loop: the A values (L2-resident) are loaded per lane as well.  grid = (ceil(N/128), nsplit).
template <int TM>
__global__ void __launch_bounds__(256) skinny_kn_kernel(const float* __restrict__ A, long sam, const float* __restrict__ Bm, long sbk,
                                                        float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                        int K, int nsplit, int kps) {
  __shared__ __attribute__((aligned(16))) float red[2][64][64];   // [wave slot][register][lane], one 32-row block at a time
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 128 + 4 * li;           // this lane's 4 columns (N % 4 == 0)
  const bool nok = n0 < N;
  const float* wcol = Bm + (nok ? n0 : 0);
  const float* arow[TM];
  bool aok[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    aok[b] = li + 32 * b < M;
    arow[b] = A + (size_t)(aok[b] ? li + 32 * b : 0) * sam;
  }
  const int k_begin = blockIdx.y * kps;
  const int k_end = min(K, k_begin + kps);
  const int nk = (k_end - k_begin) / SK_KT;          // tiles of the workgroup; wave w takes tiles w, w+4, ...
  f32x16 acc[TM][4];
#pragma unroll
  for (int b = 0; b < TM; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][c][r] = 0.f;
  vp_f32x4 w[8], wn[8], a[TM][2], an[TM][2];
  // 16 k rows per step: q = 0..1 blocks of 8 k; lane half h, step j <-> k = kb + 8 q + 4 h + j
  auto load_half = [&](int kb) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int b = 0; b < TM; ++b) an[b][q] = aok[b] ? ld4(arow[b] + kb + q * 8 + lh * 4) : zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j) wn[q * 4 + j] = ld4(wcol + (size_t)(kb + q * 8 + lh * 4 + j) * sbk);
    }
  };
  const int nh = nk > wave ? ((nk - wave + 3) / 4) * 4 : 0;   // steps of this wave: 4 per tile
  auto kb_of = [&](int h) { return k_begin + (wave + 4 * (h >> 2)) * SK_KT + (h & 3) * 16; };
  if (nh > 0) load_half(kb_of(0));
  for (int h = 0; h < nh; ++h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = wn[i];
#pragma unroll
    for (int b = 0; b < TM; ++b) { a[b][0] = an[b][0]; a[b][1] = an[b][1]; }
    if (h + 1 < nh) load_half(kb_of(h + 1));
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[b][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][q][j], w[q * 4 + j][c], acc[b][c], 0, 0, 0);
  }
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    // (wave 2, wave 3) -> LDS, added by (wave 0, wave 1); then wave 1 -> LDS, added by wave 0
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 2][c * 16 + r][lane] = acc[b][c][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][c][r] += red[wave][c * 16 + r][lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[0][c * 16 + r][lane] = acc[b][c][r];
    }
    __syncthreads();
    if (wave == 0 && nok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        vp_f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = acc[b][c][r] + red[0][c * 16 + r][lane];
        if (nsplit == 1) {
          if (bias) { v[0] += bias[n0]; v[1] += bias[n0 + 1]; v[2] += bias[n0 + 2]; v[3] += bias[n0 + 3]; }
          *reinterpret_cast<vp_f32x4*>(C + (size_t)m * ldc + n0) = v;
        } else {
          *reinterpret_cast<vp_f32x4*>(C + ((size_t)blockIdx.y * M + m) * N + n0) = v;
        }
      }
    }
  }
}

// Split count of the skinny kernels (0: shape not eligible).  mode 0 / 1 as in vp_gemm_f32.
inline int skinny_nsplit(int mode, long M, long N, long K) {
  if (mode > 1 || M > 64 || N < 128 || K % SK_KT) return 0;
  const long colblocks = (N + 127) / 128;
  const long unit = mode == 0 ? 256 : 4 * SK_KT;      // smallest K range worth a workgroup
  long blocks = 512;
  long want = (blocks + colblocks - 1) / colblocks;
  long maxs = K / unit;
  if (maxs < 1) maxs = 1;
  long s = want < maxs ? want : maxs;
  if (s > 128) s = 128;
  return (int)s;
}
inline int skinny_kps(long K, int ns) {
  const long per = (K + ns - 1) / ns;
  return (int)(((per + SK_KT - 1) / SK_KT) * SK_KT);
}

__global__ void __launch_bounds__(256) colsum_small_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += x[(size_t)r * C + c];
  out[c] = s;
}

inline int colsum_chunks(int R) {
  int n = (R + 511) / 512;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return n;
}

}  // namespace vp

using namespace vp;

extern "C" {

size_t vp_gemm_workspace_bytes(int M, int N, int K) {
  int ns = gemm_nsplit(M, N, K);
  for (int mode = 0; mode < 2; ++mode) {          // the caller does not name the mode here: cover the skinny kernels' split too
    const int s0 = skinny_nsplit(mode, M, N, K);
    if (s0) { const int kps = skinny_kps(K, s0); const int eff = (K + kps - 1) / kps; if (eff > ns) ns = eff; }
  }
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

int vp_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, int ldc,
                const float* bias, int M, int N, int K, int mode, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(A && B && C, "vp_gemm_f32: null pointer");
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "vp_gemm_f32: bad shape M=%d N=%d K=%d ldc=%d", M, N, K, ldc);
  VP_REQUIRE(mode >= 0 && mode <= 2, "vp_gemm_f32: mode must be 0|1|2");
  int ns = gemm_nsplit(M, N, K);
  // skinny path: batch-sized M, the weight matrix streamed once straight into the MFMA operand
  const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  int sk = skinny_nsplit(mode, M, N, K);
  if (sk && mode == 0 && !(sak == 1 && sbk == 1 && sam % 4 == 0 && sbn % 4 == 0 && al16(A) && al16(B))) sk = 0;
  if (sk && mode == 1 && !(sak == 1 && sbn == 1 && sam % 4 == 0 && sbk % 4 == 0 && N % 4 == 0 && ldc % 4 == 0 && al16(A) && al16(B) && al16(C))) sk = 0;
  if (sk) {
    const int kps = skinny_kps(K, sk);
    ns = (K + kps - 1) / kps;
    float* dst = C;
    if (ns > 1) {
      if (!ws || ws_bytes < (size_t)ns * M * N * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_gemm_f32: workspace too small");
      dst = (float*)ws;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)ns);
    if (mode == 0 && M <= 32) hipLaunchKernelGGL(skinny_mk_kernel<1>, grid, dim3(256), 0, s, A, sam, B, sbn, dst, ldc, bias, M, N, K, ns, kps);
    else if (mode == 0) hipLaunchKernelGGL(skinny_mk_kernel<2>, grid, dim3(256), 0, s, A, sam, B, sbn, dst, ldc, bias, M, N, K, ns, kps);
    else if (M <= 32) hipLaunchKernelGGL(skinny_kn_kernel<1>, grid, dim3(256), 0, s, A, sam, B, sbk, dst, ldc, bias, M, N, K, ns, kps);
    else hipLaunchKernelGGL(skinny_kn_kernel<2>, grid, dim3(256), 0, s, A, sam, B, sbk, dst, ldc, bias, M, N, K, ns, kps);
    int rc = check_launch("vp_gemm_f32(skinny)");
    if (rc || ns == 1) return rc;
    if (ns >= 16 && (size_t)M * N <= 65536)
      hipLaunchKernelGGL(gemm_reduce_deep_kernel, dim3((unsigned)(((size_t)M * N + 63) / 64)), dim3(256), 0, s, (const float*)ws, C, bias, M, N, ldc, ns);
    else
      hipLaunchKernelGGL(gemm_reduce_kernel, dim3(grid_for((size_t)M * N, 256)), dim3(256), 0, s, (const float*)ws, C, bias, M, N, ldc, ns);
    return check_launch("vp_gemm_f32(reduce)");
  }
  float* dst = C;
  if (ns > 1) {
    const size_t need = (size_t)ns * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(VP_ERR_WORKSPACE, "vp_gemm_f32: workspace %zu < %zu", ws_bytes, need);
    dst = (float*)ws;
  }
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) {
    auto p = make_probG<false, false>(A, sam, sak, B, sbn, sbk, dst, ldc, bias, M, N, K, ns);
    launch_igemm(p, M, N, ns, s);
  } else if (mode == 1) {
    auto p = make_probG<false, true>(A, sam, sak, B, sbn, sbk, dst, ldc, bias, M, N, K, ns);
    launch_igemm(p, M, N, ns, s);
  } else {
    auto p = make_probG<true, true>(A, sam, sak, B, sbn, sbk, dst, ldc, bias, M, N, K, ns);
    launch_igemm(p, M, N, ns, s);
  }
  int rc = check_launch("vp_gemm_f32(main)");
  if (rc || ns == 1) return rc;
  if (ns >= 16 && (size_t)M * N <= 65536)
    hipLaunchKernelGGL(gemm_reduce_deep_kernel, dim3((unsigned)(((size_t)M * N + 63) / 64)), dim3(256), 0, s, (const float*)ws, C, bias, M, N, ldc, ns);
  else
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3(grid_for((size_t)M * N, 256)), dim3(256), 0, s, (const float*)ws, C, bias, M, N, ldc, ns);
  return check_launch("vp_gemm_f32(reduce)");
}

size_t vp_colsum_workspace_bytes(int R, int C) { return (size_t)colsum_chunks(R) * C * sizeof(float); }

int vp_colsum_f32(const float* x, float* out, int R, int C, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && out && ws && R > 0 && C > 0, "vp_colsum_f32: bad arguments");
  const bool small_on = true;
  if (R <= 64 && small_on) {      // a bias gradient over the batch rows: one launch, rows summed in order
    hipLaunchKernelGGL(colsum_small_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, R, C);
    return check_launch("vp_colsum_f32(small)");
  }
  const int nchunk = colsum_chunks(R);
  if (ws_bytes < (size_t)nchunk * C * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_colsum_f32: workspace too small");
  const int rpc = (R + nchunk - 1) / nchunk;
  hipStream_t s = (hipStream_t)stream;
  if (C >= 64)
    hipLaunchKernelGGL(colsum_wide_kernel, dim3((C + 255) / 256, nchunk), dim3(256), 0, s, x, (float*)ws, R, C, rpc);
  else if (C % 4 == 0 && C >= 8 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(colsum_rows4_kernel, dim3(nchunk), dim3(256), 0, s, x, (float*)ws, R, C, rpc);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nchunk), dim3(64, 4), 0, s, x, (float*)ws, R, C, rpc);
  int rc = check_launch("vp_colsum_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, s, (const float*)ws, out, C, nchunk);
  return check_launch("vp_colsum_f32(final)");
}
}
