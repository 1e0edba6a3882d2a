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
  const int ns = gemm_nsplit(M, N, K);
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

int vp_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, int ldc,
                const float* bias, int M, int N, int K, int mode, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(A && B && C, "vp_gemm_f32: null pointer");
  VP_REQUIRE(M > 0 && N > 0 && K > 0 && ldc >= N, "vp_gemm_f32: bad shape M=%d N=%d K=%d ldc=%d", M, N, K, ldc);
  VP_REQUIRE(mode >= 0 && mode <= 2, "vp_gemm_f32: mode must be 0|1|2");
  const int ns = gemm_nsplit(M, N, K);
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
  hipLaunchKernelGGL(gemm_reduce_kernel, dim3(grid_for((size_t)M * N, 256)), dim3(256), 0, s, (const float*)ws, C, bias, M, N, ldc, ns);
  return check_launch("vp_gemm_f32(reduce)");
}

size_t vp_colsum_workspace_bytes(int R, int C) { return (size_t)colsum_chunks(R) * C * sizeof(float); }

int vp_colsum_f32(const float* x, float* out, int R, int C, void* ws, size_t ws_bytes, vp_stream stream) {
  VP_REQUIRE(x && out && ws && R > 0 && C > 0, "vp_colsum_f32: bad arguments");
  const int nchunk = colsum_chunks(R);
  if (ws_bytes < (size_t)nchunk * C * sizeof(float)) return fail(VP_ERR_WORKSPACE, "vp_colsum_f32: workspace too small");
  const int rpc = (R + nchunk - 1) / nchunk;
  hipStream_t s = (hipStream_t)stream;
  if (C >= 64)
    hipLaunchKernelGGL(colsum_wide_kernel, dim3((C + 255) / 256, nchunk), dim3(256), 0, s, x, (float*)ws, R, C, rpc);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nchunk), dim3(64, 4), 0, s, x, (float*)ws, R, C, rpc);
  int rc = check_launch("vp_colsum_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, s, (const float*)ws, out, C, nchunk);
  return check_launch("vp_colsum_f32(final)");
}
}
