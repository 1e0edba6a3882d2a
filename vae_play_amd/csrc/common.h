// Shared host-side plumbing for the C ABI: status codes, thread-local error text, checks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include "../../include/vaeplay_hip.h"
#include "env.h"

namespace vp {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return VP_OK;
}

#define VP_REQUIRE(cond, ...)                                 \
  do {                                                        \
    if (!(cond)) return ::vp::fail(VP_ERR_ARG, __VA_ARGS__);  \
  } while (0)

inline unsigned grid_for(size_t work_items, int block, unsigned cap = 256 * 8) {
  size_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// wave64 sum via DPP-free shuffles (64-lane wavefront)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace vp
