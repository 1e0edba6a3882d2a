// fp32 -> (hi, lo) 16-bit planes, x = hi + lo.
//   SPLIT_BF16 (the "bf16x3" kernels): hi = bf16_rne(x), lo = bf16_rne(x - hi): 16 significant bits, fp32's exponent range.
//   SPLIT_F16  (the "f16x2" kernels):  hi = f16_rne(x),  lo = f16_rne(x - hi):  up to 22 significant bits inside fp16's range
//   (|x| <= 65504, saturating; below 2^-14 the planes are subnormal with an absolute step of 2^-24), so a producer of
//   gradients multiplies by a power-of-two scale first and the consuming kernel divides it out of its accumulators.
// hipcc lowers the casts to v_cvt_pk_bf16_f32 / v_cvt_f16_f32 (round-to-nearest-even).
#pragma once
#include <hip/hip_runtime.h>

namespace vp {
typedef unsigned short u16_t;
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

enum : int { SPLIT_BF16 = 0, SPLIT_F16 = 1 };

__device__ __forceinline__ void split_f32(float x, u16_t& hi, u16_t& lo) {
  const __bf16 h = (__bf16)x;
  const __bf16 l = (__bf16)(x - (float)h);
  hi = __builtin_bit_cast(u16_t, h);
  lo = __builtin_bit_cast(u16_t, l);
}

__device__ __forceinline__ void split_f16(float x, u16_t& hi, u16_t& lo) {
  x = __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f);       // saturate instead of overflowing to inf
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  hi = __builtin_bit_cast(u16_t, h);
  lo = __builtin_bit_cast(u16_t, l);
}

__device__ __forceinline__ void split_f32(float x, u16_t& hi, u16_t& lo, int fmt) {
  if (fmt == SPLIT_F16) split_f16(x, hi, lo);
  else split_f32(x, hi, lo);
}

// store 4 consecutive elements into both planes (8-B stores)
__device__ __forceinline__ void store_split4(u16_t* hi_plane, size_t plane_elems, size_t idx, float a, float b, float c, float d,
                                             int fmt = SPLIT_BF16) {
  u16x4_t h, l;
  u16_t t0, t1;
  split_f32(a, t0, t1, fmt); h[0] = t0; l[0] = t1;
  split_f32(b, t0, t1, fmt); h[1] = t0; l[1] = t1;
  split_f32(c, t0, t1, fmt); h[2] = t0; l[2] = t1;
  split_f32(d, t0, t1, fmt); h[3] = t0; l[3] = t1;
  *reinterpret_cast<u16x4_t*>(hi_plane + idx) = h;
  *reinterpret_cast<u16x4_t*>(hi_plane + plane_elems + idx) = l;
}
}  // namespace vp
