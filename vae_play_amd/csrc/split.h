// fp32 -> (hi, lo) bf16 planes: x = hi + lo, hi = bf16_rne(x), lo = bf16_rne(x - hi).
// hipcc lowers the casts to v_cvt_pk_bf16_f32 (NaN-preserving, round-to-nearest-even).
#pragma once
#include <hip/hip_runtime.h>

namespace vp {
typedef unsigned short u16_t;
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_f32(float x, u16_t& hi, u16_t& lo) {
  const __bf16 h = (__bf16)x;
  const __bf16 l = (__bf16)(x - (float)h);
  hi = __builtin_bit_cast(u16_t, h);
  lo = __builtin_bit_cast(u16_t, l);
}

// store 4 consecutive elements into both planes (8-B stores)
__device__ __forceinline__ void store_split4(u16_t* hi_plane, size_t plane_elems, size_t idx, float a, float b, float c, float d) {
  u16x4_t h, l;
  u16_t t0, t1;
  split_f32(a, t0, t1); h[0] = t0; l[0] = t1;
  split_f32(b, t0, t1); h[1] = t0; l[1] = t1;
  split_f32(c, t0, t1); h[2] = t0; l[2] = t1;
  split_f32(d, t0, t1); h[3] = t0; l[3] = t1;
  *reinterpret_cast<u16x4_t*>(hi_plane + idx) = h;
  *reinterpret_cast<u16x4_t*>(hi_plane + plane_elems + idx) = l;
}
}  // namespace vp
