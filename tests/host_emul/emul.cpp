// Host emulation of the implicit-GEMM problem descriptors (TEST INFRASTRUCTURE ONLY).
//
// Compiles vae_play_amd/csrc/problems.h for the CPU and evaluates C = A * B^T with a naive
// loop through exactly the accessors the MFMA kernel uses (a_row/a_load/a_load_km/b_*/store,
// z_setup).  tests/test_index_math.py compares the result with torch's conv2d /
// conv_transpose2d / autograd on CPU, so phase decomposition, tap order, padding, strides and
// split-K bookkeeping are validated without a GPU.  This library is never loaded by the
// product package.
#include <vector>
#include <cstring>
#include <cstdio>
#include "../../vae_play_amd/csrc/problems.h"

using namespace vp;

namespace vp {
static const unsigned int host_zero_page[16] = {0};
const void* vp_zero_page() { return host_zero_page; }
}

template <class P>
static void emulate(const P& p, int M, int N, int gz) {
  for (int zi = 0; zi < gz; ++zi) {
    typename P::ZCtx z;
    p.z_setup(zi, z);
    const int kb = z.k_begin, ke = z.k_end;
    const int Kp = ke > kb ? ((ke - kb + 3) / 4) * 4 : 0;
    const int Mp = ((M + 3) / 4) * 4, Np = ((N + 3) / 4) * 4;
    std::vector<float> A((size_t)Mp * (Kp + 4), 0.f), B((size_t)Np * (Kp + 4), 0.f);
    const int ld = Kp + 4;
    if constexpr (!P::A_KM) {
      for (int m = 0; m < Mp; ++m) {
        auto row = p.a_row(m, z);
        for (int k = 0; k < Kp; k += 4) {
          vp_f32x4 v = p.a_load(row, kb + k, z);
          for (int j = 0; j < 4; ++j) A[(size_t)m * ld + k + j] = v[j];
        }
      }
    } else {
      for (int k = 0; k < Kp; ++k)
        for (int m = 0; m < Mp; m += 4) {
          vp_f32x4 v = p.a_load_km(kb + k, m, z);
          for (int j = 0; j < 4; ++j) A[(size_t)(m + j) * ld + k] = v[j];
        }
    }
    if constexpr (!P::B_KM) {
      for (int n = 0; n < Np; ++n) {
        auto row = p.b_row(n, z);
        for (int k = 0; k < Kp; k += 4) {
          vp_f32x4 v = p.b_load(row, kb + k, z);
          for (int j = 0; j < 4; ++j) B[(size_t)n * ld + k + j] = v[j];
        }
      }
    } else {
      for (int k = 0; k < Kp; ++k)
        for (int n = 0; n < Np; n += 4) {
          vp_f32x4 v = p.b_load_km(kb + k, n, z);
          for (int j = 0; j < 4; ++j) B[(size_t)(n + j) * ld + k] = v[j];
        }
    }
    // rows/cols past M/N must have been zero-filled by the accessors and dropped by store()
    for (int m = 0; m < Mp; ++m)
      for (int n = 0; n < Np; ++n) {
        double acc = 0.0;
        const float* a = &A[(size_t)m * ld];
        const float* b = &B[(size_t)n * ld];
        for (int k = 0; k < Kp; ++k) acc += (double)a[k] * (double)b[k];
        p.store(m, n, (float)acc, z);
      }
  }
}

extern "C" {
int emul_conv_gather(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int);
int emul_conv_scatter(const float*, const float*, float*, int, int, int, int, int, int, int, int, int);
int emul_conv_wgrad(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int);

int emul_conv5_gather(const float* big, const float* wp0, const float* bias, float* out, int B, int Hs, int Ws,
                      int Cb, int Cs, int stride, int act) {
  return emul_conv_gather(big, wp0, bias, out, B, Hs, Ws, Hs * stride, Ws * stride, Cb, Cs, 5, stride, act);
}

int emul_conv_gather(const float* big, const float* wp0, const float* bias, float* out, int B, int Hs, int Ws, int Hb, int Wb,
                     int Cb, int Cs, int ks, int stride, int act) {
  ConvGeom g = make_geom(B, Hs, Ws, Cs, Cb, stride, ks, Hb, Wb);
  ProbF p = make_probF(big, wp0, bias, out, g, act);
  emulate(p, p.M, p.N, 1);
  return 0;
}

int emul_conv5_scatter(const float* small, const float* wp1, float* out, int B, int Hs, int Ws, int Cs, int Cb,
                       int stride) {
  return emul_conv_scatter(small, wp1, out, B, Hs, Ws, Hs * stride, Ws * stride, Cs, Cb, 5, stride);
}

int emul_conv_scatter(const float* small, const float* wp1, float* out, int B, int Hs, int Ws, int Hb, int Wb, int Cs, int Cb,
                      int ks, int stride) {
  ConvGeom g = make_geom(B, Hs, Ws, Cs, Cb, stride, ks, Hb, Wb);
  ProbT p = make_probT(small, wp1, out, g);
  emulate(p, p.M, p.N, stride * stride);
  return 0;
}

// returns nsplit used; slab must hold wgrad_slab_floats
int emul_conv5_wgrad(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Cb, int Cs,
                     int stride, int force_split) {
  return emul_conv_wgrad(big, small, dw_ref, B, Hs, Ws, Hs * stride, Ws * stride, Cb, Cs, 5, stride, force_split);
}

int emul_conv_wgrad(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cb, int Cs,
                    int ks, int stride, int force_split) {
  ConvGeom g = make_geom(B, Hs, Ws, Cs, Cb, stride, ks, Hb, Wb);
  int ns = force_split > 0 ? force_split : wgrad_nsplit(g);
  std::vector<float> slab(wgrad_slab_floats(g, ns), 0.f);
  ProbW p = make_probW(big, small, slab.data(), g, ns);
  emulate(p, p.M, p.N, g.nt * ns);
  // reduce: dw_ref[cs][cb][tap] = sum_split slab[split][tap][cs][cb]
  for (int cs = 0; cs < Cs; ++cs)
    for (int cb = 0; cb < Cb; ++cb)
      for (int t = 0; t < g.nt; ++t) {
        double s = 0;
        for (int sp = 0; sp < ns; ++sp) s += slab[(((size_t)sp * g.nt + t) * Cs + cs) * Cb + cb];
        dw_ref[((size_t)cs * Cb + cb) * g.nt + t] = (float)s;
      }
  return ns;
}

// mode 0: NT (A MK, B MK)   1: NN (A MK, B KM)   2: TN (A KM, B KM)
int emul_gemm(const float* A, long sam, long sak, const float* Bm, long sbn, long sbk, float* C, int ldc,
              const float* bias, int M, int N, int K, int mode, int force_split) {
  int ns = force_split > 0 ? force_split : gemm_nsplit(M, N, K);
  std::vector<float> slab;
  float* dst = C;
  if (ns > 1) { slab.assign((size_t)ns * M * N, 0.f); dst = slab.data(); }
  if (mode == 0) { auto p = make_probG<false, false>(A, sam, sak, Bm, sbn, sbk, dst, ldc, bias, M, N, K, ns); emulate(p, M, N, ns); }
  else if (mode == 1) { auto p = make_probG<false, true>(A, sam, sak, Bm, sbn, sbk, dst, ldc, bias, M, N, K, ns); emulate(p, M, N, ns); }
  else { auto p = make_probG<true, true>(A, sam, sak, Bm, sbn, sbk, dst, ldc, bias, M, N, K, ns); emulate(p, M, N, ns); }
  if (ns > 1)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = bias ? bias[n] : 0.0;
        for (int sp = 0; sp < ns; ++sp) s += slab[((size_t)sp * M + m) * N + n];
        C[(size_t)m * ldc + n] = (float)s;
      }
  return ns;
}
}
