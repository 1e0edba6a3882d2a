// Stand-alone bench / cross-check of the transposed-convolution forward kernels (test infrastructure, not product):
//   old = the library's vp_conv5_scatter_bf16x3 (igemm16_kernel<ProbT16T> / igemm16p_kernel: one phase per workgroup, one tap per K-tile),
//   new = scatter5_kernel (csrc/scatter5.h: all four phases of a tile from one halo patch).  Outputs are compared with each other.
// Build: make -C tools/kbench sbench      Run (GPU box): tools/kbench/sbench B=32 img=128 reps=20 layers=dec3,enc1 clock=0|1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <string>
#include "scatter5.h"
#include "../../vae_play_amd/csrc/split.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

extern "C" int vp_conv5_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Csmall,
                                       int Cbig, int stride, void* stream);
extern "C" const char* vp_last_error(void);

namespace vp {
static void* g_zero = nullptr;
const void* vp_zero_page() { return g_zero; }
}
using namespace vp;

__global__ void fill_split(u16_t* out, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float x = ((h >> 8) * (1.f / 8388608.f) - 1.f) * scale;
    u16_t a, b;
    split_f32(x, a, b);
    out[i] = a;
    out[n + i] = b;
  }
}

template <class F>
static float time_it(F&& f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1000.f / reps;
}

static const char* arg(int argc, char** argv, const char* key, const char* dflt) {
  const size_t n = strlen(key);
  for (int i = 1; i < argc; ++i)
    if (!strncmp(argv[i], key, n) && argv[i][n] == '=') return argv[i] + n + 1;
  return dflt;
}
static bool in_list(const char* list, const char* item) {
  if (!list[0]) return true;
  std::string s(list);
  size_t p = 0;
  while (p <= s.size()) {
    size_t q = s.find(',', p);
    if (q == std::string::npos) q = s.size();
    if (q > p && strstr(item, s.substr(p, q - p).c_str())) return true;
    p = q + 1;
  }
  return false;
}

static void clock_stats(unsigned long long* dbg_dev, size_t nwg, double* mhz, double* cyc) {
  std::vector<unsigned long long> h(4 * nwg);
  CK(hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> f, c;
  for (size_t i = 0; i < nwg; ++i) {
    const double dt = (double)(h[4 * i + 2] - h[4 * i]), dr = (double)(h[4 * i + 3] - h[4 * i + 1]);
    if (dr > 0 && h[4 * i + 2]) { f.push_back(dt / dr * 100.0); c.push_back(dt); }
  }
  if (f.empty()) { *mhz = *cyc = 0; return; }
  std::sort(f.begin(), f.end()); std::sort(c.begin(), c.end());
  *mhz = f[f.size() / 2]; *cyc = c[c.size() / 2];
}

struct Layer { std::string name; int Hs, Cs, Cb; };      // small image side, small channels, big channels

int main(int argc, char** argv) {
  const int B = atoi(arg(argc, argv, "B", "32"));
  const int reps = atoi(arg(argc, argv, "reps", "20"));
  const int img = atoi(arg(argc, argv, "img", "128"));
  const char* lsel = arg(argc, argv, "layers", "");
  const bool want_clock = atoi(arg(argc, argv, "clock", "0")) != 0;
  CK(hipMalloc(&g_zero, 4096)); CK(hipMemset(g_zero, 0, 4096));

  std::vector<Layer> layers;
  const int L = img == 256 ? 5 : img == 128 ? 4 : img == 64 ? 3 : 2;
  for (int i = 1; i < L; ++i) layers.push_back({"enc" + std::to_string(i) + ".dgrad", img >> (i + 1), 64 << i, 64 << (i - 1)});
  for (int i = 0; i < L; ++i) {
    const int size = 64 << (L - 1);
    layers.push_back({"dec" + std::to_string(i) + ".fwd", 8 << i, i == 0 ? size : size >> (i - 1), size >> i});
  }
  unsigned long long* dbg = nullptr;
  const size_t DBG_WG = 1 << 14;
  CK(hipMalloc(&dbg, DBG_WG * 32));

  printf("# transposed-convolution forward, B=%d img=%d reps=%d (us per launch; TF = algorithmic TFLOP/s)\n", B, img, reps);
  for (const Layer& ly : layers) {
    if (!in_list(lsel, ly.name.c_str())) continue;
    const int Hs = ly.Hs, Hb = 2 * Hs, Cs = ly.Cs, Cb = ly.Cb;
    const size_t small_n = (size_t)B * Hs * Hs * Cs, w_n = (size_t)Cs * Cb * 25, out_n = (size_t)B * Hb * Hb * Cb;
    const double gflop = 50.0 * (double)B * Hs * Hs * Cs * Cb * 1e-9;
    u16_t *small, *w;
    float *out_old, *out_new;
    CK(hipMalloc(&small, small_n * 4)); CK(hipMalloc(&w, w_n * 4));
    CK(hipMalloc(&out_old, out_n * 4)); CK(hipMalloc(&out_new, out_n * 4));
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, small, small_n, 4242u, 1.0f);
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, w, w_n, 12345u, 0.05f);
    CK(hipDeviceSynchronize());
    auto run_old = [&]() {
      if (vp_conv5_scatter_bf16x3(small, w, out_old, B, Hs, Hs, Cs, Cb, 2, nullptr) != 0) { fprintf(stderr, "old: %s\n", vp_last_error()); exit(3); }
    };
    printf("%-11s Cs=%-3d Cb=%-3d Hs=%-3d %5.1f GF |", ly.name.c_str(), Cs, Cb, Hs, gflop);
    run_old();
    CK(hipDeviceSynchronize());
    const float us_old = time_it(run_old, reps);
    printf(" old %6.1f us %5.1f TF |", us_old, gflop / us_old * 1e3);
    if (scatter5_ok(B, Hs, Hs, Cs, Cb)) {
      CK(hipMemset(out_new, 0xff, out_n * 4));
      scatter5_launch(small, w, out_new, B, Hs, Hs, Cs, Cb, 0, nullptr, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      const float us = time_it([&] { scatter5_launch(small, w, out_new, B, Hs, Hs, Cs, Cb, 0, nullptr, nullptr); }, reps);
      const int wgs = scatter5_items(B, Hs, Hs, Cb);
      printf(" new wgs=%-4d %6.1f us %5.1f TF (%.2fx)", wgs, us, gflop / us * 1e3, us_old / us);
      if (want_clock) {
        CK(hipMemset(dbg, 0, DBG_WG * 32));
        for (int i = 0; i < 5; ++i) scatter5_launch(small, w, out_new, B, Hs, Hs, Cs, Cb, 0, nullptr, dbg);
        CK(hipDeviceSynchronize());
        double mhz, cyc; clock_stats(dbg, (size_t)((wgs + 7) / 8) * 8, &mhz, &cyc);
        printf(" clk=%.0f wg=%.1fk", mhz, cyc * 1e-3);
      }
      std::vector<float> ho(out_n), hn(out_n);
      CK(hipMemcpy(ho.data(), out_old, out_n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hn.data(), out_new, out_n * 4, hipMemcpyDeviceToHost));
      double mx = 0, ss = 0;
      size_t bad = 0;
      for (size_t i = 0; i < out_n; ++i) {
        const double d = fabs((double)hn[i] - (double)ho[i]);
        if (!(d == d)) { ++bad; continue; }
        if (d > mx) mx = d;
        ss += (double)ho[i] * ho[i];
      }
      const double rms = sqrt(ss / out_n + 1e-300);
      printf(" | new vs old: max|d|/rms = %.2e, %zu NaN%s", mx / rms, bad, (mx / rms > 3e-5 || bad) ? " !!!" : "");
    } else {
      printf(" new: shape not taken");
    }
    printf("\n");
    fflush(stdout);
    CK(hipFree(small)); CK(hipFree(w)); CK(hipFree(out_old)); CK(hipFree(out_new));
  }
  return 0;
}
