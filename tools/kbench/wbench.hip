// Stand-alone bench / cross-check of the weight-gradient kernels (test infrastructure, not product):
//   old = igemm16_kernel<ProbW16T> (one tap per workgroup; tap pairs for 64 big channels), new = wgrad5_kernel (csrc/wgrad5.h:
//   one kernel row of taps per workgroup).  Both write split-K slabs; the slabs are summed on the host in fp64 and compared with
//   each other and, on sampled outputs, with a direct fp64 evaluation of the defining sum over the split operands.
// Build: make -C tools/kbench wbench      Run (GPU box): tools/kbench/wbench B=32 img=128 reps=20 layers=dec1,dec2 clock=0|1 blocks=256
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <string>
#define VP_WGRAD5_Q 1
#include "../../vae_play_amd/csrc/wgrad5.h"
#include "../../vae_play_amd/csrc/split.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

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

static float bf16_to_f(u16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

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

struct Layer { std::string name; int Hs, Cs, Cb; };

// sum of the split slabs on the host, fp64
static std::vector<double> sum_slabs(const float* dev, int ns, size_t per) {
  std::vector<float> h((size_t)ns * per);
  CK(hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> s(per, 0.0);
  for (int i = 0; i < ns; ++i)
    for (size_t j = 0; j < per; ++j) s[j] += (double)h[(size_t)i * per + j];
  return s;
}

int main(int argc, char** argv) {
  const int B = atoi(arg(argc, argv, "B", "32"));
  const int reps = atoi(arg(argc, argv, "reps", "20"));
  const int img = atoi(arg(argc, argv, "img", "128"));
  const char* lsel = arg(argc, argv, "layers", "");
  const bool want_clock = atoi(arg(argc, argv, "clock", "0")) != 0;
  const bool want_old = atoi(arg(argc, argv, "old", "1")) != 0;
  const int check = atoi(arg(argc, argv, "check", "1"));
  const bool want_m16 = atoi(arg(argc, argv, "m16", "1")) != 0;
  CK(hipMalloc(&g_zero, 4096)); CK(hipMemset(g_zero, 0, 4096));

  std::vector<Layer> layers;
  const int L = img == 256 ? 5 : img == 128 ? 4 : img == 64 ? 3 : 2;
  std::vector<int> ch = {0};
  for (int i = 0; i < L; ++i) ch.push_back(64 << i);
  for (int i = 1; i < L; ++i) layers.push_back({"enc" + std::to_string(i), img >> (i + 1), ch[i + 1], ch[i]});
  for (int i = 0; i < L; ++i) {
    const int size = 64 << (L - 1);
    layers.push_back({"dec" + std::to_string(i), 8 << i, i == 0 ? size : size >> (i - 1), size >> i});
  }
  unsigned long long* dbg = nullptr;
  const size_t DBG_WG = 1 << 14;
  CK(hipMalloc(&dbg, DBG_WG * 32));

  printf("# weight gradients, B=%d img=%d reps=%d (us per launch of the MAIN kernel, slab reduction not included; TF = algorithmic TFLOP/s)\n", B, img, reps);
  for (const Layer& ly : layers) {
    if (!in_list(lsel, ly.name.c_str())) continue;
    const int Hs = ly.Hs, Hb = 2 * Hs, Cs = ly.Cs, Cb = ly.Cb;
    const ConvGeom g = make_geom(B, Hs, Hs, Cs, Cb, 2, 5, Hb, Hb);
    const size_t K = (size_t)B * Hs * Hs;
    const size_t small_n = K * Cs, big_n = (size_t)B * Hb * Hb * Cb;
    const double gflop = 50.0 * (double)K * Cs * Cb * 1e-9;
    const size_t per = (size_t)25 * Cs * Cb;
    u16_t *small, *big;
    CK(hipMalloc(&small, small_n * 4)); CK(hipMalloc(&big, big_n * 4));
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, small, small_n, 4242u, 0.05f);
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, big, big_n, 12345u, 1.0f);
    CK(hipDeviceSynchronize());

    // ---- old kernels (the library's dispatch, conv16_impl.h) ----
    int ns_old = wgrad_nsplit(g);
    const bool pair = Cb == 64 && Cs % 128 == 0;
    if (pair) { long n2 = 2L * ns_old, maxs = ((long)K + 511) / 512; if (n2 > maxs) n2 = maxs; if (n2 > 64) n2 = 64; if (n2 > ns_old) ns_old = (int)n2; }
    float* slab_old = nullptr;
    CK(hipMalloc(&slab_old, (size_t)ns_old * per * 4));
    auto run_old = [&]() {
      const int perk = (((int)K + ns_old - 1) / ns_old + 31) / 32 * 32;
      if (pair) {
        ProbW16T<true, 0, true> p;
        p.alpha = 1.f; p.zero = g_zero; p.g = g; p.big = big; p.big_plane = big_n; p.small = small; p.small_plane = small_n;
        p.slab = slab_old; p.M = Cs; p.N = 2 * Cb; p.K = (int)K; p.nsplit = ns_old; p.k_per_split = perk;
        hipLaunchKernelGGL((igemm16_kernel<ProbW16T<true, 0, true>, 128, 128, 2, 2, 32, true>), dim3(Cs / 128, 1, 13u * ns_old), dim3(256), 0, 0, p);
      } else {
        ProbW16 p;
        p.alpha = 1.f; p.zero = g_zero; p.g = g; p.big = big; p.big_plane = big_n; p.small = small; p.small_plane = small_n;
        p.slab = slab_old; p.M = Cs; p.N = Cb; p.K = (int)K; p.nsplit = ns_old; p.k_per_split = perk;
        launch_igemm16(p, p.M, p.N, 25 * ns_old, 0);
      }
    };
    // ---- new kernel ----
    const int bn = wgrad5_bn(g);
    int kper = 0, ns_new = 0, slabs_new = 0;
    float* slab_new = nullptr;
    if (bn) {
      ns_new = wgrad5_nsplit(g, bn, &kper);
      CK(hipMalloc(&slab_new, wgrad5_slab_floats(g, bn, ns_new) * 4));
    }
    bool m16 = false;
    auto run_new = [&](unsigned long long* d) { wgrad5_launch<0>(big, small, slab_new, g, bn, ns_new, kper, 1.f, 0, &slabs_new, d, m16); };

    printf("%-5s Cs=%-3d Cb=%-3d K=%-6zu %5.1f GF |", ly.name.c_str(), Cs, Cb, K, gflop);
    float us_old = 0;
    if (want_old) {
      run_old();
      CK(hipDeviceSynchronize());
      us_old = time_it(run_old, reps);
      printf(" old ns=%-2d slab %5.1f MB %6.1f us %5.1f TF |", ns_old, ns_old * per * 4e-6, us_old, gflop / us_old * 1e3);
    }
    for (int variant = 0; bn && variant < (bn == 128 && want_m16 ? 2 : 1); ++variant) {
      m16 = variant == 1;
      if (m16) printf("\n%-5s %40s |", "", "16x16x32 form");
      CK(hipMemset(slab_new, 0xff, wgrad5_slab_floats(g, bn, ns_new) * 4));
      run_new(nullptr);
      CK(hipDeviceSynchronize());
      const float us = time_it([&] { run_new(nullptr); }, reps);
      const int wgs = (g.Cs / 128) * (g.Cb / bn) * 5 * ns_new;
      printf(" new bn=%d ns=%-2d wgs=%-3d slab %5.1f MB %6.1f us %5.1f TF", bn, ns_new, wgs, slabs_new * per * 4e-6, us, gflop / us * 1e3);
      if (want_old) printf(" (%.2fx)", us_old / us);
      if (want_clock) {
        CK(hipMemset(dbg, 0, DBG_WG * 32));
        for (int i = 0; i < 5; ++i) run_new(dbg);
        CK(hipDeviceSynchronize());
        double mhz, cyc; clock_stats(dbg, (size_t)((wgs + 7) / 8) * 8, &mhz, &cyc);
        printf(" clk=%.0f wg=%.1fk", mhz, cyc * 1e-3);
      }
      if (check) {
        const std::vector<double> sn = sum_slabs(slab_new, slabs_new, per);
        double mx = 0, ss = 0;
        if (want_old) {
          const std::vector<double> so = sum_slabs(slab_old, ns_old, per);
          for (size_t i = 0; i < per; ++i) { const double d = fabs(sn[i] - so[i]); if (d > mx || d != d) mx = d; ss += so[i] * so[i]; }
          printf(" | new vs old: max|d|/rms = %.2e", mx / sqrt(ss / per + 1e-300));
        }
        // direct fp64 evaluation on samples
        std::vector<u16_t> hs(small_n * 2), hb(big_n * 2);
        CK(hipMemcpy(hs.data(), small, small_n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), big, big_n * 4, hipMemcpyDeviceToHost));
        double worst = 0, rms = 0;
        const int NSAMP = 48;
        std::vector<double> refs(NSAMP);
        for (int sidx = 0; sidx < NSAMP; ++sidx) {
          const int tap = (sidx * 7) % 25, cs = (sidx * 37 + 5) % Cs, cb = (sidx * 53 + 11) % Cb;
          const int r = tap / 5, q = tap % 5;
          double a = 0;
          for (int b = 0; b < B; ++b)
            for (int h = 0; h < Hs; ++h) {
              const int hh = 2 * h - 2 + r;
              if (hh < 0 || hh >= Hb) continue;
              for (int w = 0; w < Hs; ++w) {
                const int ww = 2 * w - 2 + q;
                if (ww < 0 || ww >= Hb) continue;
                const size_t si = ((size_t)(b * Hs + h) * Hs + w) * Cs + cs, bi = ((size_t)(b * Hb + hh) * Hb + ww) * Cb + cb;
                const double xs = (double)bf16_to_f(hs[si]) + (double)bf16_to_f(hs[small_n + si]);
                const double xb = (double)bf16_to_f(hb[bi]) + (double)bf16_to_f(hb[big_n + bi]);
                a += xs * xb;
              }
            }
          refs[sidx] = a;
          rms += a * a;
          const double got = sn[((size_t)tap * Cs + cs) * Cb + cb];
          worst = std::max(worst, fabs(got - a));
        }
        rms = sqrt(rms / NSAMP);
        printf(" | vs fp64 (%d samples): max|d|/rms = %.2e%s", NSAMP, worst / rms, worst / rms > 1e-4 ? " !!!" : "");
      }
    }
    if (!bn) printf(" new: shape not taken");
    printf("\n");
    fflush(stdout);
    CK(hipFree(small)); CK(hipFree(big)); CK(hipFree(slab_old));
    if (slab_new) CK(hipFree(slab_new));
  }
  return 0;
}
