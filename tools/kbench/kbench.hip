// Standalone kernel bench / cross-check for the split-bf16 implicit-GEMM kernels (test infrastructure, not product):
//   * runs every gather / scatter launch shape of the VAE step at batch B on igemm16_kernel (register staging,
//     igemm16.h) and on the tile configurations of the pipelined LDS-DMA kernel (igemm16p.h), compares the outputs
//     BIT FOR BIT and times both with HIP events;
//   * clock=1 adds the in-kernel shader clock (s_memtime / s_memrealtime stamps per workgroup) of each configuration;
//   * mode=probe answers one hardware question the design depends on (what an out-of-range buffer_load ... lds writes).
// Build: make -C tools/kbench
// Run (GPU box): tools/kbench/kbench B=32 reps=20 img=128 mode=all|probe layers=dec2,dec3 cfgs=3,6 buf=0|1|2 clock=0|1 v1=0|1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <string>
#include "../../vae_play_amd/csrc/igemm16p.h"
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

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void probe_oob(const char* src, unsigned* out, int nbytes) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2048];
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)lds)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000u;          // out of range for odd lanes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)lds)[i];
}

static void run_probe() {
  char* src; unsigned* out;
  CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 1024));
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_oob, dim3(1), dim3(64), 0, 0, src, out, 1024);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> o(256);
  CK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
  printf("probe buffer_load..lds OOB: lane0 (in range) -> %08x %08x ; lane1 (out of range) -> %08x %08x %08x %08x ; lane2 -> %08x\n",
         o[0], o[1], o[4], o[5], o[6], o[7], o[8]);
  printf("  => out-of-range LDS-DMA lanes write %s\n", o[4] == 0 ? "ZEROS" : (o[4] == 0xdeadbeefu ? "NOTHING (LDS keeps old bytes)" : "something else"));
}

struct Layer { std::string name; char fam; int Hs, Cs, Cb; };   // fam 'F' gather (small = f(big)), 'T' scatter

static int xcd_for(long M, long N, int bm, int bn) {
  const long gx = (M + bm - 1) / bm, gy = (N + bn - 1) / bn;
  return (gx % 8 == 0 && gy >= 2) ? 1 : 0;
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

static size_t count_diff(const float* a, const float* b, size_t n, double* rel = nullptr) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  double mx = 0, ss = 0;
  for (size_t i = 0; i < n; ++i) {
    bad += memcmp(&ha[i], &hb[i], 4) != 0;
    const double d = fabs((double)ha[i] - hb[i]);
    if (d > mx || d != d) mx = d;
    ss += (double)ha[i] * ha[i];
  }
  if (rel) *rel = mx / sqrt(ss / n + 1e-300);      // max |diff| relative to the reference's rms
  return bad;
}

static const char* arg(int argc, char** argv, const char* key, const char* dflt) {
  const size_t n = strlen(key);
  for (int i = 1; i < argc; ++i)
    if (!strncmp(argv[i], key, n) && argv[i][n] == '=') return argv[i] + n + 1;
  return dflt;
}
static bool in_list(const char* list, const char* item) {      // comma-separated substrings; empty list = everything
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
static bool in_ints(const char* list, int v) {
  if (!list[0]) return true;
  char buf[16]; snprintf(buf, 16, "%d", v);
  std::string s = std::string(",") + list + ",";
  return s.find(std::string(",") + buf + ",") != std::string::npos;
}

// median in-kernel shader clock (MHz) and median workgroup lifetime (cycles) from the stamps
static void clock_stats(unsigned long long* dbg_dev, size_t nwg, double* mhz, double* cyc) {
  std::vector<unsigned long long> h(4 * nwg);
  CK(hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> f, c;
  for (size_t i = 0; i < nwg; ++i) {
    const double dt = (double)(h[4 * i + 2] - h[4 * i]), dr = (double)(h[4 * i + 3] - h[4 * i + 1]);
    if (dr > 0) { f.push_back(dt / dr * 100.0); c.push_back(dt); }
  }
  if (f.empty()) { *mhz = *cyc = 0; return; }
  std::sort(f.begin(), f.end()); std::sort(c.begin(), c.end());
  *mhz = f[f.size() / 2]; *cyc = c[c.size() / 2];
}

int main(int argc, char** argv) {
  const int B = atoi(arg(argc, argv, "B", "32"));
  const int reps = atoi(arg(argc, argv, "reps", "20"));
  const int img = atoi(arg(argc, argv, "img", "128"));
  const char* mode = arg(argc, argv, "mode", "all");
  const char* lsel = arg(argc, argv, "layers", "");
  const char* csel = arg(argc, argv, "cfgs", "");
  const int bufsel = atoi(arg(argc, argv, "buf", "1"));      // 0: pointer DMA, 1: buffer DMA, 2: both
  const bool want_clock = atoi(arg(argc, argv, "clock", "0")) != 0;
  const bool want_v1 = atoi(arg(argc, argv, "v1", "1")) != 0;
  const int m16sel = atoi(arg(argc, argv, "m16", "0"));      // 0: 32x32x16 MFMA, 1: 16x16x32, 2: both
  CK(hipMalloc(&g_zero, 4096)); CK(hipMemset(g_zero, 0, 4096));
  if (!strcmp(mode, "probe") || !strcmp(mode, "all")) run_probe();
  if (!strcmp(mode, "probe")) return 0;

  std::vector<Layer> layers;
  const int L = img == 256 ? 5 : img == 128 ? 4 : img == 64 ? 3 : 2;
  std::vector<int> ch = {0};
  for (int i = 0; i < L; ++i) ch.push_back(64 << i);
  for (int i = 1; i < L; ++i) {       // encoder block i: ch[i] -> ch[i+1], output img >> (i+1)
    const int Hs = img >> (i + 1);
    layers.push_back({"enc" + std::to_string(i) + ".fwd", 'F', Hs, ch[i + 1], ch[i]});
    layers.push_back({"enc" + std::to_string(i) + ".dgrad", 'T', Hs, ch[i + 1], ch[i]});
  }
  for (int i = 0; i < L; ++i) {       // decoder block i: cin -> cout, input 8 << i
    const int size = 64 << (L - 1);
    const int cin = i == 0 ? size : size >> (i - 1), cout = size >> i;
    layers.push_back({"dec" + std::to_string(i) + ".fwd", 'T', 8 << i, cin, cout});
    layers.push_back({"dec" + std::to_string(i) + ".dgrad", 'F', 8 << i, cin, cout});
  }
  unsigned long long* dbg = nullptr;
  const size_t DBG_WG = 1 << 16;
  CK(hipMalloc(&dbg, DBG_WG * 32));

  printf("# B=%d img=%d reps=%d  (us per launch; TF = algorithmic TFLOP/s; bit = elements differing from igemm16_kernel;"
         " clk = median in-kernel MHz, wg = median workgroup lifetime in kcycles)\n", B, img, reps);
  for (const Layer& ly : layers) {
    if (!in_list(lsel, ly.name.c_str())) continue;
    const int Hs = ly.Hs, Hb = 2 * Hs, Cs = ly.Cs, Cb = ly.Cb;
    const bool isF = ly.fam == 'F';
    const double gflop = 50.0 * B * Hs * Hs * (double)Cs * Cb * 1e-9;
    // F: activation = big [B,Hb,Hb,Cb], weights P0 [Cs][25][Cb], out small [B,Hs,Hs,Cs]
    // T: activation = small [B,Hs,Hs,Cs], weights P1 [Cb][25][Cs], out big [B,Hb,Hb,Cb]
    const size_t act_n = isF ? (size_t)B * Hb * Hb * Cb : (size_t)B * Hs * Hs * Cs;
    const size_t w_n = (size_t)Cs * Cb * 25;
    const size_t out_n = isF ? (size_t)B * Hs * Hs * Cs : (size_t)B * Hb * Hb * Cb;
    if (act_n >= (1ull << 29) || out_n >= (1ull << 31)) { printf("%-12s skipped (too large)\n", ly.name.c_str()); continue; }
    u16_t *act, *w; float *out_ref, *out;
    CK(hipMalloc(&act, act_n * 4)); CK(hipMalloc(&w, w_n * 4));
    CK(hipMalloc(&out_ref, out_n * 4)); CK(hipMalloc(&out, out_n * 4));
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, act, act_n, 12345u, 1.0f);
    hipLaunchKernelGGL(fill_split, dim3(2048), dim3(256), 0, 0, w, w_n, 777u, 0.05f);
    CK(hipDeviceSynchronize());

    const ConvGeom g = make_geom(B, Hs, Hs, Cs, Cb, 2, 5, Hb, Hb);
    const long M = (long)B * Hs * Hs;
    PF16 pf; PT16 pt;
    int N, gz, nsplit, ctile, kmin;
    if (isF) {
      pf.zero = g_zero; pf.g = g;
      pf.big = act; pf.big_plane = act_n; pf.w = w; pf.w_plane = w_n;
      pf.bias = nullptr; pf.act = ACT_NONE; pf.M = (int)M; pf.N = Cs; pf.K = 25 * Cb;
      const long tiles = ((M + 127) / 128) * ((pf.N + 63) / 64);
      pf.nsplit = (Cb % 64 == 0 && tiles < 384 && pf.K >= 4096) ? 2 : 1;
      pf.k_per_split = pf.nsplit == 2 ? ((pf.K / 64 + 1) / 2) * 64 : pf.K;
      N = Cs; nsplit = pf.nsplit; gz = nsplit; ctile = Cb; kmin = pf.k_per_split;
    } else {
      pt.zero = g_zero; pt.g = g;
      pt.small = act; pt.small_plane = act_n; pt.w = w; pt.w_plane = w_n;
      pt.M = (int)M; pt.N = Cb;
      const long tiles = ((M + 127) / 128) * ((pt.N + 63) / 64) * 4;
      pt.nsplit = (Cs % 64 == 0 && tiles < 384 && 4 * Cs >= 1024) ? 2 : 1;
      N = Cb; nsplit = pt.nsplit; gz = 4 * nsplit; ctile = Cs; kmin = 4 * Cs / nsplit;
    }
    auto run_v1 = [&](float* o) {
      const Tile16 t = choose_tile16(M, N, gz);
      if (nsplit == 2) CK(hipMemsetAsync(o, 0, out_n * 4, 0));
      if (isF) { ProbF16 q = pf; q.out = o; q.xcd_map = xcd_for(M, N, t.bm, t.bn); launch_igemm16(q, M, N, gz, 0, ctile); }
      else { ProbT16 q = pt; q.out = o; q.xcd_map = xcd_for(M, N, t.bm, t.bn); launch_igemm16(q, M, N, gz, 0, ctile); }
    };
    auto run_p = [&](float* o, int cfg, bool buf, unsigned long long* d, bool m16) {
      int bm, bn; pcfg_tile(cfg, bm, bn);
      if (nsplit == 2) CK(hipMemsetAsync(o, 0, out_n * 4, 0));
      if (isF) { PF16 q = pf; q.out = o; q.dbg = d; q.xcd_map = xcd_for(M, N, bm, bn); launch_igemm16p(q, cfg, M, N, gz, 0, ctile, buf, m16); }
      else { PT16 q = pt; q.out = o; q.dbg = d; q.xcd_map = xcd_for(M, N, bm, bn); launch_igemm16p(q, cfg, M, N, gz, 0, ctile, buf, m16); }
    };
    run_v1(out_ref);
    CK(hipDeviceSynchronize());
    const Tile16 t1 = choose_tile16(M, N, gz);
    printf("%-11s %c M=%-7ld N=%-4d ct=%-4d ns=%d |", ly.name.c_str(), ly.fam, M, N, ctile, nsplit);
    if (want_v1) {
      const float us1 = time_it([&] { run_v1(out_ref); }, reps);
      printf(" v1 %dx%-3d %6.1f us %5.1f TF |", t1.bm, t1.bn, us1, gflop / us1 * 1e3);
    }
    for (int cfg = 1; cfg < PCFG_COUNT; ++cfg) {
      if (!in_ints(csel, cfg)) continue;
      int bm, bn; pcfg_tile(cfg, bm, bn);
      if (bn > N || bm > M || kmin / 32 < 4) continue;
      for (int v = 0; v < 3; ++v) {       // 0: pointer DMA, 1: buffer DMA, 2: buffer DMA + 16x16x32 MFMA
        const bool buf = v >= 1, m16 = v == 2;
        if (!m16 && m16sel == 1) continue;
        if (m16 && m16sel == 0) continue;
        if (!m16 && bufsel != 2 && (int)buf != bufsel) continue;
        CK(hipMemset(out, 0xff, out_n * 4));
        run_p(out, cfg, buf, nullptr, m16);
        CK(hipDeviceSynchronize());
        double rel = 0;
        const size_t bad = count_diff(out_ref, out, out_n, &rel);
        const float us = time_it([&] { run_p(out, cfg, buf, nullptr, m16); }, reps);
        if (m16) printf(" c%dq %dx%d %6.1f us %5.1f TF rel=%.1e%s", cfg, bm, bn, us, gflop / us * 1e3, rel, rel > 2e-5 ? " !!!" : "");
        else printf(" c%d%s %dx%d %6.1f us %5.1f TF bit=%zu%s", cfg, buf ? "b" : "", bm, bn, us, gflop / us * 1e3, bad, bad ? " !!!" : "");
        if (want_clock) {
          const size_t nwg = (size_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * gz;
          if (nwg <= DBG_WG) {
            for (int i = 0; i < 5; ++i) run_p(out, cfg, buf, dbg, m16);
            CK(hipDeviceSynchronize());
            double mhz, cyc; clock_stats(dbg, nwg, &mhz, &cyc);
            printf(" clk=%.0f wg=%.1fk n=%zu", mhz, cyc * 1e-3, nwg);
          }
        }
        printf(" |");
        fflush(stdout);
      }
    }
    printf("\n");
    CK(hipFree(act)); CK(hipFree(w)); CK(hipFree(out_ref)); CK(hipFree(out));
  }
  return 0;
}
