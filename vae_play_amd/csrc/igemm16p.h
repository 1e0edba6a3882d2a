// Pipelined LDS-DMA form of the split-bf16 implicit GEMM (gather / scatter families), gfx950.
//
// igemm16.h stages operands HBM/L2 -> VGPR -> LDS (ds_write_b128: ~79 B/clk/CU, ~830 LDS cycles per
// 128x128x64 tile against 1536 MFMA cycles) and pays two barriers per K-tile.  Here
//   * every 16-B chunk goes L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write);
//   * a ring of STAGES 32-deep K-stages is kept in flight behind a COUNTED s_waitcnt vmcnt(N) and a raw
//     s_barrier (never __syncthreads(): its fence drains the DMA queue); ONE barrier per stage;
//   * fragment reads of the next k-step are issued before the MFMAs of the current one (register double
//     buffer), so that LDS latency is hidden also at one wave per SIMD;
//   * the per-lane part of a gather address is computed ONCE per row; a stage adds a wave-uniform scalar
//     (tap shift * row pitch + channel chunk) and tests two unsigned bounds;
//   * workgroup tiles up to 256x256 with 8 waves (wave tile up to 128x64): bytes staged per MFMA fall
//     from 0.67 KB (128x128 / four 64x64 wave tiles) to 0.33 KB.
// LDS image per stage and plane: unpadded 64-B rows (32 bf16), chunk c of row r stored at physical chunk
// c ^ ((r >> 2) & 3) (applied to the per-lane SOURCE address and again on the fragment read): every
// 16-lane ds_read_b128 group then covers all 64 banks.
// K order and the hi/lo MFMA sequence are exactly those of igemm16_kernel, so results are BIT-IDENTICAL
// to it (tests compare the two kernels with ==).
#pragma once
#include <type_traits>
#include "igemm16.h"

namespace vp {

#if defined(__HIPCC__)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct KIt { int cc, tap, half; };           // wave-uniform K iterator: (channel chunk, tap, 32-half of a 64-chunk)
struct StageU { int dA, dh, dw, dB; };       // wave-uniform stage constants

// KORD = 6: igemm16's 64-deep FAST order (64-channel chunk major, tap minor, two 32-deep halves per tap);
// KORD = 5: its 32-deep FAST order (32-channel chunk major, tap minor).
template <int KORD>
__device__ __forceinline__ void kit_init(KIt& it, int k_begin, int ntap) {
  const int kt = k_begin >> KORD;
  it.cc = kt / ntap;
  it.tap = kt - it.cc * ntap;
  it.half = 0;
}
template <int KORD>
__device__ __forceinline__ void kit_next(KIt& it, int ntap) {
  if constexpr (KORD == 6) {
    it.half ^= 1;
    if (it.half) return;
  }
  if (++it.tap == ntap) { it.tap = 0; ++it.cc; }
}
template <int KORD>
__device__ __forceinline__ int kit_c0(const KIt& it) { return KORD == 6 ? it.cc * 64 + it.half * 32 : it.cc * 32; }

// ---- gather family: A = big activation planes, B = packed P0 weights [Cs][25][Cb] ----------------------
struct PF16 : ProbF16T<true> {
  unsigned long long* dbg = nullptr;   // diagnostics only (tools/kbench): per-workgroup {s_memtime, s_memrealtime} at entry and exit
  struct ARowP { int off0, h0, w0; };
  __device__ __forceinline__ int ntap(const ZCtx&) const { return 25; }
  __device__ __forceinline__ ARowP a_rowp(int m, const ZCtx&) const {
    ARowP r;
    const bool valid = m < M;
    const int mm = valid ? m : 0;
    const int b = (int)g.dHW.div((uint32_t)mm), rem = mm - b * (g.Hs * g.Ws);
    const int hs = (int)g.dW.div((uint32_t)rem), ws = rem - hs * g.Ws;
    r.h0 = valid ? g.stride * hs - 2 : -(1 << 24);
    r.w0 = g.stride * ws - 2;
    r.off0 = ((b * g.Hb + (g.stride * hs - 2)) * g.Wb + r.w0) * g.Cb;
    return r;
  }
  __device__ __forceinline__ int b_rowp(int n, const ZCtx&) const { return n < N ? n * K : -1; }
  template <int KORD>
  __device__ __forceinline__ StageU stage(const KIt& it, const ZCtx&) const {
    StageU u;
    const int rr = (int)(((unsigned)it.tap * 52429u) >> 18), qq = it.tap - 5 * rr;
    const int c0 = kit_c0<KORD>(it);
    u.dh = rr; u.dw = qq;
    u.dA = (rr * g.Wb + qq) * g.Cb + c0;
    u.dB = it.tap * g.Cb + c0;
    return u;
  }
  __device__ __forceinline__ bool a_ok(const ARowP& r, const StageU& u) const {
    return (unsigned)(r.h0 + u.dh) < (unsigned)g.Hb && (unsigned)(r.w0 + u.dw) < (unsigned)g.Wb;
  }
};

// ---- scatter family (phase-decomposed): A = small activation planes, B = packed P1 weights [Cb][25][Cs] ---
struct PT16 : ProbT16T<true> {
  unsigned long long* dbg = nullptr;
  struct ARowP { int off0, h0, w0; };
  __device__ __forceinline__ int ntap(const ZCtx& z) const { return z.th * z.tw; }
  __device__ __forceinline__ ARowP a_rowp(int m, const ZCtx&) const {
    ARowP r;
    const bool valid = m < M;
    const int mm = valid ? m : 0;
    const int b = (int)g.dHW.div((uint32_t)mm), rem = mm - b * (g.Hs * g.Ws);
    const int q = (int)g.dW.div((uint32_t)rem), p = rem - q * g.Ws;
    r.h0 = valid ? q : -(1 << 24);
    r.w0 = p;
    r.off0 = ((b * g.Hs + q) * g.Ws + p) * g.Cs;
    return r;
  }
  __device__ __forceinline__ int b_rowp(int n, const ZCtx&) const { return n < N ? n * 25 * g.Cs : -1; }
  template <int KORD>
  __device__ __forceinline__ StageU stage(const KIt& it, const ZCtx& z) const {
    StageU u;
    const int rp = div_small(it.tap, z.tw), qp = it.tap - rp * z.tw;
    const int c0 = kit_c0<KORD>(it);
    u.dh = z.bh - rp; u.dw = z.bw - qp;
    u.dA = (u.dh * g.Ws + u.dw) * g.Cs + c0;
    u.dB = ((z.r0h + g.stride * rp) * 5 + (z.r0w + g.stride * qp)) * g.Cs + c0;
    return u;
  }
  __device__ __forceinline__ bool a_ok(const ARowP& r, const StageU& u) const {
    return (unsigned)(r.h0 + u.dh) < (unsigned)g.Hs && (unsigned)(r.w0 + u.dw) < (unsigned)g.Ws;
  }
};

// MINW: waves per SIMD the register allocation must allow (1 or 2)
// BUF: DMA through buffer descriptors (32-bit per-lane offset, scalar plane offset; a padding tap is an out-of-range
// offset, which the hardware answers with zeros) instead of 64-bit per-lane pointers and the global zero page.
// M16: contract with v_mfma_f32_16x16x32_bf16 (one MFMA covers the whole 32-deep stage) instead of two k-steps of
// v_mfma_f32_32x32x16_bf16: same FLOPs per cycle, but the chip holds a higher clock on this shape when it is
// power-limited (MI355X_MICROARCH.md, DVFS give-back item 7).  Sums run over k in groups of 32 instead of 16, so
// results agree with the 32x32x16 form to rounding, not bit for bit.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <class P, int BM, int BN, int WM, int WN, int STAGES, int KORD, int MINW, bool BUF, bool M16>
__global__ void __launch_bounds__(WM * WN * 64, MINW) igemm16p_kernel(const P p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub; the buffer builtins do not exist there)
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile");
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;           // bytes per plane and stage
  constexpr int SB = 2 * A_PLANE + 2 * B_PLANE;                 // bytes per stage
  constexpr int NIA = 2 * (BM / 16) / NW, NIB = 2 * (BN / 16) / NW;   // DMA wave-instructions per wave and stage
  static_assert(NIA * NW == 2 * (BM / 16) && NIB * NW == 2 * (BN / 16) && NIA >= 1 && NIB >= 1, "DMA map");
  static_assert((NIA % 2 == 0 || NIA == 1) && (NIB % 2 == 0 || NIB == 1), "a wave owns whole (hi, lo) pairs, or one plane of one row block");
  constexpr bool EA = NIA % 2 == 0, EB = NIB % 2 == 0;
  constexpr int NRA = (NIA + 1) / 2, NRB = (NIB + 1) / 2;       // distinct rows per lane
  constexpr int NDMA = NIA + NIB;
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  static_assert(NDMA * (STAGES - 1) < 64, "vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[STAGES * SB];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* g_ptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  unsigned long long t_in = 0, r_in = 0;
  if (p.dbg) { t_in = __builtin_amdgcn_s_memtime(); r_in = __builtin_amdgcn_s_memrealtime(); }
  int tx = blockIdx.x, ty = blockIdx.y;
  if (p.xcd_map) {   // column tiles of one row tile get hardware ids equal modulo 8 (same XCD, same L2)
    const int hid = blockIdx.x + gridDim.x * blockIdx.y;
    const int r = hid & 7, q = hid >> 3;
    tx = r + 8 * (q / (int)gridDim.y);
    ty = q % (int)gridDim.y;
  }
  const int m0 = tx * BM, n0 = ty * BN;

  typename P::ZCtx z;
  p.z_setup(blockIdx.z, z);
  const int ntap = p.ntap(z);

  // DMA map.  Instruction idx = wave*NI + i -> row block rb = idx >> 1 (16 rows), plane = idx & 1;
  // lane -> row rb*16 + lane/4, physical chunk lane & 3 = logical chunk ^ ((row >> 2) & 3).
  const int a_idx0 = wave * NIA, b_idx0 = wave * NIB;
  // chunk swizzle g(row >> 2): identity serves the 32x32x16 fragment reads; the 16x16x32 reads (lane = row + 16*chunk)
  // need g = (-x) & 3 for every 16-lane ds_read_b128 group to cover all 64 banks
  auto gsw = [](int x) { return M16 ? ((-x) & 3) : (x & 3); };
  const int swz = ((lane & 3) ^ gsw(lane >> 4)) * 8;             // element offset of this lane's logical chunk
  typename P::ARowP ra[NRA];
  int rbo[NRB];
#pragma unroll
  for (int j = 0; j < NRA; ++j) ra[j] = p.a_rowp(m0 + ((a_idx0 >> 1) + j) * 16 + (lane >> 2), z);
#pragma unroll
  for (int j = 0; j < NRB; ++j) rbo[j] = p.b_rowp(n0 + ((b_idx0 >> 1) + j) * 16 + (lane >> 2), z);
  const u16* const zsrc = reinterpret_cast<const u16*>(p.zero);
  const u16* const a_hi = p.a_ptr();
  const u16* const a_lo = p.a_ptr() + p.a_plane();
  const u16* const b_hi = p.b_ptr();
  const u16* const b_lo = p.b_ptr() + p.b_plane();

  const unsigned a_plane_bytes = (unsigned)(p.a_plane() * 2), b_plane_bytes = (unsigned)(p.b_plane() * 2);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_hi, 0, BUF ? (int)(2 * a_plane_bytes) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_hi, 0, BUF ? (int)(2 * b_plane_bytes) : 0, 0x00020000);

  auto dma_stage = [&](int buf, const KIt& it) {
    unsigned char* base = lds + buf * SB;
    const StageU u = p.template stage<KORD>(it, z);
    if constexpr (BUF) {
      constexpr unsigned OOB = 0x80000000u;
      unsigned va[NRA], vb[NRB];
#pragma unroll
      for (int j = 0; j < NRA; ++j) va[j] = p.a_ok(ra[j], u) ? (unsigned)(2 * (ra[j].off0 + u.dA + swz)) : OOB;
#pragma unroll
      for (int j = 0; j < NRB; ++j) vb[j] = rbo[j] >= 0 ? (unsigned)(2 * (rbo[j] + u.dB + swz)) : OOB;
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int j = EA ? (i >> 1) : 0;
        const int pl = EA ? (i & 1) : (a_idx0 & 1), rb = (a_idx0 >> 1) + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(base + pl * A_PLANE + rb * 1024), 16, va[j], pl ? a_plane_bytes : 0u, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NIB; ++i) {
        const int j = EB ? (i >> 1) : 0;
        const int pl = EB ? (i & 1) : (b_idx0 & 1), rb = (b_idx0 >> 1) + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(base + 2 * A_PLANE + pl * B_PLANE + rb * 1024), 16, vb[j], pl ? b_plane_bytes : 0u, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int j = EA ? (i >> 1) : 0;                                   // compile-time row slot
      const int pl = EA ? (i & 1) : (a_idx0 & 1), rb = (a_idx0 >> 1) + j;   // plane, 16-row block (wave-uniform)
      const bool ok = p.a_ok(ra[j], u);
      const int e = ra[j].off0 + u.dA + swz;
      const u16* src = ok ? (pl ? a_lo : a_hi) + e : zsrc;
      __builtin_amdgcn_global_load_lds((g_ptr)src, (lds_ptr)(base + pl * A_PLANE + rb * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int j = EB ? (i >> 1) : 0;
      const int pl = EB ? (i & 1) : (b_idx0 & 1), rb = (b_idx0 >> 1) + j;
      const bool ok = rbo[j] >= 0;
      const int e = rbo[j] + u.dB + swz;
      const u16* src = ok ? (pl ? b_lo : b_hi) + e : zsrc;
      __builtin_amdgcn_global_load_lds((g_ptr)src, (lds_ptr)(base + 2 * A_PLANE + pl * B_PLANE + rb * 1024), 16, 0, 0);
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int klen = z.k_end - z.k_begin;
  const int nk = klen > 0 ? klen / 32 : 0;                       // FAST shapes: k ranges are multiples of 32
  const int arow0 = wm * (BM / WM) + li;
  const int brow0 = wn * (BN / WN) + li;
  const int sw = gsw(li >> 2);

  if constexpr (!M16) {
  struct Frags { bf16x8_t ah[TM], al[TM], bh[TN], bl[TN]; };
  auto read_frags = [&](Frags& f, int buf, int s) {
    const unsigned char* A0 = lds + buf * SB;
    const unsigned char* B0 = A0 + 2 * A_PLANE;
    const int coff = ((2 * s + lh) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned char* q = A0 + (arow0 + 32 * i) * 64 + coff;
      f.ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
      f.al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + A_PLANE));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned char* q = B0 + (brow0 + 32 * j) * 64 + coff;
      f.bh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
      f.bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + B_PLANE));
    }
  };
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
  };
  // wait until this wave's DMA pieces of the oldest outstanding stage have landed; `after` = stages issued after it
  auto wait_stage = [&](int after) {
    if (after >= STAGES - 1) wait_vmcnt<(STAGES - 1) * NDMA>();
    else if (STAGES > 2 && after == 2) wait_vmcnt<2 * NDMA>();
    else if (after == 1) wait_vmcnt<NDMA>();
    else wait_vmcnt<0>();
  };

  if (nk > 0) {
    KIt it;
    kit_init<KORD>(it, z.k_begin, ntap);
    int issued = 0;
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
      if (s < nk) { dma_stage(s, it); kit_next<KORD>(it, ntap); ++issued; }
    wait_stage(issued - 1);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frags f0, f1;
    read_frags(f0, 0, 0);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
      read_frags(f1, buf, 1);
      mfma_step(f0);
      const int nbuf = buf + 1 == STAGES ? 0 : buf + 1;
      if (t + 1 < nk) {
        wait_stage(issued - (t + 2));                   // stage t+1 landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stage t have returned
        __builtin_amdgcn_s_barrier();                   // every wave's have: stage t+1 visible, buffer `buf` free
        asm volatile("" ::: "memory");
        if (issued < nk) { dma_stage(buf, it); kit_next<KORD>(it, ntap); ++issued; }
        read_frags(f0, nbuf, 0);
      }
      mfma_step(f1);
      buf = nbuf;
    }
  }

  if (p.stat) {      // BatchNorm statistics of the tile (igemm16.h epilogue_stats32); the ring is idle: every DMA has landed
    __syncthreads();
    epilogue_stats32<BM, BN, WM, WN, TM, TN>(p.stat, (int)(gridDim.x * gridDim.z), (int)(blockIdx.z * gridDim.x) + tx, p.M, p.N, acc, lds, m0, n0,
                                             wm, wn, li, lh, tid);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + wm * (BM / WM) + 32 * i + row;
        const int n = n0 + wn * (BN / WN) + 32 * j + li;
        p.store(m, n, acc[i][j][r], z);
      }
  } else {
    // ---- 16x16x32 form: one MFMA k-step per stage; the stage is split into two halves of the wave's row tiles ----
    constexpr int TM16 = BM / WM / 16, TN16 = BN / WN / 16, TMH = TM16 / 2;
    static_assert(TM16 % 2 == 0, "two row-tile halves");
    f32x4_t acc16[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
    const int l16 = lane & 15, lc = lane >> 4;
    const int coff16 = (lc ^ gsw(l16 >> 2)) * 16;
    const int arow16 = wm * (BM / WM) + l16, brow16 = wn * (BN / WN) + l16;
    struct FA { bf16x8_t h[TMH], l[TMH]; };
    struct FB { bf16x8_t h[TN16], l[TN16]; };
    auto read_a = [&](FA& f, int buf, int half) {
      const unsigned char* A0 = lds + buf * SB;
#pragma unroll
      for (int i = 0; i < TMH; ++i) {
        const unsigned char* q = A0 + (arow16 + 16 * (half * TMH + i)) * 64 + coff16;
        f.h[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
        f.l[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + A_PLANE));
      }
    };
    auto read_b = [&](FB& f, int buf) {
      const unsigned char* B0 = lds + buf * SB + 2 * A_PLANE;
#pragma unroll
      for (int j = 0; j < TN16; ++j) {
        const unsigned char* q = B0 + (brow16 + 16 * j) * 64 + coff16;
        f.h[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
        f.l[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + B_PLANE));
      }
    };
    auto mfma_half = [&](const FA& a, const FB& b, auto half_c) {
      constexpr int HALF = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < TMH; ++i)
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
          f32x4_t c = acc16[HALF * TMH + i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l[i], b.h[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h[i], b.l[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h[i], b.h[j], c, 0, 0, 0);
          acc16[HALF * TMH + i][j] = c;
        }
    };
    auto wait_stage = [&](int after) {
      if (after >= STAGES - 1) wait_vmcnt<(STAGES - 1) * NDMA>();
      else if (STAGES > 2 && after == 2) wait_vmcnt<2 * NDMA>();
      else if (after == 1) wait_vmcnt<NDMA>();
      else wait_vmcnt<0>();
    };
    if (nk > 0) {
      KIt it;
      kit_init<KORD>(it, z.k_begin, ntap);
      int issued = 0;
#pragma unroll
      for (int s = 0; s < STAGES; ++s)
        if (s < nk) { dma_stage(s, it); kit_next<KORD>(it, ntap); ++issued; }
      wait_stage(issued - 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      FA a0, a1;
      FB b0, b1;
      read_b(b0, 0);
      read_a(a0, 0, 0);
      int buf = 0, t = 0;
      // one stage: a0 = first half of stage t's row tiles and bc = its column tiles are loaded on entry
      auto stage_body = [&](FB& bc, FB& bn) {
        read_a(a1, buf, 1);
        mfma_half(a0, bc, std::integral_constant<int, 0>{});
        const int nbuf = buf + 1 == STAGES ? 0 : buf + 1;
        if (t + 1 < nk) {
          wait_stage(issued - (t + 2));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (issued < nk) { dma_stage(buf, it); kit_next<KORD>(it, ntap); ++issued; }
          read_b(bn, nbuf);
          read_a(a0, nbuf, 0);
        }
        mfma_half(a1, bc, std::integral_constant<int, 1>{});
        buf = nbuf;
        ++t;
      };
      while (t + 1 < nk) { stage_body(b0, b1); stage_body(b1, b0); }
      if (t < nk) stage_body(b0, b1);
    }
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * (BM / WM) + 16 * i + lc * 4 + r;
          const int n = n0 + wn * (BN / WN) + 16 * j + l16;
          p.store(m, n, acc16[i][j][r], z);
        }
  }
  if (p.dbg && tid == 0) {
    unsigned long long* d = p.dbg + 4 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    d[0] = t_in; d[1] = r_in; d[2] = __builtin_amdgcn_s_memtime(); d[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// tile configurations of the pipelined kernel
enum PCfg : int {
  PCFG_NONE = 0,
  PCFG_128x128_S3,     // 4 waves 2x2, 96 KB: one workgroup per CU
  PCFG_128x128_S2,     // 4 waves 2x2, 64 KB: two workgroups per CU
  PCFG_256x128_S3,     // 8 waves 4x2, 144 KB
  PCFG_256x64_S3,      // 4 waves 4x1, 120 KB
  PCFG_256x64_S2,      // 4 waves 4x1, 80 KB: two per CU
  PCFG_256x256_S2,     // 8 waves 2x4 (wave tile 128x64), 128 KB
  PCFG_128x64_S3,      // 4 waves 2x2 (wave tile 64x32), 72 KB: two per CU
  PCFG_COUNT
};

inline void pcfg_tile(int cfg, int& bm, int& bn) {
  switch (cfg) {
    case PCFG_128x128_S3: case PCFG_128x128_S2: bm = 128; bn = 128; break;
    case PCFG_256x128_S3: bm = 256; bn = 128; break;
    case PCFG_256x64_S3: case PCFG_256x64_S2: bm = 256; bn = 64; break;
    case PCFG_256x256_S2: bm = 256; bn = 256; break;
    case PCFG_128x64_S3: bm = 128; bn = 64; break;
    default: bm = bn = 0;
  }
}

template <class P, int KORD, bool BUF, bool M16>
inline void launch_igemm16p_k(const P& p, int cfg, long M, long N, int gz, hipStream_t stream) {
  int bm, bn;
  pcfg_tile(cfg, bm, bn);
  const dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)gz);
  // The library (VP_PCFG_LIBRARY, set by conv16.hip) instantiates only what plan16 dispatches: buffer DMA, the 256x256 and 128x64
  // tiles on the 32x32x16 MFMA and the 256x128 tile on the 16x16x32 form; tools/kbench compiles every configuration.
#if defined(VP_PCFG_LIBRARY)
#define VP_PCFG_ON(plain, q) (BUF && (M16 ? (q) : (plain)))
#else
#define VP_PCFG_ON(plain, q) true
#endif
  switch (cfg) {
    case PCFG_128x128_S3: if constexpr (VP_PCFG_ON(false, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 128, 128, 2, 2, 3, KORD, 1, BUF, M16>), grid, dim3(256), 0, stream, p); break;
    case PCFG_128x128_S2: if constexpr (VP_PCFG_ON(false, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 128, 128, 2, 2, 2, KORD, 2, BUF, M16>), grid, dim3(256), 0, stream, p); break;
    case PCFG_256x128_S3: if constexpr (VP_PCFG_ON(false, true)) hipLaunchKernelGGL((igemm16p_kernel<P, 256, 128, 4, 2, 3, KORD, 2, BUF, M16>), grid, dim3(512), 0, stream, p); break;
    case PCFG_256x64_S3: if constexpr (VP_PCFG_ON(false, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 256, 64, 4, 1, 3, KORD, 1, BUF, M16>), grid, dim3(256), 0, stream, p); break;
    case PCFG_256x64_S2: if constexpr (VP_PCFG_ON(false, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 256, 64, 4, 1, 2, KORD, 2, BUF, M16>), grid, dim3(256), 0, stream, p); break;
    case PCFG_256x256_S2: if constexpr (VP_PCFG_ON(true, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 256, 256, 2, 4, 2, KORD, 2, BUF, M16>), grid, dim3(512), 0, stream, p); break;
    case PCFG_128x64_S3: if constexpr (VP_PCFG_ON(true, false)) hipLaunchKernelGGL((igemm16p_kernel<P, 128, 64, 2, 2, 3, KORD, 2, BUF, M16>), grid, dim3(256), 0, stream, p); break;
    default: break;
  }
#undef VP_PCFG_ON
}

// ctile = channel count that k is decomposed by; KORD follows igemm16's dispatch (64-deep order when ctile % 64 == 0)
template <class P>
inline void launch_igemm16p(const P& p, int cfg, long M, long N, int gz, hipStream_t stream, int ctile, bool buf = true, bool m16 = false) {
  if (m16) {       // 16x16x32 form: buffer DMA only
    if (ctile % 64 == 0) launch_igemm16p_k<P, 6, true, true>(p, cfg, M, N, gz, stream);
    else launch_igemm16p_k<P, 5, true, true>(p, cfg, M, N, gz, stream);
  } else if (buf) {
    if (ctile % 64 == 0) launch_igemm16p_k<P, 6, true, false>(p, cfg, M, N, gz, stream);
    else launch_igemm16p_k<P, 5, true, false>(p, cfg, M, N, gz, stream);
  } else {
    if (ctile % 64 == 0) launch_igemm16p_k<P, 6, false, false>(p, cfg, M, N, gz, stream);
    else launch_igemm16p_k<P, 5, false, false>(p, cfg, M, N, gz, stream);
  }
}

#endif  // __HIPCC__

}  // namespace vp
