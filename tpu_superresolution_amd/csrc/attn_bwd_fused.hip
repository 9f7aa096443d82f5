// Window attention backward for the classical SwinIR width (C = 180 -> 192, 6 heads x 32) with the q/k/v projection
// RE-COMPUTED from the saved LayerNorm output and the output-projection dgrad folded in:
//
//   q, k, v = xn1_window . Wqkv^T + b  (q scaled)          network_swinir.py:121-124  (k, v bit for bit what the forward had; q as
//                                                           fma(x W, scale, b scale): one fp32 rounding instead of two)
//   dO      = g_window . Wproj                              gradient of :143 (g = d x1 in window order, bf16)
//   P       = softmax(q k^T + bias + mask)                  :125-139
//   dV = P^T dO,  dS = P (dP - rowsum(P dP)),  dQ = dS K scale,  dK = dS^T Q,  d(bias) += dS
//
// Before, the forward wrote q/k/v (150 MB per block at cfg3) for this pass to read back, a separate GEMM turned g into
// d(attn_out) (50 MB written, 50 MB read) and the attention gradient ran as 12 288 (window, head) work items of three
// barriers each on padded LDS tiles (50 % of its LDS cycles were bank conflicts).  Here a window's gradient needs the 24 KB
// of xn1 rows and the 24 KB of g rows, and d(qkv) is the only thing written.
//
// One persistent 768-thread workgroup per CU owns a head TRIPLE (96 of the 192 attention channels) of the windows it walks;
// the two triples of a window sit on the same XCD (block ids b, b + 8) and share the rows through its L2.  The weights a
// triple needs -- 18 16-column fragments of Wqkv and 6 of Wproj^T over K = 192 (144 KB) -- live in REGISTERS, two fragments
// (48 VGPRs) per wave, for the whole kernel.  Wave w = 4 hl + r of a workgroup:
//   projection  r = 0 / 1 / 2: the q / k / v tile [64][32] of head hl from the xn1 rows; r = 3: the dO tile of head hl from the
//               g rows (48 MFMAs each, the row operand read from LDS as the B fragment)
//   phase B     query tile r of head hl: S^T and dP^T (8 MFMAs), bias from a [3][225] table in LDS, arithmetic shift mask,
//               softmax, dS; d(bias) accumulated in registers (dense [i][j], 16 VGPRs); dQ^T = K^T dS^T with dS^T taken
//               straight from the accumulators as the B operand (4 MFMAs); P and dS go to LDS as bf16 [i][j]
//   phase C     key tile r of head hl: dV^T, dK^T (8 MFMAs, transposing reads of P, dS, dO, Q)
// Three barriers per window.  Two 48 KB row slots alternate between the windows: a window's P / dS tiles overwrite ITS OWN xn1 / g
// rows once the projection has consumed them, so the other slot is free from the window's first barrier on and the rows of the
// NEXT window are fetched into it by LDS-DMA (four 1-KiB pieces per wave) under phases B and C (SRK_ABF_ISSUE_AT below).  A wave issues
// exactly three store instructions (dq, dk, dv) between its DMAs and the next window's first barrier, so `s_waitcnt vmcnt(3)`
// there waits for the DMAs and not for the stores.
//
// LDS images (148 KB): rows [64][192] bf16 with the 16-byte chunk XOR-swizzled by (row & 7) on the DMA's source address
// (conflict-free ds_read_b128 B fragments, as gemm_stream.hip); head tiles [64][32] bf16 with chunk ^ tf(row) and P / dS tiles
// [64][64] bf16 with chunk ^ ps(row), chosen so that the row-fragment ds_read_b128 (lane groups {0-3, 12-15, 20-27} ...),
// both k orders of ds_read_b64_tr_b16 (32-lane groups) and the MFMA-layout ds_write_b64 hit every bank once (the 8-byte
// writes twice: 16 rows x 8 bytes of one column group cannot do better on 64-byte rows).
#include "kernels.h"

namespace {

constexpr int H_K = 192, H_CA = 192, H_NH = 6;
constexpr int H_SLOT = 64 * H_K * 2;             // 24 576 B: 64 rows of xn1 or g
constexpr int H_TILE = 64 * 32;                  // elements of a head tile
constexpr int H_PT = 64 * 64;                    // elements of a P / dS tile
constexpr int H_TAB = 225;
constexpr int H_PAIR = 2 * H_SLOT;               // one row slot: xn1 rows | g rows; later [hl][P, dS][64][64] of the same window
constexpr int OFF_S = 0;                         // two row slots
constexpr int OFF_T = 2 * H_PAIR;                // [hl][q, k, v, dO][64][32]
constexpr int OFF_TAB = OFF_T + 12 * H_TILE * 2; // [hl][225] fp32
constexpr int OFF_PB = OFF_TAB + 2704;           // [hl][q, k, v][32] fp32 projection bias
constexpr int H_LDS = OFF_PB + 10 * 32 * 4;      // 151 440 B
static_assert(6 * H_PT * 2 == H_PAIR, "P / dS tiles of a head triple fill a row slot exactly");

#ifndef SRK_ABF_ISSUE_AT
// Where the next window's row DMAs are issued: behind barrier A (0), behind barrier B (1), behind barrier C (2) or inside phase B
// behind its MFMAs (3).  Measured in the cfg3 step: 95.2 / 91.0 / 112.0 / 91.3 us per launch -- issued into the projection, the four
// DMA instructions of every wave compete with its ds_read_b128 stream; behind C the rows land too late.
#define SRK_ABF_ISSUE_AT 1
#endif
#ifdef SRK_PROBE_ABF
// developer instrumentation (never in the shipped build): s_memrealtime (100 MHz) at the phase boundaries of workgroup 0
__device__ unsigned long long g_abf_probe[12 * 16 * 8];
#define ABF_MARK(k) do { if (blockIdx.x == 0 && lane == 0 && t < 16) g_abf_probe[(wave * 16 + t) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int srk_debug_abf_probe(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_abf_probe), sizeof(g_abf_probe)); }
#else
#define ABF_MARK(k) do {} while (0)
#endif

struct BwdFusedParams {
  const bf16_t* xn;      // [B_*64][lda] LayerNorm output, window order
  int lda;
  const bf16_t* g;       // [B_*64][ldg] gradient of the projection output, window order
  int ldg;
  const bf16_t* Wqkv;    // [576][192] packed (q rows first, head-padded)
  const float* bqkv;     // [576] or null
  float scale;
  const bf16_t* WprojT;  // [192 attention channel][192 output channel]: dO = g . WprojT^T
  const float* biasd;    // [6][64][64] dense relative-position bias
  bf16_t* dqkv;          // [B_*64][576] columns (which, head, d)
  float* slab;           // [nslab][6][64][64] partial d(bias)
  long long B_;
  WinGeom geom;
};

// chunk swizzle of a [64][32] head tile / a [64][64] P tile: functions of row bits 1..3 only
__device__ __forceinline__ int tf(int r) { return ((((r >> 2) ^ (r >> 3)) & 1) << 1) | (((r >> 3) ^ (r >> 1)) & 1); }
__device__ __forceinline__ int ps(int r) { return (((r >> 3) & 1) << 2) | (((r >> 1) & 1) << 1) | ((r >> 2) & 1); }

__device__ __forceinline__ bf16x8_t h_cat4(bf16x4_t lo, bf16x4_t hi) {
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// transposed fragment of a swizzled [64][32] head tile: this lane's column 16 dt + r16 of the eight rows rb0 .. rb0 + 3 and
// rb1 .. rb1 + 3 (the lane supplies the address of row rb + (r16 >> 2), columns 16 dt + 4 (r16 & 3) ..)
__device__ __forceinline__ bf16x8_t tile_tr(const bf16_t* tile, int rb0, int rb1, int dt, int r16) {
  const int q = r16 >> 2, c = 2 * dt + ((r16 & 3) >> 1), in = (r16 & 1) << 2;
  const int ra = rb0 + q, rb = rb1 + q;
  return h_cat4(lds_tr_read(tile + ra * 32 + ((c ^ tf(ra)) << 3) + in), lds_tr_read(tile + rb * 32 + ((c ^ tf(rb)) << 3) + in));
}
// the same for a [64][64] P / dS tile, column 16 jt + r16
__device__ __forceinline__ bf16x8_t ptile_tr(const bf16_t* tile, int rb0, int rb1, int jt, int r16) {
  const int q = r16 >> 2, c = 2 * jt + ((r16 & 3) >> 1), in = (r16 & 1) << 2;
  const int ra = rb0 + q, rb = rb1 + q;
  return h_cat4(lds_tr_read(tile + ra * 64 + ((c ^ ps(ra)) << 3) + in), lds_tr_read(tile + rb * 64 + ((c ^ ps(rb)) << 3) + in));
}

__global__ __launch_bounds__(768) void qkv_attn_bwd_kernel(const BwdFusedParams p, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned smem_base = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int tr = slot & 1, grp = slot >> 1;
  const int gidx = grp * 8 + xcd;                       // window list and d(bias) slab of this workgroup pair
  const long long first = gidx, stride = 8LL * ngrp;
  const long long nwin = first < p.B_ ? (p.B_ - first + stride - 1) / stride : 0;
  const int hl = wave >> 2, it = wave & 3;
  const int head = 3 * tr + hl;

  f32x4_t dbias[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) dbias[jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (nwin > 0) {
    bf16_t* tiles = reinterpret_cast<bf16_t*>(smem + OFF_T);
    float* tab = reinterpret_cast<float*>(smem + OFF_TAB);
    // rel-pos bias table [3][225] out of the dense [6][64][64]: offset (dy, dx) is realised by the pair i = (max(dy, 0), max(dx, 0)),
    // j = (max(-dy, 0), max(-dx, 0))   (network_swinir.py:89-103)
    for (int i = tid; i < 3 * H_TAB; i += 768) {
      const int hh = i / H_TAB, t = i - hh * H_TAB;
      const int dy = t / 15 - 7, dx = t - (t / 15) * 15 - 7;
      const int qi = (dy > 0 ? dy : 0) * 8 + (dx > 0 ? dx : 0), kj = (dy < 0 ? -dy : 0) * 8 + (dx < 0 ? -dx : 0);
      tab[i] = p.biasd[(3 * tr + hh) * 4096 + qi * 64 + kj];
    }

    // ---- this wave's two weight fragments (output columns 16 j .. 16 j + 15 of its tile) over K = 192 --------------------------
    bf16x8_t wf[2][6];
    {
      const bf16_t* wbase = it < 3 ? p.Wqkv + (long long)(it * H_CA + head * 32) * H_K : p.WprojT + (long long)(head * 32) * H_K;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 6; ++s) wf[j][s] = *reinterpret_cast<const bf16x8_t*>(wbase + (16 * j + r16) * H_K + s * 32 + g * 8);
    }
    // projection bias of the triple's q / k / v tiles -> LDS (the registers are spoken for); a dO wave adds nothing
    float* pbs = reinterpret_cast<float*>(smem + OFF_PB);
    for (int i = tid; i < 9 * 32; i += 768) {
      const int tl = i >> 5, hh = tl / 3, which = tl - 3 * hh;
      pbs[i] = p.bqkv ? p.bqkv[which * H_CA + (3 * tr + hh) * 32 + (i & 31)] : 0.f;
    }
    for (int i = tid; i < 32; i += 768) pbs[9 * 32 + i] = 0.f;        // the row a dO wave adds
    const float* pbw = pbs + (it < 3 ? hl * 3 + it : 9) * 32 + 4 * g;
    const float sc = it == 0 ? p.scale : 1.0f;

    // ---- DMA: pieces 4 wave .. 4 wave + 3 of the 48 1-KiB pieces (24 of xn1, 24 of g) ------------------------------------------
    const bool dma_g = wave >= 6;
    const bf16_t* dsrc = dma_g ? p.g : p.xn;
    const long long dld = dma_g ? p.ldg : p.lda;
    const int piece0 = dma_g ? 4 * (wave - 6) : 4 * wave;
    const unsigned ddst = smem_base + OFF_S + (dma_g ? H_SLOT : 0) + piece0 * 1024;
    // piece i of this wave covers the 16-byte chunks q = (piece0 + i) 64 + lane of the image: row q / 24, chunk q % 24.  The four
    // offsets are recomputed per window (a dozen VALU operations) rather than held: the kernel sits at its register limit
    auto issue = [&](long long b_, int slot) {
      const bf16_t* base = dsrc + b_ * 64 * dld;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q = (piece0 + i) * 64 + lane;
        asm volatile("" : "+v"(q));
        const int row = (q * 2731) >> 16, pos = q - row * 24;      // q / 24 for q < 1536
        srk_glds16<false>(base + row * (int)dld + ((pos ^ (row & 7)) << 3), __builtin_amdgcn_readfirstlane(ddst + slot * H_PAIR + i * 1024));
      }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int s = 0; s < 6; ++s) asm volatile("" ::"v"(wf[j][s]));      // retire the weight loads before the DMA ring starts
    }
    issue(first, 0);

    const int tfr = tf(r16), psr = ps(r16);
    bf16_t* mytile = tiles + (hl * 4 + it) * H_TILE;
    const bf16_t* Qs = tiles + (hl * 4 + 0) * H_TILE;
    const bf16_t* Ks = tiles + (hl * 4 + 1) * H_TILE;
    const bf16_t* Vs = tiles + (hl * 4 + 2) * H_TILE;
    const bf16_t* Os = tiles + (hl * 4 + 3) * H_TILE;
    // this lane's query i = 16 it + r16; key j = 16 jt + 4 g + e -> table index lane_idx - 30 jt - e (as attn_fused.hip)
    const float* th = tab + hl * H_TAB + ((2 * it + (r16 >> 3)) - (g >> 1) + 7) * 15 + ((r16 & 7) - 4 * (g & 1) + 7) - 93;
    const int ldq = 3 * H_CA;
    const int scol = ((g & 1) << 4) | ((g >> 1) << 3);            // column of the 16-byte store after the 16-lane-row swap
    auto store8 = [&](bf16_t* dst, uint2 x, uint2 y) {
      const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
      *reinterpret_cast<uint4*>(dst) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    };

    for (long long t = 0; t < nwin; ++t) {
      const long long b_ = first + t * stride;
      if (t == 0) srk_wait_vmcnt<0>();                    // table / weight loads and the first DMAs
      else srk_wait_vmcnt<SRK_ABF_ISSUE_AT == 2 ? 2 : 3>();   // everything but this wave's dq / dk / dv stores of window t-1 behind the DMAs
      ABF_MARK(0);
      srk_lds_barrier();                                  // A: the rows of window t are in LDS; nobody reads window t-1's tiles
      ABF_MARK(1);
      const int cur = (int)(t & 1);
      if (SRK_ABF_ISSUE_AT == 0 && t + 1 < nwin) issue(b_ + stride, cur ^ 1);      // the other slot held P / dS of window t-1: dead since this barrier
      const unsigned char* src = smem + OFF_S + cur * H_PAIR + (it < 3 ? 0 : H_SLOT);
      bf16_t* Pb = reinterpret_cast<bf16_t*>(smem + OFF_S + cur * H_PAIR) + (hl * 2 + 0) * H_PT;
      bf16_t* Db = Pb + H_PT;
      // ---- projection: this wave's [64][32] tile in four 16-row quarters -----------------------------------------------------
      // (row & 7 == r16 & 7 in every quarter: the swizzled chunk offsets are lane constants, the quarter is an immediate offset)
      const unsigned char* srow = src + r16 * (H_K * 2);
      bf16_t* trow = mytile + r16 * 32 + 4 * (g & 1);
      // The row fragments are requested half a quarter (six MFMAs) ahead of their use, in two register groups of three, fenced so
      // that the scheduler keeps the order: left to itself the compiler issues each ds_read one or two MFMAs ahead of its use and
      // every wave then sits through the LDS latency five times per quarter (a whole quarter ahead does not fit the registers).
      bf16x8_t xq[2][3];
      auto xload = [&](int mq, int half) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
          xq[half][s] = *reinterpret_cast<const bf16x8_t*>(srow + mq * (16 * H_K * 2) + ((((3 * half + s) * 4 + g) ^ (r16 & 7)) << 4));
      };
      xload(0, 0);
      xload(0, 1);
#pragma unroll
      for (int mq = 0; mq < 4; ++mq) {
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][3 * half + s], xq[half][s], acc[j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (mq < 3) xload(mq + 1, half);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {    // columns d = 16 j + 4 g .. + 3: chunk 2 j + (g >> 1), half g & 1;  (x W + b) * scale, rounded as the forward kernel rounds it
          const float4 bq = *reinterpret_cast<const float4*>(pbw + 16 * j);
          *reinterpret_cast<uint2*>(trow + mq * (16 * 32) + (((2 * j + (g >> 1)) ^ tfr) << 3)) =
              pack_bf4((acc[j][0] + bq.x) * sc, (acc[j][1] + bq.y) * sc, (acc[j][2] + bq.z) * sc, (acc[j][3] + bq.w) * sc);
        }
      }
      ABF_MARK(2);
      srk_lds_barrier();                                  // B: the twelve tiles are complete; this window's rows are consumed
      ABF_MARK(3);
      if (SRK_ABF_ISSUE_AT == 1 && t + 1 < nwin) issue(b_ + stride, cur ^ 1);

      // ---- phase B: query tile `it` of head hl against all 64 keys ---------------------------------------------------------------
      {
        f32x4_t s[4], dp[4];
        {
          // all ten fragments first, then the eight MFMAs (fenced: one LDS latency per unit instead of one per MFMA pair)
          const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(Qs + (16 * it + r16) * 32 + ((g ^ tfr) << 3));
          const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(Os + (16 * it + r16) * 32 + ((g ^ tfr) << 3));
          bf16x8_t kf[4], vf[2];
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) kf[jt] = *reinterpret_cast<const bf16x8_t*>(Ks + (16 * jt + r16) * 32 + ((g ^ tfr) << 3));
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) vf[jt] = *reinterpret_cast<const bf16x8_t*>(Vs + (16 * jt + r16) * 32 + ((g ^ tfr) << 3));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) kf[jt] = *reinterpret_cast<const bf16x8_t*>(Vs + (16 * (jt + 2) + r16) * 32 + ((g ^ tfr) << 3));   // v tiles 2, 3
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) dp[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[jt], of, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) dp[jt + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt], of, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        if (SRK_ABF_ISSUE_AT == 3 && t + 1 < nwin) issue(b_ + stride, cur ^ 1);   // behind the unit's MFMAs, in front of the softmax VALU work
        const int w = (int)((unsigned)b_ % (unsigned)p.geom.nW);      // 32-bit: the launcher bounds B_
        const int wy = w / p.geom.nWw, wx = w - wy * p.geom.nWw;
        const bool masked = p.geom.shift > 0 && (wy == p.geom.H / 8 - 1 || wx == p.geom.nWw - 1);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) s[jt] += f32x4_t{th[93 - 30 * jt], th[92 - 30 * jt], th[91 - 30 * jt], th[90 - 30 * jt]};
        if (masked) {
          const int labi = win_region_label(p.geom, w, 16 * it + r16);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (win_region_label(p.geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;   // :235 (-100, not -inf)
        }
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int jt = 1; jt < 4; ++jt) mx = fmaxf(fmaxf(mx, s[jt][0]), fmaxf(fmaxf(s[jt][1], s[jt][2]), s[jt][3]));
        mx = xrow_max4(mx);
        constexpr float L2E = 1.4426950408889634f;        // exp(x - m) = exp2(x log2e - m log2e), as softmax_numerators
        const float mxl = mx * L2E;
        f32x4_t a4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const f32x4_t tt = s[jt] * L2E - mxl;
#pragma unroll
          for (int e = 0; e < 4; ++e) s[jt][e] = __builtin_amdgcn_exp2f(tt[e]);
          a4 += s[jt];
        }
        const float inv = __builtin_amdgcn_rcpf(xrow_sum4((a4[0] + a4[1]) + (a4[2] + a4[3])));
        float dl = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s[jt][e] *= inv;
            dl += s[jt][e] * dp[jt][e];
          }
        dl = xrow_sum4(dl);
        uint2 dsp[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dp[jt][e] = s[jt][e] * (dp[jt][e] - dl);       // dS
            dbias[jt][e] += dp[jt][e];
          }
          dsp[jt] = pack_bf4(dp[jt][0], dp[jt][1], dp[jt][2], dp[jt][3]);
          const int po = (16 * it + r16) * 64 + (((2 * jt + (g >> 1)) ^ psr) << 3) + 4 * (g & 1);
          *reinterpret_cast<uint2*>(Pb + po) = pack_bf4(s[jt][0], s[jt][1], s[jt][2], s[jt][3]);
          *reinterpret_cast<uint2*>(Db + po) = dsp[jt];
        }
        // dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]: dS^T of this query tile straight from the registers (accumulator k order:
        // slot (g, jj) of 32-key step ss is key 32 ss + 16 (jj >> 2) + 4 g + (jj & 3)), K^T by transposing reads in that order
        f32x4_t aq[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8_t dsf = __builtin_bit_cast(bf16x8_t, make_uint4(dsp[2 * ss].x, dsp[2 * ss].y, dsp[2 * ss + 1].x, dsp[2 * ss + 1].y));
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            aq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr(Ks, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, r16), dsf, aq[dt], 0, 0, 0);
        }
        bf16_t* row = p.dqkv + (b_ * 64 + 16 * it + r16) * ldq + head * 32 + scol;
        store8(row, pack_bf4(aq[0][0] * p.scale, aq[0][1] * p.scale, aq[0][2] * p.scale, aq[0][3] * p.scale),
               pack_bf4(aq[1][0] * p.scale, aq[1][1] * p.scale, aq[1][2] * p.scale, aq[1][3] * p.scale));
      }
      ABF_MARK(4);
      srk_lds_barrier();                                  // C: P and dS of the three heads are complete
      ABF_MARK(5);
      if (SRK_ABF_ISSUE_AT == 2 && t + 1 < nwin) issue(b_ + stride, cur ^ 1);

      // ---- phase C: key tile jt = `it` of head hl: dV^T, dK^T summed over all 64 queries ----------------------------------------------
      {
        f32x4_t av[2], ak[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) av[dt] = ak[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // every transposing read of the unit first (24 of them, 12 fragments), then the eight MFMAs
        bf16x8_t pf[2], df[2], otf[2][2], qtf[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int rb0 = 32 * ss + 8 * g, rb1 = rb0 + 4;
          pf[ss] = ptile_tr(Pb, rb0, rb1, it, r16);      // P[i][j in tile]   (k = i)
          df[ss] = ptile_tr(Db, rb0, rb1, it, r16);      // dS[i][j in tile]  (k = i)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            otf[ss][dt] = tile_tr(Os, rb0, rb1, dt, r16);
            qtf[ss][dt] = tile_tr(Qs, rb0, rb1, dt, r16);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf[ss][dt], pf[ss], av[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[ss][dt], df[ss], ak[dt], 0, 0, 0);
          }
        bf16_t* row = p.dqkv + (b_ * 64 + 16 * it + r16) * ldq + head * 32 + scol;
        store8(row + H_CA, pack_bf4(ak[0][0], ak[0][1], ak[0][2], ak[0][3]), pack_bf4(ak[1][0], ak[1][1], ak[1][2], ak[1][3]));
        store8(row + 2 * H_CA, pack_bf4(av[0][0], av[0][1], av[0][2], av[0][3]), pack_bf4(av[1][0], av[1][1], av[1][2], av[1][3]));
      }
      ABF_MARK(6);
    }
  }
  // partial d(bias) of this workgroup's three heads, dense [i][j]; this wave owns the rows i = 16 it + r16 of head hl
  // (the lane id is re-derived here: carried across the window loop it would be the one register too many)
  int lane2;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane2));
  float* slab = p.slab + (((long long)gidx * H_NH + head) * 64 + 16 * it + (lane2 & 15)) * 64 + 4 * (lane2 >> 4);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    *reinterpret_cast<float4*>(slab + 16 * jt) = make_float4(dbias[jt][0], dbias[jt][1], dbias[jt][2], dbias[jt][3]);
}

SrkOpt g_attn_bwd_fused{OPT_ATTN_BWD_FUSED, 1};

}  // namespace

void srk_attn_bwd_fused_enable(int on) { g_attn_bwd_fused = on ? 1 : 0; }
int srk_attn_bwd_fused_enabled() { return g_attn_bwd_fused; }

static int abf_cus() { return srk_device_cus(); }

// number of d(bias) slabs the fused backward writes for B_ windows (0: the kernel does not cover this problem)
int srk_qkv_attn_bwd_slabs(long long B_, int nH, int CA, int K) {
  if (!g_attn_bwd_fused || nH != H_NH || CA != H_CA || K != H_K) return 0;
  const int cus = abf_cus();
  if (cus < 16 || B_ < cus) return 0;                  // fewer windows than CUs: the weight preload does not amortise
  return 8 * (cus / 16);
}

// SRK_NOT_COVERED (1) when the kernel does not apply: the caller then runs the output-projection dgrad GEMM and srk_launch_attn_bwd
// on saved q/k/v.
int srk_launch_qkv_attn_bwd(const bf16_t* xn, int lda, const bf16_t* Wqkv, const float* bqkv, float scale, const bf16_t* g, int ldg,
                            const bf16_t* WprojT, const float* biasd, bf16_t* dqkv, float* slab, long long B_, int nH, int CA, int K,
                            WinGeom geom, hipStream_t stream) {
  const int nslab = srk_qkv_attn_bwd_slabs(B_, nH, CA, K);
  if (nslab == 0 || lda % 8 != 0 || ldg % 8 != 0 || B_ * 64 * (long long)(lda > ldg ? lda : ldg) >= (1LL << 31)) return SRK_NOT_COVERED;
  static SrkPerDevice<int> configured_pd; int& configured = configured_pd.here();
  if (!configured) {
    const void* fn = reinterpret_cast<const void*>(&qkv_attn_bwd_kernel);
    hipFuncAttributes attr;
    configured = -1;
    // a build that spills would put scratch traffic on the counted vmcnt waits: never run it
#ifdef SRK_PROBE_ABF
    attr.localSizeBytes = 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS) == hipSuccess) configured = 1;   // instrumented build: spills tolerated
#endif
    if (configured < 0 && hipFuncGetAttributes(&attr, fn) == hipSuccess && attr.localSizeBytes == 0 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS) == hipSuccess)
      configured = 1;
  }
  if (configured < 0) return SRK_NOT_COVERED;
  BwdFusedParams bp;
  bp.xn = xn; bp.lda = lda; bp.g = g; bp.ldg = ldg; bp.Wqkv = Wqkv; bp.bqkv = bqkv; bp.scale = scale; bp.WprojT = WprojT;
  bp.biasd = biasd; bp.dqkv = dqkv; bp.slab = slab; bp.B_ = B_; bp.geom = geom;
  const int ngrp = nslab / 8;
  srk_probe_pre(FAM_ATTN_BWD, stream, 0.0);
  hipLaunchKernelGGL(qkv_attn_bwd_kernel, dim3(16 * ngrp), dim3(768), H_LDS, stream, bp, ngrp);
  srk_probe_post(FAM_ATTN_BWD, stream);
  return srk_check_launch("qkv+attention backward");
}
