// Fused QKV projection + window attention forward for the classical SwinIR width (C = 180 -> 192, 6 heads x 32):
//
//   qkv = xn1_window . Wqkv^T + b  (q scaled)        network_swinir.py:121-124
//   ao  = softmax(q k^T + bias + mask) v              network_swinir.py:125-142
//
// Run separately, the projection writes q/k/v (150 MB per launch at cfg3) and the attention kernel reads them right
// back (another 150 MB); both kernels sit on the HBM roofline.  A window is 64 consecutive rows of the window-ordered
// LayerNorm output, so one workgroup can take a window from xn1 to ao with q/k/v living only in LDS; q/k/v are still
// written out once for the backward pass, but never re-read in forward (-150 MB per block, -1 launch); in inference they
// never leave the CU.
//
// One persistent 512-thread workgroup per CU walks the windows b_ = blockIdx.x, + gridDim.x, ...  All eight waves run
// the projection and the attention:
//   projection  the 36 16-column fragments of the 576 output columns are dealt 5 / 5 / 5 / 5 / 4 / 4 / 4 / 4 to waves 0..7;
//               waves w and w + 4 share a SIMD, so every SIMD carries nine fragments.  Each wave keeps its W slice in
//               registers (<= 30 MFMA fragments, 120 VGPRs) and walks the 64 rows in four 16-row quarters; results
//               (+ bias, q scale, bf16) go to per-(q/k/v, head) LDS tiles [64][40].
//   attention   the 24 (head, 16-query tile) units of the window are dealt three per wave (4 + 4 MFMAs each, softmax in
//               registers, as attn.hip).  The rel-pos bias of a unit is loaded one unit ahead (the first one before the
//               projection), because an L2 round trip waited for in place costs more than the unit itself.
//   stores      waves 0..6 write the q/k/v tiles (1 KB contiguous per instruction) and, after the last barrier, the
//               window's 24 KB ao tile -- all from LDS.  Their next loads (the bias prefetch) are consumed microseconds
//               later, so the in-order vmcnt counter never makes them wait for these stores.
//   loading     wave 7 never stores: it issues the LDS-DMA of the next window's 64 x 192 bf16 rows (swizzled on the source
//               address, 2-slot ring) right after the first barrier, and its vmcnt(0) before the next one is "landed".
// Three raw barriers per window.  The ao tile aliases the ring slot that the window's projection has just consumed.
#include <type_traits>

#include "kernels.h"

#ifndef SRK_NT_ATTN
#define SRK_NT_ATTN 1
#endif
#ifndef SRK_NT_STORE_QKV
#define SRK_NT_STORE_QKV 1
#endif
typedef unsigned srk_u4 __attribute__((ext_vector_type(4)));

#ifdef SRK_PROBE_ATTN
// developer instrumentation (never in the shipped build): s_memrealtime (100 MHz) at the phase boundaries of workgroup 0
__device__ unsigned long long g_attn_probe[8 * 16 * 8];
#define ATTN_MARK(k) do { if (blockIdx.x == 0 && lane == 0 && t < 16) g_attn_probe[(wave * 16 + t) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int srk_debug_attn_probe(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_probe), sizeof(g_attn_probe)); }
__device__ unsigned long long g_attn_unit_probe[8 * 3 * 8];
#define UNIT_MARK(m) do { if (blockIdx.x == 0 && lane == 0 && t == 3) g_attn_unit_probe[(wave * 3 + k) * 8 + (m)] = __builtin_readcyclecounter(); } while (0)
extern "C" int srk_debug_attn_unit_probe(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_unit_probe), sizeof(g_attn_unit_probe)); }
#else
#define ATTN_MARK(k) do {} while (0)
#define UNIT_MARK(m) do {} while (0)
#endif

namespace {

constexpr int FT = 40;                 // LDS row stride (elements) of a [64][32] bf16 head tile (as attn.hip TS)
constexpr int F_K = 192, F_CA = 192, F_NH = 6;
constexpr int F_SLOT = 64 * F_K * 2;   // 24576 B: one window of xn1 / the ao tile
constexpr int F_TILE = 64 * FT;        // elements per head tile
constexpr int F_TAB = 225;             // (2 ws - 1)^2 relative-position offsets
constexpr int F_OFF_PBIAS = 2 * F_SLOT + 18 * F_TILE * 2;
constexpr int F_OFF_TAB = F_OFF_PBIAS + 3 * F_CA * 4;
constexpr int F_LDS = F_OFF_TAB + F_NH * F_TAB * 4;    // ring + head tiles + projection bias (fp32) + rel-pos bias table [6][225]

struct FusedParams {
  const bf16_t* xn;     // [B_*64][lda] window-order rows
  int lda;
  const bf16_t* Wt;     // [576][192] packed qkv weight (q rows first)
  const float* bias;    // [576] or null
  float scale;
  bf16_t* qkv;          // [3][B_][6][64][32], or null (inference: nothing reads it)
  const float* biasd;   // [6][64][64] dense relative-position bias
  bf16_t* ao;           // [B_*64][192]
  long long B_;
  WinGeom geom;
};

__device__ __forceinline__ bf16x8_t f_cat4(bf16x4_t lo, bf16x4_t hi) {
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// NF: 16-column fragments of the projection owned by this wave (5 or 4); LOADER: this wave issues the DMA and never stores
template <int NF, bool LOADER>
__device__ __forceinline__ void fused_wave(const FusedParams& p, unsigned char* smem, int wave, int lane, long long nwin) {
  bf16_t* tiles = reinterpret_cast<bf16_t*>(smem + 2 * F_SLOT);       // [which][head][64][FT]
  const float* pbias = reinterpret_cast<const float*>(smem + F_OFF_PBIAS);
  const unsigned smem_base = (unsigned)(size_t)smem;
  const int r16 = lane & 15, g = lane >> 4;
  const int f0 = wave < 4 ? 5 * wave : 20 + 4 * (wave - 4);          // first fragment of this wave

  bf16x8_t wf[NF][6];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int s = 0; s < 6; ++s)
      wf[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wt + (long long)(16 * (f0 + j) + r16) * F_K + s * 32 + g * 8);

  // loader only: element offset of the 16-byte piece lane `lane` moves in DMA instruction i (constant over the windows)
  int off[LOADER ? 24 : 1];
  if constexpr (LOADER) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const int q = i * 64 + lane;
      const int row = q / 24, pos = q - row * 24;
      off[i] = row * p.lda + ((pos ^ (row & 7)) << 3);
    }
  }
  auto issue = [&](long long t) {
    if constexpr (LOADER) {
      const bf16_t* base = p.xn + (blockIdx.x + t * gridDim.x) * 64 * (long long)p.lda;
      const unsigned dst = smem_base + (unsigned)((t & 1) * F_SLOT);
#pragma unroll
      for (int i = 0; i < 24; ++i) srk_glds16<SRK_NT_ATTN != 0>(base + off[i], __builtin_amdgcn_readfirstlane(dst + i * 1024));
    }
  };
  // rel-pos bias: this lane's query i = 16 it + r16 is the same in all three units of the wave (it = wave & 3); key
  // j = 16 jt + 4 g + e -> table index (yi - yj + 7) 15 + (xi - xj + 7) = lane_idx - 30 jt - e   (network_swinir.py:89-103)
  const int it = wave & 3, hw = wave >> 2;
  const float* tabl = reinterpret_cast<const float*>(smem + F_OFF_TAB) +
                      ((2 * it + (r16 >> 3)) - (g >> 1) + 7) * 15 + ((r16 & 7) - 4 * (g & 1) + 7) - 93;
  if constexpr (LOADER) issue(0);
  for (long long t = 0; t < nwin; ++t) {
    const long long b_ = blockIdx.x + t * gridDim.x;
    const unsigned char* As = smem + (t & 1) * F_SLOT;
    ATTN_MARK(0);
    if constexpr (LOADER) {
      srk_wait_vmcnt<0>();                      // this wave's DMAs of window t (and nothing else) have landed
      srk_lds_barrier();                        // Ba
      if (t + 1 < nwin) issue(t + 1);           // into slot (t+1)&1: the ao tile it held (window t-1) was stored before Ba(t)
    } else {
      srk_lds_barrier();                        // Ba: xn1 rows of window t are in LDS; head tiles are free
    }
    ATTN_MARK(1);
    // ---- projection: 64 rows x (16 NF) columns in four 16-row quarters ---------------------------------------------
#pragma unroll 1
    for (int mq = 0; mq < 4; ++mq) {
      f32x4_t acc[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * mq + r16;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(As + row * (F_K * 2) + (((s * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[j], 0, 0, 0);
      }
      // fragment f: columns 16 f .. 16 f + 15 -> which = f / 12, head = (f % 12) / 2, d = 16 (f & 1) + 4 g + e
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int f = f0 + j;
        const int which = f / 12, hh = (f - 12 * which) >> 1, d = 16 * (f & 1) + 4 * g;
        const float sc = which == 0 ? p.scale : 1.0f;
        const float4 bq = *reinterpret_cast<const float4*>(pbias + 16 * f + 4 * g);
        *reinterpret_cast<uint2*>(tiles + (which * F_NH + hh) * F_TILE + row * FT + d) =
            pack_bf4((acc[j][0] + bq.x) * sc, (acc[j][1] + bq.y) * sc, (acc[j][2] + bq.z) * sc, (acc[j][3] + bq.w) * sc);
      }
    }
    ATTN_MARK(2);
    srk_lds_barrier();                          // Bb: all q/k/v head tiles of the window are complete
    ATTN_MARK(3);
    // ---- q / k / v tiles -> global for the backward pass, issued BEFORE the attention so that the 72 KB drain
    // under its compute instead of arriving with the ao tile in one burst: tile tl = wave, wave + 7, wave + 14 (waves 0..6) ------------
    if constexpr (!LOADER) {
      if (p.qkv) {
#pragma unroll 1
        for (int tl = wave; tl < 18; tl += 7) {
          const int which = tl / F_NH, hh = tl - which * F_NH;
          const bf16_t* src = tiles + tl * F_TILE;
          bf16_t* dst = p.qkv + ((which * p.B_ + b_) * F_NH + hh) * 2048;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + (lane >> 2), ch = lane & 3;
            const uint4 t4 = *reinterpret_cast<const uint4*>(src + row * FT + ch * 8);
            if constexpr (SRK_NT_STORE_QKV != 0)   // next read by the backward pass
              __builtin_nontemporal_store(srk_u4{t4.x, t4.y, t4.z, t4.w}, reinterpret_cast<srk_u4*>(dst + row * 32 + ch * 8));
            else
              *reinterpret_cast<uint4*>(dst + row * 32 + ch * 8) = t4;
          }
        }
      }
    }
    ATTN_MARK(4);
    // ---- attention: units u = wave, wave + 8, wave + 16 (u = 4 head + query tile) ----------------------------------
    {
      const int w = (int)(b_ % p.geom.nW);
      const int wy = w / p.geom.nWw, wx = w - wy * p.geom.nWw;
      const bool masked = p.geom.shift > 0 && (wy == p.geom.H / 8 - 1 || wx == p.geom.nWw - 1);
      unsigned char* aot = smem + (t & 1) * F_SLOT;            // the consumed xn1 slot becomes the ao tile [64][192]
#pragma unroll 1
      for (int k = 0; k < 3; ++k) {
        const int h = hw + 2 * k;                                // unit u = wave + 8 k = 4 h + it
        UNIT_MARK(0);
        const bf16_t* Qs = tiles + (0 * F_NH + h) * F_TILE;
        const bf16_t* Ks = tiles + (1 * F_NH + h) * F_TILE;
        const bf16_t* Vs = tiles + (2 * F_NH + h) * F_TILE;
        const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(Qs + (16 * it + r16) * FT + 8 * g);
        f32x4_t s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(Ks + (16 * jt + r16) * FT + 8 * g), qf,
                                                        f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        UNIT_MARK(1);
        {
          const float* th = tabl + h * F_TAB;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
            s[jt] += f32x4_t{th[93 - 30 * jt], th[92 - 30 * jt], th[91 - 30 * jt], th[90 - 30 * jt]};
        }
        if (masked) {
          const int labi = win_region_label(p.geom, w, 16 * it + r16);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (win_region_label(p.geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;   // :235 (-100, not -inf)
        }
        UNIT_MARK(2);
        bf16x8_t pf[2];
        const float inv = softmax_numerators(s, pf);
        UNIT_MARK(3);
        f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8_t vf = f_cat4(lds_tr_read(tr_addr(Vs, FT, 32 * ss + 4 * g, 16 * dt, lane)),
                                       lds_tr_read(tr_addr(Vs, FT, 32 * ss + 16 + 4 * g, 16 * dt, lane)));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ss], o[dt], 0, 0, 0);
          }
        }
        o[0] *= inv;
        o[1] *= inv;
        UNIT_MARK(4);
        // o[dt][e] = O[i = 16 it + r16][d = 16 dt + 4 g + e]: 8 consecutive d per lane after the 16-lane-row swap
        const uint2 x = pack_bf4(o[0][0], o[0][1], o[0][2], o[0][3]), y = pack_bf4(o[1][0], o[1][1], o[1][2], o[1][3]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        *reinterpret_cast<uint4*>(aot + ((16 * it + r16) * F_CA + h * 32 + (((g & 1) << 4) | ((g >> 1) << 3))) * 2) =
            make_uint4(s0[0], s1[0], s0[1], s1[1]);
        UNIT_MARK(5);
      }
    }
    ATTN_MARK(5);
    srk_lds_barrier();                          // Bc: the ao tile is complete; head tiles are no longer read
    ATTN_MARK(6);
    if constexpr (!LOADER) {                    // the window's 24 KB of ao: 24 x 1 KB over waves 0..6
      const unsigned char* aot = smem + (t & 1) * F_SLOT;
      bf16_t* adst = p.ao + b_ * 64 * F_CA;
#pragma unroll 1
      for (int i = wave; i < 24; i += 7) {
        const int idx = i * 64 + lane;
        *reinterpret_cast<uint4*>(adst + idx * 8) = *reinterpret_cast<const uint4*>(aot + idx * 16);
      }
    }
    ATTN_MARK(7);
  }
}

__global__ __launch_bounds__(512) void qkv_attn_fwd_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long nwin = (p.B_ - blockIdx.x + gridDim.x - 1) / gridDim.x;     // windows of this workgroup
  if (nwin <= 0) return;
  // projection bias -> LDS once (no registers to spare for it; re-reading it from L2 per quarter put an exposed ~1 us
  // round trip into every quarter); visible to everyone after the first barrier of the window loop
  float* pbias = reinterpret_cast<float*>(smem + F_OFF_PBIAS);
  for (int i = tid; i < 3 * F_CA; i += 512) pbias[i] = p.bias ? p.bias[i] : 0.f;
  // rel-pos bias table [6][225] out of the dense [6][64][64] (offset (dy, dx) is realised by the pair i = (max(dy, 0), max(dx, 0)),
  // j = (max(-dy, 0), max(-dx, 0))): 5.4 KB of LDS instead of an L2 round trip per unit, which was what paced the attention phase
  float* tab = reinterpret_cast<float*>(smem + F_OFF_TAB);
  for (int i = tid; i < F_NH * F_TAB; i += 512) {
    const int h = i / F_TAB, t = i - h * F_TAB;
    const int dy = t / 15 - 7, dx = t - (t / 15) * 15 - 7;
    const int qi = (dy > 0 ? dy : 0) * 8 + (dx > 0 ? dx : 0), kj = (dy < 0 ? -dy : 0) * 8 + (dx < 0 ? -dx : 0);
    tab[i] = p.biasd[h * 4096 + qi * 64 + kj];
  }
  if (wave < 4) fused_wave<5, false>(p, smem, wave, lane, nwin);
  else if (wave < 7) fused_wave<4, false>(p, smem, wave, lane, nwin);
  else fused_wave<4, true>(p, smem, wave, lane, nwin);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Second shape of the same fusion: THREE 4-wave workgroups per CU, each owning one head pair (64 of the 192 attention channels)
// of the windows it walks.  Phase timestamps of the 8-wave kernel above (tools/attn_probe.py) show its window as a chain of
// latency-bound phases -- projection 2.8 us (MFMA pipe, VALU idle), attention 2.4-3.2 us (one dependent chain per wave: LDS ->
// MFMA -> softmax -> MFMA -> store, the SIMD two-thirds idle), stores -- with every wave of the CU in the same phase.  Three
// independent workgroups per CU drift apart, so one's projection MFMAs run under another's softmax and a third's stores, and
// each SIMD has three waves to pick from.  Per workgroup and window:
//   xn1 rows    64 x 192 bf16 (24 KB) prefetched into registers one window ahead (default cache policy: the two sibling
//               workgroups of the window sit on the same XCD and find the rows in its L2), then written XOR-swizzled to LDS
//   projection  192 output columns (q, k, v of two heads) = 12 fragments, three per wave with W in registers (72 VGPRs)
//   head tiles  six [64][32] bf16 tiles, 16-byte chunks XOR-swizzled with (row >> 2) & 3 instead of padded (3 x 51 KB of LDS)
//   attention   two units per wave (head 0 / 1 of the pair, query tile = wave), as above; ao goes to global from registers
// Two barriers per window.  Grid = 24 g workgroups: id -> xcd = id & 7, slot = id >> 3, head pair = slot % 3, group = slot / 3;
// the workgroup walks windows (8 group + xcd) + 8 g t.
constexpr int G_TILE = 64 * 32;                                       // elements per swizzled head tile
constexpr int G_OFF_TILES = F_SLOT;
constexpr int G_OFF_PBIAS = G_OFF_TILES + 6 * G_TILE * 2;
constexpr int G_OFF_TAB = G_OFF_PBIAS + 192 * 4;
constexpr int G_LDS = G_OFF_TAB + 2 * F_TAB * 4;                      // 51 720 B

__device__ __forceinline__ int g_tile_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }   // elements

__global__ __launch_bounds__(256, 3) void qkv_attn_fwd3_kernel(const FusedParams p, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int hp = slot % 3, grp = slot / 3;
  const long long first = (long long)grp * 8 + xcd, stride = (long long)ngrp * 8;
  if (first >= p.B_) return;
  const long long nwin = (p.B_ - first + stride - 1) / stride;

  bf16_t* tiles = reinterpret_cast<bf16_t*>(smem + G_OFF_TILES);       // [which][head of the pair][64][32] swizzled
  float* pbias = reinterpret_cast<float*>(smem + G_OFF_PBIAS);        // [12 fragments][16]
  float* tab = reinterpret_cast<float*>(smem + G_OFF_TAB);            // [2][225]
  for (int i = tid; i < 192; i += 256) {
    const int lf = i >> 4, which = lf >> 2, hl = (lf >> 1) & 1, half = lf & 1;
    pbias[i] = p.bias ? p.bias[which * F_CA + (2 * hp + hl) * 32 + half * 16 + (i & 15)] : 0.f;
  }
  for (int i = tid; i < 2 * F_TAB; i += 256) {                        // table out of the dense bias, as in the kernel above
    const int hl = i / F_TAB, t = i - hl * F_TAB;
    const int dy = t / 15 - 7, dx = t - (t / 15) * 15 - 7;
    const int qi = (dy > 0 ? dy : 0) * 8 + (dx > 0 ? dx : 0), kj = (dy < 0 ? -dy : 0) * 8 + (dx < 0 ? -dx : 0);
    tab[i] = p.biasd[(2 * hp + hl) * 4096 + qi * 64 + kj];
  }

  // W slice of this wave: local fragments lf = 3 wave + j -> (which, head of the pair, 16-column half)
  bf16x8_t wf[3][6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int lf = 3 * wave + j, which = lf >> 2, hl = (lf >> 1) & 1, half = lf & 1;
    const bf16_t* wr = p.Wt + (long long)(which * F_CA + (2 * hp + hl) * 32 + half * 16 + r16) * F_K;
#pragma unroll
    for (int s = 0; s < 6; ++s) wf[j][s] = *reinterpret_cast<const bf16x8_t*>(wr + s * 32 + g * 8);
  }

  // xn1 staging: thread = (row = tid >> 2, cq = tid & 3) moves the 16-byte chunks 4 i + cq, i < 6, of its row (64 contiguous
  // bytes per row and instruction); two halves of three chunks, each held in registers only between its load and its LDS write.
  // swizzled chunk (4 i + cq) ^ (row & 7) = 8 (i >> 1) + ((cq ^ (row & 7)) ^ 4 (i & 1))
  const int srow = tid >> 2, scq = tid & 3;
  const long long grow = (long long)srow * p.lda + scq * 8;
  const int lrow = srow * (F_K * 2), lc0 = (scq ^ (srow & 7)) << 4, lc1 = lc0 ^ 64;
  uint4 pre0, pre1, pre2;
#define G_PREFETCH(BW, HALF)                                                                     \
  do {                                                                                           \
    const bf16_t* base_ = p.xn + (BW) * 64 * (long long)p.lda + grow + (HALF) * 96;              \
    pre0 = *reinterpret_cast<const uint4*>(base_);                                               \
    pre1 = *reinterpret_cast<const uint4*>(base_ + 32);                                          \
    pre2 = *reinterpret_cast<const uint4*>(base_ + 64);                                          \
  } while (0)
  // chunks ii = 3 HALF + i: HALF 0 -> (0: lc0) (1: lc1) (2: 128 + lc0); HALF 1 -> (3: 128 + lc1) (4: 256 + lc0) (5: 256 + lc1)
#define G_STAGE(HALF)                                                                            \
  do {                                                                                           \
    unsigned char* d_ = smem + lrow;                                                             \
    if ((HALF) == 0) {                                                                           \
      *reinterpret_cast<uint4*>(d_ + lc0) = pre0;                                                \
      *reinterpret_cast<uint4*>(d_ + lc1) = pre1;                                                \
      *reinterpret_cast<uint4*>(d_ + 128 + lc0) = pre2;                                          \
    } else {                                                                                     \
      *reinterpret_cast<uint4*>(d_ + 128 + lc1) = pre0;                                          \
      *reinterpret_cast<uint4*>(d_ + 256 + lc0) = pre1;                                          \
      *reinterpret_cast<uint4*>(d_ + 256 + lc1) = pre2;                                          \
    }                                                                                            \
  } while (0)
  G_PREFETCH(first, 0);
  G_STAGE(0);
  G_PREFETCH(first, 1);
  G_STAGE(1);

  const int it = wave;
  const float* tabl = tab + ((2 * it + (r16 >> 3)) - (g >> 1) + 7) * 15 + ((r16 & 7) - 4 * (g & 1) + 7) - 93;

  for (long long t = 0; t < nwin; ++t) {
    const long long b_ = first + t * stride;
    srk_lds_barrier();                          // A: xn1 rows of window t (and, first time, bias + table) are in LDS; tiles are free
    const bool more = t + 1 < nwin;
    if (more) G_PREFETCH(b_ + stride, 0);
    // ---- projection: 64 rows x 48 columns per wave in four 16-row quarters ----------------------------------------------
#pragma unroll 1
    for (int mq = 0; mq < 4; ++mq) {
      f32x4_t acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * mq + r16;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(smem + row * (F_K * 2) + (((s * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int lf = 3 * wave + j, which = lf >> 2, hl = (lf >> 1) & 1, half = lf & 1;
        const float sc = which == 0 ? p.scale : 1.0f;
        const float4 bq = *reinterpret_cast<const float4*>(pbias + 16 * lf + 4 * g);
        *reinterpret_cast<uint2*>(tiles + (which * 2 + hl) * G_TILE + g_tile_off(row, 2 * half + (g >> 1)) + 4 * (g & 1)) =
            pack_bf4((acc[j][0] + bq.x) * sc, (acc[j][1] + bq.y) * sc, (acc[j][2] + bq.z) * sc, (acc[j][3] + bq.w) * sc);
      }
    }
    srk_lds_barrier();                          // B: the six head tiles are complete; the xn1 rows are no longer read
    if (more) {
      G_STAGE(0);
      G_PREFETCH(b_ + stride, 1);
    }
    // ---- q / k / v -> global for the backward pass: thread = (row, 16-byte chunk) of each tile --------------------------------
    if (p.qkv) {
      const int row = tid >> 2, ch = tid & 3;
#pragma unroll
      for (int tl = 0; tl < 6; ++tl) {
        const uint4 t4 = *reinterpret_cast<const uint4*>(tiles + tl * G_TILE + g_tile_off(row, ch));
        bf16_t* dst = p.qkv + (((tl >> 1) * p.B_ + b_) * F_NH + 2 * hp + (tl & 1)) * 2048 + row * 32 + ch * 8;
        if constexpr (SRK_NT_STORE_QKV != 0)
          __builtin_nontemporal_store(srk_u4{t4.x, t4.y, t4.z, t4.w}, reinterpret_cast<srk_u4*>(dst));
        else
          *reinterpret_cast<uint4*>(dst) = t4;
      }
    }
    // ---- attention: head hl of the pair, query tile it = wave ------------------------------------------------------------------
    {
      const int w = (int)(b_ % p.geom.nW);
      const int wy = w / p.geom.nWw, wx = w - wy * p.geom.nWw;
      const bool masked = p.geom.shift > 0 && (wy == p.geom.H / 8 - 1 || wx == p.geom.nWw - 1);
#pragma unroll 1
      for (int hl = 0; hl < 2; ++hl) {
        const bf16_t* Qs = tiles + (0 + hl) * G_TILE;
        const bf16_t* Ks = tiles + (2 + hl) * G_TILE;
        const bf16_t* Vs = tiles + (4 + hl) * G_TILE;
        const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(Qs + g_tile_off(16 * it + r16, g));
        f32x4_t s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(Ks + g_tile_off(16 * jt + r16, g)), qf,
                                                        f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        {
          const float* th = tabl + hl * F_TAB;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
            s[jt] += f32x4_t{th[93 - 30 * jt], th[92 - 30 * jt], th[91 - 30 * jt], th[90 - 30 * jt]};
        }
        if (masked) {
          const int labi = win_region_label(p.geom, w, 16 * it + r16);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (win_region_label(p.geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;   // :235 (-100, not -inf)
        }
        bf16x8_t pf[2];
        const float inv = softmax_numerators(s, pf);
        f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
        // V^T fragments: lane 4 q + c of a 16-lane group supplies row (key) rb + q, columns 16 dt + 4 c .. + 3; the rows of one
        // transposing read share (row >> 2) & 3 = g, so its chunk swizzle is a lane-group constant
        const int ll = lane & 15;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int ch = ((2 * dt + ((ll & 3) >> 1)) ^ g) << 3, in = (ll & 1) << 2;
            const bf16x8_t vf = f_cat4(lds_tr_read(Vs + (32 * ss + 4 * g + (ll >> 2)) * 32 + ch + in),
                                       lds_tr_read(Vs + (32 * ss + 16 + 4 * g + (ll >> 2)) * 32 + ch + in));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ss], o[dt], 0, 0, 0);
          }
        }
        o[0] *= inv;
        o[1] *= inv;
        const uint2 x = pack_bf4(o[0][0], o[0][1], o[0][2], o[0][3]), y = pack_bf4(o[1][0], o[1][1], o[1][2], o[1][3]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        *reinterpret_cast<uint4*>(p.ao + (b_ * 64 + 16 * it + r16) * F_CA + (2 * hp + hl) * 32 + (((g & 1) << 4) | ((g >> 1) << 3))) =
            make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    }
    if (more) G_STAGE(1);
  }
}

#ifndef SRK_ATTN_FUSED_DEFAULT
#define SRK_ATTN_FUSED_DEFAULT 2
#endif
SrkOpt g_attn_fused{OPT_ATTN_FUSED, SRK_ATTN_FUSED_DEFAULT};            // 0 separate kernels, 1 one 8-wave workgroup per CU, 2 three 4-wave workgroups per CU

}  // namespace

void srk_attn_fused_enable(int on) { g_attn_fused = on < 0 ? 0 : (on > 2 ? 2 : on); }
int srk_attn_fused_mode() { return g_attn_fused; }

// SRK_NOT_COVERED (1) when the fused kernel does not apply: the caller then runs the projection GEMM and srk_launch_attn_fwd.
int srk_launch_qkv_attn_fwd(const bf16_t* xn, int lda, const bf16_t* Wt, const float* bias, float scale, bf16_t* qkv,
                            const float* biasd, bf16_t* ao, long long B_, int nH, int CA, int K, WinGeom geom, hipStream_t stream) {
  if (!g_attn_fused || nH != F_NH || CA != F_CA || K != F_K || lda % 8 != 0) return SRK_NOT_COVERED;
  const int g_fused_cus = srk_device_cus();
  if (g_fused_cus < 1 || B_ < g_fused_cus) return SRK_NOT_COVERED;      // fewer windows than CUs: the W preload does not amortise
  static SrkPerDevice<int> configured_pd; int& configured = configured_pd.here();
  if (!configured) {
    const void* fn = reinterpret_cast<const void*>(&qkv_attn_fwd_kernel);
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, fn) != hipSuccess) {
      srk_set_error("qkv+attention: cannot query the kernel");
      return SRK_E_LAUNCH;
    }
    if (attr.localSizeBytes > 0) {
      configured = -1;              // spills would put scratch traffic on the loader's vmcnt counter
    } else {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS) != hipSuccess) {
        srk_set_error("qkv+attention: cannot reserve %d bytes of LDS", F_LDS);
        return SRK_E_LAUNCH;
      }
      configured = 1;
    }
  }
  static SrkPerDevice<int> configured3_pd; int& configured3 = configured3_pd.here();
  if (!configured3) {
    const void* fn = reinterpret_cast<const void*>(&qkv_attn_fwd3_kernel);
    hipFuncAttributes attr;
    configured3 = -1;
    if (hipFuncGetAttributes(&attr, fn) == hipSuccess && attr.localSizeBytes == 0 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) == hipSuccess)
      configured3 = 1;
  }
  if (configured < 0) return SRK_NOT_COVERED;
  FusedParams fp;
  fp.xn = xn; fp.lda = lda; fp.Wt = Wt; fp.bias = bias; fp.scale = scale; fp.qkv = qkv; fp.biasd = biasd; fp.ao = ao; fp.B_ = B_;
  fp.geom = geom;
  if (g_attn_fused == 2 && configured3 > 0) {
    long long ngrp = g_fused_cus / 8;                       // 24 workgroups per group of 8 windows in flight: 3 per CU
    if (ngrp * 8 > B_) ngrp = (B_ + 7) / 8;
    hipLaunchKernelGGL(qkv_attn_fwd3_kernel, dim3((unsigned)(24 * ngrp)), dim3(256), G_LDS, stream, fp, (int)ngrp);
    return srk_check_launch("qkv+attention (3 per CU)");
  }
  hipLaunchKernelGGL(qkv_attn_fwd_kernel, dim3(g_fused_cus), dim3(512), F_LDS, stream, fp);
  return srk_check_launch("qkv+attention");
}
