// Streaming (persistent) variant of the bf16 MFMA linear GEMM for the skinny transformer-block GEMMs
//
//   D[m][n] = sum_k A[m][k] * Wt[n][k]   (+ row epilogue),   K in {192, 384, 576},  N a multiple of 192
//
// These GEMMs have an arithmetic intensity of 100-150 FLOP/B -- below the MI355X ridge -- so the job is to keep
// HBM streaming, not to feed the MFMA.  The tile-per-workgroup kernel (gemm.hip) serialises, per tile, a chain of
// dependent memory round trips (K-loop prefetch -> LDS -> MFMA -> epilogue loads -> stores) and leaves the memory
// system idle for most of it.  Here one 512-thread workgroup per CU walks a strided list of BM-row tiles for ONE
// 192-column slice of W:
//
//   * the W slice lives in REGISTERS as MFMA fragments for the whole kernel (no per-tile W traffic at all);
//   * every global read of the loop -- the A rows AND the epilogue's row operands (fp32 residual rows, LayerNorm
//     input + gradient-stream rows, bf16 pre-activations, per-row statistics) -- is an LDS-DMA (global_load_lds) into
//     an R-deep ring of LDS slots, issued by waves 4-7 R-1 (R-2) tiles ahead of its use and retired with a counted
//     s_waitcnt vmcnt, so that >= 64 KB per CU is in flight at all times without costing VGPRs; those waves never
//     store, so their vmcnt counter sees only DMAs and the counted wait is exact (CDNA4 counts stores on vmcnt too);
//   * the accumulators go through an LDS tile T and are re-read ROW-major (16 lanes per row, the layout of
//     gemm_rowep.h) together with the row operands of the slot by waves 0-3, which only read LDS and store -- they
//     never wait on vmcnt; raw s_barrier with s_waitcnt lgkmcnt(0) (a __syncthreads would drain the DMAs).
//
// Two kernels share the loader and the epilogue:
//   gemm_stream_split_kernel  role split: waves 4-7 (front) = DMA + MFMA of tile i into T[i & 1], waves 0-3 (back) = row
//                             epilogue of tile i-1 from T[(i-1) & 1]; ONE barrier per tile, the MFMA pipe and the
//                             epilogue's VALU / LDS / store work overlap on every SIMD.  Default for K <= 384.
//   gemm_stream_kernel        symmetric: waves 0-3 run MFMA + epilogue, waves 4-7 load (two barriers per tile); with KS2
//                             both groups run half of K and the loaders add their half into T (three barriers).  Used
//                             where the split variant does not fit: K = 576 (216 VGPRs of W per wave) and the
//                             LayerNorm-backward epilogues (two fp32 row operands per tile leave no LDS for a second T).
// Per-(epilogue, K) choices and tile heights were measured with tools/stream_sweep.py (srk_set_option overrides).
//
// The A slot image is [BM][K] bf16 row-major with the 16-byte chunk index XOR-swizzled by (row & 7).  An LDS-DMA
// writes lane-linear (wave-uniform base + 16 * lane), so the swizzle is applied on the per-lane SOURCE address.
#include <hip/hip_runtime.h>

#include "gemm.h"
#include "gemm_rowep.h"

namespace {

constexpr int SBN = 192;         // columns per workgroup
constexpr int SBNP = SBN + 4;    // T row pitch in floats (conflict-free float4 writes from the MFMA layout)
constexpr int NC = SBN / 64;     // float4 per lane per row in the row-major phase
constexpr int LDS_BUDGET = 160 * 1024;

#ifndef SRK_NT_GEMM
#define SRK_NT_GEMM 1
#endif
#ifndef SRK_NT_STORE_U
#define SRK_NT_STORE_U 1
#endif
typedef unsigned srk_u2 __attribute__((ext_vector_type(2)));
typedef float srk_f4 __attribute__((ext_vector_type(4)));
#ifndef SRK_NT_STORE_ALL
#define SRK_NT_STORE_ALL 0
#endif
__device__ __forceinline__ void st_u2(bf16_t* ptr, uint2 v) {
  if constexpr (SRK_NT_STORE_ALL != 0) __builtin_nontemporal_store(srk_u2{v.x, v.y}, reinterpret_cast<srk_u2*>(ptr));
  else *reinterpret_cast<uint2*>(ptr) = v;
}
__device__ __forceinline__ void st_f4(float* ptr, float4 v) {
  if constexpr (SRK_NT_STORE_ALL != 0) __builtin_nontemporal_store(srk_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<srk_f4*>(ptr));
  else *reinterpret_cast<float4*>(ptr) = v;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) { srk_glds16<SRK_NT_GEMM != 0>(gsrc, lds_dst); }
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  srk_wait_vmcnt<N>();
}
// LDS operations of this wave complete, then the workgroup barrier (no vmcnt drain: DMAs and stores stay in flight)
__device__ __forceinline__ void lds_barrier() { srk_lds_barrier(); }

template <int EP, int KC, int BM>
struct StreamCfg {
  static constexpr int K = 64 * KC;
  static constexpr int NE32 = (EP == EP_PROJ_RES || EP == EP_RES) ? 1 : (EP == EP_LNBWD ? 2 : 0);   // fp32 row operands
  static constexpr int NE16 = (EP == EP_DGELU) ? 1 : 0;                                              // bf16 row operands
  static constexpr bool AUX = (EP == EP_PROJ_RES || EP == EP_RES || EP == EP_LNBWD);               // per-row scalars
  static constexpr int A_BYTES = BM * K * 2;
  static constexpr int E32_BYTES = BM * SBN * 4;
  static constexpr int E16_BYTES = BM * SBN * 2;
  static constexpr int AUX_BYTES = AUX ? 6 * 64 * 4 : 0;          // 4 DMA'd float arrays (one per loader wave) + row maps tok / ro
  static constexpr int SLOT = A_BYTES + NE32 * E32_BYTES + NE16 * E16_BYTES + AUX_BYTES;
  static constexpr int T_BYTES = BM * SBNP * 4;
  static constexpr int RED_BYTES = (EP == EP_LNBWD) ? 2 * SBN * 4 : 0;
  // DMA instructions per loader wave per tile (each loader wave moves a quarter of every operand image)
  static constexpr int CPW_A = BM * (K / 8) / 4, NI_A = (CPW_A + 63) / 64;
  static constexpr int CPW_E32 = BM * (SBN / 4) / 4, NI_E32 = (CPW_E32 + 63) / 64;
  static constexpr int CPW_E16 = BM * (SBN / 8) / 4, NI_E16 = (CPW_E16 + 63) / 64;
  static constexpr int P = NI_A + NE32 * NI_E32 + NE16 * NI_E16 + (AUX ? 1 : 0);
  static constexpr int R_LDS = (LDS_BUDGET - T_BYTES - RED_BYTES) / SLOT;
  static constexpr int R_CNT = 2 + 63 / P;                         // vmcnt is a 6-bit counter
  static constexpr int R0 = R_LDS < R_CNT ? R_LDS : R_CNT;
  static constexpr int R = R0 > 8 ? 8 : (R0 < 2 ? 2 : R0);
  static constexpr bool VALID = R0 >= 2;                             // the ring needs at least one tile in flight
  static constexpr int LDS = T_BYTES + RED_BYTES + R * SLOT;
  static_assert(P * (R - 2) <= 63, "vmcnt overflow");
};

// ---- loader: all DMAs of one tile ------------------------------------------------------------
// Which 16-byte piece of each operand image a lane moves in DMA instruction i never changes, so everything that depends
// on the lane only is computed once per kernel (IssueState); what depends on the tile is uniform (a tile of BM <= 64
// rows lies inside one window and one sample) and is computed once per tile on values the compiler keeps in SGPRs.  Per
// DMA instruction this leaves a handful of VALU operations: timestamps showed the former per-instruction address
// arithmetic (two integer divisions per window-ordered row) costing 1.3-1.4 us per 16-row tile of the LayerNorm-
// backward epilogue, on the critical path of the tile loop.
template <int EP, int KC, int BM>
struct IssueState {
  using C = StreamCfg<EP, KC, BM>;
  int aoff[C::NI_A];                                   // A piece: element offset relative to row m0
  int erow[C::NE32 > 0 ? C::NI_E32 : 1];               // fp32 row operands: tile row ...
  int eoff[C::NE32 > 0 ? C::NI_E32 : 1];               // ... and element offset (row * ldo + column for raster rows)
  int hoff[C::NE16 > 0 ? C::NI_E16 : 1];               // bf16 row operand: element offset relative to row m0
  float r_nW, r_nWw, r_rps, r_ohw, r_oW;               // reciprocals of the launch-constant divisors (fdiv24)
};

// x / d for 0 <= x < 2^24, d > 0, rcp ~ 1 / d: the float quotient is off by at most one, two fix-ups make it exact
// (~11 VALU operations instead of the ~30 of the generic 32-bit division expansion; the launcher bounds M by 2^24)
__device__ __forceinline__ int fdiv24(int x, int d, float rcp, int& rem) {
  int q = (int)((float)x * rcp);
  int r = x - q * d;
  if (r < 0) { q -= 1; r += d; }
  if (r >= d) { q += 1; r -= d; }
  rem = r;
  return q;
}

template <int EP, int KC, int BM>
__device__ __forceinline__ void issue_init(const GemmParams& p, int n0, int lw, int lane, IssueState<EP, KC, BM>& is) {
  using C = StreamCfg<EP, KC, BM>;
  constexpr int CRA = C::K / 8;
#pragma unroll
  for (int i = 0; i < C::NI_A; ++i) {
    const int qq = lw * C::CPW_A + i * 64 + lane;
    const int row = qq / CRA, pos = qq - row * CRA;
    is.aoff[i] = row * p.lda + ((pos ^ (row & 7)) << 3);
  }
  if constexpr (C::NE32 > 0) {
#pragma unroll
    for (int i = 0; i < C::NI_E32; ++i) {
      const int qq = lw * C::CPW_E32 + i * 64 + lane;
      const int row = qq / (SBN / 4), pos = qq - row * (SBN / 4);
      is.erow[i] = row;
      is.eoff[i] = n0 + pos * 4;
    }
  }
  if constexpr (C::NE16 > 0) {
#pragma unroll
    for (int i = 0; i < C::NI_E16; ++i) {
      const int qq = lw * C::CPW_E16 + i * 64 + lane;
      const int row = qq / (SBN / 8), pos = qq - row * (SBN / 8);
      is.hoff[i] = row * p.ldo + n0 + pos * 8;
    }
  }
  const WinGeom& og = (EP == EP_LNBWD) ? p.geom : p.xn_geom;
  is.r_nW = 1.0f / (float)max(p.geom.nW, 1);
  is.r_nWw = 1.0f / (float)max(p.geom.nWw, 1);
  is.r_rps = 1.0f / (float)max(p.rows_per_sample, 1);
  is.r_ohw = 1.0f / (float)max(og.H * og.W, 1);
  is.r_oW = 1.0f / (float)max(og.W, 1);
}

// the window / sample a tile lies in (uniform)
struct TileGeom {
  int tokbase, ybase, xbase, pbase, b;   // window-ordered rows: raster token = tokbase + y * W + x
};

template <int EP, int KC, int BM>
__device__ __forceinline__ void stream_issue_tile(const GemmParams& p, const IssueState<EP, KC, BM>& is, int m0, int n0, unsigned slot, int lw,
                                                  int lane, unsigned char* smem_ptr, unsigned smem_base) {
  using C = StreamCfg<EP, KC, BM>;
  const bf16_t* abase = p.A + (long long)m0 * p.lda;
#pragma unroll
  for (int i = 0; i < C::NI_A; ++i)
    if (i * 64 + lane < C::CPW_A) glds16(abase + is.aoff[i], __builtin_amdgcn_readfirstlane(slot + (lw * C::CPW_A + i * 64) * 16));

  // rows in window order (proj epilogue; LayerNorm backward behind the qkv dgrad): the tile's window
  bool win = false;
  if constexpr (EP == EP_PROJ_RES) win = true;
  if constexpr (EP == EP_LNBWD) win = p.ln_rows_window != 0;
  TileGeom tg = {0, 0, 0, 0, 0};
  if (win) {
    const int b_ = m0 >> 6;
    int w, wx;
    const int b = fdiv24(b_, p.geom.nW, is.r_nW, w);
    const int wy = fdiv24(w, p.geom.nWw, is.r_nWw, wx);
    tg.b = b;
    tg.tokbase = b * p.geom.H * p.geom.W;
    tg.ybase = wy * 8 + p.geom.shift;
    tg.xbase = wx * 8 + p.geom.shift;
    tg.pbase = m0 & 63;
  }
  auto row_yx = [&](int row, int& y, int& x) {      // window-ordered tile row -> image coordinates
    const int pp = tg.pbase + row;
    y = tg.ybase + (pp >> 3);
    x = tg.xbase + (pp & 7);
    if (y >= p.geom.H) y -= p.geom.H;
    if (x >= p.geom.W) x -= p.geom.W;
  };

  unsigned off = slot + C::A_BYTES;
  if constexpr (C::NE32 > 0) {
#pragma unroll
    for (int i = 0; i < C::NI_E32; ++i) {
      if (i * 64 + lane < C::CPW_E32) {
        long long t = m0 + is.erow[i];
        if (win) {
          int y, x;
          row_yx(is.erow[i], y, x);
          t = tg.tokbase + y * p.geom.W + x;
        }
        const long long eo = t * p.ldo + is.eoff[i];
#pragma unroll
        for (int e = 0; e < C::NE32; ++e) {
          const float* base = (EP == EP_LNBWD) ? (e == 0 ? p.ln_x : p.outf) : p.res;
          glds16(base + eo, __builtin_amdgcn_readfirstlane(off + e * C::E32_BYTES + (lw * C::CPW_E32 + i * 64) * 16));
        }
      }
    }
    off += C::NE32 * C::E32_BYTES;
  }
  if constexpr (C::NE16 > 0) {
    const bf16_t* hbase = p.aux + (long long)m0 * p.ldo;
#pragma unroll
    for (int i = 0; i < C::NI_E16; ++i)
      if (i * 64 + lane < C::CPW_E16) glds16(hbase + is.hoff[i], __builtin_amdgcn_readfirstlane(off + (lw * C::CPW_E16 + i * 64) * 16));
    off += C::E16_BYTES;
  }
  if constexpr (C::AUX) {
    // array lw of the slot: LNBWD {mean, rstd, rowscale, -}; residual epilogues {rowscale, -, -, -}; a "-" (or a null
    // rowscale) still issues one harmless load so that every loader wave retires the same number of DMAs per tile
    if (lane < BM) {
      const int m = m0 + lane;
      int t = m, y = 0, x = 0;
      if (win) {
        row_yx(lane, y, x);
        t = tg.tokbase + y * p.geom.W + x;
      }
      // the tile lies inside one sample (rows_per_sample % BM == 0, checked by the launcher)
      int srem;
      const int samp = p.rowscale ? fdiv24(win ? tg.tokbase : m0, p.rows_per_sample, is.r_rps, srem) : 0;
      const float* src = reinterpret_cast<const float*>(p.Wt) + lane;
      if constexpr (EP == EP_LNBWD) {
        const int st = p.ln_stats_by_m ? m : t;
        if (lw == 0) src = p.ln_mean + st;
        if (lw == 1) src = p.ln_rstd + st;
        if (lw == 2 && p.rowscale) src = p.rowscale + samp;
      } else {
        if (lw == 0 && p.rowscale) src = p.rowscale + samp;
      }
      glds4(src, __builtin_amdgcn_readfirstlane(off + lw * 256));
      // row maps of the tile (consumers would otherwise redo these integer divisions per row-quad): tok = raster token of
      // tile row `lane`, ro = row of the bf16 side output (fused LayerNorm output / windowed gradient copy)
      if (lw == 3) {
        int ro = t;
        bool to_win = false;
        if constexpr (EP == EP_LNBWD) to_win = p.ln_out_window != 0;
        else to_win = p.xn_out != nullptr && p.xn_window != 0;
        if (to_win) {
          const WinGeom& og = (EP == EP_LNBWD) ? p.geom : p.xn_geom;
          if (win) {
            // image coordinates are already known: only the target frame's shift and window index remain
            int yy = y - og.shift, xx = x - og.shift;
            if (yy < 0) yy += og.H;
            if (xx < 0) xx += og.W;
            ro = ((tg.b * og.nW + (yy >> 3) * og.nWw + (xx >> 3)) << 6) | ((yy & 7) << 3) | (xx & 7);
          } else {
            // token_to_win_row(og, t) with the two divisions by launch constants done through their reciprocals
            int rem, xx;
            const int bb = fdiv24(t, og.H * og.W, is.r_ohw, rem);
            int yy = fdiv24(rem, og.W, is.r_oW, xx) - og.shift;
            xx -= og.shift;
            if (yy < 0) yy += og.H;
            if (xx < 0) xx += og.W;
            ro = ((bb * og.nW + (yy >> 3) * og.nWw + (xx >> 3)) << 6) | ((yy & 7) << 3) | (xx & 7);
          }
        }
        int* maps = reinterpret_cast<int*>(smem_ptr + (off - smem_base) + 4 * 256);
        maps[lane] = t;
        maps[64 + lane] = ro;
      }
    }
  }
}

// ---- per-lane state of the row-major epilogue (lane j16 of a 16-lane group owns columns 64c + 4 j16 .. +3) ----------
struct EpState {
  float4 bias[NC];
  float gm[NC][4], cg[NC][4], cb[NC][4];   // EP_LNBWD: gamma, dgamma / dbeta partials
  float lg[NC][4], lb[NC][4];              // fused forward LayerNorm: gamma / beta
  float invC;
  bool has_scale;
};

#define EP_STATE_REFS(st)                       \
  float4(&bias)[NC] = st.bias;                  \
  float(&gm)[NC][4] = st.gm;                    \
  float(&cg)[NC][4] = st.cg;                    \
  float(&cb)[NC][4] = st.cb;                    \
  float(&lg)[NC][4] = st.lg;                    \
  float(&lb)[NC][4] = st.lb;                    \
  float& invC = st.invC;                        \
  bool& has_scale = st.has_scale;               \
  (void)bias; (void)gm; (void)cg; (void)cb; (void)lg; (void)lb; (void)invC; (void)has_scale

template <int EP>
__device__ __forceinline__ void ep_init(const GemmParams& p, int n0, int j16, EpState& st) {
  EP_STATE_REFS(st);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    bias[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EP != EP_DGELU && EP != EP_LNBWD) {
      if (p.bias) bias[c] = *reinterpret_cast<const float4*>(p.bias + n0 + 64 * c + 4 * j16);
    }
  }
  if constexpr (EP == EP_LNBWD) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 64 * c + 4 * j16 + e;
        gm[c][e] = n < p.ln_C ? p.ln_gamma[n] : 0.f;
        cg[c][e] = 0.f;
        cb[c][e] = 0.f;
      }
  }
  invC = EP == EP_LNBWD ? 1.0f / (float)p.ln_C : 0.f;
  if constexpr (EP == EP_PROJ_RES || EP == EP_RES) {
    if (p.xn_out) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 64 * c + 4 * j16 + e;
          lg[c][e] = n < p.xn_C ? p.xn_gamma[n] : 0.f;
          lb[c][e] = n < p.xn_C ? p.xn_beta[n] : 0.f;
        }
    }
  }
  has_scale = p.rowscale != nullptr;
}

// row-major epilogue of one tile: T = accumulators [BM][SBNP], slot = the tile's LDS slot (A image, row operands, row
// scalars, row maps); the calling wave (0..3) handles rows 16 ps + 4 wave + (lane >> 4)
template <int EP, int KC, int BM, int NB = 4>
__device__ __forceinline__ void stream_epilogue_tile(const GemmParams& p, const float* T, const unsigned char* slot, int m0, int n0,
                                                     int wave, int lane, EpState& st) {
  using C = StreamCfg<EP, KC, BM>;
  constexpr int MF = BM / (4 * NB);          // passes: NB waves x 4 rows per pass
  EP_STATE_REFS(st);
  const int sub = lane >> 4, j16 = lane & 15;
  // ---- row-major epilogue: 16 lanes per row, lane j16 holds columns 64c + 4 j16 .. +3 -----------------
  const unsigned char* e0 = slot + C::A_BYTES;
  const float* auxf = reinterpret_cast<const float*>(slot + C::A_BYTES + C::NE32 * C::E32_BYTES + C::NE16 * C::E16_BYTES);
  const int* maps = reinterpret_cast<const int*>(auxf + 4 * 64);
#pragma unroll
  for (int ps = 0; ps < MF; ++ps) {
    const int lr = ps * (4 * NB) + wave * 4 + sub;
    const int m = m0 + lr;
    float4 v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      v[c] = *reinterpret_cast<const float4*>(T + lr * SBNP + 64 * c + 4 * j16);
      v[c].x += bias[c].x; v[c].y += bias[c].y; v[c].z += bias[c].z; v[c].w += bias[c].w;
    }
    if constexpr (EP == EP_BF16) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        st_u2(p.outb + (long long)m * p.ldo + n0 + 64 * c + 4 * j16, pack_bf4(v[c].x, v[c].y, v[c].z, v[c].w));
    } else if constexpr (EP == EP_QKV) {
      const long long b_ = m >> 6;
      const int tok = m & 63;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int n = n0 + 64 * c + 4 * j16;
        const int which = n / p.CA, rem = n - which * p.CA;
        const int h = rem >> 5, d = rem & 31;
        const float s = which == 0 ? p.scale : 1.0f;
        bf16_t* dst = p.outb + ((((long long)which * p.B_ + b_) * p.nH + h) * 64 + tok) * 32 + d;
        *reinterpret_cast<uint2*>(dst) = pack_bf4(v[c].x * s, v[c].y * s, v[c].z * s, v[c].w * s);
      }
    } else if constexpr (EP == EP_GELU) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const long long o = (long long)m * p.ldo + n0 + 64 * c + 4 * j16;
        const uint2 pu = pack_bf4(v[c].x, v[c].y, v[c].z, v[c].w);
        if (p.outb) {                          // the pre-activation is only kept for a backward pass (inference callers pass null)
          if constexpr (SRK_NT_STORE_U != 0)   // u is next read by the backward pass: keep it out of the caches
            __builtin_nontemporal_store(srk_u2{pu.x, pu.y}, reinterpret_cast<srk_u2*>(p.outb + o));
          else
            st_u2(p.outb + o, pu);
        }
        st_u2(p.outb2 + o, gelu_pack4(v[c].x, v[c].y, v[c].z, v[c].w));
      }
    } else if constexpr (EP == EP_DGELU) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const uint2 ua = *reinterpret_cast<const uint2*>(e0 + lr * (SBN * 2) + (64 * c + 4 * j16) * 2);
        float u0, u1, u2, u3;
        unpack_bf2(ua.x, u0, u1);
        unpack_bf2(ua.y, u2, u3);
        st_u2(p.outb + (long long)m * p.ldo + n0 + 64 * c + 4 * j16,  dgelu_mul_pack4(v[c].x, v[c].y, v[c].z, v[c].w, u0, u1, u2, u3));
      }
    } else if constexpr (EP == EP_PROJ_RES || EP == EP_RES) {
      const long long t_ = maps[lr];
      const float f = has_scale ? auxf[lr] : 1.0f;
      float4 o[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 rv = *reinterpret_cast<const float4*>(e0 + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
        o[c] = make_float4(rv.x + v[c].x * f, rv.y + v[c].y * f, rv.z + v[c].z * f, rv.w + v[c].w * f);
        st_f4(p.outf + t_ * p.ldo + n0 + 64 * c + 4 * j16, o[c]);
        if constexpr (EP == EP_RES) {
          if (p.outb) st_u2(p.outb + t_ * p.ldo + n0 + 64 * c + 4 * j16, pack_bf4(o[c].x, o[c].y, o[c].z, o[c].w));
        }
      }
      if (p.xn_out) fused_ln_row_at<NC>(p, o, maps[64 + lr], j16, lg, lb);
    } else if constexpr (EP == EP_LNBWD) {
      const long long t_ = maps[lr];
      const float mean = auxf[lr], rstd = auxf[64 + lr];
      const float f = has_scale ? auxf[128 + lr] : 1.0f;
      float xh[NC][4], dy[NC][4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 xv = *reinterpret_cast<const float4*>(e0 + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float dv[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dy[c][e] = dv[e];
          xh[c][e] = 64 * c + 4 * j16 + e < p.ln_C ? (xs[e] - mean) * rstd : 0.f;
          const float dg = dv[e] * gm[c][e];
          s1 += dg;
          s2 += dg * xh[c][e];
        }
      }
      s1 = wave_sum16(s1) * invC;
      s2 = wave_sum16(s2) * invC;
      const long long ro = maps[64 + lr];
      float* dst = p.ln_skip ? p.ln_skip : p.outf;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 old = *reinterpret_cast<const float4*>(e0 + C::E32_BYTES + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
        float o[4] = {old.x, old.y, old.z, old.w};
        if (p.ln_skip) {             // the layer's skip gradient: a plain load (one launch per layer; no registers to prefetch it into)
          const float4 sk = *reinterpret_cast<const float4*>(p.ln_skip + t_ * p.ldo + 64 * c + 4 * j16);
          o[0] += sk.x; o[1] += sk.y; o[2] += sk.z; o[3] += sk.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (64 * c + 4 * j16 + e < p.ln_C) {
            o[e] += rstd * (dy[c][e] * gm[c][e] - s1 - xh[c][e] * s2);
            cg[c][e] += dy[c][e] * xh[c][e];
            cb[c][e] += dy[c][e];
          }
        }
        st_f4(dst + t_ * p.ldo + 64 * c + 4 * j16, make_float4(o[0], o[1], o[2], o[3]));
        if (p.outb)
          st_u2(p.outb + ro * p.ldo + 64 * c + 4 * j16, pack_bf4(o[0] * f, o[1] * f, o[2] * f, o[3] * f));
      }
    }
  }
}

// ---- the kernel --------------------------------------------------------------------------------
template <int EP, int KC, int BM, bool KS2>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const GemmParams p, int nchunk, int groups_per_xcd) {
  using C = StreamCfg<EP, KC, BM>;
  constexpr int K = C::K, R = C::R, MF = BM / 16;
  constexpr int KST = KS2 ? K / 64 : K / 32;          // 32-wide k-steps per MFMA wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* T = reinterpret_cast<float*>(smem);
  float* colred = reinterpret_cast<float*>(smem + C::T_BYTES);
  const unsigned smem_base = (unsigned)(size_t)smem;            // LDS byte address of the dynamic segment
  constexpr int SLOTS_OFF = C::T_BYTES + C::RED_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;

  // workgroup -> (W slice, tile list): siblings of one tile list (the slices that re-read the same A rows) sit on
  // the same XCD (blockIdx % 8) and walk the tiles in lock step, so A comes from HBM once and from that L2 after
  const int xcd = blockIdx.x & 7, sx = blockIdx.x >> 3;
  const int chunk = sx % nchunk, gx = sx / nchunk;
  if (gx >= groups_per_xcd) return;
  const int n0 = chunk * SBN;
  const int Gm = 8 * groups_per_xcd, gi = gx * 8 + xcd;
  const int ntm = p.M / BM;
  const int nt = gi < ntm ? (ntm - gi + Gm - 1) / Gm : 0;
  if (nt == 0) return;

  const bool mfma_wave = KS2 || wave < 4;
  const int wn = wave & 3, kh = KS2 ? (wave >> 2) : 0;

  // ---- W slice -> registers ---------------------------------------------------------------------
  bf16x8_t wf[3][KST];
  if (mfma_wave) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < KST; ++s)
        wf[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wt + (long long)(n0 + wn * 48 + 16 * j + r16) * K + (kh * KST + s) * 32 + g * 8);
    // make the compiler retire these loads HERE: left alone it waits for them at their first use inside the tile
    // loop, and that s_waitcnt vmcnt(0) would drain the loaders' DMA ring on every iteration
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < KST; ++s) asm volatile("" ::"v"(wf[j][s]));
  }
  if constexpr (EP == EP_LNBWD) {
    for (int i = tid; i < 2 * SBN; i += 512) colred[i] = 0.f;
  }

  if (wave >= 4) {
    // =================================== loader waves =============================================
    const int lw = wave - 4;
    IssueState<EP, KC, BM> is;
    issue_init<EP, KC, BM>(p, n0, lw, lane, is);
    for (int s = 0; s < R - 1 && s < nt; ++s)
      stream_issue_tile<EP, KC, BM>(p, is, (gi + s * Gm) * BM, n0, smem_base + SLOTS_OFF + s * C::SLOT, lw, lane, smem, smem_base);
    for (int t = 0; t < nt; ++t) {
      // tile t has landed once at most the (R-2) tiles issued after it are outstanding
      if (t + R - 2 < nt) wait_vmcnt<C::P*(R - 2)>(); else wait_vmcnt<0>();
      lds_barrier();                                                                 // B1
      // KS2: the upper-half MFMA is on the tile's critical path (the consumers wait for it at Bm), the DMA issue
      // (~1.3 us of address arithmetic per tile) is not: it runs behind B2, beside the consumers' epilogue
      if constexpr (!KS2) {
        if (t + R - 1 < nt)
          stream_issue_tile<EP, KC, BM>(p, is, (gi + (t + R - 1) * Gm) * BM, n0, smem_base + SLOTS_OFF + ((t + R - 1) % R) * C::SLOT, lw, lane, smem, smem_base);
      }
      if constexpr (KS2) {
        const unsigned char* As = smem + SLOTS_OFF + (t % R) * C::SLOT;
        f32x4_t acc[MF][3];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
          for (int i = 0; i < MF; ++i) {
            const int row = 16 * i + r16;
            const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(As + row * (K * 2) + ((((KST + s) * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[i][j], 0, 0, 0);
          }
        lds_barrier();                                                               // Bm: lower half is in T
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float4* tp = reinterpret_cast<float4*>(T + (16 * i + r16) * SBNP + wn * 48 + 16 * j + 4 * g);
            float4 o = *tp;
            o.x += acc[i][j][0]; o.y += acc[i][j][1]; o.z += acc[i][j][2]; o.w += acc[i][j][3];
            *tp = o;
          }
      }
      lds_barrier();                                                                 // B2
      if constexpr (KS2) {
        if (t + R - 1 < nt)
          stream_issue_tile<EP, KC, BM>(p, is, (gi + (t + R - 1) * Gm) * BM, n0, smem_base + SLOTS_OFF + ((t + R - 1) % R) * C::SLOT, lw, lane, smem, smem_base);
      }
    }
  } else {
    // =================================== consumer waves ===========================================
    EpState st;
    ep_init<EP>(p, n0, r16, st);

    for (int t = 0; t < nt; ++t) {
      const int m0 = (gi + t * Gm) * BM;
      const unsigned char* slot = smem + SLOTS_OFF + (t % R) * C::SLOT;
      lds_barrier();                                                                 // B1: tile t is in its slot
      {
        f32x4_t acc[MF][3];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
          for (int i = 0; i < MF; ++i) {
            const int row = 16 * i + r16;
            const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(slot + row * (K * 2) + (((s * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[i][j], 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            *reinterpret_cast<float4*>(T + (16 * i + r16) * SBNP + wn * 48 + 16 * j + 4 * g) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
      if constexpr (KS2) lds_barrier();                                              // Bm
      lds_barrier();                                                                 // B2: T is complete

      stream_epilogue_tile<EP, KC, BM>(p, T, slot, m0, n0, wave, lane, st);
    }
    if constexpr (EP == EP_LNBWD) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(&colred[64 * c + 4 * r16 + e], st.cg[c][e]);
          atomicAdd(&colred[SBN + 64 * c + 4 * r16 + e], st.cb[c][e]);
        }
    }
  }
  if constexpr (EP == EP_LNBWD) {
    // dgamma / dbeta: one global atomic per column per workgroup (partials of all its tiles)
    lds_barrier();
    for (int n = tid; n < p.ln_C; n += 512) {
      atomicAdd(p.ln_dgamma + n, colred[n]);
      atomicAdd(p.ln_dbeta + n, colred[SBN + n]);
    }
  }
}

// ---- role-split variant: MFMA on the loader waves, epilogue on the consumer waves, ONE barrier per tile ---------------
// Iteration i: the front waves (4-7: loaders, each also owns a 48-column slice of W in registers) run the MFMA of tile i
// into T[i & 1] while the back waves (0-3) run the row epilogue of tile i-1 from T[(i-1) & 1]; the MFMA pipe and the
// epilogue's VALU / LDS / store work then overlap on every SIMD instead of alternating.  A slot stays busy one iteration
// longer (until its epilogue is done), so the ring keeps R-2 tiles in flight.
template <int EP, int KC, int BM>
struct SplitCfg {
  using C = StreamCfg<EP, KC, BM>;
  static constexpr int R_LDS = (LDS_BUDGET - 2 * C::T_BYTES - C::RED_BYTES) / C::SLOT;
  static constexpr int R_CNT = 3 + 63 / C::P;
  static constexpr int R0 = R_LDS < R_CNT ? R_LDS : R_CNT;
  static constexpr int R = R0 > 8 ? 8 : (R0 < 3 ? 3 : R0);
  static constexpr bool VALID = R0 >= 3 && KC <= 6;        // K = 576 would need 216 VGPRs of W per front wave
  static constexpr int LDS = 2 * C::T_BYTES + C::RED_BYTES + R * C::SLOT;
  static_assert(C::P * (R - 3) <= 63, "vmcnt overflow");
};

template <int EP, int KC, int BM, int NB>
__global__ __launch_bounds__(64 * (NB + 4)) void gemm_stream_split_kernel(const GemmParams p, int nchunk, int groups_per_xcd) {
  using C = StreamCfg<EP, KC, BM>;
  using S = SplitCfg<EP, KC, BM>;
  constexpr int K = C::K, R = S::R, MF = BM / 16, KST = K / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* colred = reinterpret_cast<float*>(smem + 2 * C::T_BYTES);
  const unsigned smem_base = (unsigned)(size_t)smem;
  constexpr int SLOTS_OFF = 2 * C::T_BYTES + C::RED_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, sx = blockIdx.x >> 3;
  const int chunk = sx % nchunk, gx = sx / nchunk;
  if (gx >= groups_per_xcd) return;
  const int n0 = chunk * SBN;
  const int Gm = 8 * groups_per_xcd, gi = gx * 8 + xcd;
  const int ntm = p.M / BM;
  const int nt = gi < ntm ? (ntm - gi + Gm - 1) / Gm : 0;
  if (nt == 0) return;
  if constexpr (EP == EP_LNBWD) {
    for (int i = tid; i < 2 * SBN; i += 64 * (NB + 4)) colred[i] = 0.f;
  }

  if (wave >= NB) {
    // =================================== front waves: DMA + MFMA ===================================
    const int lw = wave - NB, wn = lw;
    bf16x8_t wf[3][KST];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < KST; ++s)
        wf[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wt + (long long)(n0 + wn * 48 + 16 * j + r16) * K + s * 32 + g * 8);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < KST; ++s) asm volatile("" ::"v"(wf[j][s]));     // retire the loads before the DMA ring starts

    IssueState<EP, KC, BM> is;
    issue_init<EP, KC, BM>(p, n0, lw, lane, is);
    for (int s = 0; s < R - 2 && s < nt; ++s)
      stream_issue_tile<EP, KC, BM>(p, is, (gi + s * Gm) * BM, n0, smem_base + SLOTS_OFF + s * C::SLOT, lw, lane, smem, smem_base);
    for (int i = 0; i < nt; ++i) {
      // this wave's DMAs of tile i have landed once at most the R-3 tiles issued after it are outstanding
      if (i + R - 3 < nt) wait_vmcnt<C::P*(R - 3)>(); else wait_vmcnt<0>();
      lds_barrier();                        // barrier(i): tile i complete in LDS; epilogue(i-2) done -> its slot and T are free
      if (i + R - 2 < nt)
        stream_issue_tile<EP, KC, BM>(p, is, (gi + (i + R - 2) * Gm) * BM, n0, smem_base + SLOTS_OFF + ((i + R - 2) % R) * C::SLOT, lw, lane, smem,
                                      smem_base);
      const unsigned char* slot = smem + SLOTS_OFF + (i % R) * C::SLOT;
      float* T = reinterpret_cast<float*>(smem + (i & 1) * C::T_BYTES);
      f32x4_t acc[MF][3];
#pragma unroll
      for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[a][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KST; ++s)
#pragma unroll
        for (int a = 0; a < MF; ++a) {
          const int row = 16 * a + r16;
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(slot + row * (K * 2) + (((s * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[a][j], 0, 0, 0);
        }
#pragma unroll
      for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          *reinterpret_cast<float4*>(T + (16 * a + r16) * SBNP + wn * 48 + 16 * j + 4 * g) =
              make_float4(acc[a][j][0], acc[a][j][1], acc[a][j][2], acc[a][j][3]);
    }
    lds_barrier();                          // barrier(nt)
  } else {
    // =================================== back waves: row epilogue ==================================
    EpState st;
    ep_init<EP>(p, n0, r16, st);
    for (int i = 0; i <= nt; ++i) {
      lds_barrier();                        // barrier(i): T[(i-1) & 1] holds tile i-1
      if (i > 0) {
        const int t = i - 1;
        stream_epilogue_tile<EP, KC, BM, NB>(p, reinterpret_cast<const float*>(smem + (t & 1) * C::T_BYTES),
                                         smem + SLOTS_OFF + (t % R) * C::SLOT, (gi + t * Gm) * BM, n0, wave, lane, st);
      }
    }
    if constexpr (EP == EP_LNBWD) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(&colred[64 * c + 4 * r16 + e], st.cg[c][e]);
          atomicAdd(&colred[SBN + 64 * c + 4 * r16 + e], st.cb[c][e]);
        }
    }
  }
  if constexpr (EP == EP_LNBWD) {
    lds_barrier();
    for (int n = tid; n < p.ln_C; n += 64 * (NB + 4)) {
      atomicAdd(p.ln_dgamma + n, colred[n]);
      atomicAdd(p.ln_dbeta + n, colred[SBN + n]);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Fused MLP forward: x_out = x1 + f * (gelu(xn2 W1^T + b1) W2^T + b2)  [+ bf16 copy, + the next LayerNorm]
// (network_swinir.py:25-28 Mlp.forward and the second residual of SwinTransformerBlock.forward :277)
//
// One persistent 512-thread workgroup per CU walks 16-row tiles.  The two weight matrices live in registers for the whole
// kernel: the FRONT waves (4-7) each hold a 96-column slice of W1 (6 x 6 fragments), the BACK waves (0-3) each a 48-column
// slice of W2 over the full K = 384 (3 x 12 fragments).  Per tile:
//   front  DMA ring (xn2 rows, fp32 residual rows, row scalars / maps -- the EP_RES loader of the streaming GEMM) ->
//          fc1 MFMA -> bias + GELU in registers (MFMA layout) -> u and h = gelu(u) as bf16 into the double-buffered LDS
//          tiles Us / Hs (swizzled A-operand layout of fc2); the front waves never store to global memory, so their counted
//          vmcnt waits see only DMAs;
//   back   fc2 MFMA from Hs -> T2 -> the row-major fc2 epilogue of the streaming GEMM (bias, DropPath factor, residual,
//          fp32 / bf16 stores, fused LayerNorm of the new row) and, when training, the coalesced global stores of the u / h
//          rows from LDS (both are only read by the backward pass: nt stores).
// Two barriers per iteration; in iteration i the front works on tile i while the back finishes tiles i-1 (fc2) and i-2
// (epilogue), so MFMA, GELU VALU work, LDS traffic and the global stores of three tiles overlap on every SIMD.  The hidden
// activations never travel to HBM between fc1 and fc2 (inference: they never leave the CU at all).  Same MFMA order and
// rounding points as the separate fc1 / fc2 kernels: results are bit-identical to them.
// ------------------------------------------------------------------------------------------------
struct MlpCfg {
  using C = StreamCfg<EP_RES, 3, 16>;
  static constexpr int BM = 16, K1 = 192, HP = 384, R = 5;
  static constexpr int T2_OFF = 0;
  static constexpr int B1_OFF = C::T_BYTES;                 // [3][192] fp32: fc2 bias, gamma / beta of the fused LayerNorm (zero-padded)
  static constexpr int UH_BYTES = BM * HP * 2;              // one bf16 [16][384] tile
  static constexpr int US_OFF = B1_OFF + 3 * SBN * 4;       // Us[2], Hs[2]: double-buffered (front writes tile i while the back reads i-1)
  static constexpr int HS_OFF = US_OFF + 2 * UH_BYTES;
  static constexpr int SLOTS_OFF = HS_OFF + 2 * UH_BYTES;
  static constexpr int LDS = SLOTS_OFF + R * C::SLOT;
  static_assert(LDS <= LDS_BUDGET, "fused MLP: LDS budget");
  static_assert(C::P * (R - 4) <= 63, "vmcnt overflow");
};

template <bool DG>        // DG: the u tile carries gelu'(u) for mlp_fused_bwd_kernel<true>
__global__ __launch_bounds__(512) void mlp_fused_fwd_kernel(const GemmParams p, int groups_per_xcd) {
  using M = MlpCfg;
  using C = M::C;
  constexpr int R = M::R, BM = M::BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned smem_base = (unsigned)(size_t)smem;
  float* T2 = reinterpret_cast<float*>(smem + M::T2_OFF);
  float* b1s = reinterpret_cast<float*>(smem + M::B1_OFF);
  unsigned char* Us = smem + M::US_OFF;
  unsigned char* Hs = smem + M::HS_OFF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, gx = blockIdx.x >> 3;
  if (gx >= groups_per_xcd) return;
  const int Gm = 8 * groups_per_xcd, gi = gx * 8 + xcd;
  const int ntm = p.M / BM;
  const int nt = gi < ntm ? (ntm - gi + Gm - 1) / Gm : 0;
  if (nt == 0) return;
  // row vectors of the back waves' epilogue live in LDS (W2 fills their registers): fc2 bias | LayerNorm gamma | beta
  for (int i = tid; i < 3 * SBN; i += 512) {
    const int which = i / SBN, n = i - which * SBN;
    float v = 0.f;
    if (which == 0) v = p.bias2 ? p.bias2[n] : 0.f;
    else if (p.xn_out && n < p.xn_C) v = which == 1 ? p.xn_gamma[n] : p.xn_beta[n];
    b1s[i] = v;
  }
  __syncthreads();

  if (wave >= 4) {
    // =================================== front waves: DMA + fc1 + GELU ===============================
    const int lw = wave - 4;
    bf16x8_t wf[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int s = 0; s < 6; ++s)
        wf[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wt + (long long)(lw * 96 + 16 * j + r16) * M::K1 + s * 32 + g * 8);
    float4 b1[6];                                                       // fc1 bias of this lane's columns
#pragma unroll
    for (int j = 0; j < 6; ++j) b1[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + lw * 96 + 16 * j + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
      for (int s = 0; s < 6; ++s) asm volatile("" ::"v"(wf[j][s]));     // retire the loads before the DMA ring starts
      asm volatile("" ::"v"(b1[j].x), "v"(b1[j].y), "v"(b1[j].z), "v"(b1[j].w));
    }

    IssueState<EP_RES, 3, BM> is;
    issue_init<EP_RES, 3, BM>(p, 0, lw, lane, is);
    for (int s = 0; s < R - 3 && s < nt; ++s)
      stream_issue_tile<EP_RES, 3, BM>(p, is, (gi + s * Gm) * BM, 0, smem_base + M::SLOTS_OFF + s * C::SLOT, lw, lane, smem, smem_base);
    for (int i = 0; i < nt + 2; ++i) {
      if (i < nt) {
        // tile i has landed once at most the R-4 tiles issued after it are outstanding
        if (i + R - 4 < nt) wait_vmcnt<C::P*(R - 4)>(); else wait_vmcnt<0>();
      }
      lds_barrier();                                            // A(i)
      if (i < nt) {
        if (i + R - 3 < nt)
          stream_issue_tile<EP_RES, 3, BM>(p, is, (gi + (i + R - 3) * Gm) * BM, 0, smem_base + M::SLOTS_OFF + ((i + R - 3) % R) * C::SLOT, lw,
                                           lane, smem, smem_base);
        const unsigned char* slot = smem + M::SLOTS_OFF + (i % R) * C::SLOT;
        f32x4_t acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(slot + r16 * (M::K1 * 2) + (((s * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[j], 0, 0, 0);
        }
        // lane holds u[row r16][n .. n+3], n = 96 lw + 16 j + 4 g  ->  bias, GELU, bf16 into this tile's Us / Hs
        unsigned char* us = Us + (i & 1) * M::UH_BYTES;
        unsigned char* hs = Hs + (i & 1) * M::UH_BYTES;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 bv = b1[j];
          const float v0 = acc[j][0] + bv.x, v1 = acc[j][1] + bv.y, v2 = acc[j][2] + bv.z, v3 = acc[j][3] + bv.w;
          const int n = lw * 96 + 16 * j + 4 * g;               // 16-byte chunk n / 8, half (n & 4)
          const int off = r16 * (M::HP * 2) + ((((n >> 3) ^ (r16 & 7)) << 4) | ((n & 4) << 1));
          if constexpr (DG) {
            float g0, g1, g2, g3, d0, d1, d2, d3;
            gelu_both(v0, g0, d0); gelu_both(v1, g1, d1); gelu_both(v2, g2, d2); gelu_both(v3, g3, d3);
            *reinterpret_cast<uint2*>(us + off) = pack_bf4(d0, d1, d2, d3);
            *reinterpret_cast<uint2*>(hs + off) = pack_bf4(g0, g1, g2, g3);
          } else {
            *reinterpret_cast<uint2*>(us + off) = pack_bf4(v0, v1, v2, v3);
            *reinterpret_cast<uint2*>(hs + off) = gelu_pack4(v0, v1, v2, v3);
          }
        }
      }
      lds_barrier();                                            // B(i): (back-wave hand-over of T2)
    }
  } else {
    // =================================== back waves: fc2 + row epilogue + stores ======================
    const int wn = wave;
    bf16x8_t w2[3][12];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < 12; ++s)
        w2[j][s] = *reinterpret_cast<const bf16x8_t*>(p.W2 + (long long)(wn * 48 + 16 * j + r16) * M::HP + s * 32 + g * 8);
    const int btid = tid;        // 0..255
    const int sub = lane >> 4, j16 = lane & 15, lr = wave * 4 + sub;     // row epilogue: 16 lanes per row, 4 rows per wave
    const bool has_scale = p.rowscale != nullptr;
    for (int i = 0; i < nt + 2; ++i) {
      lds_barrier();                                            // A(i)
      if (i >= 2) {                                             // fc2 epilogue of tile i-2 (T2 holds it since B(i-1))
        const int t = i - 2;
        const unsigned char* slot = smem + M::SLOTS_OFF + (t % R) * C::SLOT;
        const unsigned char* e0 = slot + C::A_BYTES;                    // fp32 residual rows of the tile
        const float* auxf = reinterpret_cast<const float*>(slot + C::A_BYTES + C::E32_BYTES);
        const int* maps = reinterpret_cast<const int*>(auxf + 4 * 64);
        const long long t_ = maps[lr];
        const float f = has_scale ? auxf[lr] : 1.0f;
        float4 o[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(T2 + lr * SBNP + 64 * c + 4 * j16);
          const float4 bv = *reinterpret_cast<const float4*>(b1s + 64 * c + 4 * j16);
          const float4 rv = *reinterpret_cast<const float4*>(e0 + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
          o[c] = make_float4(rv.x + (v.x + bv.x) * f, rv.y + (v.y + bv.y) * f, rv.z + (v.z + bv.z) * f, rv.w + (v.w + bv.w) * f);
          st_f4(p.outf + t_ * p.ldo + 64 * c + 4 * j16, o[c]);
          if (p.outb) st_u2(p.outb + t_ * p.ldo + 64 * c + 4 * j16, pack_bf4(o[c].x, o[c].y, o[c].z, o[c].w));
        }
        if (p.xn_out) {
          float lg[NC][4], lb[NC][4];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float4 gv = *reinterpret_cast<const float4*>(b1s + SBN + 64 * c + 4 * j16);
            const float4 bb = *reinterpret_cast<const float4*>(b1s + 2 * SBN + 64 * c + 4 * j16);
            lg[c][0] = gv.x; lg[c][1] = gv.y; lg[c][2] = gv.z; lg[c][3] = gv.w;
            lb[c][0] = bb.x; lb[c][1] = bb.y; lb[c][2] = bb.z; lb[c][3] = bb.w;
          }
          fused_ln_row_at<NC>(p, o, maps[64 + lr], j16, lg, lb);
        }
      }
      f32x4_t acc[3];
      const bool have = i >= 1 && i - 1 < nt;
      if (have) {                                               // fc2 of tile i-1 from Hs
        const unsigned char* us = Us + ((i - 1) & 1) * M::UH_BYTES;
        const unsigned char* hs = Hs + ((i - 1) & 1) * M::UH_BYTES;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(hs + r16 * (M::HP * 2) + (((s * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][s], xf, acc[j], 0, 0, 0);
        }
        if (p.u_out) {                                          // training: u / h rows leave the CU once, as 768-byte rows
          const long long m0 = (long long)(gi + (i - 1) * Gm) * BM;
          int bt = btid;
          asm volatile("" : "+v"(bt));                          // recompute the chunk map per tile: W2 owns the registers
#pragma unroll 1
          for (int it = 0; it < 3; ++it) {
            const int idx = it * 256 + bt;                      // 16 rows x 48 chunks
            const int row = idx / 48, c = idx - row * 48;
            const int off = row * (M::HP * 2) + ((c ^ (row & 7)) << 4);
            const uint4 uv = *reinterpret_cast<const uint4*>(us + off);
            const uint4 hv = *reinterpret_cast<const uint4*>(hs + off);
            typedef unsigned srk_u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(srk_u4{uv.x, uv.y, uv.z, uv.w}, reinterpret_cast<srk_u4*>(p.u_out + (m0 + row) * p.HP + c * 8));
            __builtin_nontemporal_store(srk_u4{hv.x, hv.y, hv.z, hv.w}, reinterpret_cast<srk_u4*>(p.h_out + (m0 + row) * p.HP + c * 8));
          }
        }
      }
      lds_barrier();                                            // B(i): T2's readers (epilogue of tile i-2) are done
      if (have) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          *reinterpret_cast<float4*>(T2 + r16 * SBNP + wn * 48 + 16 * j + 4 * g) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused MLP backward (the mirror of mlp_fused_fwd_kernel):
//   dh  = d x2 . W2            d u = dh * gelu'(u)                         (network_swinir.py:25-28 Mlp.forward backwards)
//   dxn = d u . W1             LayerNorm(norm2) backward of dxn on x1 -> gx += ..., bf16 copy (window order, DropPath factor)
// Run separately (EP_DGELU GEMM, then the EP_LNBWD GEMM) d u makes a round trip through HBM (100 MB written, 100 MB read back
// per block at cfg3); here it is written once -- the fc1 weight gradient reads it -- and consumed from LDS.
//
// One persistent 512-thread workgroup per CU walks 16-row tiles; both weight matrices live in registers: the FRONT waves (4-7)
// each hold a 96-column slice of W2^T [384][192] (6 x 6 fragments), the BACK waves (0-3) a 48-column slice of W1^T [192][384]
// over K = 384 (3 x 12 fragments).  The row operands arrive by LDS-DMA through TWO rings, because the tile's operands are
// consumed two iterations apart: the front ring (3 slots: d x2 rows + the u rows, 18 KB) feeds iteration i, the back ring
// (2 slots: x1 rows, gradient-stream rows, row statistics / maps, 25.5 KB) feeds the LayerNorm-backward epilogue in iteration
// i + 2.  The front waves issue both (front tile i + 2 behind barrier A, back tile i behind barrier B) and never store, so
// their counted vmcnt wait -- everything but the youngest front tile and the youngest back tile -- is exact.  Per iteration:
//   front  MFMA of tile i, d u = acc * gelu'(u) in the MFMA register layout (u read from the slot's swizzled image) -> Ds[i & 1]
//   back   LayerNorm-backward row epilogue of tile i - 2 (T2 + its back slot), MFMA of tile i - 1 from Ds, the d u rows of tile
//          i - 1 to global (768-byte rows), accumulators -> T2 behind barrier B
// Same MFMA order and rounding points as the separate kernels.
struct MlpBwdCfg {
  using CB = StreamCfg<EP_LNBWD, 0, 16>;                    // back slot: two fp32 row operands + row scalars / maps, no A image
  static constexpr int BM = 16, K1 = 192, HP = 384, RF = 3, RB = 2;
  static constexpr int A_BYTES = BM * K1 * 2;               // 6 144
  static constexpr int U_BYTES = BM * HP * 2;               // 12 288
  static constexpr int FSLOT = A_BYTES + U_BYTES;           // 18 432
  static constexpr int PF = 2 + 3;                          // DMA instructions per front wave and front tile (A: 96 pieces, u: 192)
  static constexpr int PB = CB::P;                          // ... and back tile (3 + 3 + 1)
  static constexpr int T2_OFF = 0;
  static constexpr int RED_OFF = CB::T_BYTES;               // dgamma / dbeta column partials, one private [2][192] per back wave
  static constexpr int GAM_OFF = RED_OFF + 4 * 2 * SBN * 4; // gamma of norm2, zero-padded [192] (W1 fills the back waves' registers)
  static constexpr int DS_OFF = GAM_OFF + SBN * 4;          // Ds[2]: d u tiles (swizzled A-operand layout of the second GEMM)
  static constexpr int FS_OFF = DS_OFF + 2 * U_BYTES;
  static constexpr int BS_OFF = FS_OFF + RF * FSLOT;
  static constexpr int LDS = BS_OFF + RB * CB::SLOT;
  static_assert(CB::A_BYTES == 0 && CB::NI_A == 0, "back slot carries no A image");
  static_assert(LDS <= LDS_BUDGET, "fused MLP backward: LDS budget");
  static_assert(PF + PB <= 63, "vmcnt overflow");
};

template <bool DG>        // DG: aux holds gelu'(u) (written by mlp_fused_fwd_kernel<true>)
__global__ __launch_bounds__(512) void mlp_fused_bwd_kernel(const GemmParams p, int groups_per_xcd) {
  using M = MlpBwdCfg;
  using CB = M::CB;
  constexpr int BM = M::BM;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned smem_base = (unsigned)(size_t)smem;
  float* T2 = reinterpret_cast<float*>(smem + M::T2_OFF);
  float* colred = reinterpret_cast<float*>(smem + M::RED_OFF);
  unsigned char* Ds = smem + M::DS_OFF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, gx = blockIdx.x >> 3;
  if (gx >= groups_per_xcd) return;
  const int Gm = 8 * groups_per_xcd, gi = gx * 8 + xcd;
  const int ntm = p.M / BM;
  const int nt = gi < ntm ? (ntm - gi + Gm - 1) / Gm : 0;
  if (nt == 0) return;
  float* gam = reinterpret_cast<float*>(smem + M::GAM_OFF);
  for (int i = tid; i < 4 * 2 * SBN; i += 512) colred[i] = 0.f;
  for (int i = tid; i < SBN; i += 512) gam[i] = i < p.ln_C ? p.ln_gamma[i] : 0.f;
  __syncthreads();

  if (wave >= 4) {
    // =================================== front waves: both DMA rings + dh + gelu' ===================
    const int lw = wave - 4;
    bf16x8_t wf[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int s = 0; s < 6; ++s)
        wf[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wt + (long long)(lw * 96 + 16 * j + r16) * M::K1 + s * 32 + g * 8);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int s = 0; s < 6; ++s) asm volatile("" ::"v"(wf[j][s]));     // retire the loads before the DMA rings start

    // front tile: this wave's quarter of the A image (96 16-byte pieces: 2 instructions) and of the u image (192 pieces: 3)
    int aoff[2], uoff[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qq = lw * 96 + i * 64 + lane;
      const int row = qq / 24, pos = qq - row * 24;
      aoff[i] = row * p.lda + ((pos ^ (row & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int qq = lw * 192 + i * 64 + lane;
      const int row = qq / 48, pos = qq - row * 48;
      uoff[i] = row * p.HP + ((pos ^ (row & 7)) << 3);
    }
    auto issue_front = [&](int t) {
      const long long m0 = (long long)(gi + t * Gm) * BM;
      const unsigned slot = smem_base + M::FS_OFF + (t % M::RF) * M::FSLOT;
      const bf16_t* ab = p.A + m0 * p.lda;
      const bf16_t* ub = p.aux + m0 * p.HP;
      glds16(ab + aoff[0], __builtin_amdgcn_readfirstlane(slot + (lw * 96) * 16));
      // (the second A instruction is half empty -- 96 pieces per wave -- but still one instruction on vmcnt)
      if (lane < 32) glds16(ab + aoff[1], __builtin_amdgcn_readfirstlane(slot + (lw * 96 + 64) * 16));
#pragma unroll
      for (int i = 0; i < 3; ++i) glds16(ub + uoff[i], __builtin_amdgcn_readfirstlane(slot + M::A_BYTES + (lw * 192 + i * 64) * 16));
    };
    IssueState<EP_LNBWD, 0, BM> is;
    issue_init<EP_LNBWD, 0, BM>(p, 0, lw, lane, is);
    auto issue_back = [&](int t) {
      stream_issue_tile<EP_LNBWD, 0, BM>(p, is, (gi + t * Gm) * BM, 0, smem_base + M::BS_OFF + (t % M::RB) * CB::SLOT, lw, lane, smem, smem_base);
    };

    for (int s = 0; s < M::RF - 1 && s < nt; ++s) issue_front(s);
    for (int i = 0; i < nt + 2; ++i) {
      // landed before A(i): front tile i and back tile i - 2; younger and still allowed in flight: front tile i + 1, back tile i - 1
      if (i + 1 < nt && i >= 1) wait_vmcnt<M::PF + M::PB>();
      else if (i + 1 < nt) wait_vmcnt<M::PF>();
      else wait_vmcnt<0>();
      lds_barrier();                                            // A(i)
      if (i < nt) {
        if (i + M::RF - 1 < nt) issue_front(i + M::RF - 1);     // its slot held tile i - 1, consumed before B(i - 1)
        const unsigned char* slot = smem + M::FS_OFF + (i % M::RF) * M::FSLOT;
        const unsigned char* uimg = slot + M::A_BYTES;
        f32x4_t acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(slot + r16 * (M::K1 * 2) + (((s * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][s], xf, acc[j], 0, 0, 0);
        }
        // lane holds dh[row r16][n .. n+3], n = 96 lw + 16 j + 4 g  ->  times gelu'(u) -> bf16 into this tile's Ds
        unsigned char* ds = Ds + (i & 1) * M::U_BYTES;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int n = lw * 96 + 16 * j + 4 * g;               // 16-byte chunk n / 8, half (n & 4)
          const int off = r16 * (M::HP * 2) + ((((n >> 3) ^ (r16 & 7)) << 4) | ((n & 4) << 1));
          const uint2 ua = *reinterpret_cast<const uint2*>(uimg + off);
          float u0, u1, u2, u3;
          unpack_bf2(ua.x, u0, u1);
          unpack_bf2(ua.y, u2, u3);
          if constexpr (DG) *reinterpret_cast<uint2*>(ds + off) = pack_bf4(acc[j][0] * u0, acc[j][1] * u1, acc[j][2] * u2, acc[j][3] * u3);
          else *reinterpret_cast<uint2*>(ds + off) = dgelu_mul_pack4(acc[j][0], acc[j][1], acc[j][2], acc[j][3], u0, u1, u2, u3);
        }
      }
      lds_barrier();                                            // B(i)
      if (i < nt) issue_back(i);                                // its slot held tile i - 2, whose epilogue ran before B(i)
    }
  } else {
    // =================================== back waves: d u . W1 + LayerNorm backward + d u stores =====
    const int wn = wave;
    bf16x8_t w2[3][12];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < 12; ++s)
        w2[j][s] = *reinterpret_cast<const bf16x8_t*>(p.W2 + (long long)(wn * 48 + 16 * j + r16) * M::HP + s * 32 + g * 8);
    const int btid = tid;        // 0..255
    const int sub = lane >> 4, j16 = lane & 15, lr = wave * 4 + sub;     // row epilogue: 16 lanes per row, 4 rows per wave
    const float invC = 1.0f / (float)p.ln_C;
    const bool has_scale = p.rowscale != nullptr;
    for (int i = 0; i < nt + 2; ++i) {
      lds_barrier();                                            // A(i)
      if (i >= 2) {                                             // LayerNorm-backward epilogue of tile i-2 (T2 holds it since B(i-1))
        // (the EP_LNBWD row epilogue of stream_epilogue_tile with gamma and the dgamma / dbeta partials in LDS instead of registers)
        const int t = i - 2;
        const unsigned char* slot = smem + M::BS_OFF + (t % M::RB) * CB::SLOT;
        const float* auxf = reinterpret_cast<const float*>(slot + 2 * CB::E32_BYTES);
        const int* maps = reinterpret_cast<const int*>(auxf + 4 * 64);
        const long long t_ = maps[lr], ro = maps[64 + lr];
        const float mean = auxf[lr], rstd = auxf[64 + lr];
        const float f = has_scale ? auxf[128 + lr] : 1.0f;
        float dy[NC][4], xh[NC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 dv = *reinterpret_cast<const float4*>(T2 + lr * SBNP + 64 * c + 4 * j16);
          const float4 xv = *reinterpret_cast<const float4*>(slot + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
          const float4 gv = *reinterpret_cast<const float4*>(gam + 64 * c + 4 * j16);
          const float dvs[4] = {dv.x, dv.y, dv.z, dv.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dy[c][e] = dvs[e];
            xh[c][e] = 64 * c + 4 * j16 + e < p.ln_C ? (xs[e] - mean) * rstd : 0.f;
            const float dg = dvs[e] * gs[e];
            s1 += dg;
            s2 += dg * xh[c][e];
          }
        }
        s1 = wave_sum16(s1) * invC;
        s2 = wave_sum16(s2) * invC;
        // dgamma / dbeta: the wave's four rows are summed across its 16-lane groups (two permlane swaps), then the first group adds
        // them to the wave's private partial in LDS with plain read-modify-write (LDS float atomics -- 24 per lane and tile, four
        // lanes per address -- made this kernel 2.4 x slower than the two kernels it replaces)
        float* myred = colred + wave * (2 * SBN);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 old = *reinterpret_cast<const float4*>(slot + CB::E32_BYTES + lr * (SBN * 4) + (64 * c + 4 * j16) * 4);
          const float4 gv = *reinterpret_cast<const float4*>(gam + 64 * c + 4 * j16);
          const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
          float o[4] = {old.x, old.y, old.z, old.w};
          float pg[4], pb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool live = 64 * c + 4 * j16 + e < p.ln_C;     // xh and gamma are zero in the pad columns; dy need not be
            o[e] += live ? rstd * (dy[c][e] * gs[e] - s1 - xh[c][e] * s2) : 0.f;
            pg[e] = xrow_sum4(dy[c][e] * xh[c][e]);
            pb[e] = xrow_sum4(live ? dy[c][e] : 0.f);
          }
          if (sub == 0) {
            float4* rg = reinterpret_cast<float4*>(myred + 64 * c + 4 * j16);
            float4* rb = reinterpret_cast<float4*>(myred + SBN + 64 * c + 4 * j16);
            const float4 a = *rg, b = *rb;
            *rg = make_float4(a.x + pg[0], a.y + pg[1], a.z + pg[2], a.w + pg[3]);
            *rb = make_float4(b.x + pb[0], b.y + pb[1], b.z + pb[2], b.w + pb[3]);
          }
          st_f4(p.outf + t_ * p.ldo + 64 * c + 4 * j16, make_float4(o[0], o[1], o[2], o[3]));
          if (p.outb) st_u2(p.outb + ro * p.ldo + 64 * c + 4 * j16, pack_bf4(o[0] * f, o[1] * f, o[2] * f, o[3] * f));
        }
      }
      f32x4_t acc[3];
      const bool have = i >= 1 && i - 1 < nt;
      if (have) {                                               // d u . W1 of tile i-1 from Ds
        const unsigned char* ds = Ds + ((i - 1) & 1) * M::U_BYTES;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(ds + r16 * (M::HP * 2) + (((s * 4 + g) ^ (r16 & 7)) << 4));
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][s], xf, acc[j], 0, 0, 0);
        }
        {                                                       // d u rows leave the CU once (the fc1 weight gradient reads them)
          const long long m0 = (long long)(gi + (i - 1) * Gm) * BM;
          int bt = btid;
          asm volatile("" : "+v"(bt));                          // recompute the chunk map per tile: W1 owns the registers
#pragma unroll 1
          for (int it = 0; it < 3; ++it) {
            const int idx = it * 256 + bt;                      // 16 rows x 48 chunks
            const int row = idx / 48, c = idx - row * 48;
            const uint4 dv = *reinterpret_cast<const uint4*>(ds + row * (M::HP * 2) + ((c ^ (row & 7)) << 4));
            *reinterpret_cast<uint4*>(p.u_out + (m0 + row) * p.HP + c * 8) = dv;
          }
        }
      }
      lds_barrier();                                            // B(i): T2's readers (epilogue of tile i-2) are done
      if (have) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          *reinterpret_cast<float4*>(T2 + r16 * SBNP + wn * 48 + 16 * j + 4 * g) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      }
    }
  }
  // dgamma / dbeta: one global atomic per column per workgroup
  lds_barrier();
  for (int n = tid; n < p.ln_C; n += 512) {
    atomicAdd(p.ln_dgamma + n, (colred[n] + colred[2 * SBN + n]) + (colred[4 * SBN + n] + colred[6 * SBN + n]));
    atomicAdd(p.ln_dbeta + n, (colred[SBN + n] + colred[3 * SBN + n]) + (colred[5 * SBN + n] + colred[7 * SBN + n]));
  }
}

SrkOpt g_mlp_fused_enabled{OPT_MLP_FUSED, 1};
SrkOpt g_mlp_bwd_fused_enabled{OPT_MLP_BWD_FUSED, 1};
SrkOpt g_mlp_dgelu_store{OPT_MLP_DGELU_STORE, 1};   // fused MLP pair of the training plan: keep gelu'(u) instead of u between the passes

SrkOpt g_stream_enabled{OPT_GEMM_STREAM, -1};     // -1: read SRK_GEMM_STREAM once
thread_local int g_num_cus = 0;      // CU count of the device the current launch goes to (refreshed by every launcher)
SrkOpt g_tune_bm{OPT_TUNE_BM, 0};             // tuning overrides (srk_set_option): rows per tile 16/32/64, 0 = per-epilogue default
SrkOpt g_tune_ks2{OPT_TUNE_KS2, -1};           // split K over the two wave groups: 0/1, -1 = default
SrkOpt g_tune_split{OPT_TUNE_SPLIT, -1};         // role-split kernel (MFMA on the loader waves): 0/1, -1 = default
SrkOpt g_tune_nb{OPT_TUNE_NB, 0};             // role-split kernel: epilogue waves 4/8, 0 = default

template <typename KernelT>
int stream_configure(KernelT kernel, int lds, int* state) {
  if (*state) return SRK_OK;
  const void* fn = reinterpret_cast<const void*>(kernel);
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, fn) != hipSuccess) {
    srk_set_error("gemm(stream): cannot query the kernel");
    return SRK_E_LAUNCH;
  }
  // a variant that spills would put scratch traffic on the loaders' vmcnt counter: never run it
  if (attr.localSizeBytes > 0) {
    *state = -1;
    return SRK_OK;
  }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
    srk_set_error("gemm(stream): cannot reserve %d bytes of LDS", lds);
    return SRK_E_LAUNCH;
  }
  *state = 1;
  return SRK_OK;
}

template <int EP, int KC, int BM, int NB>
int launch_split(const GemmParams& p, hipStream_t stream) {
  using S = SplitCfg<EP, KC, BM>;
  if constexpr (!S::VALID || BM % (4 * NB) != 0 || (NB == 8 && (KC > 3 || EP == EP_LNBWD))) {   // 12 waves: <= 168 VGPRs
    return SRK_NOT_COVERED;
  } else {
    static SrkPerDevice<int> configured_pd; int& configured = configured_pd.here();     // 1 usable, -1 not usable
    const int rc = stream_configure(&gemm_stream_split_kernel<EP, KC, BM, NB>, S::LDS, &configured);
    if (rc) return rc;
    if (configured < 0) return SRK_NOT_COVERED;
    const int nchunk = p.N / SBN;
    const int per_xcd = g_num_cus / 8;
    srk_probe_pre(FAM_GEMM_LINEAR, stream, p.flops, p.bytes);
    hipLaunchKernelGGL((gemm_stream_split_kernel<EP, KC, BM, NB>), dim3(per_xcd * 8), dim3(64 * (NB + 4)), S::LDS, stream, p, nchunk, per_xcd / nchunk);
    srk_probe_post(FAM_GEMM_LINEAR, stream);
    return srk_check_launch("gemm(stream-split)");
  }
}

template <int EP, int KC, int BM, bool KS2>
int launch_stream(const GemmParams& p, hipStream_t stream) {
  using C = StreamCfg<EP, KC, BM>;
  if constexpr (!C::VALID || (KC == 9 && !KS2)) {
    return SRK_NOT_COVERED;
  } else {
    static SrkPerDevice<int> configured_pd; int& configured = configured_pd.here();     // 1 usable, -1 not usable
    const int rcc = stream_configure(&gemm_stream_kernel<EP, KC, BM, KS2>, C::LDS, &configured);
    if (rcc) return rcc;
    if (configured < 0) return SRK_NOT_COVERED;
    const int nchunk = p.N / SBN;
    const int per_xcd = g_num_cus / 8;
    const int groups_per_xcd = per_xcd / nchunk;
    srk_probe_pre(FAM_GEMM_LINEAR, stream, p.flops, p.bytes);
    hipLaunchKernelGGL((gemm_stream_kernel<EP, KC, BM, KS2>), dim3(per_xcd * 8), dim3(512), C::LDS, stream, p, nchunk, groups_per_xcd);
    srk_probe_post(FAM_GEMM_LINEAR, stream);
    return srk_check_launch("gemm(stream)");
  }
}

template <int EP, int KC>
int launch_cfg(const GemmParams& p, hipStream_t stream, int bm, bool ks2) {
  if (bm == 16) return ks2 ? launch_stream<EP, KC, 16, true>(p, stream) : launch_stream<EP, KC, 16, false>(p, stream);
  if (bm == 32) return ks2 ? launch_stream<EP, KC, 32, true>(p, stream) : launch_stream<EP, KC, 32, false>(p, stream);
  if (bm == 64) return ks2 ? launch_stream<EP, KC, 64, true>(p, stream) : launch_stream<EP, KC, 64, false>(p, stream);
  return SRK_NOT_COVERED;
}

template <int EP, int KC>
int launch_split_bm(const GemmParams& p, hipStream_t stream, int bm, int nb) {
  if (nb == 8) {
    if (bm == 32) return launch_split<EP, KC, 32, 8>(p, stream);
    if (bm == 64) return launch_split<EP, KC, 64, 8>(p, stream);
    return SRK_NOT_COVERED;
  }
  if (bm == 16) return launch_split<EP, KC, 16, 4>(p, stream);
  if (bm == 32) return launch_split<EP, KC, 32, 4>(p, stream);
  if (bm == 64) return launch_split<EP, KC, 64, 4>(p, stream);
  return SRK_NOT_COVERED;
}

// per-(epilogue, K) defaults measured on MI355X (tools/stream_sweep.py); the overrides fall back to them
struct StreamChoice {
  int bm;        // rows per tile
  bool ks2;      // symmetric kernel: split K over the two wave groups
  bool split;    // role-split kernel
  int nb = 4;    // role-split kernel: epilogue (back) waves, 4 or 8
};

template <int EP, int KC>
int pick(const GemmParams& p, hipStream_t stream, StreamChoice def) {
  StreamChoice c = def;
  if (g_tune_bm) c.bm = g_tune_bm;
  if (g_tune_ks2 >= 0) c.ks2 = g_tune_ks2 != 0;
  if (g_tune_split >= 0) c.split = g_tune_split != 0;
  if (g_tune_nb > 0) c.nb = g_tune_nb;
  int rc = c.split ? launch_split_bm<EP, KC>(p, stream, c.bm, c.nb) : launch_cfg<EP, KC>(p, stream, c.bm, c.ks2);
  if (rc == SRK_NOT_COVERED) rc = def.split ? launch_split_bm<EP, KC>(p, stream, def.bm, def.nb) : launch_cfg<EP, KC>(p, stream, def.bm, def.ks2);
  return rc;
}

template <int EP>
int dispatch_k(const GemmParams& p, hipStream_t stream, StreamChoice k192, StreamChoice k384) {
  if (p.K == 192) return pick<EP, 3>(p, stream, k192);
  if (p.K == 384) return pick<EP, 6>(p, stream, k384);
  return SRK_NOT_COVERED;
}

}  // namespace

// 1: use the streaming kernel where it applies (default); 0: always use the tile kernel of gemm.hip
void srk_gemm_stream_enable(int on) { g_stream_enabled = on ? 1 : 0; }
void srk_gemm_stream_tune(int bm, int ks2, int split, int nb) {
  g_tune_bm = bm;
  g_tune_ks2 = ks2;
  g_tune_split = split;
  g_tune_nb = nb;
}
int srk_gemm_stream_enabled() {
  if (g_stream_enabled < 0) {
    const char* e = getenv("SRK_GEMM_STREAM");
    g_stream_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return g_stream_enabled;
}
void srk_gemm_stream_tune_get(int* bm, int* ks2, int* split, int* nb) {
  *bm = g_tune_bm; *ks2 = g_tune_ks2; *split = g_tune_split; *nb = g_tune_nb;
}

// Returns SRK_NOT_COVERED when the streaming kernel does not cover this problem (the caller then uses the tile kernel).
int srk_launch_gemm_stream(int epilogue, const GemmParams& p, hipStream_t stream) {
  if (g_stream_enabled < 0) {
    const char* e = getenv("SRK_GEMM_STREAM");
    g_stream_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!g_stream_enabled) return SRK_NOT_COVERED;
  g_num_cus = srk_device_cus() & ~7;          // of the CURRENT device (cached per device id)
  if (g_num_cus < 8) return SRK_NOT_COVERED;
  if (p.N % SBN != 0 || p.M % 64 != 0 || p.lda % 8 != 0 || p.N / SBN > g_num_cus / 8) return SRK_NOT_COVERED;
  if (p.M < 64 * g_num_cus) return SRK_NOT_COVERED;            // too few tiles to fill the persistent grid
  if (p.rowscale && (p.rows_per_sample <= 0 || p.rows_per_sample % 64 != 0)) return SRK_NOT_COVERED;   // a tile lies inside one sample
  if (p.M >= (1 << 24)) return SRK_NOT_COVERED;                // row / token indices go through fdiv24
  switch (epilogue) {
    case EP_BF16: return dispatch_k<EP_BF16>(p, stream, {32, false, true}, {32, false, true});
    case EP_QKV: return dispatch_k<EP_QKV>(p, stream, {64, false, false}, {32, false, true});   // front-bound when split: 3 slices share A
    case EP_GELU: return dispatch_k<EP_GELU>(p, stream, {32, false, true}, {32, false, true});
    case EP_DGELU: return dispatch_k<EP_DGELU>(p, stream, {32, false, true}, {32, false, true});
    case EP_PROJ_RES: if (p.N != SBN) return SRK_NOT_COVERED; return dispatch_k<EP_PROJ_RES>(p, stream, {16, false, true}, {16, false, true});
    case EP_RES: if (p.N != SBN) return SRK_NOT_COVERED; return dispatch_k<EP_RES>(p, stream, {16, false, true}, {16, false, true});
    case EP_LNBWD:
      if (p.N != SBN) return SRK_NOT_COVERED;
      if (p.K == 576) return pick<EP_LNBWD, 9>(p, stream, {16, true, false});
      return dispatch_k<EP_LNBWD>(p, stream, {16, true, false}, {16, true, false});
    default: return SRK_NOT_COVERED;
  }
}

void srk_mlp_fused_enable(int on) { g_mlp_fused_enabled = on ? 1 : 0; }
int srk_mlp_fused_enabled() { return g_mlp_fused_enabled; }

int srk_launch_mlp_fused(const GemmParams& p, hipStream_t stream) {
  if (!g_mlp_fused_enabled) return SRK_NOT_COVERED;
  if (g_stream_enabled < 0) {
    const char* e = getenv("SRK_GEMM_STREAM");
    g_stream_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!g_stream_enabled) return SRK_NOT_COVERED;
  g_num_cus = srk_device_cus() & ~7;          // of the CURRENT device (cached per device id)
  if (g_num_cus < 8) return SRK_NOT_COVERED;
  if (p.K != MlpCfg::K1 || p.HP != MlpCfg::HP || p.N != SBN || p.lda % 8 != 0 || p.ldo != SBN) return SRK_NOT_COVERED;
  if (p.M % 64 != 0 || p.M < 64 * g_num_cus || p.M >= (1 << 24)) return SRK_NOT_COVERED;
  if (p.rowscale && (p.rows_per_sample <= 0 || p.rows_per_sample % 64 != 0)) return SRK_NOT_COVERED;
  if (!p.A || !p.Wt || !p.W2 || !p.res || !p.outf || (p.u_out != nullptr) != (p.h_out != nullptr)) return SRK_NOT_COVERED;
  if (p.u_dgelu && !p.u_out) return SRK_NOT_COVERED;
  static SrkPerDevice<int> configured_pd[2]; int& configured = configured_pd[p.u_dgelu ? 1 : 0].here();
  const int rc = p.u_dgelu ? stream_configure(&mlp_fused_fwd_kernel<true>, MlpCfg::LDS, &configured)
                           : stream_configure(&mlp_fused_fwd_kernel<false>, MlpCfg::LDS, &configured);
  if (rc) return rc;
  if (configured < 0) return SRK_NOT_COVERED;       // the build spilled: never run it (scratch traffic would break the counted waits)
  srk_probe_pre(FAM_GEMM_LINEAR, stream, p.flops, p.bytes);
  if (p.u_dgelu) hipLaunchKernelGGL(mlp_fused_fwd_kernel<true>, dim3(g_num_cus), dim3(512), MlpCfg::LDS, stream, p, g_num_cus / 8);
  else hipLaunchKernelGGL(mlp_fused_fwd_kernel<false>, dim3(g_num_cus), dim3(512), MlpCfg::LDS, stream, p, g_num_cus / 8);
  srk_probe_post(FAM_GEMM_LINEAR, stream);
  return srk_check_launch("mlp_fused");
}

void srk_mlp_bwd_fused_enable(int on) { g_mlp_bwd_fused_enabled = on ? 1 : 0; }
void srk_mlp_dgelu_store_enable(int on) { g_mlp_dgelu_store = on ? 1 : 0; }
int srk_mlp_dgelu_store_enabled() { return g_mlp_dgelu_store; }
int srk_mlp_bwd_fused_enabled() { return g_mlp_bwd_fused_enabled; }

// A = d x2 bf16 [M][lda], Wt = W2^T [384][192], aux = u [M][HP], u_out = d u [M][HP] (written), W2 = W1^T [192][384]; the
// LayerNorm-backward fields, outf / outb / geom / rowscale as for EP_LNBWD.  SRK_NOT_COVERED -> run the two GEMMs.
int srk_launch_mlp_fused_bwd(const GemmParams& p, hipStream_t stream) {
  if (!g_mlp_bwd_fused_enabled) return SRK_NOT_COVERED;
  if (g_stream_enabled < 0) {
    const char* e = getenv("SRK_GEMM_STREAM");
    g_stream_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!g_stream_enabled) return SRK_NOT_COVERED;
  g_num_cus = srk_device_cus() & ~7;          // of the CURRENT device (cached per device id)
  if (g_num_cus < 8) return SRK_NOT_COVERED;
  if (p.K != MlpBwdCfg::K1 || p.HP != MlpBwdCfg::HP || p.N != SBN || p.lda % 8 != 0 || p.ldo != SBN) return SRK_NOT_COVERED;
  if (p.M % 64 != 0 || p.M < 64 * g_num_cus || p.M >= (1 << 24)) return SRK_NOT_COVERED;
  if (p.rowscale && (p.rows_per_sample <= 0 || p.rows_per_sample % 64 != 0)) return SRK_NOT_COVERED;
  if (!p.A || !p.Wt || !p.W2 || !p.aux || !p.u_out || !p.outf || !p.ln_x || !p.ln_mean || !p.ln_rstd || !p.ln_gamma || !p.ln_dgamma ||
      !p.ln_dbeta || p.ln_rows_window != 0)
    return SRK_NOT_COVERED;
  static SrkPerDevice<int> configured_pd[2]; int& configured = configured_pd[p.u_dgelu ? 1 : 0].here();
  const int rc = p.u_dgelu ? stream_configure(&mlp_fused_bwd_kernel<true>, MlpBwdCfg::LDS, &configured)
                           : stream_configure(&mlp_fused_bwd_kernel<false>, MlpBwdCfg::LDS, &configured);
  if (rc) return rc;
  if (configured < 0) return SRK_NOT_COVERED;       // the build spilled: never run it (scratch traffic would break the counted waits)
  srk_probe_pre(FAM_GEMM_LINEAR, stream, p.flops, p.bytes);
  if (p.u_dgelu) hipLaunchKernelGGL(mlp_fused_bwd_kernel<true>, dim3(g_num_cus), dim3(512), MlpBwdCfg::LDS, stream, p, g_num_cus / 8);
  else hipLaunchKernelGGL(mlp_fused_bwd_kernel<false>, dim3(g_num_cus), dim3(512), MlpBwdCfg::LDS, stream, p, g_num_cus / 8);
  srk_probe_post(FAM_GEMM_LINEAR, stream);
  return srk_check_launch("mlp_fused_bwd");
}
