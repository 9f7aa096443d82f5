// One Swin block of the LIGHT width (SwinIR-light: C <= 64, 6 heads x d <= 16, hidden <= 128) as ONE kernel, inference:
//
//   x1 = x  + proj(softmax(q k^T + bias + mask) v),   q, k, v = qkv(LN1(x)) per (shifted) 8 x 8 window     network_swinir.py:239-276
//   x2 = x1 + fc2(gelu(fc1(LN2(x1))))                                                                    :277, :25-28
//
// At this width everything a window needs fits one workgroup: its 64 residual rows (16 KB fp32), the q / k / v head tiles
// (only the first 16 of the 32 padded head columns are non-zero: 36 KB), the hidden tile (16 KB, on top of the dead head tiles)
// and every weight fragment in registers.  The layer-per-launch path moves each of those tensors through L2 / HBM and pays five
// launches of a few microseconds of work each (cfg2: 83 us per block at 1.4 % of the MFMA peak, latency all the way); here a
// block is one launch, one read and one write of the residual stream ("each Swin block is ONE kernel").
//
// One 256-thread workgroup per window, three workgroups per CU (53 KB of LDS, <= 168 VGPRs): a window is a chain of eight short
// dependent phases, so what fills the CU is independent windows, not more waves per window (512-thread workgroups at two per CU
// needed two rounds for cfg2's 576 windows: 39 us per block against 2 x 19.5 us per workgroup).  The residual rows live in
// registers (row layout: thread = 4 channels of rows tid >> 4 + 16 i); GEMM results in MFMA layout reach them through an fp32
// exchange tile on top of the dead head tiles.  Phases (a barrier between each):
//   0  rows of the window gathered through the roll + window-partition map, LayerNorm1 -> Xn (bf16, swizzled)
//   1  projection: 18 live 16-column fragments (q, k, v x 6 heads; 5 / 5 / 4 / 4 per wave), W in registers, + bias, q scaled
//      -> head tiles [64][16]
//   2  attention: 24 (head, 16-query tile) units, six per wave; q k^T on the 16-deep MFMA (only 16 head columns are live),
//      rel-pos bias rows from the dense table (L2, one head ahead), softmax_numerators, P.V on the 16 live columns -> ao [64][6 x 16]
//   3  proj over the compact K = 96 -> exchange tile E
//   4  x1 = x + E + bias (registers), LayerNorm2 -> Xn
//   5  fc1 + bias + GELU -> H [64][128] (bf16, swizzled)
//   6  fc2 -> exchange tile E2
//   7  x2 = x1 + E2 + bias -> global through the inverse map (256-byte rows), optional bf16 copy for the RSTB conv
// Rounding points are those of the layer-per-launch path (LN outputs, q/k/v, P, attention output, hidden: bf16; residual
// stream fp32).
#include <type_traits>

#include "kernels.h"

#ifdef SRK_PROBE_LIGHT
// developer instrumentation (never in the shipped build): s_memrealtime (100 MHz) at the phase boundaries of two workgroups
__device__ unsigned long long g_light_probe[2 * 4 * 16];
#define LIGHT_MARK(k) do { if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0) g_light_probe[((blockIdx.x ? 1 : 0) * 4 + wave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int srk_debug_light_probe(void* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_light_probe), sizeof(g_light_probe)); }
#else
#define LIGHT_MARK(k) do {} while (0)
#endif

namespace {

constexpr int L_OFF_R1 = 0;            // 12 KB: Xn bf16 [64][64] (chunk ^ (row & 7)) in phases 0-1 and 4-5, ao bf16 [64][96] in phases 2-3
constexpr int L_OFF_R2 = 12288;        // 36 KB: 18 head tiles bf16 [64][16] (1-2); E fp32 [64][64] (3-4); H bf16 [64][128] at +0 (5-6)
constexpr int L_OFF_E2 = L_OFF_R2 + 16384;   //   and E2 fp32 [64][64] at +16 KB (6-7)
constexpr int L_LDS = 49152;           // three workgroups per CU (LDS is granted in blocks: 54.5 KB with a bias table in LDS fitted only two)
constexpr int L_TILE = 64 * 16;        // elements per head tile

struct LightParams {
  const float* x;        // [T][64] fp32 token-major
  float* y;              // [T][64] fp32
  bf16_t* yb;            // [T][64] bf16 copy of y, or null
  const float *n1w, *n1b, *n2w, *n2b;   // [C]
  const bf16_t *Wqkv, *Wproj, *W1, *W2; // packed [576][64], [64][192], [128][64], [64][128]
  const float *bqkv, *bproj, *b1, *b2;  // [576], [64], [128], [64]
  const float* biasd;    // [6][64][64]
  float scale;
  int C;
  long long B_;
  WinGeom geom;
};

__global__ __launch_bounds__(256, 3) void swin_block_light_kernel(const LightParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Xn = reinterpret_cast<bf16_t*>(smem + L_OFF_R1);
  bf16_t* AO = reinterpret_cast<bf16_t*>(smem + L_OFF_R1);
  bf16_t* tiles = reinterpret_cast<bf16_t*>(smem + L_OFF_R2);
  float* E = reinterpret_cast<float*>(smem + L_OFF_R2);
  bf16_t* Hs = reinterpret_cast<bf16_t*>(smem + L_OFF_R2);
  float* E2 = reinterpret_cast<float*>(smem + L_OFF_E2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const long long b_ = blockIdx.x;
  const int C = p.C;
  const float invC = 1.0f / (float)C;

  LIGHT_MARK(0);
  // ---- phase 0: gather + LayerNorm1 ----------------------------------------------------------------------------------
  const int j4 = 4 * (tid & 15), rr = tid >> 4;    // this thread: channels j4 .. j4 + 3 of rows rr + 16 i
  int tok[4];
  float4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    tok[i] = win_row_to_token(p.geom, (int)(b_ * 64) + 16 * i + rr);
    xv[i] = *reinterpret_cast<const float4*>(p.x + (long long)tok[i] * 64 + j4);
  }
  // projection W slice of this wave: fragment f = which * 6 + head (the first 16 columns of the head); 5 / 5 / 4 / 4
  const int nfq = wave < 2 ? 5 : 4, fq0 = wave < 2 ? 5 * wave : 10 + 4 * (wave - 2);
  bf16x8_t wq[5][2];
  float4 bqv[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int f = fq0 + (j < nfq ? j : 0), which = f / 6, hh = f - 6 * which;
    const bf16_t* wr = p.Wqkv + (long long)(which * 192 + hh * 32 + r16) * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) wq[j][s] = *reinterpret_cast<const bf16x8_t*>(wr + s * 32 + g * 8);
    bqv[j] = p.bqkv ? *reinterpret_cast<const float4*>(p.bqkv + which * 192 + hh * 32 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto layer_norm = [&](const float4 (&v)[4], const float* gamma, const float* beta) {
    const float4 gm = make_float4(j4 < C ? gamma[j4] : 0.f, j4 + 1 < C ? gamma[j4 + 1] : 0.f, j4 + 2 < C ? gamma[j4 + 2] : 0.f,
                                  j4 + 3 < C ? gamma[j4 + 3] : 0.f);
    const float4 bt = make_float4(j4 < C ? beta[j4] : 0.f, j4 + 1 < C ? beta[j4 + 1] : 0.f, j4 + 2 < C ? beta[j4 + 2] : 0.f,
                                  j4 + 3 < C ? beta[j4 + 3] : 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * i + rr;
      const float mean = wave_sum16((v[i].x + v[i].y) + (v[i].z + v[i].w)) * invC;      // pad columns are zero
      const float d0 = j4 < C ? v[i].x - mean : 0.f, d1 = j4 + 1 < C ? v[i].y - mean : 0.f;
      const float d2 = j4 + 2 < C ? v[i].z - mean : 0.f, d3 = j4 + 3 < C ? v[i].w - mean : 0.f;
      const float rstd = rsqrtf(wave_sum16((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * invC + 1e-5f);
      *reinterpret_cast<uint2*>(Xn + swz_off(row, j4 >> 3) + (j4 & 4)) =
          pack_bf4(d0 * rstd * gm.x + bt.x, d1 * rstd * gm.y + bt.y, d2 * rstd * gm.z + bt.z, d3 * rstd * gm.w + bt.w);
    }
  };
  LIGHT_MARK(1);
  layer_norm(xv, p.n1w, p.n1b);
  LIGHT_MARK(2);
  srk_lds_barrier();
  LIGHT_MARK(3);

  // ---- phase 1: projection -> head tiles [f][64][16] ------------------------------------------------------------------
#pragma unroll 1
  for (int mq = 0; mq < 4; ++mq) {
    f32x4_t acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int row = 16 * mq + r16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(Xn + swz_off(row, 4 * s + g));
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[j][s], xf, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < nfq) {
        const int f = fq0 + j;
        const float sc = f < 6 ? p.scale : 1.0f;
        *reinterpret_cast<uint2*>(tiles + f * L_TILE + row * 16 + 4 * g) =
            pack_bf4((acc[j][0] + bqv[j].x) * sc, (acc[j][1] + bqv[j].y) * sc, (acc[j][2] + bqv[j].z) * sc, (acc[j][3] + bqv[j].w) * sc);
      }
    }
  }
  // proj W slice (phase 3): in flight under the attention.  This wave owns the token rows 16 wave .. in every GEMM below.
  bf16x8_t wp[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 3; ++s)          // compact k = 32 s + 8 g + i -> head 2 s + (g >> 1), d = 8 (g & 1) + i
      wp[j][s] = *reinterpret_cast<const bf16x8_t*>(p.Wproj + (long long)(16 * j + r16) * 192 + (2 * s + (g >> 1)) * 32 + 8 * (g & 1));
  LIGHT_MARK(4);
  srk_lds_barrier();
  LIGHT_MARK(5);

  // ---- phase 2: attention, query tile it = wave, heads 0..5 ---------------------------------------------------------------
  {
    const int it = wave;
    // rel-pos bias rows of this lane's query (dense [6][64][64], L2-resident): loaded one head ahead of their use
    const float* bl = p.biasd + (16 * it + r16) * 64 + 4 * g;
    auto load_bias = [&](f32x4_t (&bv)[4], int h) {
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const float4 t4 = *reinterpret_cast<const float4*>(bl + h * 4096 + 16 * jt);
        bv[jt] = f32x4_t{t4.x, t4.y, t4.z, t4.w};
      }
    };
    const int w = (int)(b_ % p.geom.nW);
    const int wy = w / p.geom.nWw, wx = w - wy * p.geom.nWw;
    const bool masked = p.geom.shift > 0 && (wy == p.geom.H / 8 - 1 || wx == p.geom.nWw - 1);
    const int ll = lane & 15;
    auto units = [&](auto masked_c) {
      f32x4_t bnext[4];
      load_bias(bnext, 0);
#pragma unroll
      for (int h = 0; h < 6; ++h) {
        const bf16_t* Qs = tiles + (0 + h) * L_TILE;
        const bf16_t* Ks = tiles + (6 + h) * L_TILE;
        const bf16_t* Vs = tiles + (12 + h) * L_TILE;
        // q k^T over the 16 live head columns: the 16-deep MFMA (k = 4 g .. 4 g + 3 per lane), no zero padding to feed
        const bf16x4_t qf = *reinterpret_cast<const bf16x4_t*>(Qs + (16 * it + r16) * 16 + 4 * g);
        f32x4_t s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          s[jt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*reinterpret_cast<const bf16x4_t*>(Ks + (16 * jt + r16) * 16 + 4 * g), qf,
                                                          f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) s[jt] += bnext[jt];
        if (h < 5) load_bias(bnext, h + 1);
        if constexpr (decltype(masked_c)::value) {
          const int labi = win_region_label(p.geom, w, 16 * it + r16);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (win_region_label(p.geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;   // :235 (-100, not -inf)
        }
        bf16x8_t pf[2];
        const float inv = softmax_numerators(s, pf);
        f32x4_t o = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {     // V^T fragment: rows d = 0..15, k = key in accumulator order (as tr_frag_acc)
          const bf16x4_t lo = lds_tr_read(Vs + (32 * ss + 4 * g + (ll >> 2)) * 16 + ((ll & 3) << 2));
          const bf16x4_t hi = lds_tr_read(Vs + (32 * ss + 16 + 4 * g + (ll >> 2)) * 16 + ((ll & 3) << 2));
          const bf16x8_t vf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ss], o, 0, 0, 0);
        }
        o *= inv;
        // O[i = 16 it + r16][d = 4 g + e]; the ao tile lies on Xn, whose last readers (phase 1) are behind the barrier above
        *reinterpret_cast<uint2*>(AO + (16 * it + r16) * 96 + h * 16 + 4 * g) = pack_bf4(o[0], o[1], o[2], o[3]);
      }
    };
    if (masked) units(std::true_type{}); else units(std::false_type{});
  }
  LIGHT_MARK(6);
  srk_lds_barrier();
  LIGHT_MARK(7);

  // ---- phase 3: proj over the compact K = 96 -> E ---------------------------------------------------------------------------
  const int row = 16 * wave + r16;         // this lane's token row in the GEMMs
  {
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(AO + row * 96 + 32 * s + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[j][s], af, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)             // acc[j][e] = out[token row][n = 16 j + 4 g + e]
      *reinterpret_cast<float4*>(E + row * 64 + 16 * j + 4 * g) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  // fc1 W slice (phase 5): in flight under phase 4
  bf16x8_t w1[8][2];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) w1[j][s] = *reinterpret_cast<const bf16x8_t*>(p.W1 + (long long)(16 * j + r16) * 64 + 32 * s + 8 * g);
  LIGHT_MARK(8);
  srk_lds_barrier();
  LIGHT_MARK(9);

  // ---- phase 4: x1 = x + proj + bias, LayerNorm2 -> Xn ------------------------------------------------------------------------
  {
    const float4 bv = p.bproj ? *reinterpret_cast<const float4*>(p.bproj + j4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 e = *reinterpret_cast<const float4*>(E + (16 * i + rr) * 64 + j4);
      xv[i] = make_float4(xv[i].x + e.x + bv.x, xv[i].y + e.y + bv.y, xv[i].z + e.z + bv.z, xv[i].w + e.w + bv.w);
    }
    layer_norm(xv, p.n2w, p.n2b);
  }
  LIGHT_MARK(10);
  srk_lds_barrier();

  // ---- phase 5: fc1 + bias + GELU -> H [64][128] ----------------------------------------------------------------------
  bf16x8_t w2[4][4];                        // fc2 W slice (phase 6): requested after the fc1 MFMAs, in flight under the GELU epilogue
  {
    f32x4_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(Xn + swz_off(row, 4 * s + g));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j][s], xf, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) w2[j][s] = *reinterpret_cast<const bf16x8_t*>(p.W2 + (long long)(16 * j + r16) * 128 + 32 * s + 8 * g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {           // hidden column n = 16 j + 4 g + e -> 16-byte chunk n >> 3 of the 256-byte row
      const int n = 16 * j + 4 * g;
      const float4 bv = p.b1 ? *reinterpret_cast<const float4*>(p.b1 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<uint2*>(Hs + row * 128 + (((n >> 3) ^ (row & 7)) << 3) + (n & 4)) =
          gelu_pack4(acc[j][0] + bv.x, acc[j][1] + bv.y, acc[j][2] + bv.z, acc[j][3] + bv.w);
    }
  }
  LIGHT_MARK(11);
  srk_lds_barrier();
  LIGHT_MARK(12);

  // ---- phase 6: fc2 -> E2 ------------------------------------------------------------------------------------------------
  {
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8_t hf = *reinterpret_cast<const bf16x8_t*>(Hs + row * 128 + (((4 * s + g) ^ (row & 7)) << 3));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[j][s], hf, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(E2 + row * 64 + 16 * j + 4 * g) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  LIGHT_MARK(13);
  srk_lds_barrier();

  // ---- phase 7: x2 = x1 + fc2 + bias, rows back to their tokens (window reverse + un-roll) -------------------------------------
  {
    const float4 bv = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + j4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 e = *reinterpret_cast<const float4*>(E2 + (16 * i + rr) * 64 + j4);
      const float4 v = make_float4(xv[i].x + e.x + bv.x, xv[i].y + e.y + bv.y, xv[i].z + e.z + bv.z, xv[i].w + e.w + bv.w);
      *reinterpret_cast<float4*>(p.y + (long long)tok[i] * 64 + j4) = v;
      if (p.yb) *reinterpret_cast<uint2*>(p.yb + (long long)tok[i] * 64 + j4) = pack_bf4(v.x, v.y, v.z, v.w);
    }
  }
  LIGHT_MARK(14);
}

SrkOpt g_block_light{OPT_BLOCK_LIGHT, 1};

}  // namespace

void srk_block_light_enable(int on) { g_block_light = on ? 1 : 0; }
int srk_block_light_enabled() { return g_block_light; }

// SRK_NOT_COVERED (1) when the shape is not the light one (the caller runs the layer-per-launch path).
int srk_launch_swin_block_light(const float* x, float* y, bf16_t* yb, const float* n1w, const float* n1b, const float* n2w, const float* n2b,
                                const bf16_t* Wqkv, const bf16_t* Wproj, const bf16_t* W1, const bf16_t* W2, const float* bqkv,
                                const float* bproj, const float* b1, const float* b2, const float* biasd, float scale, int C, int CP, int HP,
                                int nH, int dh, int HID, long long B_, WinGeom geom, hipStream_t stream) {
  if (!g_block_light || CP != 64 || HP != 128 || nH != 6 || dh > 16 || C > 64 || B_ < 1 || B_ >= (1LL << 24)) return SRK_NOT_COVERED;
  static SrkPerDevice<int> configured_pd; int& configured = configured_pd.here();
  if (!configured) {
    const void* fn = reinterpret_cast<const void*>(&swin_block_light_kernel);
    hipFuncAttributes attr;
    configured = -1;
    if (hipFuncGetAttributes(&attr, fn) == hipSuccess && attr.localSizeBytes == 0 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS) == hipSuccess)
      configured = 1;
  }
  if (configured < 0) return SRK_NOT_COVERED;
  LightParams p;
  p.x = x; p.y = y; p.yb = yb; p.n1w = n1w; p.n1b = n1b; p.n2w = n2w; p.n2b = n2b; p.Wqkv = Wqkv; p.Wproj = Wproj; p.W1 = W1; p.W2 = W2;
  p.bqkv = bqkv; p.bproj = bproj; p.b1 = b1; p.b2 = b2; p.biasd = biasd; p.scale = scale; p.C = C; p.B_ = B_; p.geom = geom;
  // algorithmic (un-padded) work of one block: qkv + proj + MLP GEMMs and the two attention products; bytes: the residual stream
  // in and out (+ the bf16 copy) and the weights once
  const double T = 64.0 * (double)B_;
  const double flops = T * (2.0 * 3 * C * C + 2.0 * C * C + 4.0 * C * HID + 4.0 * 64 * C);
  const double bytes = T * C * (4.0 + 4.0 + (yb ? 2.0 : 0.0)) + 2.0 * (4.0 * C * C + 2.0 * C * HID);
  srk_probe_pre(FAM_GEMM_LINEAR, stream, flops, bytes);
  hipLaunchKernelGGL(swin_block_light_kernel, dim3((unsigned)B_), dim3(256), L_LDS, stream, p);
  srk_probe_post(FAM_GEMM_LINEAR, stream);
  return srk_check_launch("swin_block_light");
}
