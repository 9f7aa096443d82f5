// (Shifted-)window multi-head attention, forward and backward, for 8x8 windows (N = 64 tokens)
// and head_dim padded to 32.  Reference: WindowAttention.forward network_swinir.py:114-145 (the
// q/k/v projections and the output projection are separate GEMMs; this file is
// softmax(q k^T + bias + mask) v and its gradient).
//
// One wave owns one (window, head) pair at a time and walks `wpw` consecutive windows of the same
// head, keeping that head's dense relative-position bias (64x64 fp32, 64 VGPRs) in registers.
// All matrix products are v_mfma_f32_16x16x32_bf16.  The scores are computed transposed
// (S^T[j][i] = k_j . q_i) so that a softmax row (fixed query i, all keys j) lives in 16 registers
// of 4 lanes: the row max / row sum need two cross-lane steps (xor 16, xor 32), and the normalised
// probabilities are already in MFMA B-operand form (k = key index) for the P.V product.
// Operands that are needed "transposed" (V^T, Q^T, K^T, dO^T, P, dS) are read from small
// wave-private LDS tiles with ds_read_b64_tr_b16.
#include "common.h"
#include "wgrad.h"

namespace {

constexpr int TS = 40;   // LDS row stride (elements) of a [64][32] bf16 tile (80 B: 16-B aligned rows)
constexpr int PS = 72;   // LDS row stride of the [64][64] bf16 P / dS tile (144 B)

__device__ __forceinline__ void stage_tile_64x32(bf16_t* dst, const bf16_t* src, long long row_stride, int lane) {
  // 64 rows x 64 B; lane handles 4 x 16 B
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = lane + 64 * t;
    const int row = idx >> 2, ch = idx & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(src + row * row_stride + ch * 8);
    *reinterpret_cast<uint4*>(dst + row * TS + ch * 8) = v;
  }
}

__device__ __forceinline__ bf16x8_t cat4(bf16x4_t lo, bf16x4_t hi) {
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// transposed fragment: lane (r16, g) gets T[k = k0 + 8g + jj][col c0 + r16], jj = 0..7 (natural k order)
__device__ __forceinline__ bf16x8_t tr_frag_nat(const bf16_t* tile, int stride, int k0, int c0, int lane) {
  const int g = lane >> 4;
  return cat4(lds_tr_read(tr_addr(tile, stride, k0 + 8 * g, c0, lane)),
              lds_tr_read(tr_addr(tile, stride, k0 + 8 * g + 4, c0, lane)));
}

// transposed fragment in "accumulator k order": jj<4 -> k = k0 + 4g + jj ; jj>=4 -> k = k0 + 16 + 4g + (jj-4)
__device__ __forceinline__ bf16x8_t tr_frag_acc(const bf16_t* tile, int stride, int k0, int c0, int lane) {
  const int g = lane >> 4;
  return cat4(lds_tr_read(tr_addr(tile, stride, k0 + 4 * g, c0, lane)),
              lds_tr_read(tr_addr(tile, stride, k0 + 16 + 4 * g, c0, lane)));
}

__device__ __forceinline__ bf16x8_t row_frag(const bf16_t* tile, int stride, int row, int k0) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * stride + k0);
}

__device__ __forceinline__ float xmax4(float v) { return xrow_max4(v); }
__device__ __forceinline__ float xsum4(float v) { return xrow_sum4(v); }

// scores^T + bias + mask -> probabilities^T (in place), T-layout: s[jt][it][e] = S[i=16it+r16][j=16jt+4g+e]
struct BiasRegs {   // dense bias of one head held in 64 VGPRs (forward: reused across the windows of a wave)
  f32x4_t v[4][4];
  __device__ __forceinline__ f32x4_t get(int jt, int it, int, int) const { return v[jt][it]; }
};
struct BiasMem {    // re-read from L2 per window (backward: registers are needed for d(bias))
  const float* base;  // biasd + h*4096
  __device__ __forceinline__ f32x4_t get(int jt, int it, int r16, int g) const {
    const float4 b = *reinterpret_cast<const float4*>(base + (16 * it + r16) * 64 + 16 * jt + 4 * g);
    return f32x4_t{b.x, b.y, b.z, b.w};
  }
};

template <typename Bias>
__device__ __forceinline__ void softmax_T(f32x4_t (&s)[4][4], const Bias& bias, const WinGeom& geom, int w, int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  const int wy = w / geom.nWw, wx = w - wy * geom.nWw;
  const bool masked = geom.shift > 0 && (wy == geom.H / 8 - 1 || wx == geom.nWw - 1);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) s[jt][it] += bias.get(jt, it, r16, g);
  if (masked) {
    int labi[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) labi[it] = win_region_label(geom, w, 16 * it + r16);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int labj = win_region_label(geom, w, 16 * jt + 4 * g + e);
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (labj != labi[it]) s[jt][it][e] += -100.0f;   // network_swinir.py:235 (-100, not -inf)
      }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[jt][it][e]);
    mx = xmax4(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __expf(s[jt][it][e] - mx);
        s[jt][it][e] = pv;
        sum += pv;
      }
    sum = xsum4(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[jt][it][e] *= inv;
  }
}

__device__ __forceinline__ void load_bias_T(f32x4_t (&bias)[4][4], const float* biasd, int h, int lane) {  // -> BiasRegs::v
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float4 b = *reinterpret_cast<const float4*>(biasd + ((h * 64) + 16 * it + r16) * 64 + 16 * jt + 4 * g);
      bias[jt][it] = f32x4_t{b.x, b.y, b.z, b.w};
    }
}

// ------------------------------------------------------------------------------------------------
// forward
#ifndef SRK_NT_ATTN_BWD
#define SRK_NT_ATTN_BWD 1
#endif
typedef unsigned srk_v4u __attribute__((ext_vector_type(4)));
// 16 bytes that this launch reads exactly once (saved q / k / v, the incoming gradient): streaming cache policy
__device__ __forceinline__ uint4 ld_once16(const bf16_t* p) {
  if constexpr (SRK_NT_ATTN_BWD != 0) {
    const srk_v4u v = __builtin_nontemporal_load(reinterpret_cast<const srk_v4u*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}

// ------------------------------------------------------------------------------------------------
// One 4-wave workgroup owns one (window, head) at a time; wave w handles the 16 queries of tile it = w against all
// 64 keys (4 + 4 MFMAs).  K and V are staged once per window in LDS (V is read back transposed); the next window's
// tiles are prefetched into registers.  ~64 VGPRs and 10 KB of LDS per workgroup -> up to 8 workgroups per CU.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ biasd,
                                                       bf16_t* __restrict__ ao, long long B_, int nH, int CA,
                                                       WinGeom geom, int wpw) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * TS], Vs[64 * TS];
  const int tid = threadIdx.x, lane = tid & 63, it = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y;
  const long long w_begin = (long long)blockIdx.x * wpw;
  const int srow = tid >> 2, sch = tid & 3;
  const float* bias_h = biasd + h * 4096;

  uint4 prek = make_uint4(0, 0, 0, 0), prev = prek, preq = prek;
#define ATTN_FWD_PREFETCH(BW)                                                                                  \
  do {                                                                                                         \
    prek = ld_once16(qkv + ((1 * B_ + (BW)) * nH + h) * 2048 + srow * 32 + sch * 8);                           \
    prev = ld_once16(qkv + ((2 * B_ + (BW)) * nH + h) * 2048 + srow * 32 + sch * 8);                           \
    preq = ld_once16(qkv + ((0 * B_ + (BW)) * nH + h) * 2048 + (16 * it + r16) * 32 + 8 * g);                       \
  } while (0)
  if (w_begin < B_) ATTN_FWD_PREFETCH(w_begin);

  for (int wi = 0; wi < wpw; ++wi) {
    const long long b_ = w_begin + wi;
    if (b_ >= B_) break;
    *reinterpret_cast<uint4*>(Ks + srow * TS + sch * 8) = prek;
    *reinterpret_cast<uint4*>(Vs + srow * TS + sch * 8) = prev;
    const bf16x8_t qf = __builtin_bit_cast(bf16x8_t, preq);
    __syncthreads();
    if (wi + 1 < wpw && b_ + 1 < B_) ATTN_FWD_PREFETCH(b_ + 1);

    f32x4_t s[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, TS, 16 * jt + r16, 8 * g), qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    {
      const int w = (int)b_ % geom.nW;
      const int wy = w / geom.nWw, wx = w - wy * geom.nWw;
      const bool masked = geom.shift > 0 && (wy == geom.H / 8 - 1 || wx == geom.nWw - 1);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_h + (16 * it + r16) * 64 + 16 * jt + 4 * g);
        s[jt] += f32x4_t{bv.x, bv.y, bv.z, bv.w};
      }
      if (masked) {
        const int labi = win_region_label(geom, w, 16 * it + r16);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (win_region_label(geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;   // :235 (-100, not -inf)
      }
    }
    bf16x8_t pf[2];
    const float inv = softmax_numerators(s, pf);      // unnormalised exp(s - max) as bf16; the outputs are scaled (as attn_fused.hip)
    f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_acc(Vs, TS, 32 * ss, 16 * dt, lane), pf[ss], o[dt], 0, 0, 0);
    }
    o[0] *= inv;
    o[1] *= inv;
    // o[dt][e] = O[i = 16 it + r16][d = 16 dt + 4 g + e]
    // exchange the dt = 0 quad of the odd 16-lane rows with the dt = 1 quad of the even ones (v_permlane16_swap): every lane
    // then owns 8 consecutive d -> one 16-byte store, 64-byte runs per row (see the backward kernel)
    {
      const uint2 x = pack_bf4(o[0][0], o[0][1], o[0][2], o[0][3]), y = pack_bf4(o[1][0], o[1][1], o[1][2], o[1][3]);
      const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
      bf16_t* dst = ao + (b_ * 64 + 16 * it + r16) * CA + h * 32 + (((g & 1) << 4) | ((g >> 1) << 3));
      *reinterpret_cast<uint4*>(dst) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// One 4-wave workgroup owns one (window, head) at a time (walking `wpw` windows of the head):
//   phase 1  wave w = query tile it: S^T[:, it], dP^T[:, it] (8 MFMAs), softmax, dS; P and dS go to LDS as [i][j]
//   phase 2  wave w = key tile jt:   dV^T[:, jt], dK^T[:, jt] (sum over all 64 queries) and dQ^T[:, it = w]
// 160 VGPRs per wave and 38 KB of LDS per workgroup -> 3 workgroups (12 waves) per CU hide each other's
// latencies; the next window's q/k/v/dO tiles are prefetched into registers during phase 2.
#ifndef SRK_ATTN_BWD_WGS
#define SRK_ATTN_BWD_WGS 3
#endif
__global__ __launch_bounds__(256, SRK_ATTN_BWD_WGS) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ biasd,
                                                       const bf16_t* __restrict__ dao, bf16_t* __restrict__ dqkv,
                                                       float* __restrict__ dbias_slab, long long B_, int nH, int CA,
                                                       WinGeom geom, int wpw, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[64 * TS], Ks[64 * TS], Vs[64 * TS], Os[64 * TS], Pb[64 * PS], Db[64 * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.y;
  const long long w_begin = (long long)blockIdx.x * wpw;
  const int ldq = 3 * CA;
  const int srow = tid >> 2, sch = tid & 3;          // tile staging: 64 rows x 4 chunks of 16 B
  const float* bias_h = biasd + h * 4096;

  f32x4_t dbias[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) dbias[jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 preq = make_uint4(0, 0, 0, 0), prek = preq, prev = preq, preo = preq;
#define ATTN_BWD_PREFETCH(BW)                                                                                     \
  do {                                                                                                            \
    preq = ld_once16(qkv + ((0 * B_ + (BW)) * nH + h) * 2048 + srow * 32 + sch * 8);                              \
    prek = ld_once16(qkv + ((1 * B_ + (BW)) * nH + h) * 2048 + srow * 32 + sch * 8);                              \
    prev = ld_once16(qkv + ((2 * B_ + (BW)) * nH + h) * 2048 + srow * 32 + sch * 8);                              \
    preo = ld_once16(dao + ((BW) * 64 + srow) * CA + h * 32 + sch * 8);                                           \
  } while (0)
  if (w_begin < B_) ATTN_BWD_PREFETCH(w_begin);

  for (int wi = 0; wi < wpw; ++wi) {
    const long long b_ = w_begin + wi;
    if (b_ >= B_) break;
    *reinterpret_cast<uint4*>(Qs + srow * TS + sch * 8) = preq;
    *reinterpret_cast<uint4*>(Ks + srow * TS + sch * 8) = prek;
    *reinterpret_cast<uint4*>(Vs + srow * TS + sch * 8) = prev;
    *reinterpret_cast<uint4*>(Os + srow * TS + sch * 8) = preo;
    srk_lds_barrier();      // LDS only: no vmcnt(0) drain of the prefetch loads / dqkv stores in flight
    if (wi + 1 < wpw && b_ + 1 < B_) ATTN_BWD_PREFETCH(b_ + 1);

    // ---- phase 1: this wave's 16 queries (it = wave) against all 64 keys -----------------------
    const int it = wave;
    f32x4_t s[4], dp[4];
    {
      const bf16x8_t qf = row_frag(Qs, TS, 16 * it + r16, 8 * g);
      const bf16x8_t of = row_frag(Os, TS, 16 * it + r16, 8 * g);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const bf16x8_t kf = row_frag(Ks, TS, 16 * jt + r16, 8 * g);
        const bf16x8_t vf = row_frag(Vs, TS, 16 * jt + r16, 8 * g);
        s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
    }
    {  // + bias + mask, softmax over the 64 keys of column i = 16 it + r16
      const int w = (int)b_ % geom.nW;
      const int wy = w / geom.nWw, wx = w - wy * geom.nWw;
      const bool masked = geom.shift > 0 && (wy == geom.H / 8 - 1 || wx == geom.nWw - 1);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_h + (16 * it + r16) * 64 + 16 * jt + 4 * g);
        s[jt] += f32x4_t{bv.x, bv.y, bv.z, bv.w};
      }
      if (masked) {
        const int labi = win_region_label(geom, w, 16 * it + r16);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (win_region_label(geom, w, 16 * jt + 4 * g + e) != labi) s[jt][e] += -100.0f;
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[jt][e]);
      mx = xmax4(mx);
      constexpr float L2E = 1.4426950408889634f;      // exp(x - m) = exp2(x log2e - m log2e), as softmax_numerators
      const float mxl = mx * L2E;
      f32x4_t a4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f32x4_t t = s[jt] * L2E - mxl;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[jt][e] = __builtin_amdgcn_exp2f(t[e]);
        a4 += s[jt];
      }
      const float inv = __builtin_amdgcn_rcpf(xsum4((a4[0] + a4[1]) + (a4[2] + a4[3])));
      float dl = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[jt][e] *= inv;
          dl += s[jt][e] * dp[jt][e];
        }
      dl = xsum4(dl);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dp[jt][e] = s[jt][e] * (dp[jt][e] - dl);       // dS
          dbias[jt][e] += dp[jt][e];
        }
        *reinterpret_cast<uint2*>(Pb + (16 * it + r16) * PS + 16 * jt + 4 * g) = pack_bf4(s[jt][0], s[jt][1], s[jt][2], s[jt][3]);
        *reinterpret_cast<uint2*>(Db + (16 * it + r16) * PS + 16 * jt + 4 * g) = pack_bf4(dp[jt][0], dp[jt][1], dp[jt][2], dp[jt][3]);
      }
    }
    srk_lds_barrier();      // LDS only: no vmcnt(0) drain of the prefetch loads / dqkv stores in flight

    // ---- phase 2: this wave's 16 keys (jt = wave) for dV / dK, its 16 queries for dQ --------------
    {
      f32x4_t av[2], ak[2], aq[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) av[dt] = ak[dt] = aq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8_t pf = tr_frag_nat(Pb, PS, 32 * ss, 16 * wave, lane);     // P[i][j in tile]  (k = i)
        const bf16x8_t df = tr_frag_nat(Db, PS, 32 * ss, 16 * wave, lane);     // dS[i][j in tile] (k = i)
        const bf16x8_t dr = row_frag(Db, PS, 16 * wave + r16, 32 * ss + 8 * g); // dS[i in tile][j] (k = j)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_nat(Os, TS, 32 * ss, 16 * dt, lane), pf, av[dt], 0, 0, 0);
          ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_nat(Qs, TS, 32 * ss, 16 * dt, lane), df, ak[dt], 0, 0, 0);
          aq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_nat(Ks, TS, 32 * ss, 16 * dt, lane), dr, aq[dt], 0, 0, 0);
        }
      }
      // The MFMA layout gives a lane d = 4g .. 4g+3 (dt = 0) and 16 + 4g .. (dt = 1) of row r16: stored directly that is
      // two 8-byte stores per lane and 32-byte runs per row.  v_permlane16_swap exchanges the dt = 0 quad of the odd
      // 16-lane rows with the dt = 1 quad of the even ones, after which every lane owns 8 consecutive d (16 bytes): one
      // store per q / k / v, 64-byte runs per row.
      const int col = ((g & 1) << 4) | ((g >> 1) << 3);
      bf16_t* row = dqkv + (b_ * 64 + 16 * wave + r16) * ldq + h * 32 + col;
      auto store8 = [&](bf16_t* dst, uint2 x, uint2 y) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        *reinterpret_cast<uint4*>(dst) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      };
      store8(row, pack_bf4(aq[0][0] * scale, aq[0][1] * scale, aq[0][2] * scale, aq[0][3] * scale),
             pack_bf4(aq[1][0] * scale, aq[1][1] * scale, aq[1][2] * scale, aq[1][3] * scale));
      store8(row + CA, pack_bf4(ak[0][0], ak[0][1], ak[0][2], ak[0][3]), pack_bf4(ak[1][0], ak[1][1], ak[1][2], ak[1][3]));
      store8(row + 2 * CA, pack_bf4(av[0][0], av[0][1], av[0][2], av[0][3]), pack_bf4(av[1][0], av[1][1], av[1][2], av[1][3]));
    }
    srk_lds_barrier();      // LDS only: no vmcnt(0) drain of the prefetch loads / dqkv stores in flight
  }

  // per-workgroup partial d(bias) slab, dense [i][j]; this wave owns the rows i = 16 wave + r16
  float* slab = dbias_slab + (((long long)blockIdx.x * nH + h) * 64) * 64;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    *reinterpret_cast<float4*>(slab + (16 * wave + r16) * 64 + 16 * jt + 4 * g) =
        make_float4(dbias[jt][0], dbias[jt][1], dbias[jt][2], dbias[jt][3]);
}

__global__ __launch_bounds__(256) void rpb_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dtable,
                                                         int nslab, int nH) {
  __shared__ float part[4][64];
  rpb_reduce_block(slab, dtable, nslab, nH, blockIdx.x, blockIdx.y, part);     // wgrad.h
}

// dense bias[h][i][j] = table[rpi(i,j)][h]   (network_swinir.py:127-129)
__global__ void rpb_expand_kernel(const float* __restrict__ table, float* __restrict__ biasd, int nH) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nH * 4096) return;
  const int h = idx >> 12, i = (idx >> 6) & 63, j = idx & 63;
  const int t = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
  biasd[idx] = table[t * nH + h];
}

}  // namespace

int srk_launch_attn_fwd(const bf16_t* qkv, const float* biasd, bf16_t* ao, long long B_, int nH, WinGeom geom,
                        hipStream_t stream) {
  // windows per workgroup: as few as possible while every workgroup is resident at once (8 per CU)
  const long long slots = 8 * 256;
  long long wpw = (B_ * nH + slots - 1) / slots;
  if (wpw < 1) wpw = 1;
  while (((B_ + wpw - 1) / wpw) * nH > slots && wpw < B_) ++wpw;
  dim3 grid((unsigned)((B_ + wpw - 1) / wpw), nH);
  srk_probe_pre(FAM_ATTN_FWD, stream, 0.0);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, stream, qkv, biasd, ao, B_, nH, nH * 32, geom, (int)wpw);
  srk_probe_post(FAM_ATTN_FWD, stream);
  return srk_check_launch("attn_fwd");
}

int srk_attn_bwd_slabs(long long B_, int nH, int* wpw_out) {
  // windows per workgroup: as few as possible while all workgroups are resident at once (3 per CU: 160 VGPRs)
  const long long slots = SRK_ATTN_BWD_WGS * 256;
  long long wpw = (B_ * nH + slots - 1) / slots;
  if (wpw < 1) wpw = 1;
  while (((B_ + wpw - 1) / wpw) * nH > slots && wpw < B_) ++wpw;
  if (wpw_out) *wpw_out = (int)wpw;
  return (int)((B_ + wpw - 1) / wpw);
}

int srk_launch_attn_bwd(const bf16_t* qkv, const float* biasd, const bf16_t* dao, bf16_t* dqkv, float* dbias_slab,
                        float* dtable, long long B_, int nH, WinGeom geom, float scale, hipStream_t stream) {
  int wpw;
  const int nslab = srk_attn_bwd_slabs(B_, nH, &wpw);
  dim3 grid(nslab, nH);
  srk_probe_pre(FAM_ATTN_BWD, stream, 0.0);
  hipLaunchKernelGGL(attn_bwd_kernel, grid, dim3(256), 0, stream, qkv, biasd, dao, dqkv, dbias_slab, B_, nH, nH * 32, geom,
                     wpw, scale);
  srk_probe_post(FAM_ATTN_BWD, stream);
  int rc = srk_check_launch("attn_bwd");
  if (rc || dtable == nullptr) return rc;      // dtable == null: the caller reduces the slabs later (RpbJob, wgrad.h)
  return srk_launch_rpb_reduce(dbias_slab, dtable, nslab, nH, stream);
}

int srk_launch_rpb_reduce(const float* slab, float* dtable, int nslab, int nH, hipStream_t stream) {
  hipLaunchKernelGGL(rpb_reduce_kernel, dim3(64, nH), dim3(256), 0, stream, slab, dtable, nslab, nH);
  return srk_check_launch("rpb_reduce");
}

int srk_launch_rpb_expand(const float* table, float* biasd, int nH, hipStream_t stream) {
  hipLaunchKernelGGL(rpb_expand_kernel, dim3(cdiv(nH * 4096, 256)), dim3(256), 0, stream, table, biasd, nH);
  return srk_check_launch("rpb_expand");
}
