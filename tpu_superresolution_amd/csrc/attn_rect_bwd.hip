// Backward of the rectangular-window attention of DAT's Spatial_Attention (dat_arch.py:138-262 on windows cut by img2windows
// :15-23, called per branch from Adaptive_Spatial_Attention.forward :366-446): wh x ww windows of 256 or 128 tokens (8 x 32, 32 x 8,
// 8 x 16, 16 x 8, ...), optional cyclic shift with the arithmetic shift mask, a DENSE additive bias [heads][N][N] (the dynamic
// position bias gathered through relative_position_index, :219-224), on a window frame Hp x Wp that zero-pads the H x W map at the
// bottom / right (:376-384).  The forward is srk_win_attention_fwd_padded (csrc/attn256.hip).
//
//   S = scale q k^T + bias (+ mask),  P = softmax(S),  O = P v
//   dV = P^T dO,  dP = dO v^T,  dS = P (dP - rowsum(P dP)),  dq = scale dS k,  dk = scale dS^T q,  d bias += dS  (summed over windows)
//
// Structure as csrc/attn256_bwd.hip (one 256-thread workgroup per (window, head); K, V, Q, dO of the window in LDS; pass 1 with lane =
// query produces the row statistics, dS and dq; pass 2 with lane = key rebuilds P and dS from the statistics and accumulates dk, dv),
// without its 16 x 16 geometry: query / key coordinates come from divisions by the window width, and the bias gradient is dense --
// with a scratch buffer every workgroup stores its dS tile [N][N] with 16-byte stores and a second kernel sums the windows of the
// launch into d bias (2 x 4 N^2 bytes per (window, head) of streaming traffic instead of N^2 float atomics on N^2 addresses: 4 x
// faster at DAT x4 size); without one every dS element is added to d bias[head][q][k] with a global float atomic.  Padded tokens are zero vectors: as keys they take part in the softmax with score = bias (their dk / dv have
// no destination), as queries their dO is zero, so they contribute nothing and their dq is not stored.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

// [rows][32] bf16 tiles without padding, the four 16-byte chunks of a row XOR-swizzled (the scheme of attn256_bwd.hip's self-attention
// form): conflict-free for the row-per-lane fragment reads and for the transposing reads, and 4 x 256 x 64 B + statistics = 68 KB per
// workgroup, so that TWO workgroups share a CU (the 40-element pitch needed 86 KB: one workgroup, one wave per SIMD).
constexpr int RP = 32;
__device__ __forceinline__ int tfs(int r) { return ((((r >> 2) ^ (r >> 3)) & 1) << 1) | (((r >> 3) ^ (r >> 1)) & 1); }
__device__ __forceinline__ int chunk_off(int r, int c) { return r * 32 + ((c ^ tfs(r)) << 3); }      // element offset of 16-byte chunk c of row r

struct RectBwdParams {
  const bf16_t* qkv;    // [T][ldq]
  const bf16_t* dout;   // [T][ldo]
  const float* bias;    // [nH][NQ][NQ]
  bf16_t* dqkv;         // [T][ldq]
  float* dbias;         // [nH][NQ][NQ], accumulated
  float* tiles;         // null, or scratch [windows][nH][NQ][NQ]: every workgroup stores its dS tile, a second kernel sums the windows
  const float* biasT;   // null, or [nH][NQ(key)][NQ(query)]: the bias transposed (behind the tiles in the scratch) for pass 2, whose lanes are keys
  int ldq, ldo, CA;
  int B, H, W, Hp, Wp;
  int wh, ww, sy, sx;
  int nWh, nWw, nH;
  float scale;
};

__device__ __forceinline__ int region3r(int v, int n, int w, int s) { return v < n - w ? 0 : (v < n - s ? 1 : 2); }

__device__ __forceinline__ bf16x8_t r_cat4(bf16x4_t lo, bf16x4_t hi) {
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// transposed fragment in accumulator k order (slots (g, jj): rows k0 + 4 g + jj, jj < 4, and k0 + 16 + 4 g + jj - 4)
// k0 is a multiple of 32: the swizzle of rows k0 + 4 g + (ll >> 2) (+ 16) depends on the lane only -> `tro` = the lane's element offset
// inside a 32-row block for c0 = 0 (tr_lane_off), c0 = 16 flips chunk bit 1
__device__ __forceinline__ int tr_lane_off(int lane, int c0) {
  const int ll = lane & 15, g = lane >> 4;
  const int ra = 4 * g + (ll >> 2);
  return chunk_off(ra, (c0 >> 3) + ((ll & 3) >> 1)) + ((ll & 1) << 2);
}
__device__ __forceinline__ bf16x8_t r_tr_acc(const bf16_t* tile, int k0, int tro) {
  return r_cat4(lds_tr_read(tile + k0 * 32 + tro), lds_tr_read(tile + (k0 + 16) * 32 + tro));
}

template <int QT>        // 16-query tiles per wave: 4 (256-token windows) or 2 (128-token windows)
__global__ __launch_bounds__(256, 2) void win_rect_attn_bwd_kernel(const RectBwdParams p) {
  constexpr int NQ = 64 * QT, NT = NQ / 16;
  constexpr float L2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);          // [NQ][RP]
  bf16_t* Vs = Ks + NQ * RP;
  bf16_t* Qs = Vs + NQ * RP;
  bf16_t* Os = Qs + NQ * RP;
  float* stats = reinterpret_cast<float*>(Os + NQ * RP);   // [NQ][4]: max * log2e, 1 / sum, rowsum(P dP)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int fro = r16 * 32 + ((g ^ tfs(r16)) << 3);           // fragment read of row 16 x + r16, chunk g: + 16 x * 32
  const int tro[2] = {tr_lane_off(lane, 0), tr_lane_off(lane, 16)};
  const int h = blockIdx.x % p.nH;
  const int wflat = blockIdx.x / p.nH;
  const int nW = p.nWh * p.nWw;
  const int b = wflat / nW, w = wflat - b * nW;
  const int wy = w / p.nWw, wx = w - wy * p.nWw;
  const long long tok0 = (long long)b * p.H * p.W;

  // raster token of window-local position l (roll + partition on the Hp x Wp frame), or -1 in the zero padding
  auto token = [&](int l) -> long long {
    const int ly = l / p.ww, lx = l - ly * p.ww;
    int y = wy * p.wh + ly + p.sy, x = wx * p.ww + lx + p.sx;
    if (y >= p.Hp) y -= p.Hp;
    if (x >= p.Wp) x -= p.Wp;
    if (y >= p.H || x >= p.W) return -1;
    return tok0 + (long long)y * p.W + x;
  };

  for (int kk = tid; kk < NQ; kk += 256) {
    const long long t = token(kk);
    uint4 qv[4], kv[4], vv[4], ov[4];
    if (t >= 0) {
      const bf16_t* row = p.qkv + t * p.ldq + h * 32;
      const bf16_t* orow = p.dout + t * p.ldo + h * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qv[c] = *reinterpret_cast<const uint4*>(row + 8 * c);
        kv[c] = *reinterpret_cast<const uint4*>(row + p.CA + 8 * c);
        vv[c] = *reinterpret_cast<const uint4*>(row + 2 * p.CA + 8 * c);
        ov[c] = *reinterpret_cast<const uint4*>(orow + 8 * c);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) qv[c] = kv[c] = vv[c] = ov[c] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<uint4*>(Qs + chunk_off(kk, c)) = qv[c];
      *reinterpret_cast<uint4*>(Ks + chunk_off(kk, c)) = kv[c];
      *reinterpret_cast<uint4*>(Vs + chunk_off(kk, c)) = vv[c];
      *reinterpret_cast<uint4*>(Os + chunk_off(kk, c)) = ov[c];
    }
  }
  __syncthreads();

  const bool masked = p.sy > 0 || p.sx > 0;
  const bool need_mask = masked && (wy == p.nWh - 1 || wx == p.nWw - 1);
  auto label = [&](int l) {      // region label of window-local position l in the shifted frame (dat_arch.py:318-364)
    const int ly = l / p.ww, lx = l - ly * p.ww;
    return region3r(wy * p.wh + ly, p.Hp, p.wh, p.sy) * 3 + region3r(wx * p.ww + lx, p.Wp, p.ww, p.sx);
  };
  const float* bias_h = p.bias + (long long)h * NQ * NQ;
  const float* biasT_h = p.biasT ? p.biasT + (long long)h * NQ * NQ : nullptr;
  float* dbias_h = p.tiles ? p.tiles + ((long long)wflat * p.nH + h) * NQ * NQ : p.dbias + (long long)h * NQ * NQ;
  const bool to_tiles = p.tiles != nullptr;

  // =========================== pass 1: lane = query r16 of the tile, registers = keys 16 j + 4 g + e ===========================
#pragma unroll 1
  for (int qt = 0; qt < QT; ++qt) {
    const int ql = wave * (16 * QT) + qt * 16 + r16;
    const int qlab = need_mask ? label(ql) : 0;
    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(Qs + (ql - r16) * 32 + fro);
    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(Os + (ql - r16) * 32 + fro);
    f32x4_t s[NT], dp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + 16 * j * 32 + fro);
      const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + 16 * j * 32 + fro);
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    const float* brow = bias_h + (long long)ql * NQ + 4 * g;
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4 bv = *reinterpret_cast<const float4*>(brow + 16 * j);
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = s[j][e] * p.scale + bb[e];
        if (need_mask && label(16 * j + 4 * g + e) != qlab) v += -100.0f;       // dat_arch.py:357-362 (-100, not -inf)
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = xrow_max4(mx);
    const float mxl = mx * L2E;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[j][e] = __builtin_amdgcn_exp2f(s[j][e] * L2E - mxl);
        sum += s[j][e];
      }
    const float inv = __builtin_amdgcn_rcpf(xrow_sum4(sum));
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[j][e] *= inv;
        dl += s[j][e] * dp[j][e];
      }
    dl = xrow_sum4(dl);
    if (g == 0) *reinterpret_cast<float4*>(stats + ql * 4) = make_float4(mxl, inv, dl, 0.f);
    float* drow = dbias_h + (long long)ql * NQ + 4 * g;
    f32x4_t aq[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int jj = 0; jj < NT / 2; ++jj) {
      uint2 lohi[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = 2 * jj + u;
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[e] = s[j][e] * (dp[j][e] - dl);      // dS = d bias[h][q][k] of this window
        if (to_tiles) {
          *reinterpret_cast<float4*>(drow + 16 * j) = make_float4(ds[0], ds[1], ds[2], ds[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(drow + 16 * j + e, ds[e]);    // summed over the windows of the launch
        }
        lohi[u] = pack_bf4(ds[0], ds[1], ds[2], ds[3]);
      }
      const bf16x8_t dsf = __builtin_bit_cast(bf16x8_t, make_uint4(lohi[0].x, lohi[0].y, lohi[1].x, lohi[1].y));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) aq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r_tr_acc(Ks, 32 * jj, tro[dt]), dsf, aq[dt], 0, 0, 0);
    }
    const long long qt_tok = token(ql);
    if (qt_tok >= 0) {
      bf16_t* qdst = p.dqkv + qt_tok * p.ldq + h * 32;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<uint2*>(qdst + 16 * dt + 4 * g) =
            pack_bf4(aq[dt][0] * p.scale, aq[dt][1] * p.scale, aq[dt][2] * p.scale, aq[dt][3] * p.scale);
    }
  }
  __syncthreads();      // row statistics of all queries are in LDS

  // =========================== pass 2: lane = key r16 of tile j, registers = queries 16 qt + 4 g + e ===========================
#pragma unroll 1
  for (int j = wave; j < NT; j += 4) {
    const int kl = 16 * j + r16;
    const int klab = need_mask ? label(kl) : 0;
    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (kl - r16) * 32 + fro);
    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + (kl - r16) * 32 + fro);
    f32x4_t av[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    f32x4_t ak[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
    for (int qp = 0; qp < NT / 2; ++qp) {
      uint2 pl[2], dl2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * qp + u;
        const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qs + 16 * qt * 32 + fro);
        const bf16x8_t oa = *reinterpret_cast<const bf16x8_t*>(Os + 16 * qt * 32 + fro);
        // sa[e] = q . k of (query 16 qt + 4 g + e, key kl);  da[e] = dO . v of the same pair
        const f32x4_t sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const f32x4_t da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // bias[q][kl] for the lane's four queries: one 16-byte load from the transposed copy; gathered from the [q][k] layout these
        // were four dependent-latency scalar loads per MFMA pair, 111 of the kernel's 268 us (null experiment)
        float bq[4];
        if (biasT_h) {
          const float4 bt = *reinterpret_cast<const float4*>(biasT_h + (long long)kl * NQ + 16 * qt + 4 * g);
          bq[0] = bt.x; bq[1] = bt.y; bq[2] = bt.z; bq[3] = bt.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[e] = bias_h[(long long)(16 * qt + 4 * g + e) * NQ + kl];
        }
        float pv[4], dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = 16 * qt + 4 * g + e;
          const float4 st = *reinterpret_cast<const float4*>(stats + q * 4);
          float v = sa[e] * p.scale + bq[e];
          if (need_mask && label(q) != klab) v += -100.0f;
          pv[e] = __builtin_amdgcn_exp2f(v * L2E - st.x) * st.y;              // P
          dv[e] = pv[e] * (da[e] - st.z);                                      // dS
        }
        pl[u] = pack_bf4(pv[0], pv[1], pv[2], pv[3]);
        dl2[u] = pack_bf4(dv[0], dv[1], dv[2], dv[3]);
      }
      const bf16x8_t pfb = __builtin_bit_cast(bf16x8_t, make_uint4(pl[0].x, pl[0].y, pl[1].x, pl[1].y));
      const bf16x8_t dfb = __builtin_bit_cast(bf16x8_t, make_uint4(dl2[0].x, dl2[0].y, dl2[1].x, dl2[1].y));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r_tr_acc(Os, 32 * qp, tro[dt]), pfb, av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r_tr_acc(Qs, 32 * qp, tro[dt]), dfb, ak[dt], 0, 0, 0);
      }
    }
    const long long t = token(kl);
    if (t >= 0) {
      bf16_t* dst = p.dqkv + t * p.ldq + h * 32;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        *reinterpret_cast<uint2*>(dst + p.CA + 16 * dt + 4 * g) =
            pack_bf4(ak[dt][0] * p.scale, ak[dt][1] * p.scale, ak[dt][2] * p.scale, ak[dt][3] * p.scale);
        *reinterpret_cast<uint2*>(dst + 2 * p.CA + 16 * dt + 4 * g) = pack_bf4(av[dt][0], av[dt][1], av[dt][2], av[dt][3]);
      }
    }
  }
}

// d bias[i] += sum over the windows of tiles[w][i]; grid (n4 / 256, window groups of 32): a few float atomics per element
__global__ __launch_bounds__(256) void rect_dbias_reduce_kernel(const float* __restrict__ tiles, float* __restrict__ dbias, long long n, int windows) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int w0 = blockIdx.y * 32, w1 = min(w0 + 32, windows);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int w = w0; w < w1; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(tiles + (long long)w * n + i4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  atomicAdd(dbias + i4, acc.x);
  atomicAdd(dbias + i4 + 1, acc.y);
  atomicAdd(dbias + i4 + 2, acc.z);
  atomicAdd(dbias + i4 + 3, acc.w);
}

// biasT[h][k][q] = bias[h][q][k]: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void rect_bias_transpose_kernel(const float* __restrict__ bias, float* __restrict__ biasT, int N) {
  __shared__ float t[32][33];
  const int h = blockIdx.z, q0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = bias + (long long)h * N * N;
  float* dst = biasT + (long long)h * N * N;
  for (int r = ty; r < 32; r += 8) t[r][tx] = src[(long long)(q0 + r) * N + k0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) dst[(long long)(k0 + r) * N + q0 + tx] = t[tx][r];
}

template <int QT>
int launch_rect(const RectBwdParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)4 * 64 * QT * RP * sizeof(bf16_t) + (size_t)64 * QT * 4 * sizeof(float);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&win_rect_attn_bwd_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      srk_set_error("window attention backward: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const long long grid = (long long)p.B * p.nWh * p.nWw * p.nH;
  SRK_REQUIRE(grid > 0 && grid < (1LL << 31), SRK_E_SHAPE, "window attention backward: bad grid %lld", grid);
  RectBwdParams pp = p;
  if (p.tiles) {        // the transposed bias lives behind the dS tiles of the scratch (srk_win_attention_bwd_padded_scratch counts it)
    constexpr int N = 64 * QT;
    float* bt = p.tiles + grid * N * N;
    hipLaunchKernelGGL(rect_bias_transpose_kernel, dim3(N / 32, N / 32, p.nH), dim3(256), 0, stream, p.bias, bt, N);
    pp.biasT = bt;
  }
  hipLaunchKernelGGL((win_rect_attn_bwd_kernel<QT>), dim3((unsigned)grid), dim3(256), lds, stream, pp);
  if (p.tiles) {
    const int windows = p.B * p.nWh * p.nWw;
    const long long n = (long long)p.nH * 64 * QT * 64 * QT;
    hipLaunchKernelGGL(rect_dbias_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256), (windows + 31) / 32), dim3(256), 0, stream, p.tiles, p.dbias, n,
                       windows);
  }
  return srk_check_launch("win_rect_attn_bwd");
}

}  // namespace

int srk_launch_win_attn_bwd_padded(const bf16_t* qkv, int ldq, int CA, const float* bias, const bf16_t* dout, int ldo, bf16_t* dqkv,
                                   float* dbias, float* tiles, int B, int H, int W, int Hp, int Wp, int wh, int ww, int sy, int sx, int nH,
                                   float scale, hipStream_t stream) {
  SRK_REQUIRE(qkv && bias && dout && dqkv && dbias, SRK_E_NULL, "window attention backward: null pointer");
  SRK_REQUIRE(wh > 0 && ww > 0 && (wh * ww == 256 || wh * ww == 128), SRK_E_UNSUPPORTED,
              "window attention backward: the window must hold 256 or 128 tokens (got %dx%d)", wh, ww);
  SRK_REQUIRE(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp % wh == 0 && Wp % ww == 0, SRK_E_SHAPE,
              "window attention backward: window frame %dx%d must cover the %dx%d map and be a multiple of the %dx%d window", Hp, Wp, H, W, wh,
              ww);
  SRK_REQUIRE(nH > 0 && CA >= nH * 32 && CA % 32 == 0 && ldq >= 3 * CA && ldq % 8 == 0 && ldo >= nH * 32 && ldo % 8 == 0, SRK_E_SHAPE,
              "window attention backward: bad layout nH=%d CA=%d ldq=%d ldo=%d", nH, CA, ldq, ldo);
  SRK_REQUIRE(sy >= 0 && sy < wh && sx >= 0 && sx < ww, SRK_E_SHAPE, "shift_size must in 0-window_size");
  RectBwdParams p;
  p.qkv = qkv; p.dout = dout; p.bias = bias; p.dqkv = dqkv; p.dbias = dbias; p.tiles = tiles; p.ldq = ldq; p.ldo = ldo; p.CA = CA; p.B = B; p.H = H; p.W = W;
  p.Hp = Hp; p.Wp = Wp; p.wh = wh; p.ww = ww; p.sy = sy; p.sx = sx; p.nWh = Hp / wh; p.nWw = Wp / ww; p.nH = nH; p.scale = scale; p.biasT = nullptr;
  return wh * ww == 256 ? launch_rect<4>(p, stream) : launch_rect<2>(p, stream);
}
