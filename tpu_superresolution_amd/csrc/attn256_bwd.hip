// Backward of the 256-query window attention of attn256.hip (HAT): window self-attention on 16 x 16 windows (optionally
// shifted, arithmetic mask) and the overlapping cross-attention (16 x 16 queries against 24 x 24 zero-padded keys).
// Reference: hat_arch.py:163-197 (WindowAttention.forward), :403-439 (OCAB.forward), bias tables through
// relative_position_index_SA / _OCA (:881-918; negative rpi_oca entries wrap like torch indexing).
//
//   S = scale q k^T + table[rpi] (+ mask),  P = softmax(S),  O = P v
//   dV = P^T dO,  dP = dO v^T,  dS = P (dP - rowsum(P dP)),  dq = scale dS k,  dk = scale dS^T q,  d table[rpi] += dS
//
// q, k, v and dO are read in raster token order straight from the projection output / the incoming gradient (roll, window
// partition and unfold live in the row addresses, as in the forward); d qkv is written in the same raster layout.
//
// One 256-thread workgroup per (window, head) holds K, V (NK rows), Q and dO (256 rows) of the window in LDS and makes two passes:
//   pass 1 (a wave owns 64 queries, 16 at a time; lane = query, registers = keys -- the forward's layout): S^T and dP^T, softmax,
//          row statistics (max, 1/sum, rowsum(P dP)) to LDS, dS; dq^T = K^T dS^T with dS^T taken from the accumulators as the B
//          operand; d table: self-attention: the query row qy and the key row ky of a (query tile, key tile) pair are constants,
//          so dS is first summed in REGISTERS per row offset qy - ky (19 offsets per wave x 16 x 16 column pairs: 76 VGPRs) and only
//          those sums are scattered along their column offsets with LDS float atomics at the end of the pass (LDS float atomics are
//          slow on this part: one per (query, key) pair -- 65 536 per workgroup -- took 2/3 of the kernel); the overlapping form
//          (24-wide key rows: a key tile straddles two rows) still adds every pair;
//   pass 2 (a wave owns every fourth 16-key tile; lane = key, registers = queries): S and dP recomputed per pair of query tiles,
//          P and dS rebuilt from the saved row statistics (no cross-lane reduction), dV^T += dO^T P, dK^T += Q^T dS.
// Self-attention windows partition the tokens: dk / dv are stored.  Overlapping key windows share tokens (up to four windows per
// token): dk / dv are added with fp32 atomics into a zeroed [T][2 CA] buffer that srk_launch_win256_attn_bwd converts afterwards
// (keys in the zero padding have no token and drop out).  The table gradient leaves as one partial column per workgroup,
// summed by win256_table_reduce_kernel (32-window slices, one float atomic per element and slice).
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

// LDS layout of the [rows][32] bf16 tiles.  Overlapping form: row pitch 40 elements (as attn256.hip: 16-byte fragment reads of 16
// consecutive rows hit 64 distinct banks).  Self-attention: unpadded 64-byte rows with the 16-byte chunk XOR-swizzled by tfs(row) --
// the swizzle of attn_bwd_fused.hip, conflict-free for the row-fragment and both transposing read patterns -- so that the four
// tiles take 64 KB (76 KB with the statistics and table columns: room for a second workgroup per CU once the pass-1 register
// footprint -- ~380 with the 76 row-offset sums -- fits 256; until then the kernel runs one workgroup per CU).
template <bool PAD>
struct TileAddr {
  static constexpr int PITCH = PAD ? 40 : 32;
  static __device__ __forceinline__ int tfs(int r) { return ((((r >> 2) ^ (r >> 3)) & 1) << 1) | (((r >> 3) ^ (r >> 1)) & 1); }
  // element offset of 16-byte chunk c (0..3) of row r
  static __device__ __forceinline__ int chunk(int r, int c) { return PAD ? r * 40 + c * 8 : r * 32 + ((c ^ tfs(r)) << 3); }
};

struct Win256BwdParams {
  const bf16_t* qkv;    // [T][ldq]
  const bf16_t* dout;   // [T][ldo] gradient of the attention output
  const float* table;   // [table_rows][nH]
  bf16_t* dqkv;         // [T][ldq] (q, k, v column blocks as qkv)
  float* dkv32;         // overlapping form: [T][2 CA] fp32, zeroed by the launcher
  float* tpart;         // [workgroup][table_rows] partial table gradient
  int table_rows;
  int ldq, ldo, CA;
  int B, H, W;
  int sy, sx;
  int nWh, nWw, nH;
  float scale;
};

__device__ __forceinline__ int region3(int v, int n, int w, int s) { return v < n - w ? 0 : (v < n - s ? 1 : 2); }

__device__ __forceinline__ bf16x8_t b_cat4(bf16x4_t lo, bf16x4_t hi) {
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// transposed fragment in accumulator k order (slots (g, jj): rows k0 + 4 g + jj, jj < 4, and k0 + 16 + 4 g + jj - 4), columns
// c0 .. c0 + 15 (c0 = 0 or 16): the lane supplies the address of row rb + (ll >> 2), columns c0 + 4 (ll & 3) ..
template <bool PAD>
__device__ __forceinline__ bf16x8_t tr_acc(const bf16_t* tile, int k0, int c0, int lane) {
  const int g = lane >> 4, ll = lane & 15;
  const int ra = k0 + 4 * g + (ll >> 2), rb = ra + 16;
  const int c = (c0 >> 3) + ((ll & 3) >> 1), in = (ll & 1) << 2;
  return b_cat4(lds_tr_read(tile + TileAddr<PAD>::chunk(ra, c) + in), lds_tr_read(tile + TileAddr<PAD>::chunk(rb, c) + in));
}

template <int NT, bool OCA>        // NT = key tiles of 16 (16: 256 keys, 36: 576 keys)
__global__ __launch_bounds__(256, 1) void win256_attn_bwd_kernel(const Win256BwdParams p) {
  using TA = TileAddr<OCA>;
  constexpr int BP = TA::PITCH;
  constexpr int TABN = OCA ? 1536 : 964;                 // floats reserved per table column (1521 / 961 rows)
  constexpr int NK = NT * 16;
  constexpr int KW = OCA ? 24 : 16;
  constexpr int PAD = OCA ? 4 : 0;
  constexpr float L2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);          // [NK][BP]
  bf16_t* Vs = Ks + NK * BP;                             // [NK][BP]
  bf16_t* Qs = Vs + NK * BP;                             // [256][BP]
  bf16_t* Os = Qs + 256 * BP;                            // [256][BP]
  float* stats = reinterpret_cast<float*>(Os + 256 * BP);   // [256][4]: max * log2e, 1 / sum, rowsum(P dP)
  float* tab = stats + 256 * 4;                          // [table_rows]
  float* tabg = tab + TABN;                              // [table_rows] gradient
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.nH;
  const int wflat = blockIdx.x / p.nH;
  const int nW = p.nWh * p.nWw;
  const int b = wflat / nW, w = wflat - b * nW;
  const int wy = w / p.nWw, wx = w - wy * p.nWw;
  const long long tok0 = (long long)b * p.H * p.W;

  // raster token of window-local query ql (roll + partition folded in)
  auto q_token = [&](int ql) {
    int y = wy * 16 + (ql >> 4) + p.sy, x = wx * 16 + (ql & 15) + p.sx;
    if (y >= p.H) y -= p.H;
    if (x >= p.W) x -= p.W;
    return tok0 + (long long)y * p.W + x;
  };
  // raster token of window-local key kl, or -1 in the zero padding of an overlapping window
  auto k_token = [&](int kl) -> long long {
    const int ky = kl / KW, kx = kl - ky * KW;
    int y, x;
    if constexpr (OCA) {
      y = wy * 16 - PAD + ky;
      x = wx * 16 - PAD + kx;
      if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) return -1;
    } else {
      y = wy * 16 + ky + p.sy;
      x = wx * 16 + kx + p.sx;
      if (y >= p.H) y -= p.H;
      if (x >= p.W) x -= p.W;
    }
    return tok0 + (long long)y * p.W + x;
  };

  // ---- stage K, V, Q, dO; table column of this head; zero the table gradient ------------------------------------------------
  for (int kk = tid; kk < NK; kk += 256) {
    const long long t = k_token(kk);
    uint4 kv[4], vv[4];
    if (t >= 0) {
      const bf16_t* row = p.qkv + t * p.ldq + h * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        kv[c] = *reinterpret_cast<const uint4*>(row + p.CA + 8 * c);
        vv[c] = *reinterpret_cast<const uint4*>(row + 2 * p.CA + 8 * c);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) kv[c] = vv[c] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<uint4*>(Ks + TA::chunk(kk, c)) = kv[c];
      *reinterpret_cast<uint4*>(Vs + TA::chunk(kk, c)) = vv[c];
    }
  }
  {
    const long long t = q_token(tid);
    const bf16_t* qrow = p.qkv + t * p.ldq + h * 32;
    const bf16_t* orow = p.dout + t * p.ldo + h * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<uint4*>(Qs + TA::chunk(tid, c)) = *reinterpret_cast<const uint4*>(qrow + 8 * c);
      *reinterpret_cast<uint4*>(Os + TA::chunk(tid, c)) = *reinterpret_cast<const uint4*>(orow + 8 * c);
    }
  }
  for (int i = tid; i < p.table_rows; i += 256) {
    tab[i] = p.table[(long long)i * p.nH + h];
    tabg[i] = 0.f;
  }
  __syncthreads();

  const bool masked = !OCA && (p.sy > 0 || p.sx > 0);
  const bool need_mask = masked && (wy == p.nWh - 1 || wx == p.nWw - 1);
  auto label = [&](int ly, int lx) {      // region label of window-local position (ly, lx) in the shifted frame (hat_arch.py:921-941)
    return region3(wy * 16 + ly, p.H, 16, p.sy) * 3 + region3(wx * 16 + lx, p.W, 16, p.sx);
  };
  // relative position index of (query (qy, qx), key (ky, kx)): hat_arch.py:881-894 / :896-918 in closed form
  auto rpi = [&](int qy, int qx, int ky, int kx) {
    if constexpr (OCA) {
      int idx = (ky - qy - 7) * 39 + (kx - qx - 7);
      if (idx < 0) idx += p.table_rows;                      // torch's negative-index wrap
      return idx;
    } else {
      return (qy - ky + 15) * 31 + (qx - kx + 15);
    }
  };

  // =========================== pass 1: lane = query r16 of the tile, registers = keys 16 j + 4 g + e ===========================
  // self-attention: dsum[qt - j + 15][e] = sum of dS[(qy, qx = r16)][(ky = j, kx = 4 g + e)] over this wave's pairs with qy - ky =
  // 4 wave + qt - j  (qt and j are unrolled: the index is a constant)
  constexpr int NDY = OCA ? 1 : 19;
  float dsum[NDY][4];
#pragma unroll
  for (int d = 0; d < NDY; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) dsum[d][e] = 0.f;
  constexpr int QT_UNROLL = OCA ? 1 : 4;
#pragma unroll QT_UNROLL
  for (int qt = 0; qt < 4; ++qt) {
    const int ql = wave * 64 + qt * 16 + r16;
    const int qy = ql >> 4, qx = ql & 15;
    const int qlab = need_mask ? label(qy, qx) : 0;
    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(Qs + TA::chunk(ql, g));
    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(Os + TA::chunk(ql, g));
    // the 36-tile form cannot hold S^T and dP^T at once (2 x 144 registers): dP^T tiles are produced twice, where they are used
    f32x4_t s[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + TA::chunk(16 * j + r16, g));
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    auto dp_tile = [&](int j) {
      const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + TA::chunk(16 * j + r16, g));
      return __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    };
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = 16 * j + 4 * g;
      const int ky = c / KW, kx0 = c - ky * KW;          // no carry inside the quad: KW is a multiple of 4
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = s[j][e] * p.scale + tab[rpi(qy, qx, ky, kx0 + e)];
        if (need_mask && label(ky, kx0 + e) != qlab) v += -100.0f;       // hat_arch.py:939 (-100, not -inf)
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
    for (int j = 0; j < NT; ++j) {
      const f32x4_t dpj = dp_tile(j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[j][e] *= inv;
        dl += s[j][e] * dpj[e];
      }
    }
    dl = xrow_sum4(dl);
    if (g == 0) *reinterpret_cast<float4*>(stats + ql * 4) = make_float4(mxl, inv, dl, 0.f);
    f32x4_t aq[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int jj = 0; jj < NT / 2; ++jj) {
      uint2 lohi[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = 2 * jj + u;
        const int c = 16 * j + 4 * g;
        const int ky = c / KW, kx0 = c - ky * KW;
        const f32x4_t dpj = dp_tile(j);
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ds[e] = s[j][e] * (dpj[e] - dl);                                     // dS
          if constexpr (OCA) atomicAdd(&tabg[rpi(qy, qx, ky, kx0 + e)], ds[e]);   // d table[rpi] (LDS)
          else dsum[OCA ? 0 : qt - j + 15][e] += ds[e];
        }
        lohi[u] = pack_bf4(ds[0], ds[1], ds[2], ds[3]);
      }
      const bf16x8_t dsf = __builtin_bit_cast(bf16x8_t, make_uint4(lohi[0].x, lohi[0].y, lohi[1].x, lohi[1].y));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) aq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_acc<OCA>(Ks, 32 * jj, 16 * dt, lane), dsf, aq[dt], 0, 0, 0);
    }
    // aq[dt][e] = dq[query r16][d = 16 dt + 4 g + e]
    bf16_t* qdst = p.dqkv + q_token(ql) * p.ldq + h * 32;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      *reinterpret_cast<uint2*>(qdst + 16 * dt + 4 * g) =
          pack_bf4(aq[dt][0] * p.scale, aq[dt][1] * p.scale, aq[dt][2] * p.scale, aq[dt][3] * p.scale);
  }
  if constexpr (!OCA) {
    // rows qy - ky = 4 wave + d - 15 of the table gradient: column offset qx - kx = r16 - 4 g - e
#pragma unroll
    for (int d = 0; d < NDY; ++d) {
      const int dyv = 4 * wave + d - 15;                    // in -15 .. 15 for the pairs that exist
      if (dyv < -15 || dyv > 15) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&tabg[(dyv + 15) * 31 + (r16 - 4 * g - e + 15)], dsum[d][e]);
    }
  }
  __syncthreads();      // row statistics of all 256 queries and every d table add are in LDS

  // =========================== pass 2: lane = key r16 of tile j, registers = queries 16 qt + 4 g + e ===========================
#pragma unroll 1
  for (int j = wave; j < NT; j += 4) {
    const int kl = 16 * j + r16;
    const int ky = kl / KW, kx = kl - ky * KW;
    const int klab = need_mask ? label(ky, kx) : 0;
    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + TA::chunk(kl, g));
    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vs + TA::chunk(kl, g));
    f32x4_t av[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    f32x4_t ak[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
    for (int qp = 0; qp < 8; ++qp) {
      uint2 pl[2], dl2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * qp + u;                                           // = qy of every query of the tile (16-wide windows)
        const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qs + TA::chunk(16 * qt + r16, g));
        const bf16x8_t oa = *reinterpret_cast<const bf16x8_t*>(Os + TA::chunk(16 * qt + r16, g));
        // sa[e] = q . k of (query 16 qt + 4 g + e, key kl);  da[e] = dO . v of the same pair
        const f32x4_t sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const f32x4_t da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        float pv[4], dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int qx = 4 * g + e;
          const float4 st = *reinterpret_cast<const float4*>(stats + (16 * qt + qx) * 4);
          float v = sa[e] * p.scale + tab[rpi(qt, qx, ky, kx)];
          if (need_mask && label(qt, qx) != klab) v += -100.0f;
          pv[e] = __builtin_amdgcn_exp2f(v * L2E - st.x) * st.y;              // P
          dv[e] = pv[e] * (da[e] - st.z);                                      // dS
        }
        pl[u] = pack_bf4(pv[0], pv[1], pv[2], pv[3]);
        dl2[u] = pack_bf4(dv[0], dv[1], dv[2], dv[3]);
      }
      // B operands over the 32 queries of the pair in accumulator k order; A = dO^T / Q^T by transposing reads in that order
      const bf16x8_t pfb = __builtin_bit_cast(bf16x8_t, make_uint4(pl[0].x, pl[0].y, pl[1].x, pl[1].y));
      const bf16x8_t dfb = __builtin_bit_cast(bf16x8_t, make_uint4(dl2[0].x, dl2[0].y, dl2[1].x, dl2[1].y));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_acc<OCA>(Os, 32 * qp, 16 * dt, lane), pfb, av[dt], 0, 0, 0);
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_acc<OCA>(Qs, 32 * qp, 16 * dt, lane), dfb, ak[dt], 0, 0, 0);
      }
    }
    // av[dt][e] = dv[key kl][d = 16 dt + 4 g + e], ak likewise (dk = scale dS^T q)
    const long long t = k_token(kl);
    if (t >= 0) {
      if constexpr (OCA) {
        float* dst = p.dkv32 + t * (2 * p.CA) + h * 32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(dst + 16 * dt + 4 * g + e, ak[dt][e] * p.scale);
            atomicAdd(dst + p.CA + 16 * dt + 4 * g + e, av[dt][e]);
          }
      } else {
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
  // partial table gradient of this workgroup (complete since the barrier above)
  float* tp = p.tpart + (long long)blockIdx.x * p.table_rows;
  for (int i = tid; i < p.table_rows; i += 256) tp[i] = tabg[i];
}

// d table[i][h] += sum over the (window) workgroups of head h: blockIdx.z takes a 32-window slice (eight loads in flight per thread) and adds
// its sum with one float atomic per element (24 workgroups walking all the windows one dependent load at a time took 65 us)
__global__ __launch_bounds__(256) void win256_table_reduce_kernel(const float* __restrict__ tpart, float* __restrict__ dtable, int nwin,
                                                                   int nH, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (i >= rows) return;
  const int w0 = blockIdx.z * 32, w1 = min(nwin, w0 + 32);
  const float* src = tpart + (long long)h * rows + i;
  const long long stride = (long long)nH * rows;
  float a = 0.f;
  int wdx = w0;
  for (; wdx + 8 <= w1; wdx += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(wdx + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; wdx < w1; ++wdx) a += src[wdx * stride];
  atomicAdd(dtable + (long long)i * nH + h, a);
}

// dqkv[t][CA + c] = bf16(dkv32[t][c]), c < 2 CA
__global__ __launch_bounds__(256) void win256_dkv_cast_kernel(const float* __restrict__ dkv32, bf16_t* __restrict__ dqkv, long long T, int CA,
                                                              int ldq) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 each
  const int per_row = 2 * CA / 4;
  if (i >= T * per_row) return;
  const long long t = i / per_row;
  const int c = (int)(i - t * per_row) * 4;
  const float4 v = *reinterpret_cast<const float4*>(dkv32 + t * (2 * CA) + c);
  *reinterpret_cast<uint2*>(dqkv + t * ldq + CA + c) = pack_bf4(v.x, v.y, v.z, v.w);
}

template <int NT, bool OCA>
int launch_bwd(const Win256BwdParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * NT * 16 + 512) * TileAddr<OCA>::PITCH * sizeof(bf16_t) + 256 * 4 * sizeof(float) +
                         2 * (OCA ? 1536 : 964) * sizeof(float);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&win256_attn_bwd_kernel<NT, OCA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      srk_set_error("win256 attention backward: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const long long grid = (long long)p.B * p.nWh * p.nWw * p.nH;
  SRK_REQUIRE(grid > 0 && grid < (1LL << 31), SRK_E_SHAPE, "win256 attention backward: bad grid %lld", grid);
  hipLaunchKernelGGL((win256_attn_bwd_kernel<NT, OCA>), dim3((unsigned)grid), dim3(256), lds, stream, p);
  return srk_check_launch("win256_attn_bwd");
}

}  // namespace

size_t srk_win256_attn_bwd_scratch(int B, int H, int W, int nH, int CA, int table_rows, int overlap) {
  if (B <= 0 || H <= 0 || W <= 0 || nH <= 0) return 0;
  const size_t nwg = (size_t)B * (H / 16) * (W / 16) * nH;
  size_t bytes = (nwg * (size_t)table_rows + 3) / 4 * 4 * sizeof(float);       // the accumulation image behind it starts 16-byte aligned
  if (overlap > 0) bytes += (size_t)B * H * W * 2 * CA * sizeof(float);
  return (bytes + 255) / 256 * 256;
}

int srk_launch_win256_attn_bwd(const bf16_t* qkv, int ldq, int CA, const float* table, int table_rows, const bf16_t* dout, int ldo,
                               bf16_t* dqkv, float* dtable, void* scratch, int B, int H, int W, int sy, int sx, int nH, float scale,
                               int overlap, hipStream_t stream) {
  SRK_REQUIRE(qkv && table && dout && dqkv && dtable && scratch, SRK_E_NULL, "win256 attention backward: null pointer");
  SRK_REQUIRE(B > 0 && H % 16 == 0 && W % 16 == 0, SRK_E_SHAPE, "win256 attention backward: %dx%d is not a multiple of the 16x16 window", H, W);
  SRK_REQUIRE(nH > 0 && CA >= nH * 32 && CA % 32 == 0 && ldq >= 3 * CA && ldq % 8 == 0 && ldo >= nH * 32 && ldo % 8 == 0, SRK_E_SHAPE,
              "win256 attention backward: bad layout nH=%d CA=%d ldq=%d ldo=%d", nH, CA, ldq, ldo);
  SRK_REQUIRE(sy >= 0 && sy < 16 && sx >= 0 && sx < 16, SRK_E_SHAPE, "shift_size must in 0-window_size");
  SRK_REQUIRE(table_rows == (overlap > 0 ? 39 * 39 : 31 * 31), SRK_E_UNSUPPORTED,
              "win256 attention backward: built for 16x16 windows with the bias table (961 rows; 1521 for the overlapping form), got %d rows",
              table_rows);
  SRK_REQUIRE(overlap == 0 || (overlap == 8 && sy == 0 && sx == 0), SRK_E_UNSUPPORTED,
              "overlapping cross-attention backward is built for overlap 8 (24x24 keys), no shift");
  Win256BwdParams p;
  p.qkv = qkv; p.dout = dout; p.table = table; p.dqkv = dqkv; p.table_rows = table_rows; p.ldq = ldq; p.ldo = ldo; p.CA = CA;
  p.B = B; p.H = H; p.W = W; p.sy = sy; p.sx = sx; p.nWh = H / 16; p.nWw = W / 16; p.nH = nH; p.scale = scale;
  const long long nwin = (long long)B * p.nWh * p.nWw;
  p.tpart = static_cast<float*>(scratch);
  p.dkv32 = p.tpart + (nwin * nH * table_rows + 3) / 4 * 4;
  const long long T = (long long)B * H * W;
  int rc;
  if (overlap > 0) {
    rc = srk_launch_zero_f32(p.dkv32, T * 2 * CA, stream);      // (not hipMemsetAsync: see misc.hip, graph replays)
    if (rc) return rc;
    rc = launch_bwd<36, true>(p, stream);
    if (rc) return rc;
    const long long n4 = T * (2 * CA / 4);
    hipLaunchKernelGGL(win256_dkv_cast_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p.dkv32, dqkv, T, CA, ldq);
    rc = srk_check_launch("win256_dkv_cast");
  } else {
    rc = launch_bwd<16, false>(p, stream);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(win256_table_reduce_kernel, dim3((table_rows + 255) / 256, nH, (unsigned)((nwin + 31) / 32)), dim3(256), 0, stream, p.tpart, dtable, (int)nwin, nH,
                     table_rows);
  return srk_check_launch("win256_table_reduce");
}
