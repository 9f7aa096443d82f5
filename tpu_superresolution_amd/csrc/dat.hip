// DAT building blocks (reference dat_arch.py) for the host-orchestrated inference path (tpu_superresolution_amd/dat_arch.py).
// Token-major layouts throughout ([B*H*W][ld] bf16, channels padded per head to 32 / per tensor to 64); everything here is
// HBM-bound elementwise / stencil / small-reduction work: coalesced 16-byte accesses per lane, no GEMM reshaping.
//
//   dwconv3x3_kernel        depth-wise 3x3 conv (+ folded bias / BatchNorm affine, optional GELU, optional gating multiply):
//                           the DW-conv branches of both attention blocks (dat_arch.py:310-314, :463-467) and SGFN's spatial
//                           gate x1 * DWconv(LN(x2)) (:48-54)
//   rowln_bf16_kernel       LayerNorm of a bf16 column slice (SpatialGate.norm :46 over hidden / 2 channels)
//   spatial_gate_kernel     spatial_interaction (:322-327): sigmoid(w3 . gelu(W0 x + b0) + b3) per token (BatchNorm folded)
//   dual_gate_combine_kernel  attened_x * sigmoid(map_a) + conv_x * sigmoid(map_b) (:430-436, :518-524)
//   chan_gram_* / chan_apply  Adaptive_Channel_Attention (:481-505): q, k L2-normalised over the tokens, (C/h x C/h) logits *
//                           temperature, softmax, applied to v -- Gram partials per 256-token chunk, fixed-order finish
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  unpack_bf2(v.x, f[0], f[1]);
  unpack_bf2(v.y, f[2], f[3]);
  unpack_bf2(v.z, f[4], f[5]);
  unpack_bf2(v.w, f[6], f[7]);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

__device__ __forceinline__ void unpack4(const uint2& v, float (&f)[4]) {
  unpack_bf2(v.x, f[0], f[1]);
  unpack_bf2(v.y, f[2], f[3]);
}

// Depth-wise 3 x 3 conv (+ folded bias / BatchNorm affine, optional GELU, optional gating multiply), LDS-tiled: a workgroup owns an 8 x 16
// pixel tile of 64 channels.  w: fp32 [C][9] (zero rows for pad channels), scale / shift: fp32 [C].  The 10 x 18 halo tile (23 KB) arrives
// by LDS-DMA -- every input element is fetched once per tile (1.4 x with the halo), and no registers hold loads in flight.  (Earlier
// forms: nine neighbour loads + weight re-reads per pixel, 350 us per launch at cfg5; then a register form with a sliding window that
// issued its 30 loads per thread up front and ran at three waves per SIMD, 31.6 us for 67 MB of compulsory traffic.)
// Thread = (4-channel group, tile column): it walks the eight rows of its column with a 3 x 3 window of packed pairs that takes three 8-byte LDS reads per output (a wave reads 512
// contiguous bytes: conflict-free).  Any H, W (edges are predicated); channels beyond C read zeros and are not stored.
constexpr int DT_H = 8, DT_W = 16, DT_CB = 64;
constexpr int DT_HH = DT_H + 2, DT_HW = DT_W + 2;
constexpr int DT_PIECES = DT_HH * DT_HW * 8;                 // 16-byte pieces of the halo tile (8 per pixel)
constexpr int DT_FILL = (DT_PIECES + 255) / 256;             // DMA instructions per wave
__device__ uint4 g_dw_zero[1];                               // 16 zero bytes: DMA source outside the image / beyond the channels

__global__ __launch_bounds__(256) void dwconv3x3_tile_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const bf16_t* __restrict__ mul, int ldm, bf16_t* __restrict__ out, int ldo, int B,
                                                             int H, int W, int C, int act, int tilesx, int tilesy) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[DT_FILL * 256 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tx = bid % tilesx;
  bid /= tilesx;
  const int ty = bid % tilesy, b = bid / tilesy;
  const int c0 = blockIdx.y * DT_CB, x0 = tx * DT_W, y0 = ty * DT_H;
  const unsigned lds_base = (unsigned)(size_t)tile;
  const bf16_t* zero16 = reinterpret_cast<const bf16_t*>(g_dw_zero);
#pragma unroll
  for (int it = 0; it < DT_FILL; ++it) {
    const int p = it * 256 + wave * 64 + lane;
    const int pix = p >> 3, ch = p & 7;
    const int py = pix / DT_HW, px = pix - py * DT_HW;
    const int yy = y0 + py - 1, xx = x0 + px - 1;
    const bool ok = p < DT_PIECES && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && c0 + ch * 8 < C;
    const bf16_t* src = ok ? x + ((long long)(b * H + yy) * W + xx) * ldx + c0 + ch * 8 : zero16;
    srk_glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((it * 256 + wave * 64) * 16)));
  }
  const int cg = tid & 15, col = tid >> 4;
  const int c = c0 + cg * 4;
  const bool live = c < C && x0 + col < W;
  srk_f32x2_t wr[9][2], sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ca = live ? c + 2 * h : 0, cb = live ? c + 2 * h + 1 : 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][h] = srk_f32x2_t{w[ca * 9 + t], w[cb * 9 + t]};
    sc[h] = srk_f32x2_t{scale[ca], scale[cb]};
    sh[h] = srk_f32x2_t{shift[ca], shift[cb]};
  }
  uint2 mraw[DT_H];
  const long long pix0 = ((long long)(b * H + y0) * W + x0 + col);
  if (mul && live) {
#pragma unroll
    for (int it = 0; it < DT_H; ++it)
      mraw[it] = y0 + it < H ? *reinterpret_cast<const uint2*>(mul + (pix0 + (long long)it * W) * ldm + c) : make_uint2(0u, 0u);
  }
  srk_wait_vmcnt<0>();
  __syncthreads();
  const unsigned char* tl = tile + col * 128 + cg * 8;          // halo pixel (r, col + dx) at + (r * DT_HW + dx) * 128
  srk_f32x2_t win[3][3][2];                                     // [row slot][dx][channel pair]
  auto take_row = [&](int slot, int r) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const uint2 v = *reinterpret_cast<const uint2*>(tl + (r * DT_HW + dx) * 128);
      win[slot][dx][0] = srk_f32x2_t{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
      win[slot][dx][1] = srk_f32x2_t{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    }
  };
  take_row(0, 0);
  take_row(1, 1);
#pragma unroll
  for (int it = 0; it < DT_H; ++it) {
    take_row((it + 2) % 3, it + 2);
    srk_f32x2_t acc[2] = {srk_f32x2_t{0.f, 0.f}, srk_f32x2_t{0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] = __builtin_elementwise_fma(wr[r * 3 + dx][h], win[(it + r) % 3][dx][h], acc[h]);
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const srk_f32x2_t a = __builtin_elementwise_fma(acc[h], sc[h], sh[h]);
      v[2 * h] = a[0];
      v[2 * h + 1] = a[1];
    }
    if (act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
    }
    if (mul) {
      float m[4];
      unpack4(mraw[it], m);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= m[e];
    }
    if (live && y0 + it < H)
      *reinterpret_cast<uint2*>(out + (pix0 + (long long)it * W) * ldo + c) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}


// LayerNorm over C channels of a bf16 row slice, eps 1e-5: LPR lanes per row (8 channels each: 16 / 32 / 64 lanes for C <= 128 / 256 / 512),
// so a wave works on 64 / LPR rows at once, and every lane group keeps two rows in flight.  (One row per wave and iteration -- 24 of
// 64 lanes busy at C = 192, one dependent load per iteration -- ran at 1.8 TB/s: 28 us per launch at DAT x4 size.)
template <int LPR>
__global__ __launch_bounds__(256) void rowln_bf16_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ out, int ldo, long long rows,
                                                         int C, int CPo) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane % LPR, sub = lane / LPR;
  const int c0 = li * 8;
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gm[e] = c0 + e < C ? gamma[c0 + e] : 0.f;
    bt[e] = c0 + e < C ? beta[c0 + e] : 0.f;
  }
  const float invC = 1.0f / (float)C;
  auto group_sum = [&](float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
  };
  const long long stride = (long long)gridDim.x * 4 * RPW;
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * RPW + sub; r0 < rows; r0 += 2 * stride) {
    float v[2][8];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long r = r0 + u * stride;
      ok[u] = r < rows;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
      if (ok[u] && c0 < C) unpack8(*reinterpret_cast<const uint4*>(x + r * ldx + c0), v[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (c0 + e >= C) v[u][e] = 0.f;
        s += v[u][e];
      }
      const float mean = group_sum(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[u][e] = c0 + e < C ? v[u][e] - mean : 0.f;
        q += v[u][e] * v[u][e];
      }
      const float rstd = rsqrtf(group_sum(q) * invC + 1e-5f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = v[u][e] * rstd * gm[e] + bt[e];
      if (ok[u] && c0 < CPo) *reinterpret_cast<uint4*>(out + (r0 + u * stride) * ldo + c0) = pack8(v[u]);
    }
  }
}

// per token: g = sigmoid(b3 + sum_s w3[s] * gelu(b0[s] + sum_c W0[s][c] x[c])).  16 lanes per token (4 tokens per wave),
// lane j holds channels 64 i + 4 j .. +3 (the LayerNorm layout).  W0: fp32 [S][CP] (zero at pad channels), S <= 16.
template <int NV>
__global__ __launch_bounds__(256) void spatial_gate_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ W0,
                                                           const float* __restrict__ b0, const float* __restrict__ w3, float b3,
                                                           const float* __restrict__ b3_dev, int S, float* __restrict__ gate, long long rows) {
  constexpr int CP = NV * 64;
  __shared__ float Ws[16 * 256];
  if (b3_dev) b3 += *b3_dev;           // the last bias from device memory (training: no host read of a parameter per block)
  for (int i = threadIdx.x; i < S * CP; i += 256) Ws[i] = W0[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, sub = lane >> 4;
  for (long long m = ((long long)blockIdx.x * 4 + wave) * 4 + sub; m < rows; m += (long long)gridDim.x * 16) {
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint2 u = *reinterpret_cast<const uint2*>(x + m * ldx + 64 * i + 4 * j);
      unpack_bf2(u.x, v[i][0], v[i][1]);
      unpack_bf2(u.y, v[i][2], v[i][3]);
    }
    float acc = b3;
    for (int s = 0; s < S; ++s) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(Ws + s * CP + 64 * i + 4 * j);
        d += v[i][0] * wv.x + v[i][1] * wv.y + v[i][2] * wv.z + v[i][3] * wv.w;
      }
      d = wave_sum16(d) + b0[s];
      acc += w3[s] * gelu_f(d);
    }
    if (j == 0) gate[m] = 1.0f / (1.0f + __expf(-acc));
  }
}

// out = a * ga + b * gb;  tok_gate_on_a != 0: ga = tgate[token], gb = cgate[sample][c]; else ga = cgate[sample][c], gb = tgate[token]
__global__ __launch_bounds__(256) void dual_gate_combine_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                                const float* __restrict__ cgate, const float* __restrict__ tgate,
                                                                bf16_t* __restrict__ out, long long rows, int rows_per_sample, int CP,
                                                                int tok_gate_on_a) {
  const int c8 = CP / 8;
  const long long total = rows * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    const long long m = i / c8;
    float av[8], bv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(a + m * CP + cg * 8), av);
    unpack8(*reinterpret_cast<const uint4*>(b + m * CP + cg * 8), bv);
    const float tg = tgate[m];
    const float* cgp = cgate + (m / rows_per_sample) * CP + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cgv = cgp[e];
      o[e] = tok_gate_on_a ? av[e] * tg + bv[e] * cgv : av[e] * cgv + bv[e] * tg;
    }
    *reinterpret_cast<uint4*>(out + m * CP + cg * 8) = pack8(o);
  }
}

// ---- channel attention -----------------------------------------------------------------------------------------------------
constexpr int GCH = 256;            // tokens per Gram chunk
constexpr int GSZ = 32 * 32 + 64;   // partial: G[32][32], |q|^2 [32], |k|^2 [32]

// Gram partials on the matrix cores, grid (chunks, heads, B): G = x^T y is a [32 x 256] x [256 x 32] product per (chunk, head).  The chunk's x and y rows go to
// LDS as bf16 [256][40] (16-byte loads, one token row per thread); wave w owns the 16 x 16 output tile (i tile w >> 1, j tile w & 1) and
// walks the 256 tokens in eight k-steps with transposing LDS reads (ds_read_b64_tr_b16: lane = channel, registers = tokens) for both
// operands; the two diagonal waves also form x^T x and y^T y, whose diagonals are the squared column norms.  fp32 accumulation of exact
// bf16 products.  (The VALU form this replaces -- 1088 dot products of length 256 out of padded fp32 LDS tiles -- took 67.7 us per
// launch at DAT x4 size.)
constexpr int GP = 40;              // LDS row pitch (elements) of the bf16 tiles
__global__ __launch_bounds__(256) void chan_gram_mfma_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ y, int ldy,
                                                             float* __restrict__ partial, int N) {
  __shared__ __attribute__((aligned(16))) bf16_t xs[GCH * GP];
  __shared__ __attribute__((aligned(16))) bf16_t ys[GCH * GP];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int n0 = chunk * GCH;
  const int tid = threadIdx.x;
  {
    uint4 xv[4], yv[4];
    if (n0 + tid < N) {
      const long long row = (long long)b * N + n0 + tid;
      const bf16_t* xr = x + row * ldx + h * 32;
      const bf16_t* yr = y + row * ldy + h * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xv[c] = *reinterpret_cast<const uint4*>(xr + 8 * c);
        yv[c] = *reinterpret_cast<const uint4*>(yr + 8 * c);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[c] = yv[c] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<uint4*>(xs + tid * GP + 8 * c) = xv[c];
      *reinterpret_cast<uint4*>(ys + tid * GP + 8 * c) = yv[c];
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int it = wave >> 1, jt = wave & 1;
  f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f}, accx = acc, accy = acc;
  auto frag = [&](const bf16_t* tile, int ss, int ct) {
    const bf16x4_t lo = lds_tr_read(tr_addr(tile, GP, 32 * ss + 4 * g, 16 * ct, lane));
    const bf16x4_t hi = lds_tr_read(tr_addr(tile, GP, 32 * ss + 16 + 4 * g, 16 * ct, lane));
    return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
#pragma unroll
  for (int ss = 0; ss < GCH / 32; ++ss) {
    const bf16x8_t xf = frag(xs, ss, it), yf = frag(ys, ss, jt);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, xf, acc, 0, 0, 0);          // D[j = 4 g + e][i = r16]
    if (it == jt) {
      accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, xf, accx, 0, 0, 0);
      accy = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, yf, accy, 0, 0, 0);
    }
  }
  float* o = partial + (((long long)b * gridDim.y + h) * gridDim.x + chunk) * GSZ;
  *reinterpret_cast<float4*>(o + (16 * it + r16) * 32 + 16 * jt + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  if (it == jt && (r16 >> 2) == g) {
    const int e = r16 & 3;
    o[1024 + 16 * it + r16] = e == 0 ? accx[0] : (e == 1 ? accx[1] : (e == 2 ? accx[2] : accx[3]));
    o[1056 + 16 * jt + r16] = e == 0 ? accy[0] : (e == 1 ? accy[1] : (e == 2 ? accy[2] : accy[3]));
  }
}

// grid (heads, B): sum the chunk partials in order, normalise, temperature, softmax over j (d real columns) -> A [B][h][32][32]
__global__ __launch_bounds__(256) void chan_attn_finish_kernel(const float* __restrict__ partial, int nchunk, int d, const float* __restrict__ temperature,
                                                               float* __restrict__ A) {
  __shared__ float G[GSZ];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* base = partial + ((long long)b * gridDim.x + h) * nchunk * GSZ;
  for (int p = tid; p < GSZ; p += 256) {
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += base[(long long)c * GSZ + p];
    G[p] = s;
  }
  __syncthreads();
  float* out = A + ((long long)b * gridDim.x + h) * 1024;
  // thread = (row i, four columns): the row's softmax through three lane exchanges over its eight threads (32 threads walking whole rows
  // out of private arrays took 26 us per launch)
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  const float qn = fmaxf(sqrtf(G[1024 + i]), 1e-12f);                     // F.normalize: x / max(||x||, eps)
  const float t = temperature[h];
  float l[4], mx = -3.0e38f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = j0 + e;
    const float kn = fmaxf(sqrtf(G[1024 + 32 + j]), 1e-12f);
    l[e] = (j < d && i < d) ? G[i * 32 + j] / (qn * kn) * t : -3.0e38f;
    mx = fmaxf(mx, l[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  mx = fmaxf(mx, __shfl_xor(mx, 4));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    l[e] = (j0 + e < d && i < d) ? __expf(l[e] - mx) : 0.f;
    sum += l[e];
  }
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  sum += __shfl_xor(sum, 4);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  *reinterpret_cast<float4*>(out + i * 32 + j0) = make_float4(l[0] * inv, l[1] * inv, l[2] * inv, l[3] * inv);
}

// out[n][32 h + i] = sum_j A[b][h][i][j] v[n][32 h + j] on the matrix cores: one MFMA 16x16x32 per (16 tokens, head, 16 output
// channels), K = the 32 (padded) channels of the head.  a operand = A rows as bf16 (the window-attention kernels round P the same
// way), b operand = the tokens' v fragments straight from the qkv rows; D[i = 4 g + e][token = lane & 15].  One wave = 16 tokens.
__global__ __launch_bounds__(256) void chan_apply_kernel(const bf16_t* __restrict__ qkv, int ldq, int CA, const float* __restrict__ A, int nH,
                                                         bf16_t* __restrict__ out, int ldo, int N) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int n = blockIdx.x * 64 + wave * 16 + r16;
  const bool ok = n < N;
  const long long row = (long long)b * N + (ok ? n : 0);
  for (int h = 0; h < nH; ++h) {
    bf16x8_t vf = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) vf = *reinterpret_cast<const bf16x8_t*>(qkv + row * ldq + 2 * CA + h * 32 + 8 * g);
    const float* Ah = A + ((long long)b * nH + h) * 1024;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float4 a0 = *reinterpret_cast<const float4*>(Ah + (16 * mt + r16) * 32 + 8 * g);
      const float4 a1 = *reinterpret_cast<const float4*>(Ah + (16 * mt + r16) * 32 + 8 * g + 4);
      const uint2 lo = pack_bf4(a0.x, a0.y, a0.z, a0.w), hi = pack_bf4(a1.x, a1.y, a1.z, a1.w);
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, u32x4{lo.x, lo.y, hi.x, hi.y});
      const f32x4_t d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, vf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      if (ok) *reinterpret_cast<uint2*>(out + row * ldo + h * 32 + 16 * mt + 4 * g) = pack_bf4(d[0], d[1], d[2], d[3]);
    }
  }
}

inline int grid_for(long long n, int block = 256, int cap = 16384) {
  long long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int srk_dwconv3x3(const uint16_t* x, int ldx, const float* w, const float* scale, const float* shift, const uint16_t* mul, int ldm, uint16_t* out,
                  int ldo, int B, int H, int W, int C8, int act, srk_stream_t stream) {
  SRK_REQUIRE(x && w && scale && shift && out, SRK_E_NULL, "dwconv3x3: null pointer");
  SRK_REQUIRE(B > 0 && H > 0 && W > 0 && C8 > 0 && ldx >= 8 * C8 && ldo >= 8 * C8 && ldx % 8 == 0 && ldo % 8 == 0 && (mul == nullptr || ldm % 8 == 0),
              SRK_E_SHAPE, "dwconv3x3: bad shape / strides (16-byte pieces)");
  SRK_REQUIRE(C8 <= 64, SRK_E_SHAPE, "dwconv3x3: at most 512 channels (got %d)", 8 * C8);
  const int tilesx = (W + DT_W - 1) / DT_W, tilesy = (H + DT_H - 1) / DT_H;       // any H, W: edges are predicated
  const long long nb = (long long)B * tilesx * tilesy;
  SRK_REQUIRE(nb < (1LL << 31), SRK_E_SHAPE, "dwconv3x3: too many tiles");
  hipLaunchKernelGGL(dwconv3x3_tile_kernel, dim3((unsigned)nb, (unsigned)((8 * C8 + DT_CB - 1) / DT_CB)), dim3(256), 0, (hipStream_t)stream, x, ldx, w,
                     scale, shift, mul, ldm, out, ldo, B, H, W, 8 * C8, act, tilesx, tilesy);
  return srk_check_launch("dwconv3x3");
}

int srk_rowln_bf16(const uint16_t* x, int ldx, const float* gamma, const float* beta, uint16_t* out, int ldo, int64_t rows, int C, int CP_out,
                   srk_stream_t stream) {
  SRK_REQUIRE(x && gamma && beta && out, SRK_E_NULL, "rowln: null pointer");
  SRK_REQUIRE(rows > 0 && C > 0 && C <= 512 && CP_out >= C && CP_out <= 512 && CP_out % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, SRK_E_SHAPE,
              "rowln: C=%d (<= 512) CP_out=%d", C, CP_out);
  // few, long-lived workgroups: every thread loads its 16 gamma / beta values once and then walks many rows
  const int span = C > CP_out ? C : CP_out;
  if (span <= 128)
    hipLaunchKernelGGL(rowln_bf16_kernel<16>, dim3(grid_for(rows, 32, 2048)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, out, ldo,
                       (long long)rows, C, CP_out);
  else if (span <= 256)
    hipLaunchKernelGGL(rowln_bf16_kernel<32>, dim3(grid_for(rows, 16, 2048)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, out, ldo,
                       (long long)rows, C, CP_out);
  else
    hipLaunchKernelGGL(rowln_bf16_kernel<64>, dim3(grid_for(rows, 8, 2048)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, out, ldo,
                       (long long)rows, C, CP_out);
  return srk_check_launch("rowln_bf16");
}

static int spatial_gate_impl(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, float b3, const float* b3_dev, int S,
                             float* gate, int64_t rows, int CP, srk_stream_t stream) {
  SRK_REQUIRE(x && W0 && b0 && w3 && gate, SRK_E_NULL, "spatial_gate: null pointer");
  SRK_REQUIRE(rows > 0 && S > 0 && S <= 16 && ldx % 4 == 0, SRK_E_SHAPE, "spatial_gate: S=%d (<= 16)", S);
  const int grid = grid_for(rows, 16, 1024);
#define SG_CASE(NV)                                                                                                                     \
  if (CP == 64 * NV) {                                                                                                                  \
    hipLaunchKernelGGL(spatial_gate_kernel<NV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, W0, b0, w3, b3, b3_dev, S, gate, \
                       (long long)rows);                                                                                                \
    return srk_check_launch("spatial_gate");                                                                                            \
  }
  SG_CASE(1) SG_CASE(2) SG_CASE(3) SG_CASE(4)
#undef SG_CASE
  srk_set_error("spatial_gate: CP=%d unsupported (64/128/192/256)", CP);
  return SRK_E_UNSUPPORTED;
}

int srk_spatial_gate(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, float b3, int S, float* gate, int64_t rows,
                     int CP, srk_stream_t stream) {
  return spatial_gate_impl(x, ldx, W0, b0, w3, b3, nullptr, S, gate, rows, CP, stream);
}

int srk_spatial_gate_dev(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, const float* b3, int S, float* gate,
                         int64_t rows, int CP, srk_stream_t stream) {
  SRK_REQUIRE(b3, SRK_E_NULL, "spatial_gate: null pointer");
  return spatial_gate_impl(x, ldx, W0, b0, w3, 0.f, b3, S, gate, rows, CP, stream);
}

int srk_dual_gate_combine(const uint16_t* a, const uint16_t* b, const float* cgate, const float* tgate, uint16_t* out, int64_t rows,
                          int rows_per_sample, int CP, int tok_gate_on_a, srk_stream_t stream) {
  SRK_REQUIRE(a && b && cgate && tgate && out, SRK_E_NULL, "dual_gate_combine: null pointer");
  SRK_REQUIRE(rows > 0 && rows_per_sample > 0 && rows % rows_per_sample == 0 && CP % 8 == 0, SRK_E_SHAPE, "dual_gate_combine: bad shape");
  hipLaunchKernelGGL(dual_gate_combine_kernel, dim3(grid_for(rows * (CP / 8))), dim3(256), 0, (hipStream_t)stream, a, b, cgate, tgate, out,
                     (long long)rows, rows_per_sample, CP, tok_gate_on_a);
  return srk_check_launch("dual_gate_combine");
}

size_t srk_channel_attention_workspace(int B, int N, int num_heads) {
  if (B <= 0 || N <= 0 || num_heads <= 0) return 0;
  return ((size_t)B * num_heads * ((N + GCH - 1) / GCH) * GSZ + (size_t)B * num_heads * 1024) * sizeof(float);
}

}  // extern "C"

// partial [B][heads][chunks][1088] of x^T y, column norms^2 of x and of y per 256-token chunk (head h at column 32 h of each operand)
int srk_launch_chan_gram(const bf16_t* x, int ldx, const bf16_t* y, int ldy, float* partial, int B, int N, int nH, hipStream_t stream) {
  SRK_REQUIRE(x && y && partial, SRK_E_NULL, "chan_gram: null pointer");
  SRK_REQUIRE(B > 0 && B < 65536 && N > 0 && nH > 0 && nH < 65536 && ldx % 8 == 0 && ldy % 8 == 0, SRK_E_SHAPE, "chan_gram: bad shape (16-byte rows)");
  hipLaunchKernelGGL(chan_gram_mfma_kernel, dim3((N + GCH - 1) / GCH, nH, B), dim3(256), 0, stream, x, ldx, y, ldy, partial, N);
  return srk_check_launch("chan_gram");
}

extern "C" {

int srk_channel_attention_fwd(const uint16_t* qkv, int ldq, int CA, const float* temperature, void* workspace, uint16_t* out, int ldo, int B,
                              int N, int num_heads, int head_dim, srk_stream_t stream) {
  SRK_REQUIRE(qkv && temperature && workspace && out, SRK_E_NULL, "channel_attention: null pointer");
  SRK_REQUIRE(B > 0 && B < 65536 && N > 0 && num_heads > 0 && num_heads <= 8 && head_dim > 0 && head_dim <= 32 && CA == num_heads * 32 &&
                  ldq >= 3 * CA && ldq % 8 == 0 && ldo >= CA && ldo % 4 == 0,
              SRK_E_SHAPE, "channel_attention: bad shape B=%d N=%d heads=%d d=%d CA=%d", B, N, num_heads, head_dim, CA);
  const int nchunk = (N + GCH - 1) / GCH;
  float* partial = static_cast<float*>(workspace);
  float* A = partial + (size_t)B * num_heads * nchunk * GSZ;
  {
    const int rc = srk_launch_chan_gram(qkv, ldq, qkv + CA, ldq, partial, B, N, num_heads, (hipStream_t)stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(chan_attn_finish_kernel, dim3(num_heads, B), dim3(256), 0, (hipStream_t)stream, partial, nchunk, head_dim, temperature, A);
  hipLaunchKernelGGL(chan_apply_kernel, dim3((N + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, qkv, ldq, CA, A, num_heads, out, ldo, N);
  return srk_check_launch("channel_attention");
}

}  // extern "C"
