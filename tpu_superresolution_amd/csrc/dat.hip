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

// one thread = one pixel x 8 channels.  w: fp32 [C8*8][9] (zero rows for pad channels), scale / shift: fp32 [C8*8]
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const bf16_t* __restrict__ mul, int ldm, bf16_t* __restrict__ out, int ldo, int B,
                                                        int H, int W, int C8, int act) {
  const long long total = (long long)B * H * W * C8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % C8);
    const long long pix = i / C8;
    const int xw = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = xw + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (pix + (long long)dy * W + dx) * ldx + cg * 8), v);
        const int tap = (dy + 1) * 3 + dx + 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w[(cg * 8 + e) * 9 + tap], v[e], acc[e]);
      }
    }
    float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (mul) unpack8(*reinterpret_cast<const uint4*>(mul + pix * ldm + cg * 8), m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[e] * scale[cg * 8 + e] + shift[cg * 8 + e];
      if (act == 1) v = gelu_f(v);
      acc[e] = v * m[e];
    }
    *reinterpret_cast<uint4*>(out + pix * ldo + cg * 8) = pack8(acc);
  }
}

// LayerNorm over C channels of a bf16 row slice; one wave per row, lane holds up to 8 values (C <= 512), eps 1e-5
__global__ __launch_bounds__(256) void rowln_bf16_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ out, int ldo, long long rows,
                                                         int C, int CPo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * 8;
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gm[e] = c0 + e < C ? gamma[c0 + e] : 0.f;
    bt[e] = c0 + e < C ? beta[c0 + e] : 0.f;
  }
  const float invC = 1.0f / (float)C;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) unpack8(*reinterpret_cast<const uint4*>(x + r * ldx + c0), v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (c0 + e >= C) v[e] = 0.f;
      s += v[e];
    }
    const float mean = wave_sum64(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = c0 + e < C ? v[e] - mean : 0.f;
      q += v[e] * v[e];
    }
    const float rstd = rsqrtf(wave_sum64(q) * invC + 1e-5f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gm[e] + bt[e];
    if (c0 < CPo) *reinterpret_cast<uint4*>(out + r * ldo + c0) = pack8(v);
  }
}

// per token: g = sigmoid(b3 + sum_s w3[s] * gelu(b0[s] + sum_c W0[s][c] x[c])).  16 lanes per token (4 tokens per wave),
// lane j holds channels 64 i + 4 j .. +3 (the LayerNorm layout).  W0: fp32 [S][CP] (zero at pad channels), S <= 16.
template <int NV>
__global__ __launch_bounds__(256) void spatial_gate_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ W0,
                                                           const float* __restrict__ b0, const float* __restrict__ w3, float b3, int S,
                                                           float* __restrict__ gate, long long rows) {
  constexpr int CP = NV * 64;
  __shared__ float Ws[16 * 256];
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

// grid (chunks, heads, B); qkv bf16 [T][ldq], q at column 32 h, k at CA + 32 h
__global__ __launch_bounds__(256) void chan_gram_partial_kernel(const bf16_t* __restrict__ qkv, int ldq, int CA, float* __restrict__ partial, int N) {
  __shared__ float qs[GCH][33], ks[GCH][33];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int n0 = chunk * GCH;
  const int tid = threadIdx.x;
  for (int i = tid; i < GCH * 8; i += 256) {            // 4-channel pieces: 8 per row, q and k
    const int r = i >> 3, c = (i & 7) * 4;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (n0 + r < N) {
      const bf16_t* row = qkv + ((long long)b * N + n0 + r) * ldq + h * 32 + c;
      const uint2 qu = *reinterpret_cast<const uint2*>(row);
      const uint2 ku = *reinterpret_cast<const uint2*>(row + CA);
      unpack_bf2(qu.x, q0, q1); unpack_bf2(qu.y, q2, q3);
      unpack_bf2(ku.x, k0, k1); unpack_bf2(ku.y, k2, k3);
    }
    qs[r][c] = q0; qs[r][c + 1] = q1; qs[r][c + 2] = q2; qs[r][c + 3] = q3;
    ks[r][c] = k0; ks[r][c + 1] = k1; ks[r][c + 2] = k2; ks[r][c + 3] = k3;
  }
  __syncthreads();
  float* o = partial + (((long long)b * gridDim.y + h) * gridDim.x + chunk) * GSZ;
  for (int p = tid; p < 32 * 32; p += 256) {           // thread owns 4 (i, j) pairs
    const int i = p >> 5, j = p & 31;
    float s = 0.f;
    for (int r = 0; r < GCH; ++r) s = fmaf(qs[r][i], ks[r][j], s);
    o[p] = s;
  }
  if (tid < 64) {
    const int c = tid & 31;
    float s = 0.f;
    if (tid < 32) for (int r = 0; r < GCH; ++r) s = fmaf(qs[r][c], qs[r][c], s);
    else for (int r = 0; r < GCH; ++r) s = fmaf(ks[r][c], ks[r][c], s);
    o[1024 + tid] = s;
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
  if (tid < 32) {
    const int i = tid;
    float row[32];
    float mx = -3.0e38f;
    const float qn = fmaxf(sqrtf(G[1024 + i]), 1e-12f);                   // F.normalize: x / max(||x||, eps)
    for (int j = 0; j < 32; ++j) {
      const float kn = fmaxf(sqrtf(G[1024 + 32 + j]), 1e-12f);
      row[j] = j < d && i < d ? G[i * 32 + j] / (qn * kn) * temperature[h] : -3.0e38f;
      mx = fmaxf(mx, row[j]);
    }
    float sum = 0.f;
    for (int j = 0; j < 32; ++j) {
      row[j] = (j < d && i < d) ? __expf(row[j] - mx) : 0.f;
      sum += row[j];
    }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int j = 0; j < 32; ++j) out[i * 32 + j] = row[j] * inv;
  }
}

// out[n][32 h + i] = sum_j A[b][h][i][j] v[n][32 h + j];  workgroup = 64 tokens of one sample, thread = (token, head) x 8 outputs
__global__ __launch_bounds__(256) void chan_apply_kernel(const bf16_t* __restrict__ qkv, int ldq, int CA, const float* __restrict__ A, int nH,
                                                         bf16_t* __restrict__ out, int ldo, int N) {
  extern __shared__ float sm[];
  float* As = sm;                      // [nH][32][33]
  float* vs = sm + nH * 32 * 33;       // [64][CA + 1]
  const int b = blockIdx.y, n0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int i = tid; i < nH * 1024; i += 256) As[(i >> 5) * 33 + (i & 31)] = A[(long long)b * nH * 1024 + i];
  for (int i = tid; i < 64 * (CA / 4); i += 256) {
    const int r = i / (CA / 4), c = (i - r * (CA / 4)) * 4;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (n0 + r < N) {
      const uint2 u = *reinterpret_cast<const uint2*>(qkv + ((long long)b * N + n0 + r) * ldq + 2 * CA + c);
      unpack_bf2(u.x, v0, v1); unpack_bf2(u.y, v2, v3);
    }
    float* d = vs + r * (CA + 1) + c;
    d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
  }
  __syncthreads();
  const int groups = CA / 8;            // 8-channel output groups per token
  for (int w = tid; w < 64 * groups; w += 256) {
    const int r = w / groups, g8 = w - r * groups;
    if (n0 + r >= N) continue;
    const int h = g8 >> 2, i0 = (g8 & 3) * 8;
    const float* vrow = vs + r * (CA + 1) + h * 32;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* arow = As + (h * 32 + i0 + e) * 33;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) s = fmaf(arow[j], vrow[j], s);
      o[e] = s;
    }
    *reinterpret_cast<uint4*>(out + ((long long)b * N + n0 + r) * ldo + g8 * 8) = pack8(o);
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
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for((long long)B * H * W * C8)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, scale, shift, mul,
                     ldm, out, ldo, B, H, W, C8, act);
  return srk_check_launch("dwconv3x3");
}

int srk_rowln_bf16(const uint16_t* x, int ldx, const float* gamma, const float* beta, uint16_t* out, int ldo, int64_t rows, int C, int CP_out,
                   srk_stream_t stream) {
  SRK_REQUIRE(x && gamma && beta && out, SRK_E_NULL, "rowln: null pointer");
  SRK_REQUIRE(rows > 0 && C > 0 && C <= 512 && CP_out >= C && CP_out <= 512 && CP_out % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, SRK_E_SHAPE,
              "rowln: C=%d (<= 512) CP_out=%d", C, CP_out);
  hipLaunchKernelGGL(rowln_bf16_kernel, dim3(grid_for(rows, 4, 16384)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, out, ldo,
                     (long long)rows, C, CP_out);
  return srk_check_launch("rowln_bf16");
}

int srk_spatial_gate(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, float b3, int S, float* gate, int64_t rows,
                     int CP, srk_stream_t stream) {
  SRK_REQUIRE(x && W0 && b0 && w3 && gate, SRK_E_NULL, "spatial_gate: null pointer");
  SRK_REQUIRE(rows > 0 && S > 0 && S <= 16 && ldx % 4 == 0, SRK_E_SHAPE, "spatial_gate: S=%d (<= 16)", S);
  const int grid = grid_for(rows, 16, 8192);
#define SG_CASE(NV)                                                                                                                     \
  if (CP == 64 * NV) {                                                                                                                  \
    hipLaunchKernelGGL(spatial_gate_kernel<NV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, W0, b0, w3, b3, S, gate, (long long)rows); \
    return srk_check_launch("spatial_gate");                                                                                            \
  }
  SG_CASE(1) SG_CASE(2) SG_CASE(3) SG_CASE(4)
#undef SG_CASE
  srk_set_error("spatial_gate: CP=%d unsupported (64/128/192/256)", CP);
  return SRK_E_UNSUPPORTED;
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

int srk_channel_attention_fwd(const uint16_t* qkv, int ldq, int CA, const float* temperature, void* workspace, uint16_t* out, int ldo, int B,
                              int N, int num_heads, int head_dim, srk_stream_t stream) {
  SRK_REQUIRE(qkv && temperature && workspace && out, SRK_E_NULL, "channel_attention: null pointer");
  SRK_REQUIRE(B > 0 && B < 65536 && N > 0 && num_heads > 0 && num_heads <= 8 && head_dim > 0 && head_dim <= 32 && CA == num_heads * 32 &&
                  ldq >= 3 * CA && ldq % 4 == 0 && ldo >= CA && ldo % 8 == 0,
              SRK_E_SHAPE, "channel_attention: bad shape B=%d N=%d heads=%d d=%d CA=%d", B, N, num_heads, head_dim, CA);
  const int nchunk = (N + GCH - 1) / GCH;
  float* partial = static_cast<float*>(workspace);
  float* A = partial + (size_t)B * num_heads * nchunk * GSZ;
  hipLaunchKernelGGL(chan_gram_partial_kernel, dim3(nchunk, num_heads, B), dim3(256), 0, (hipStream_t)stream, qkv, ldq, CA, partial, N);
  hipLaunchKernelGGL(chan_attn_finish_kernel, dim3(num_heads, B), dim3(256), 0, (hipStream_t)stream, partial, nchunk, head_dim, temperature, A);
  const size_t lds = ((size_t)num_heads * 32 * 33 + 64 * (size_t)(CA + 1)) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&chan_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      srk_set_error("channel_attention: cannot reserve LDS");
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  hipLaunchKernelGGL(chan_apply_kernel, dim3((N + 63) / 64, B), dim3(256), lds, (hipStream_t)stream, qkv, ldq, CA, A, num_heads, out, ldo, N);
  return srk_check_launch("channel_attention");
}

}  // extern "C"
