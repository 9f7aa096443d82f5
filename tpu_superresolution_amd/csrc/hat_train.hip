// Training-side building blocks for the host-orchestrated models (HAT: tpu_superresolution_amd/hat_arch.py) and their C ABI.
//
// New kernels (reference hat_arch.py:41-75 CAB / ChannelAttention, HAB.forward :322 `x = shortcut + attn_x + conv_x * conv_scale`):
//   cab_pool_bwd_partial_kernel   per (sample, 256-token chunk): sum_t c2[t][c] and sum_t g[t][c] c2[t][c]  (fixed-order partials)
//   channel_gate_bwd_kernel       one workgroup per sample: the squeeze-excite MLP forwards again from the pooled mean, then
//                                 d gate -> d z2 -> dW2, db2 -> d relu -> dW1, db1 -> d mean / HW
//   cab_dconv_kernel              d c2[t][c] = bf16(g[t][c] gate[b][c] + dmean[b][c])
// The rest of this file exposes launchers that the SwinIR executor uses internally (LayerNorm backward, the small image-head
// convolution gradients, element-wise helpers, the conv weight gradient with a pixel-shuffled output gradient) so that a Python
// orchestration can run a whole backward pass through the C ABI.
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm.h"
#include "kernels.h"
#include "wgrad.h"

size_t srk_win256_attn_bwd_scratch(int B, int H, int W, int nH, int CA, int table_rows, int overlap);
int srk_launch_win256_attn_bwd(const bf16_t* qkv, int ldq, int CA, const float* table, int table_rows, const bf16_t* dout, int ldo,
                               bf16_t* dqkv, float* dtable, void* scratch, int B, int H, int W, int sy, int sx, int nH, float scale,
                               int overlap, hipStream_t stream);

namespace {

constexpr int TM_ROWS = 256;     // tokens per workgroup of the pooling passes (as hat.hip)

// partial[b][chunk][0][c] = sum c2, partial[b][chunk][1][c] = sum g * c2 over the chunk's tokens
__global__ __launch_bounds__(256) void cab_pool_bwd_partial_kernel(const bf16_t* __restrict__ c2, const float* __restrict__ g,
                                                                  float* __restrict__ partial, int HW, int CP) {
  __shared__ float red[2][8][256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c8 = CP / 8;
  const int tid = threadIdx.x;
  const int piece = tid % 32, rg = tid / 32;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, accg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (piece < c8) {
    const int r0 = chunk * TM_ROWS;
    for (int r = r0 + rg; r < min(r0 + TM_ROWS, HW); r += 8) {
      const long long o = ((long long)b * HW + r) * CP + piece * 8;
      const uint4 v = *reinterpret_cast<const uint4*>(c2 + o);
      const float4 g0 = *reinterpret_cast<const float4*>(g + o), g1 = *reinterpret_cast<const float4*>(g + o + 4);
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack_bf2(u[e], lo, hi);
        acc[2 * e] += lo;
        acc[2 * e + 1] += hi;
        accg[2 * e] += lo * gv[2 * e];
        accg[2 * e + 1] += hi * gv[2 * e + 1];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][rg][piece * 8 + e] = acc[e];
    red[1][rg][piece * 8 + e] = accg[e];
  }
  __syncthreads();
  if (tid < CP) {
    float s = 0.f, sg = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += red[0][k][tid];
      sg += red[1][k][tid];
    }
    float* dst = partial + (((long long)b * gridDim.x + chunk) * 2) * CP;
    dst[tid] = s;
    dst[CP + tid] = sg;
  }
}

// gate = out_scale * sigmoid(W2 relu(W1 m + b1) + b2)  (hat_arch.py:50-54); dgate[c] = sum_t g c2.  One workgroup per sample.
__global__ __launch_bounds__(256) void channel_gate_bwd_kernel(const float* __restrict__ partial, int nchunk, int HW, int C, int CP, int S,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2, float out_scale,
                                                               float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                               float* __restrict__ db2, float* __restrict__ dmean) {
  __shared__ float mean[256], dgate[256], dz2[256];
  __shared__ float z1[64], a[64], dz1[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < CP) {
    float s = 0.f, sg = 0.f;
    for (int k = 0; k < nchunk; ++k) {
      const float* src = partial + (((long long)b * nchunk + k) * 2) * CP;
      s += src[tid];
      sg += src[CP + tid];
    }
    mean[tid] = s / (float)HW;
    dgate[tid] = sg;
  }
  __syncthreads();
  if (tid < S) {
    float v = b1[tid];
    for (int c = 0; c < C; ++c) v += w1[tid * C + c] * mean[c];
    z1[tid] = v;
    a[tid] = v > 0.f ? v : 0.f;
  }
  __syncthreads();
  if (tid < CP) {
    float d = 0.f;
    if (tid < C) {
      float v = b2[tid];
      for (int s = 0; s < S; ++s) v += w2[tid * S + s] * a[s];
      const float sg = 1.0f / (1.0f + __expf(-v));
      d = dgate[tid] * out_scale * sg * (1.0f - sg);
      atomicAdd(db2 + tid, d);
      for (int s = 0; s < S; ++s) atomicAdd(dw2 + tid * S + s, d * a[s]);
    }
    dz2[tid] = d;
  }
  __syncthreads();
  if (tid < S) {
    float da = 0.f;
    for (int c = 0; c < C; ++c) da += w2[c * S + tid] * dz2[c];
    const float d = z1[tid] > 0.f ? da : 0.f;
    dz1[tid] = d;
    atomicAdd(db1 + tid, d);
  }
  __syncthreads();
  for (int i = tid; i < S * C; i += 256) atomicAdd(dw1 + i, dz1[i / C] * mean[i % C]);
  if (tid < CP) {
    float dm = 0.f;
    if (tid < C)
      for (int s = 0; s < S; ++s) dm += w1[s * C + tid] * dz1[s];
    dmean[(long long)b * CP + tid] = dm / (float)HW;
  }
}

// d c2[t][c] = bf16(g[t][c] * gate[b][c] + dmean[b][c]); pad columns: gate == dmean == 0
__global__ __launch_bounds__(256) void cab_dconv_kernel(const float* __restrict__ g, const float* __restrict__ gate, const float* __restrict__ dmean,
                                                        bf16_t* __restrict__ out, long long rows, int rows_per_sample, int CP) {
  const long long n4 = rows * (CP / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long t = i / (CP / 4);
    const int c = (int)(i - t * (CP / 4)) * 4;
    const long long b = t / rows_per_sample;
    const float4 gv = *reinterpret_cast<const float4*>(g + t * CP + c);
    const float4 gt = *reinterpret_cast<const float4*>(gate + b * CP + c);
    const float4 dm = *reinterpret_cast<const float4*>(dmean + b * CP + c);
    *reinterpret_cast<uint2*>(out + t * CP + c) = pack_bf4(gv.x * gt.x + dm.x, gv.y * gt.y + dm.y, gv.z * gt.z + dm.z, gv.w * gt.w + dm.w);
  }
}

// dst[m][c] = bf16(src[m][c] * f[sample(m)])
__global__ __launch_bounds__(256) void rowscale_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const float* __restrict__ f,
                                                            long long rows, int rows_per_sample, int CP) {
  const long long n4 = rows * (CP / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long t = i / (CP / 4);
    const float s = f[t / rows_per_sample];
    const uint2 v = reinterpret_cast<const uint2*>(src)[i];
    float a, b, c, d;
    unpack_bf2(v.x, a, b);
    unpack_bf2(v.y, c, d);
    reinterpret_cast<uint2*>(dst)[i] = pack_bf4(a * s, b * s, c * s, d * s);
  }
}

// a[i] = b[i] + c[i] (fp32) -- fan-in of two gradient streams
__global__ __launch_bounds__(256) void add3_f32_kernel(float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(b)[i], y = reinterpret_cast<const float4*>(c)[i];
    reinterpret_cast<float4*>(a)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace

extern "C" {

size_t srk_cab_bwd_workspace(int B, int HW, int CP) { return (size_t)B * ((HW + TM_ROWS - 1) / TM_ROWS) * 2 * CP * sizeof(float); }

int srk_cab_bwd(const uint16_t* conv, const float* g, const float* gate, void* workspace, const float* w1, const float* b1, const float* w2,
                const float* b2, float out_scale, float* dw1, float* db1, float* dw2, float* db2, float* dmean, uint16_t* d_conv, int B, int HW,
                int C, int CP, int S, srk_stream_t stream) {
  SRK_REQUIRE(conv && g && gate && workspace && w1 && b1 && w2 && b2 && dw1 && db1 && dw2 && db2 && dmean && d_conv, SRK_E_NULL,
              "cab_bwd: null pointer");
  SRK_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0 && C <= CP && CP % 64 == 0 && CP <= 256 && S > 0 && S <= 64, SRK_E_SHAPE,
              "cab_bwd: bad shape B=%d HW=%d C=%d CP=%d S=%d", B, HW, C, CP, S);
  const int nchunk = (HW + TM_ROWS - 1) / TM_ROWS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cab_pool_bwd_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, conv, g, static_cast<float*>(workspace), HW, CP);
  hipLaunchKernelGGL(channel_gate_bwd_kernel, dim3(B), dim3(256), 0, st, static_cast<const float*>(workspace), nchunk, HW, C, CP, S, w1, b1, w2,
                     b2, out_scale, dw1, db1, dw2, db2, dmean);
  const long long rows = (long long)B * HW;
  const long long n4 = rows * (CP / 4);
  const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
  hipLaunchKernelGGL(cab_dconv_kernel, dim3(grid), dim3(256), 0, st, g, gate, dmean, d_conv, rows, HW, CP);
  return srk_check_launch("cab_bwd");
}

size_t srk_win256_attention_bwd_scratch(int B, int H, int W, int num_heads, int CA, int table_rows, int overlap) {
  return srk_win256_attn_bwd_scratch(B, H, W, num_heads, CA, table_rows, overlap);
}

int srk_win256_attention_bwd(const uint16_t* qkv, int ldq, int CA, const float* table, int table_rows, const uint16_t* d_out, int ldo,
                             uint16_t* d_qkv, float* d_table, void* scratch, int B, int H, int W, int shift_y, int shift_x, int num_heads,
                             float scale, int overlap, srk_stream_t stream) {
  return srk_launch_win256_attn_bwd(qkv, ldq, CA, table, table_rows, d_out, ldo, d_qkv, d_table, scratch, B, H, W, shift_y, shift_x, num_heads,
                                    scale, overlap, (hipStream_t)stream);
}

int srk_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* gx,
                      uint16_t* gx_bf16, float* dgamma, float* dbeta, int rows, int C, int CP, int accumulate, srk_stream_t stream) {
  SRK_REQUIRE(dy && x && mean && rstd && gamma && gx && dgamma && dbeta, SRK_E_NULL, "layernorm_bwd: null pointer");
  SRK_REQUIRE(rows > 0 && C > 0 && C <= CP && CP % 64 == 0 && CP <= 256, SRK_E_SHAPE, "layernorm_bwd: rows=%d C=%d CP=%d", rows, C, CP);
  return srk_launch_ln_bwd(dy, x, mean, rstd, gamma, gx, gx_bf16, dgamma, dbeta, rows, C, CP, nullptr, 0, 0, 0, accumulate, nullptr, rows,
                           (hipStream_t)stream);
}

int srk_add_f32_bf16(float* a, const float* b, uint16_t* ab_bf16, int64_t n, srk_stream_t stream) {
  SRK_REQUIRE(a && b && ab_bf16 && n > 0 && n % 4 == 0, SRK_E_SHAPE, "add_f32_bf16: bad arguments");
  return srk_launch_add_f32_bf16(a, b, ab_bf16, n, (hipStream_t)stream);
}

int srk_add_bf16_into_f32(float* a, const uint16_t* b, int64_t n, srk_stream_t stream) {
  SRK_REQUIRE(a && b && n > 0 && n % 4 == 0, SRK_E_SHAPE, "add_bf16_into_f32: bad arguments");
  return srk_launch_add_bf16_into_f32(a, b, n, (hipStream_t)stream);
}

int srk_add_f32(float* out, const float* a, const float* b, int64_t n, srk_stream_t stream) {
  SRK_REQUIRE(out && a && b && n > 0 && n % 4 == 0, SRK_E_SHAPE, "add_f32: bad arguments");
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
  hipLaunchKernelGGL(add3_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, a, b, n4);
  return srk_check_launch("add_f32");
}

int srk_rowscale_bf16(const uint16_t* src, uint16_t* dst, const float* f, int64_t rows, int rows_per_sample, int CP, srk_stream_t stream) {
  SRK_REQUIRE(src && dst && f && rows > 0 && rows_per_sample > 0 && CP % 4 == 0, SRK_E_SHAPE, "rowscale_bf16: bad arguments");
  const long long n4 = rows * (CP / 4);
  const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
  hipLaunchKernelGGL(rowscale_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, f, (long long)rows, rows_per_sample, CP);
  return srk_check_launch("rowscale_bf16");
}

int srk_img_grad_prep(const float* d_pred, float* gy, int B, int Cimg, int Hc, int Wc, int H, int W, int r, int CoP, float inv_range,
                      srk_stream_t stream) {
  SRK_REQUIRE(d_pred && gy, SRK_E_NULL, "img_grad_prep: null pointer");
  return srk_launch_img_grad_prep(d_pred, gy, B, Cimg, Hc, Wc, H, W, r, CoP, inv_range, (hipStream_t)stream);
}

int srk_smallconv_wgrad(const uint16_t* x, const float* gy, float* dw, float* db, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                        srk_stream_t stream) {
  SRK_REQUIRE(x && gy && dw && db, SRK_E_NULL, "smallconv_wgrad: null pointer");
  return srk_launch_smallconv_wgrad(x, gy, dw, db, B, H, W, Cin, CinP, Co, CoP, (hipStream_t)stream);
}

int srk_smallconv_dgrad(const float* gy, const float* weight, uint16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                        srk_stream_t stream) {
  SRK_REQUIRE(gy && weight && dx, SRK_E_NULL, "smallconv_dgrad: null pointer");
  return srk_launch_smallconv_dgrad(gy, weight, dx, B, H, W, Cin, CinP, Co, CoP, (hipStream_t)stream);
}

int srk_stem_wgrad(const float* img4, const float* gy, float* dw, float* db, int B, int H, int W, int Cin, int C, int CP, srk_stream_t stream) {
  SRK_REQUIRE(img4 && gy && dw && db, SRK_E_NULL, "stem_wgrad: null pointer");
  return srk_launch_stem_wgrad(img4, gy, dw, db, B, H, W, Cin, C, CP, (hipStream_t)stream);
}

int srk_conv3x3_wgrad_ps_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int B, int H, int W, int CinP, int N, int r, int Cs,
                              srk_stream_t stream) {
  SRK_REQUIRE(y && x && dw, SRK_E_NULL, "conv3x3_wgrad_ps: null pointer");
  WgradParams p = {};
  p.Y = y; p.ldy = N; p.X = x; p.ldx = CinP; p.M = B * H * W; p.N = N; p.K = CinP; p.dW = dw; p.ldw = 9 * CinP; p.db = db;
  p.conv = 1; p.B = B; p.H = H; p.W = W; p.r = r; p.Cs = Cs;
  return srk_launch_wgrad(p, (hipStream_t)stream);
}

int srk_mlp_fused_fwd_train(const uint16_t* xn, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* res,
                            float* out, uint16_t* out_bf16, uint16_t* u_out, uint16_t* h_out, uint16_t* xn_next, float* xn_mean,
                            float* xn_rstd, const float* xn_gamma, const float* xn_beta, int xn_C, const float* rowscale, int rows_per_sample,
                            int M, srk_stream_t stream) {
  SRK_REQUIRE(xn && w1 && w2 && res && out && u_out && h_out, SRK_E_NULL, "mlp_fused_train: null pointer");
  GemmParams p = {};
  p.A = xn; p.lda = 192; p.Wt = w1; p.K = 192; p.bias = b1; p.W2 = w2; p.bias2 = b2; p.HP = 384; p.M = M; p.N = 192; p.ldo = 192;
  p.res = res; p.outf = out; p.outb = out_bf16; p.u_out = u_out; p.h_out = h_out;
  p.rowscale = rowscale; p.rows_per_sample = rows_per_sample;
  if (xn_next) {
    SRK_REQUIRE(xn_mean && xn_rstd && xn_gamma && xn_beta, SRK_E_NULL, "mlp_fused_train: fused LayerNorm needs mean / rstd / gamma / beta");
    p.xn_out = xn_next; p.xn_mean = xn_mean; p.xn_rstd = xn_rstd; p.xn_gamma = xn_gamma; p.xn_beta = xn_beta; p.xn_C = xn_C;
  }
  p.flops = 4.0 * M * 180.0 * 360.0;
  const int rc = srk_launch_mlp_fused(p, (hipStream_t)stream);
  if (rc == SRK_NOT_COVERED) {
    srk_set_error("mlp_fused_train: shape not covered (needs C 180/192, hidden 360/384, M %% 64 == 0 and M >= 64 * #CUs; got M=%d)", M);
    return SRK_E_UNSUPPORTED;
  }
  return rc;
}

// Backward of Mlp + residual + the LayerNorm in front of it as ONE kernel (csrc/gemm_stream.hip: mlp_fused_bwd_kernel), for the
// host-orchestrated backward passes: d u = (g W2) * gelu'(u) (written: the fc1 weight gradient reads it), d xn = d u W1, LayerNorm
// backward of d xn through (ln_x, mean, rstd, gamma): gx += d x (fp32 gradient stream, in place), gxb = bf16(gx * rowscale[sample]),
// d_gamma / d_beta ACCUMULATED.  g bf16 [M][192], w2t bf16 [384][192] (= fc2.weight^T packed), u bf16 [M][384], w1t bf16 [192][384].
int srk_mlp_fused_bwd(const uint16_t* g, const uint16_t* w2t, const uint16_t* u, uint16_t* du_out, const uint16_t* w1t, const float* ln_x,
                      const float* ln_mean, const float* ln_rstd, const float* ln_gamma, float* gx, uint16_t* gxb, const float* rowscale,
                      int rows_per_sample, float* d_gamma, float* d_beta, int C, int M, srk_stream_t stream) {
  SRK_REQUIRE(g && w2t && u && du_out && w1t && ln_x && ln_mean && ln_rstd && ln_gamma && gx && d_gamma && d_beta, SRK_E_NULL,
              "mlp_fused_bwd: null pointer");
  GemmParams p = {};
  p.A = g; p.lda = 192; p.Wt = w2t; p.K = 192; p.HP = 384; p.aux = u; p.u_out = du_out; p.W2 = w1t; p.M = M; p.N = 192; p.ldo = 192;
  p.outf = gx; p.outb = gxb; p.rowscale = rowscale; p.rows_per_sample = rows_per_sample;
  p.ln_x = ln_x; p.ln_mean = ln_mean; p.ln_rstd = ln_rstd; p.ln_gamma = ln_gamma; p.ln_dgamma = d_gamma; p.ln_dbeta = d_beta; p.ln_C = C;
  p.ln_rows_window = 0; p.ln_stats_by_m = 0; p.ln_out_window = 0;
  p.flops = 8.0 * M * 180.0 * 360.0;
  const int rc = srk_launch_mlp_fused_bwd(p, (hipStream_t)stream);
  if (rc == SRK_NOT_COVERED) {
    srk_set_error("mlp_fused_bwd: shape not covered (needs C 180/192, hidden 360/384, M %% 64 == 0, M >= 64 * #CUs, rows_per_sample %% 64 == 0; "
                  "got M=%d)", M);
    return SRK_E_UNSUPPORTED;
  }
  return rc;
}

}  // extern "C"
