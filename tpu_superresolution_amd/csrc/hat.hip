// HAT building blocks beyond the SwinIR kernels (reference hat_arch.py) and the generic GEMM / conv entry point the Python
// orchestration of HAT uses.  Token stream layout as everywhere: fp32 / bf16 [B*H*W][CP], CP = channels padded to 64.
//
//   token_mean_kernel     AdaptiveAvgPool2d(1) of ChannelAttention (hat_arch.py:50): per-sample mean over the H*W tokens of a
//                         bf16 [T][CP] map -> fp32 [B][CP] (fixed-order partial sums, no atomics: reproducible)
//   channel_gate_kernel   the squeeze-excite MLP of ChannelAttention (:51-54): sigmoid(W2 relu(W1 m + b1) + b2) * conv_scale
//   cab_add_ln_kernel     HAB.forward :322: x = (shortcut + attn_x) + conv_x * gate  [fp32, in place on the proj epilogue's output]
//                         fused with norm2 (:323) -> bf16 operand of the MLP
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace {

constexpr int TM_ROWS = 256;     // tokens per workgroup of the first pooling pass

// pass 1: partial[b][chunk][c] = sum over TM_ROWS tokens; 256 threads: thread = (row group of 4 tokens..., 8-channel piece)
__global__ __launch_bounds__(256) void token_mean_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int HW, int CP) {
  __shared__ float red[8][256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c8 = CP / 8;                    // 16-byte pieces per row (24 for CP 192)
  const int tid = threadIdx.x;
  const int piece = tid % 32, rg = tid / 32;           // up to 32 pieces per row; 8 row groups
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (piece < c8) {
    const int r0 = chunk * TM_ROWS;
    for (int r = r0 + rg; r < min(r0 + TM_ROWS, HW); r += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + ((long long)b * HW + r) * CP + piece * 8);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack_bf2(u[e], lo, hi);
        acc[2 * e] += lo;
        acc[2 * e + 1] += hi;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rg][piece * 8 + e] = acc[e];
  __syncthreads();
  if (tid < CP) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][tid];
    partial[((long long)b * gridDim.x + chunk) * CP + tid] = s;
  }
}

// pass 2 + the gate MLP: one workgroup per sample.  w1 [S][C], b1 [S], w2 [C][S], b2 [C]  (1x1 convs, fp32 parameters)
__global__ __launch_bounds__(256) void channel_gate_kernel(const float* __restrict__ partial, int nchunk, int HW, int C, int CP, int S,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, float out_scale,
                                                           float* __restrict__ gate, int act) {
  __shared__ float mean[256];
  __shared__ float z[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < CP) {
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += partial[((long long)b * nchunk + k) * CP + tid];
    mean[tid] = s / (float)HW;
  }
  __syncthreads();
  if (tid < S) {
    float a = b1[tid];
    for (int c = 0; c < C; ++c) a += w1[tid * C + c] * mean[c];
    z[tid] = act == 1 ? gelu_f(a) : (a > 0.f ? a : 0.f);
  }
  __syncthreads();
  if (tid < CP) {
    float g = 0.f;
    if (tid < C) {
      float a = b2[tid];
      for (int s = 0; s < S; ++s) a += w2[tid * S + s] * z[s];
      g = out_scale / (1.0f + __expf(-a));
    }
    gate[(long long)b * CP + tid] = g;
  }
}

// x[t][c] += conv[t][c] * gate[b][c]; optional LayerNorm of the new row -> bf16.  16 lanes per row, NV float4 per lane.
template <int NV>
__global__ __launch_bounds__(256) void cab_add_ln_kernel(float* __restrict__ x, const bf16_t* __restrict__ conv, const float* __restrict__ gate,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ xn, long long rows, int rows_per_sample, int C) {
  constexpr int CP = NV * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, sub = lane >> 4;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 64 * i + 4 * j;
    gm[i] = make_float4(c < C ? gamma[c] : 0.f, c + 1 < C ? gamma[c + 1] : 0.f, c + 2 < C ? gamma[c + 2] : 0.f, c + 3 < C ? gamma[c + 3] : 0.f);
    bt[i] = make_float4(c < C ? beta[c] : 0.f, c + 1 < C ? beta[c + 1] : 0.f, c + 2 < C ? beta[c + 2] : 0.f, c + 3 < C ? beta[c + 3] : 0.f);
  }
  const float invC = 1.0f / (float)C;
  for (long long m = ((long long)blockIdx.x * 4 + wave) * 4 + sub; m < rows; m += (long long)gridDim.x * 16) {
    const long long b = m / rows_per_sample;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long long o = m * CP + 64 * i + 4 * j;
      v[i] = *reinterpret_cast<const float4*>(x + o);
      const uint2 cu = *reinterpret_cast<const uint2*>(conv + o);
      const float4 gv = *reinterpret_cast<const float4*>(gate + b * CP + 64 * i + 4 * j);
      float c0, c1, c2, c3;
      unpack_bf2(cu.x, c0, c1);
      unpack_bf2(cu.y, c2, c3);
      v[i].x += c0 * gv.x; v[i].y += c1 * gv.y; v[i].z += c2 * gv.z; v[i].w += c3 * gv.w;     // pad columns: gate == 0
      *reinterpret_cast<float4*>(x + o) = v[i];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (xn == nullptr) continue;
    const float mean = wave_sum16(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 64 * i + 4 * j;
      const float d0 = c < C ? v[i].x - mean : 0.f, d1 = c + 1 < C ? v[i].y - mean : 0.f;
      const float d2 = c + 2 < C ? v[i].z - mean : 0.f, d3 = c + 3 < C ? v[i].w - mean : 0.f;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      v[i] = make_float4(d0, d1, d2, d3);
    }
    const float rstd = rsqrtf(wave_sum16(q) * invC + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      *reinterpret_cast<uint2*>(xn + m * CP + 64 * i + 4 * j) =
          pack_bf4(v[i].x * rstd * gm[i].x + bt[i].x, v[i].y * rstd * gm[i].y + bt[i].y, v[i].z * rstd * gm[i].z + bt[i].z,
                   v[i].w * rstd * gm[i].w + bt[i].w);
  }
}

}  // namespace

extern "C" {

size_t srk_channel_gate_workspace(int B, int HW, int CP) { return (size_t)B * ((HW + TM_ROWS - 1) / TM_ROWS) * CP * sizeof(float); }

int srk_channel_gate_act(const uint16_t* x, void* workspace, const float* w1, const float* b1, const float* w2, const float* b2, float out_scale,
                         float* gate, int B, int HW, int C, int CP, int S, int act, srk_stream_t stream);
int srk_channel_gate(const uint16_t* x, void* workspace, const float* w1, const float* b1, const float* w2, const float* b2, float out_scale,
                     float* gate, int B, int HW, int C, int CP, int S, srk_stream_t stream) {
  return srk_channel_gate_act(x, workspace, w1, b1, w2, b2, out_scale, gate, B, HW, C, CP, S, 0, stream);
}

int srk_channel_gate_act(const uint16_t* x, void* workspace, const float* w1, const float* b1, const float* w2, const float* b2, float out_scale,
                         float* gate, int B, int HW, int C, int CP, int S, int act, srk_stream_t stream) {
  SRK_REQUIRE(x && workspace && w1 && b1 && w2 && b2 && gate, SRK_E_NULL, "channel_gate: null pointer");
  SRK_REQUIRE(B > 0 && B < 65536 && HW > 0 && C > 0 && C <= CP && CP % 64 == 0 && CP <= 256 && S > 0 && S <= 64, SRK_E_SHAPE,
              "channel_gate: bad shape B=%d HW=%d C=%d CP=%d S=%d", B, HW, C, CP, S);
  const int nchunk = (HW + TM_ROWS - 1) / TM_ROWS;
  hipLaunchKernelGGL(token_mean_partial_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, static_cast<float*>(workspace), HW, CP);
  hipLaunchKernelGGL(channel_gate_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace), nchunk, HW, C, CP, S,
                     w1, b1, w2, b2, out_scale, gate, act);
  return srk_check_launch("channel_gate");
}

int srk_cab_add_ln(float* x, const uint16_t* conv, const float* gate, const float* gamma, const float* beta, uint16_t* xn, int64_t rows,
                   int rows_per_sample, int C, int CP, srk_stream_t stream) {
  SRK_REQUIRE(x && conv && gate, SRK_E_NULL, "cab_add_ln: null pointer");
  SRK_REQUIRE(xn == nullptr || (gamma && beta), SRK_E_NULL, "cab_add_ln: LayerNorm output without gamma / beta");
  SRK_REQUIRE(rows > 0 && rows_per_sample > 0 && rows % rows_per_sample == 0 && C > 0 && C <= CP, SRK_E_SHAPE, "cab_add_ln: bad shape");
  const int grid = (int)((rows + 15) / 16 < 8192 ? (rows + 15) / 16 : 8192);
  const float* gm = gamma ? gamma : gate;      // never dereferenced beyond C when xn is null (values unused)
  const float* bt = beta ? beta : gate;
#define CAB_CASE(NV)                                                                                                                    \
  if (CP == 64 * NV) {                                                                                                                  \
    hipLaunchKernelGGL(cab_add_ln_kernel<NV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, conv, gate, gm, bt, xn, (long long)rows, \
                       rows_per_sample, C);                                                                                             \
    return srk_check_launch("cab_add_ln");                                                                                              \
  }
  CAB_CASE(1) CAB_CASE(2) CAB_CASE(3) CAB_CASE(4)
#undef CAB_CASE
  srk_set_error("cab_add_ln: CP=%d unsupported (64/128/192/256)", CP);
  return SRK_E_UNSUPPORTED;
}

int srk_win256_attention_fwd(const uint16_t* qkv, int ldq, int CA, const float* bias, int table_rows, uint16_t* out, int ldo, int B, int H,
                             int W, int wh, int ww, int shift_y, int shift_x, int num_heads, float scale, int overlap, srk_stream_t stream) {
  return srk_launch_win256_attn_fwd(qkv, ldq, CA, bias, table_rows, out, ldo, B, H, W, wh, ww, shift_y, shift_x, num_heads, scale, overlap,
                                    (hipStream_t)stream);
}

int srk_win_attention_fwd_padded(const uint16_t* qkv, int ldq, int CA, const float* bias, int table_rows, uint16_t* out, int ldo, int B, int H,
                                 int W, int Hp, int Wp, int wh, int ww, int shift_y, int shift_x, int num_heads, float scale, int overlap,
                                 srk_stream_t stream) {
  return srk_launch_win_attn_fwd_padded(qkv, ldq, CA, bias, table_rows, out, ldo, B, H, W, Hp, Wp, wh, ww, shift_y, shift_x, num_heads, scale,
                                        overlap, (hipStream_t)stream);
}

int srk_swin_block_fwd(const float* x, float* y, uint16_t* y_bf16, const float* norm1_w, const float* norm1_b, const float* norm2_w,
                       const float* norm2_b, const uint16_t* wqkv, const float* bqkv, const uint16_t* wproj, const float* bproj,
                       const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* bias_dense, float scale,
                       int C, int num_heads, int head_dim, int hidden, int B, int H, int W, int shift, srk_stream_t stream) {
  SRK_REQUIRE(x && y && norm1_w && norm1_b && norm2_w && norm2_b && wqkv && wproj && w1 && w2 && bias_dense, SRK_E_NULL,
              "swin_block_fwd: null operand");
  SRK_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (shift == 0 || shift == 4), SRK_E_SHAPE,
              "swin_block_fwd: B=%d H=%d W=%d shift=%d", B, H, W, shift);
  SRK_REQUIRE(C > 0 && C <= 64 && hidden > 0 && hidden <= 128 && num_heads * head_dim == C, SRK_E_SHAPE,
              "swin_block_fwd: C=%d heads=%d x %d hidden=%d", C, num_heads, head_dim, hidden);
  WinGeom g;
  g.H = H; g.W = W; g.nWw = W / 8; g.nW = (H / 8) * (W / 8); g.shift = shift;
  const int rc = srk_launch_swin_block_light(x, y, y_bf16, norm1_w, norm1_b, norm2_w, norm2_b, wqkv, wproj, w1, w2, bqkv, bproj, b1, b2,
                                             bias_dense, scale, C, 64, 128, num_heads, head_dim, hidden, (long long)B * g.nW, g,
                                             (hipStream_t)stream);
  if (rc == SRK_NOT_COVERED) {
    srk_set_error("swin_block_fwd: only the light width (6 heads x d <= 16) has a whole-block kernel");
    return SRK_E_UNSUPPORTED;
  }
  return rc;
}

// ---- generic GEMM / implicit-GEMM conv entry -----------------------------------------------------------------------------------
int srk_gemm_ex(const srk_gemm_args* a, srk_stream_t stream) {
  SRK_REQUIRE(a != nullptr, SRK_E_NULL, "gemm_ex: null argument block");
  SRK_REQUIRE(a->loader == SRK_LD_ROWS || a->loader == SRK_LD_CONV3 || a->loader == SRK_LD_CONV3_PS, SRK_E_UNSUPPORTED, "gemm_ex: loader %d",
              a->loader);
  GemmParams p = {};
  p.A = static_cast<const bf16_t*>(a->A); p.lda = a->lda; p.Wt = static_cast<const bf16_t*>(a->W);
  p.M = a->M; p.N = a->N; p.K = a->K; p.B = a->B; p.H = a->H; p.W = a->Wd; p.CinP = a->CinP; p.r = a->r; p.Cs = a->Cs;
  p.bias = a->bias; p.outf = a->outf; p.outb = static_cast<bf16_t*>(a->outb); p.outb2 = static_cast<bf16_t*>(a->outb2);
  p.res = a->res; p.aux = static_cast<const bf16_t*>(a->aux); p.ldo = a->ldo; p.scale = a->scale;
  p.inv_range = a->inv_range; p.Cimg = a->Cimg; p.Hc = a->Hc; p.Wc = a->Wc;
  for (int i = 0; i < 4; ++i) p.mean[i] = a->mean[i];
  p.xn_out = static_cast<bf16_t*>(a->xn_out); p.xn_mean = a->xn_mean; p.xn_rstd = a->xn_rstd; p.xn_gamma = a->xn_gamma;
  p.xn_beta = a->xn_beta; p.xn_C = a->xn_C;
  p.rowscale = a->rowscale; p.rows_per_sample = a->rows_per_sample;
  SRK_REQUIRE(a->rowscale == nullptr || ((a->epilogue == SRK_EP_RES || a->epilogue == SRK_EP_LNBWD) && a->rows_per_sample > 0), SRK_E_SHAPE,
              "gemm_ex: rowscale goes with SRK_EP_RES / SRK_EP_LNBWD and rows_per_sample > 0");
  p.flops = 2.0 * a->M * (double)a->N * a->K;
  switch (a->epilogue) {
    case SRK_EP_BF16: case SRK_EP_GELU: case SRK_EP_RES: case SRK_EP_LRELU: case SRK_EP_PS: case SRK_EP_IMG: case SRK_EP_PS_IMG: case SRK_EP_RES_BF16:
      break;
    case SRK_EP_DGELU: case SRK_EP_DLRELU:
      SRK_REQUIRE(a->aux != nullptr && a->outb != nullptr, SRK_E_NULL, "gemm_ex: the activation-gradient epilogues need aux and outb");
      break;
    case SRK_EP_F32_BF16:
      SRK_REQUIRE(a->outf != nullptr, SRK_E_NULL, "gemm_ex: SRK_EP_F32_BF16 needs outf");
      break;
    case SRK_EP_LNBWD:
      SRK_REQUIRE(a->loader == SRK_LD_ROWS && a->outf && a->ln_x && a->ln_mean && a->ln_rstd && a->ln_gamma && a->ln_dgamma && a->ln_dbeta, SRK_E_NULL,
                  "gemm_ex: SRK_EP_LNBWD needs LD_ROWS, outf and the LayerNorm operands");
      SRK_REQUIRE((a->N == 64 || a->N == 128 || a->N == 192) && a->ln_C > 0 && a->ln_C <= a->N && a->ldo == a->N && a->bias == nullptr, SRK_E_SHAPE,
                  "gemm_ex: SRK_EP_LNBWD needs N = 64 / 128 / 192 = ldo (one tile holds a whole row), no bias");
      p.ln_x = a->ln_x; p.ln_mean = a->ln_mean; p.ln_rstd = a->ln_rstd; p.ln_gamma = a->ln_gamma; p.ln_dgamma = a->ln_dgamma;
      p.ln_dbeta = a->ln_dbeta; p.ln_C = a->ln_C;
      break;
    default:
      srk_set_error("gemm_ex: epilogue %d is not exposed", a->epilogue);
      return SRK_E_UNSUPPORTED;
  }
  if (a->epilogue == SRK_EP_MLP_FUSED) return SRK_E_UNSUPPORTED;
  return srk_launch_gemm(a->loader, a->epilogue, p, (hipStream_t)stream);
}

int srk_mlp_fused_fwd(const uint16_t* xn, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* res,
                      float* out, uint16_t* out_bf16, uint16_t* xn_next, float* xn_mean, float* xn_rstd, const float* xn_gamma,
                      const float* xn_beta, int xn_C, int M, srk_stream_t stream) {
  SRK_REQUIRE(xn && w1 && w2 && res && out, SRK_E_NULL, "mlp_fused: null pointer");
  GemmParams p = {};
  p.A = xn; p.lda = 192; p.Wt = w1; p.K = 192; p.bias = b1; p.W2 = w2; p.bias2 = b2; p.HP = 384; p.M = M; p.N = 192; p.ldo = 192;
  p.res = res; p.outf = out; p.outb = out_bf16;
  if (xn_next) {
    SRK_REQUIRE(xn_mean && xn_rstd && xn_gamma && xn_beta, SRK_E_NULL, "mlp_fused: fused LayerNorm needs mean / rstd / gamma / beta");
    p.xn_out = xn_next; p.xn_mean = xn_mean; p.xn_rstd = xn_rstd; p.xn_gamma = xn_gamma; p.xn_beta = xn_beta; p.xn_C = xn_C;
  }
  p.flops = 4.0 * M * 180.0 * 360.0;
  const int rc = srk_launch_mlp_fused(p, (hipStream_t)stream);
  if (rc == SRK_NOT_COVERED) {
    srk_set_error("mlp_fused: shape not covered (needs C 180/192, hidden 360/384, M %% 64 == 0 and M >= 64 * #CUs; got M=%d)", M);
    return SRK_E_UNSUPPORTED;
  }
  return rc;
}

int srk_img_prep(const float* x, float* out, int B, int Cimg, int H0, int W0, int H, int W, float range, const float* mean3, srk_stream_t stream) {
  SRK_REQUIRE(x && out && mean3, SRK_E_NULL, "img_prep: null pointer");
  SRK_REQUIRE(H >= H0 && W >= W0 && H - H0 < H0 && W - W0 < W0 && Cimg >= 1 && Cimg <= 3, SRK_E_SHAPE, "img_prep: bad geometry");
  return srk_launch_img_prep(x, out, B, Cimg, H0, W0, H, W, range, mean3, (hipStream_t)stream);
}

int srk_stem_conv(const float* img4, const float* weight, const float* bias, float* out, int B, int H, int W, int Cin, int C, int CP,
                  srk_stream_t stream) {
  SRK_REQUIRE(img4 && weight && bias && out, SRK_E_NULL, "stem_conv: null pointer");
  return srk_launch_stem_conv(img4, weight, bias, out, B, H, W, Cin, C, CP, (hipStream_t)stream);
}

}  // extern "C"
