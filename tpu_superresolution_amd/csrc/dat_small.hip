// DAT training: the two small functions that sit between token passes, each as ONE launch forward and ONE backward (they were ~35 and
// ~50 tiny torch kernels per block, forward + autograd: the hipGraph replay hides their launch cost, not their serial run time).
//
//   channel_interaction (dat_arch.py:315-321 on the pooled 1 x 1 map):  pooled mean [B][C] -> 1x1 conv C -> S -> BatchNorm2d over the
//       BATCH (the map is 1 x 1) -> GELU -> 1x1 conv S -> C -> sigmoid, with the head-padded token layout gathered / scattered here
//   channel-attention matrix (dat_arch.py:497-503):  A = softmax_j(temperature * G_ij / (|q_i| |k_j|)) from the Gram partials of
//       srk_chan_gram, and its backward (d G, the diagonal terms of d q / d k through the norms, d temperature)
//
// Everything here is a few thousand flops: one workgroup (channel interaction) or one per (sample, head) (attention matrix), fp32.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int CI_THREADS = 512;
constexpr int CI_MAX_LDS = 150 * 1024;      // dynamic LDS of the one workgroup: both weight matrices and every [B][C] / [B][S] array

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct CiParams {
  const float* pooled;   // [B][ldp] sums over the tokens of a sample, head-padded channel order
  int ldp;
  float inv_hw;
  const int* pad_of;     // [C] padded position of real channel c
  const float *W1, *b1, *gamma, *beta, *W2, *b2;      // [S][C], [S], [S], [S], [C][S], [C]
  float eps;
  int B, C, S, CA;
};

// LDS layout (floats): W1 [S][C] | W2 [C][S] | pm [B][C] | zh [B][S] | a [B][S] | mean [64] | rstd [64] | var [64]  (+ backward: dpre [B][C] |
// dy [B][S] | sdg [64] | sdb [64]).  The weights are staged once with coalesced loads: read from global inside the dot-product loops,
// every thread walked a chain of ~180 L2 round trips (69 us per backward launch).
struct CiLds {
  float *W1, *W2, *pm, *zh, *a, *mean, *rstd, *var, *dpre, *dy, *sdg, *sdb;
};
__host__ __device__ inline size_t ci_lds_floats(int B, int C, int S, bool bwd) {
  return (size_t)2 * S * C + (size_t)B * C + 2 * (size_t)B * S + 3 * 64 + (bwd ? (size_t)B * C + (size_t)B * S + 2 * 64 : 0);
}
__device__ __forceinline__ CiLds ci_carve(float* base, int B, int C, int S) {
  CiLds l;
  l.W1 = base; l.W2 = l.W1 + S * C; l.pm = l.W2 + S * C; l.zh = l.pm + B * C; l.a = l.zh + B * S; l.mean = l.a + B * S; l.rstd = l.mean + 64;
  l.var = l.rstd + 64; l.dpre = l.var + 64; l.dy = l.dpre + B * C; l.sdg = l.dy + B * S; l.sdb = l.sdg + 64;
  return l;
}

// weights + pm -> LDS, y = pm W1^T + b1, batch statistics, a = gelu(BatchNorm(y)); leaves zh (normalised y) [B][S], a [B][S], mean / rstd / var [S]
__device__ void ci_forward_part(const CiParams& p, const float* pm_src, const CiLds& l) {
  const int tid = threadIdx.x, B = p.B, C = p.C, S = p.S;
  for (int i = tid; i < S * C; i += CI_THREADS) {
    l.W1[i] = p.W1[i];
    l.W2[i] = p.W2[i];
  }
  for (int i = tid; i < B * C; i += CI_THREADS) {
    const int b = i / C, c = i - b * C;
    l.pm[i] = pm_src ? pm_src[i] : p.pooled[(long long)b * p.ldp + p.pad_of[c]] * p.inv_hw;
  }
  __syncthreads();
  for (int i = tid; i < B * S; i += CI_THREADS) {
    const int b = i / S, s = i - b * S;
    float acc = p.b1[s];
    const float* w = l.W1 + s * C;
    const float* x = l.pm + b * C;
    for (int c = 0; c < C; ++c) acc = fmaf(x[c], w[c], acc);
    l.zh[i] = acc;                                  // y for now
  }
  __syncthreads();
  if (tid < S) {
    float m = 0.f;
    for (int b = 0; b < B; ++b) m += l.zh[b * S + tid];
    m /= (float)B;
    float v = 0.f;
    for (int b = 0; b < B; ++b) {
      const float d = l.zh[b * S + tid] - m;
      v = fmaf(d, d, v);
    }
    v /= (float)B;
    l.mean[tid] = m;
    l.rstd[tid] = rsqrtf(v + p.eps);
    l.var[tid] = v;
  }
  __syncthreads();
  for (int i = tid; i < B * S; i += CI_THREADS) {
    const int s = i % S;
    const float z = (l.zh[i] - l.mean[s]) * l.rstd[s];
    l.zh[i] = z;
    l.a[i] = gelu_f(fmaf(z, p.gamma[s], p.beta[s]));
  }
  __syncthreads();
}

__global__ __launch_bounds__(CI_THREADS) void channel_interaction_fwd_kernel(const CiParams p, float* __restrict__ pm_out,
                                                                             float* __restrict__ cgate, float* __restrict__ running_mean,
                                                                             float* __restrict__ running_var, float momentum) {
  extern __shared__ __attribute__((aligned(16))) float ci_smem[];
  const CiLds l = ci_carve(ci_smem, p.B, p.C, p.S);
  const int tid = threadIdx.x, B = p.B, C = p.C, S = p.S;
  for (int i = tid; i < B * p.CA; i += CI_THREADS) cgate[i] = 0.f;   // the padding channels of the gate
  ci_forward_part(p, nullptr, l);
  if (running_mean != nullptr && tid < S) {                           // nn.BatchNorm2d's buffers: unbiased variance
    running_mean[tid] = (1.0f - momentum) * running_mean[tid] + momentum * l.mean[tid];
    running_var[tid] = (1.0f - momentum) * running_var[tid] + momentum * l.var[tid] * ((float)B / fmaxf((float)B - 1.0f, 1.0f));
  }
  for (int i = tid; i < B * C; i += CI_THREADS) {
    const int b = i / C, c = i - b * C;
    float acc = p.b2[c];
    const float* w = l.W2 + c * S;
    const float* x = l.a + b * S;
    for (int s = 0; s < S; ++s) acc = fmaf(x[s], w[s], acc);
    pm_out[i] = l.pm[i];
    cgate[(long long)b * p.CA + p.pad_of[c]] = sigmoid_f(acc);
  }
}

// d cgate [B][ldg] (head-padded) -> parameter gradients and d pooled [B][CA] (= d pm / HW, head-padded, padding zero)
__global__ __launch_bounds__(CI_THREADS) void channel_interaction_bwd_kernel(const CiParams p, const float* __restrict__ pm_in,
                                                                             const float* __restrict__ dcg, int ldg, float* __restrict__ dW1,
                                                                             float* __restrict__ db1, float* __restrict__ dgamma,
                                                                             float* __restrict__ dbeta, float* __restrict__ dW2,
                                                                             float* __restrict__ db2, float* __restrict__ dpool) {
  extern __shared__ __attribute__((aligned(16))) float ci_smem[];
  const CiLds l = ci_carve(ci_smem, p.B, p.C, p.S);
  const int tid = threadIdx.x, B = p.B, C = p.C, S = p.S;
  for (int i = tid; i < B * p.CA; i += CI_THREADS) dpool[i] = 0.f;
  ci_forward_part(p, pm_in, l);
  // d(pre-sigmoid) = d cgate * o (1 - o)
  for (int i = tid; i < B * C; i += CI_THREADS) {
    const int b = i / C, c = i - b * C;
    float acc = p.b2[c];
    const float* w = l.W2 + c * S;
    const float* x = l.a + b * S;
    for (int s = 0; s < S; ++s) acc = fmaf(x[s], w[s], acc);
    const float o = sigmoid_f(acc);
    l.dpre[i] = dcg[(long long)b * ldg + p.pad_of[c]] * o * (1.0f - o);
  }
  __syncthreads();
  for (int i = tid; i < C * S; i += CI_THREADS) {                     // d W2 [C][S]
    const int c = i / S, s = i - c * S;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = fmaf(l.dpre[b * C + c], l.a[b * S + s], acc);
    dW2[i] = acc;
  }
  for (int c = tid; c < C; c += CI_THREADS) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += l.dpre[b * C + c];
    db2[c] = acc;
  }
  for (int i = tid; i < B * S; i += CI_THREADS) {                     // d a -> d z (through GELU and the BatchNorm affine)
    const int b = i / S, s = i - b * S;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(l.dpre[b * C + c], l.W2[c * S + s], acc);
    l.dy[i] = acc * dgelu_shared_exp(fmaf(l.zh[i], p.gamma[s], p.beta[s]));        // d(BatchNorm output)
  }
  __syncthreads();
  if (tid < S) {
    float g = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) {
      g = fmaf(l.dy[b * S + tid], l.zh[b * S + tid], g);
      bsum += l.dy[b * S + tid];
    }
    dgamma[tid] = g;
    dbeta[tid] = bsum;
    l.sdg[tid] = g;
    l.sdb[tid] = bsum;
  }
  __syncthreads();
  for (int i = tid; i < B * S; i += CI_THREADS) {                     // BatchNorm backward over the batch: d y
    const int s = i % S;
    l.dy[i] = p.gamma[s] * l.rstd[s] * (l.dy[i] - l.sdb[s] / (float)B - l.zh[i] * l.sdg[s] / (float)B);
  }
  __syncthreads();
  for (int i = tid; i < S * C; i += CI_THREADS) {                     // d W1 [S][C]
    const int s = i / C, c = i - s * C;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = fmaf(l.dy[b * S + s], l.pm[b * C + c], acc);
    dW1[i] = acc;
  }
  if (tid < S) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += l.dy[b * S + tid];
    db1[tid] = acc;
  }
  for (int i = tid; i < B * C; i += CI_THREADS) {                     // d pm -> d pooled (/ HW), head-padded
    const int b = i / C, c = i - b * C;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc = fmaf(l.dy[b * S + s], l.W1[s * C + c], acc);
    dpool[(long long)b * p.CA + p.pad_of[c]] = acc * p.inv_hw;
  }
}

// ---- channel-attention matrix -------------------------------------------------------------------------------------------------
constexpr int GSZ = 1088;        // floats per Gram partial: G [32][32], sum q^2 [32], sum k^2 [32]

// one workgroup per (b, h): chunk partials -> G | sq | sk (saved for the backward), A = softmax over the dh real key channels
__global__ __launch_bounds__(256) void chan_attn_matrix_fwd_kernel(const float* __restrict__ partial, int nchunk, const float* __restrict__ temp,
                                                                   int nH, int dh, float* __restrict__ gram, float* __restrict__ A) {
  __shared__ float g[GSZ];
  const int bh = blockIdx.x, h = bh % nH, tid = threadIdx.x;
  const float* src = partial + (long long)bh * nchunk * GSZ;
  for (int i = tid; i < GSZ; i += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = 0;
    for (; c + 4 <= nchunk; c += 4) {
      a0 += src[(long long)c * GSZ + i];
      a1 += src[(long long)(c + 1) * GSZ + i];
      a2 += src[(long long)(c + 2) * GSZ + i];
      a3 += src[(long long)(c + 3) * GSZ + i];
    }
    for (; c < nchunk; ++c) a0 += src[(long long)c * GSZ + i];
    const float v = (a0 + a1) + (a2 + a3);
    g[i] = v;
    gram[(long long)bh * GSZ + i] = v;
  }
  __syncthreads();
  const float t = temp[h];
  // thread = (row i, four columns); the row's max / sum through three lane exchanges over its eight threads
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  const float nq = sqrtf(fmaxf(g[1024 + i], 1e-24f));               // F.normalize: norm clamped at 1e-12
  float l[4], mx = -3.0e38f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = j0 + e;
    const float nk = sqrtf(fmaxf(g[1056 + j], 1e-24f));
    l[e] = (i < dh && j < dh) ? g[i * 32 + j] / (nq * nk) * t : -3.0e38f;
    mx = fmaxf(mx, l[e]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  mx = fmaxf(mx, __shfl_xor(mx, 4));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    l[e] = (i < dh && j0 + e < dh) ? __expf(l[e] - mx) : 0.f;
    sum += l[e];
  }
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  sum += __shfl_xor(sum, 4);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  *reinterpret_cast<float4*>(A + ((long long)bh * 32 + i) * 32 + j0) = make_float4(l[0] * inv, l[1] * inv, l[2] * inv, l[3] * inv);
}

// d A (chunk partials of the Gram kernel run on (d out, v)) -> d G, d G^T, 2 d(sum q^2), 2 d(sum k^2) (the diagonal coefficients of
// srk_chan_apply_mat), d temperature per (b, h)
__global__ __launch_bounds__(256) void chan_attn_matrix_bwd_kernel(const float* __restrict__ dpartial, int nchunk, const float* __restrict__ gram,
                                                                   const float* __restrict__ A, const float* __restrict__ temp, int nH, int dh,
                                                                   float* __restrict__ dG, float* __restrict__ dGt, float* __restrict__ dsq2,
                                                                   float* __restrict__ dsk2, float* __restrict__ dtemp) {
  __shared__ float dA[1024], dL[1024], LL[1024], rq[32], rk[32], red[32];
  const int bh = blockIdx.x, h = bh % nH, tid = threadIdx.x;
  const float* src = dpartial + (long long)bh * nchunk * GSZ;
  const float* g = gram + (long long)bh * GSZ;
  for (int i = tid; i < 1024; i += 256) {
    float a0 = 0.f, a1 = 0.f;
    int c = 0;
    for (; c + 2 <= nchunk; c += 2) {
      a0 += src[(long long)c * GSZ + i];
      a1 += src[(long long)(c + 1) * GSZ + i];
    }
    for (; c < nchunk; ++c) a0 += src[(long long)c * GSZ + i];
    dA[i] = a0 + a1;
  }
  if (tid < 32) {
    rq[tid] = 1.0f / sqrtf(fmaxf(g[1024 + tid], 1e-24f));
    rk[tid] = 1.0f / sqrtf(fmaxf(g[1056 + tid], 1e-24f));
  }
  __syncthreads();
  const float t = temp[h];
  const float* Ab = A + (long long)bh * 1024;
  if (tid < 32) {                                                     // softmax backward, one row per thread
    const int i = tid;
    float dot = 0.f;
    if (i < dh)
      for (int j = 0; j < dh; ++j) dot = fmaf(dA[i * 32 + j], Ab[i * 32 + j], dot);
    float dt = 0.f;
    for (int j = 0; j < 32; ++j) {
      const bool live = i < dh && j < dh;
      const float dl = live ? Ab[i * 32 + j] * (dA[i * 32 + j] - dot) : 0.f;
      const float cosv = live ? g[i * 32 + j] * rq[i] * rk[j] : 0.f;   // L / temperature
      dL[i * 32 + j] = dl;
      LL[i * 32 + j] = dl * cosv * t;                                   // d L * L
      dt = fmaf(dl, cosv, dt);
    }
    red[i] = dt;
  }
  __syncthreads();
  for (int idx = tid; idx < 1024; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    const float v = dL[idx] * t * rq[i] * rk[j];
    dG[(long long)bh * 1024 + idx] = v;
    dGt[(long long)bh * 1024 + j * 32 + i] = v;
  }
  if (tid < 32) {
    // d(sum q_i^2): through |q_i| = sqrt(max(sq, 1e-24)): - sum_j dL L / sq (clamped entries get no gradient)
    const int i = tid;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < 32; ++j) {
      a += LL[i * 32 + j];
      b += LL[j * 32 + i];
    }
    const float sq = g[1024 + i], sk = g[1056 + i];
    dsq2[(long long)bh * 32 + i] = (i < dh && sq > 1e-24f) ? -a / sq : 0.f;
    dsk2[(long long)bh * 32 + i] = (i < dh && sk > 1e-24f) ? -b / sk : 0.f;
  }
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += red[i];
    dtemp[bh] = s;
  }
}

}  // namespace

#define REQS(cond, ...) SRK_REQUIRE(cond, SRK_E_SHAPE, __VA_ARGS__)

extern "C" {

static int ci_configure(const void* fn, size_t bytes, int* state) {
  if (*state >= (int)bytes) return SRK_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, CI_MAX_LDS) != hipSuccess) {
    srk_set_error("channel_interaction: cannot reserve %d bytes of LDS", CI_MAX_LDS);
    return SRK_E_LAUNCH;
  }
  *state = CI_MAX_LDS;
  return SRK_OK;
}

/* 1 when srk_channel_interaction_fwd / _bwd cover the shape (everything of the one workgroup fits its LDS) */
int srk_channel_interaction_covered(int B, int C, int S) {
  return B > 1 && C > 0 && S > 0 && S <= 64 && ci_lds_floats(B, C, S, true) * 4 <= (size_t)CI_MAX_LDS;
}

int srk_channel_interaction_fwd(const float* pooled, int ldp, float inv_hw, const int* pad_of, const float* W1, const float* b1,
                                const float* gamma, const float* beta, float eps, const float* W2, const float* b2, float* running_mean,
                                float* running_var, float momentum, float* pm_out, float* cgate, int B, int C, int S, int CA,
                                srk_stream_t stream) {
  SRK_REQUIRE(pooled && pad_of && W1 && b1 && gamma && beta && W2 && b2 && pm_out && cgate, SRK_E_NULL, "channel_interaction_fwd: null pointer");
  REQS(srk_channel_interaction_covered(B, C, S) && CA >= C && ldp >= CA && (running_mean == nullptr) == (running_var == nullptr),
       "channel_interaction_fwd: B=%d C=%d S=%d (B > 1 as nn.BatchNorm2d in training; S <= 64; the arrays must fit %d bytes of LDS)", B, C, S,
       CI_MAX_LDS);
  static SrkPerDevice<int> st_pd;
  const size_t lds = ci_lds_floats(B, C, S, false) * 4;
  const int rc = ci_configure(reinterpret_cast<const void*>(&channel_interaction_fwd_kernel), lds, &st_pd.here());
  if (rc) return rc;
  CiParams p{pooled, ldp, inv_hw, pad_of, W1, b1, gamma, beta, W2, b2, eps, B, C, S, CA};
  hipLaunchKernelGGL(channel_interaction_fwd_kernel, dim3(1), dim3(CI_THREADS), lds, (hipStream_t)stream, p, pm_out, cgate, running_mean,
                     running_var, momentum);
  return srk_check_launch("channel_interaction_fwd");
}

int srk_channel_interaction_bwd(const float* pm, const float* dcgate, int ldg, float inv_hw, const int* pad_of, const float* W1,
                                const float* b1, const float* gamma, const float* beta, float eps, const float* W2, const float* b2,
                                float* dW1, float* db1, float* dgamma, float* dbeta, float* dW2, float* db2, float* dpool, int B, int C, int S,
                                int CA, srk_stream_t stream) {
  SRK_REQUIRE(pm && dcgate && pad_of && W1 && b1 && gamma && beta && W2 && b2 && dW1 && db1 && dgamma && dbeta && dW2 && db2 && dpool, SRK_E_NULL,
              "channel_interaction_bwd: null pointer");
  REQS(srk_channel_interaction_covered(B, C, S) && CA >= C && ldg >= CA, "channel_interaction_bwd: B=%d C=%d S=%d", B, C, S);
  static SrkPerDevice<int> st_pd;
  const size_t lds = ci_lds_floats(B, C, S, true) * 4;
  const int rc = ci_configure(reinterpret_cast<const void*>(&channel_interaction_bwd_kernel), lds, &st_pd.here());
  if (rc) return rc;
  CiParams p{nullptr, 0, inv_hw, pad_of, W1, b1, gamma, beta, W2, b2, eps, B, C, S, CA};
  hipLaunchKernelGGL(channel_interaction_bwd_kernel, dim3(1), dim3(CI_THREADS), lds, (hipStream_t)stream, p, pm, dcgate, ldg, dW1, db1, dgamma,
                     dbeta, dW2, db2, dpool);
  return srk_check_launch("channel_interaction_bwd");
}

int srk_chan_attn_matrix_fwd(const float* partial, int nchunk, const float* temperature, float* gram, float* A, int B, int num_heads, int dh,
                             srk_stream_t stream) {
  SRK_REQUIRE(partial && temperature && gram && A, SRK_E_NULL, "chan_attn_matrix_fwd: null pointer");
  REQS(B > 0 && num_heads > 0 && nchunk > 0 && dh > 0 && dh <= 32, "chan_attn_matrix_fwd: dh=%d (<= 32)", dh);
  hipLaunchKernelGGL(chan_attn_matrix_fwd_kernel, dim3(B * num_heads), dim3(256), 0, (hipStream_t)stream, partial, nchunk, temperature, num_heads, dh,
                     gram, A);
  return srk_check_launch("chan_attn_matrix_fwd");
}

int srk_chan_attn_matrix_bwd(const float* dpartial, int nchunk, const float* gram, const float* A, const float* temperature, float* dG, float* dGt,
                             float* dsq2, float* dsk2, float* dtemp, int B, int num_heads, int dh, srk_stream_t stream) {
  SRK_REQUIRE(dpartial && gram && A && temperature && dG && dGt && dsq2 && dsk2 && dtemp, SRK_E_NULL, "chan_attn_matrix_bwd: null pointer");
  REQS(B > 0 && num_heads > 0 && nchunk > 0 && dh > 0 && dh <= 32, "chan_attn_matrix_bwd: dh=%d (<= 32)", dh);
  hipLaunchKernelGGL(chan_attn_matrix_bwd_kernel, dim3(B * num_heads), dim3(256), 0, (hipStream_t)stream, dpartial, nchunk, gram, A, temperature,
                     num_heads, dh, dG, dGt, dsq2, dsk2, dtemp);
  return srk_check_launch("chan_attn_matrix_bwd");
}

}  // extern "C"
