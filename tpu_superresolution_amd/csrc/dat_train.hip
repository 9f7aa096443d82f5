// Training-side kernels for DAT (reference dat_arch.py), token-sized work only.  DAT's blocks are full of small per-channel or
// per-sample functions of token REDUCTIONS (train-mode BatchNorm :301-313 / :464-476, the squeeze-excite channel interaction, the
// channel attention's d x d matrices :497-503, the dynamic position bias MLP :93-130).  The split used here: these kernels
// produce the reductions over tokens (fixed-order partials) and apply per-channel / per-sample coefficient vectors to token
// tensors; the tiny functions in between (a few hundred floats) run on the host side of the C ABI (tpu_superresolution_amd/
// dat_train.py), forward and backward.
//
//   chan_stats            partial[sample][chunk][0][c] = sum_t p[t][c], [1][c] = sum_t p[t][c] q[t][c]   (the pooled mean, BatchNorm statistics
//                         with q = p; the gradients of a per-channel scale / shift with p = dy, q = x)
//   affine_act            out = act(x * s[b][c] + t[b][c])                      (BatchNorm apply + GELU)
//   dgelu_affine          out = dy * gelu'(x * s[c] + t[c])
//   lincomb2              out (+)= A[b][c] p + B[b][c] q + C[b][c]              (per-sample or per-channel coefficient vectors; BatchNorm backward,
//                         the broadcast of a pooled gradient, plain sums)
//   mul_bwd               da = dy * b, db = dy * a                              (SpatialGate's x1 * x2, :54)
//   dwconv3x3_wgrad       depth-wise 3x3 weight / bias gradient partials
//   dual_gate_bwd         backward of dual_gate_combine (dat.hip): gated copies of the gradient, the channel map's gradient partials,
//                         the spatial map's gradient through its sigmoid
//   spatial_gate_stats / _bwd_stats / _bwd_apply   the spatial interaction (1x1 conv C -> C/16, BatchNorm, GELU, 1x1 conv -> 1, :318-323)
//   rowln_bwd             LayerNorm backward on bf16 rows (SpatialGate.norm, :44-54)
//   chan_gram / chan_apply_mat   the channel attention's token reductions and the application of its d x d matrices (:497-508)
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int ST_ROWS = 256;     // tokens per workgroup of the reduction passes

inline int grid_cap(long long n, int block = 256, int cap = 16384) {
  long long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// thread = (row group rg of 8, 8-channel piece); 256 threads cover 32 pieces x 8 row groups; pieces >= C8 idle
__global__ __launch_bounds__(256) void chan_stats_kernel(const bf16_t* __restrict__ p, int ldp, const bf16_t* __restrict__ q, int ldq,
                                                         float* __restrict__ partial, long long rows, int C8) {
  // rows = rows of ONE sample; blockIdx.y = sample (its rows follow each other); partial [sample][chunk][2][CP]
  __shared__ float red[2][8][256];
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int CP = C8 * 8;
  p += (long long)blockIdx.y * rows * ldp;
  q += (long long)blockIdx.y * rows * ldq;
  for (int c0 = 0; c0 < C8; c0 += 32) {
    const int piece = c0 + tid % 32, rg = tid / 32;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (piece < C8) {
      const long long r0 = (long long)chunk * ST_ROWS;
      for (long long r = r0 + rg; r < r0 + ST_ROWS && r < rows; r += 8) {
        const uint4 pv = *reinterpret_cast<const uint4*>(p + r * ldp + piece * 8);
        const uint4 qv = *reinterpret_cast<const uint4*>(q + r * ldq + piece * 8);
        const unsigned pu[4] = {pv.x, pv.y, pv.z, pv.w}, qu[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float p0, p1, q0, q1;
          unpack_bf2(pu[e], p0, p1);
          unpack_bf2(qu[e], q0, q1);
          a[2 * e] += p0; a[2 * e + 1] += p1;
          b[2 * e] += p0 * q0; b[2 * e + 1] += p1 * q1;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[0][rg][(tid % 32) * 8 + e] = a[e];
      red[1][rg][(tid % 32) * 8 + e] = b[e];
    }
    __syncthreads();
    const int c = c0 * 8 + tid;
    if (c < CP && tid < 256) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s0 += red[0][k][tid];
        s1 += red[1][k][tid];
      }
      float* dst = partial + ((long long)blockIdx.y * gridDim.x + chunk) * 2 * CP;
      dst[c] = s0;
      dst[CP + c] = s1;
    }
  }
}

// Sum of the R partial rows for 64 channels per workgroup: thread = (row group rg of 4, channel), eight independent loads in flight per
// thread (one dependent add per load made this a chain of R memory round trips), row groups combined through LDS in a fixed order.
// Returns the channel of this thread when it holds the sums (rg == 0 and c < Cn), else -1.
__device__ __forceinline__ int bn_sum_rows(const float* __restrict__ partial, int R, int rs, int ld, int Cn, float& s1, float& s2) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < Cn) {
    const float* base = partial + c;
    int r = rg;
    for (; r + 28 < R; r += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a1[u] += base[(long long)(r + 4 * u) * rs];
        a2[u] += base[(long long)(r + 4 * u) * rs + ld];
      }
    }
    for (; r < R; r += 4) {
      a1[0] += base[(long long)r * rs];
      a2[0] += base[(long long)r * rs + ld];
    }
  }
  red[0][rg][cl] = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
  red[1][rg][cl] = ((a2[0] + a2[1]) + (a2[2] + a2[3])) + ((a2[4] + a2[5]) + (a2[6] + a2[7]));
  __syncthreads();
  s1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  s2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  return (rg == 0 && c < Cn) ? c : -1;
}

// out[o][i] = sum_r in[o][r][i]: the fixed-order sum of the R partial rows a token pass leaves behind (thread = (row group of 4, column);
// eight independent loads in flight per thread, row groups combined through LDS).  torch's reduction of the same [R][n] shapes takes
// ~9.5 us per call, seven calls per DAT block.
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ in, int R, int n, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  const float* base = in + (long long)blockIdx.y * R * n + i;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int r = rg;
    for (; r + 28 < R; r += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += base[(long long)(r + 4 * u) * n];
    }
    for (; r < R; r += 4) a[0] += base[(long long)r * n];
  }
  red[rg][cl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (rg == 0 && i < n) out[(long long)blockIdx.y * n + i] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// BatchNorm (training) between two token passes, one thread per channel: the R partial rows (rs floats apart; the two sums at + 0 and + ld)
// of chan_stats / spatial_gate_train are summed in a fixed order, then  mean = s1 / n, var = s2 / n - mean^2, rstd = (var + eps)^-1/2, scale = gamma rstd, shift = beta - mean scale
// -> coef [4][ld] = scale, shift, mean, rstd; the running estimates move in place as nn.BatchNorm2d's do (momentum, unbiased variance);
// real_of[c] = index of padded channel c in the BatchNorm's own (un-padded) buffers, -1 for padding (null: identity).
__global__ __launch_bounds__(256) void bn_train_coeffs_kernel(const float* __restrict__ partial, int R, int rs, int ld, int Cn, float n,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              float* __restrict__ coef, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum, const int* __restrict__ real_of) {
  float s1, s2;
  const int c = bn_sum_rows(partial, R, rs, ld, Cn, s1, s2);
  if (c < 0) return;
  const float mean = s1 / n;
  const float var = fmaxf(s2 / n - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  coef[c] = sc;
  coef[ld + c] = beta[c] - mean * sc;
  coef[2 * ld + c] = mean;
  coef[3 * ld + c] = rstd;
  const int rc = real_of ? real_of[c] : c;
  if (running_mean != nullptr && rc >= 0) {
    running_mean[rc] = (1.0f - momentum) * running_mean[rc] + momentum * mean;
    running_var[rc] = (1.0f - momentum) * running_var[rc] + momentum * var * (n / fmaxf(n - 1.0f, 1.0f));
  }
}

// ... and its backward: S1 = sum dz, S2 = sum dz x (partial rows as above) -> coef [5][ld] = A, B, C of d x = A dz + B x + C, d gamma, d beta
__global__ __launch_bounds__(256) void bn_train_bwd_coeffs_kernel(const float* __restrict__ partial, int R, int rs, int ld, int Cn, float n,
                                                                  const float* __restrict__ fwd_coef, float* __restrict__ coef) {
  float S1, S2;
  const int c = bn_sum_rows(partial, R, rs, ld, Cn, S1, S2);
  if (c < 0) return;
  const float sc = fwd_coef[c], mean = fwd_coef[2 * ld + c], rstd = fwd_coef[3 * ld + c];
  const float dgamma = rstd * (S2 - mean * S1);
  coef[c] = sc;
  coef[ld + c] = -sc * rstd * dgamma / n;
  coef[2 * ld + c] = (sc / n) * (mean * rstd * dgamma - S1);
  coef[3 * ld + c] = dgamma;
  coef[4 * ld + c] = S1;
}

// coefficient index of row t, channel c: (rps > 0 ? t / rps : 0) * CP + c
// The GELU-derivative pass (mode 1) keeps the one-piece-per-thread grid-stride form: it is VALU-bound (erf + exp per element), and with
// the row-walking mapping below it lost occupancy (37 -> 49 us with four rows per lane, 113 us with one).
template <int MODE>
__global__ __launch_bounds__(256) void token_elementwise_flat_kernel(const bf16_t* __restrict__ p, int ldp, const bf16_t* __restrict__ q, int ldq,
                                                                const float* __restrict__ A, const float* __restrict__ Bv,
                                                                const float* __restrict__ Cv, bf16_t* __restrict__ out, int ldo,
                                                                long long rows, int C8, int rps, int flag) {
  const unsigned n = (unsigned)(rows * C8);              // < 2^31 (checked by the launchers): 32-bit index arithmetic, no 64-bit divisions
  const int CP = C8 * 8;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned tu = i / (unsigned)C8;
    const int c = (int)(i - tu * (unsigned)C8) * 8;
    const long long t = tu;
    const long long co = (rps > 0 ? (long long)(tu / (unsigned)rps) * CP : 0) + c;
    float pf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p) {
      const uint4 pv = *reinterpret_cast<const uint4*>(p + t * ldp + c);
      unpack_bf2(pv.x, pf[0], pf[1]); unpack_bf2(pv.y, pf[2], pf[3]); unpack_bf2(pv.z, pf[4], pf[5]); unpack_bf2(pv.w, pf[6], pf[7]);
    }
    if (q) {
      const uint4 qv = *reinterpret_cast<const uint4*>(q + t * ldq + c);
      unpack_bf2(qv.x, qf[0], qf[1]); unpack_bf2(qv.y, qf[2], qf[3]); unpack_bf2(qv.z, qf[4], qf[5]); unpack_bf2(qv.w, qf[6], qf[7]);
    }
    float o[8];
    if constexpr (MODE == 0) {            // out = act(p * A + B)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = pf[e] * A[co + e] + Bv[co + e];
        o[e] = flag ? gelu_f(v) : v;
      }
    } else if constexpr (MODE == 1) {     // out = p(dy) * gelu'(q(x) * A + B)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = pf[e] * dgelu_shared_exp(qf[e] * A[co + e] + Bv[co + e]);
    } else if constexpr (MODE == 2) {     // out (+)= A p + B q + C   (null A / B: coefficient 1; null p / q: no such term)
      float old[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (flag) {
        const uint4 ov = *reinterpret_cast<const uint4*>(out + t * ldo + c);
        unpack_bf2(ov.x, old[0], old[1]); unpack_bf2(ov.y, old[2], old[3]); unpack_bf2(ov.z, old[4], old[5]); unpack_bf2(ov.w, old[6], old[7]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = old[e] + (A ? A[co + e] * pf[e] : pf[e]) + (Bv ? Bv[co + e] * qf[e] : qf[e]) + (Cv ? Cv[co + e] : 0.f);
    }
    *reinterpret_cast<uint4*>(out + t * ldo + c) =
        make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
  }
}

// Thread = (row lane, 8-channel piece): the piece is fixed for the thread's lifetime, so per-channel coefficients (rps == 0) live in
// registers, and the thread walks EW_ROWS rows per lane with all their loads issued before the first use.  (One 16-byte piece per thread
// with 24 coefficient loads and two integer divisions each ran these passes at 2.6 TB/s.)
template <int MODE>      // 0 affine (+ optional GELU), 1 dgelu_affine, 2 lincomb2
__global__ __launch_bounds__(256) void token_elementwise_kernel(const bf16_t* __restrict__ p, int ldp, const bf16_t* __restrict__ q, int ldq,
                                                                const float* __restrict__ A, const float* __restrict__ Bv,
                                                                const float* __restrict__ Cv, bf16_t* __restrict__ out, int ldo,
                                                                long long rows, int C8, int rps, int flag, int cw, int lanes) {
  constexpr int EW_ROWS = 4;
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw;
  const int piece = blockIdx.y * cw + tx;
  if (ty >= lanes || piece >= C8) return;
  const int c = piece * 8, CP = C8 * 8;
  float ca[8], cb[8], cc[8];
  auto load_coef = [&](long long co) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ca[e] = A ? A[co + e] : 1.0f;
      cb[e] = Bv ? Bv[co + e] : (MODE == 2 ? 1.0f : 0.0f);
      cc[e] = Cv ? Cv[co + e] : 0.0f;
    }
  };
  if (rps <= 0) load_coef(c);
  const long long step = (long long)gridDim.x * lanes;
  for (long long t0 = (long long)blockIdx.x * lanes + ty; t0 < rows; t0 += step * EW_ROWS) {
    uint4 pv[EW_ROWS], qv[EW_ROWS], ov[EW_ROWS];
#pragma unroll
    for (int r = 0; r < EW_ROWS; ++r) {
      const long long t = t0 + r * step;
      pv[r] = qv[r] = ov[r] = make_uint4(0u, 0u, 0u, 0u);
      if (t < rows) {
        if (p) pv[r] = *reinterpret_cast<const uint4*>(p + t * ldp + c);
        if (q) qv[r] = *reinterpret_cast<const uint4*>(q + t * ldq + c);
        if (MODE == 2 && flag) ov[r] = *reinterpret_cast<const uint4*>(out + t * ldo + c);
      }
    }
#pragma unroll
    for (int r = 0; r < EW_ROWS; ++r) {
      const long long t = t0 + r * step;
      if (t >= rows) break;
      if (rps > 0) load_coef((long long)((unsigned)t / (unsigned)rps) * CP + c);      // rows < 2^31 (launchers)
      float pf[8], qf[8], o[8];
      unpack_bf2(pv[r].x, pf[0], pf[1]); unpack_bf2(pv[r].y, pf[2], pf[3]); unpack_bf2(pv[r].z, pf[4], pf[5]); unpack_bf2(pv[r].w, pf[6], pf[7]);
      unpack_bf2(qv[r].x, qf[0], qf[1]); unpack_bf2(qv[r].y, qf[2], qf[3]); unpack_bf2(qv[r].z, qf[4], qf[5]); unpack_bf2(qv[r].w, qf[6], qf[7]);
      if constexpr (MODE == 0) {            // out = act(p * A + B)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = pf[e] * ca[e] + cb[e];
          o[e] = flag ? gelu_f(v) : v;
        }
      } else if constexpr (MODE == 1) {     // out = p(dy) * gelu'(q(x) * A + B)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = pf[e] * dgelu_shared_exp(qf[e] * ca[e] + cb[e]);
      } else {                              // out (+)= A p + B q + C   (null A / B: coefficient 1; null p / q: no such term)
        float old[8];
        unpack_bf2(ov[r].x, old[0], old[1]); unpack_bf2(ov[r].y, old[2], old[3]); unpack_bf2(ov[r].z, old[4], old[5]); unpack_bf2(ov[r].w, old[6], old[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = old[e] + ca[e] * pf[e] + cb[e] * qf[e] + cc[e];
      }
      *reinterpret_cast<uint4*>(out + t * ldo + c) =
          make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
  }
}

// launch geometry of token_elementwise_kernel: cw pieces x lanes row lanes per workgroup, grid.y piece groups
struct EwGeom { int cw, lanes; dim3 grid; };
inline EwGeom ew_geom(long long rows, int C8, int EW_ROWS) {
  EwGeom g;
  g.cw = C8 < 256 ? C8 : 256;
  g.lanes = 256 / g.cw;
  const long long per_block = (long long)g.lanes * EW_ROWS;
  long long gx = (rows + per_block - 1) / per_block;
  if (gx > 65535) gx = 65535;
  g.grid = dim3((unsigned)(gx < 1 ? 1 : gx), (unsigned)((C8 + g.cw - 1) / g.cw));
  return g;
}

// da = dy * b, db = dy * a
__global__ __launch_bounds__(256) void mul_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ a, int lda,
                                                      const bf16_t* __restrict__ b, int ldb, bf16_t* __restrict__ da, int ldda,
                                                      bf16_t* __restrict__ db, int lddb, long long rows, int C8) {
  const unsigned n = (unsigned)(rows * C8);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned tu = i / (unsigned)C8;
    const int c = (int)(i - tu * (unsigned)C8) * 8;
    const long long t = tu;
    const uint4 dv = *reinterpret_cast<const uint4*>(dy + t * lddy + c);
    const uint4 av = *reinterpret_cast<const uint4*>(a + t * lda + c);
    const uint4 bv = *reinterpret_cast<const uint4*>(b + t * ldb + c);
    const unsigned du[4] = {dv.x, dv.y, dv.z, dv.w}, au[4] = {av.x, av.y, av.z, av.w}, bu[4] = {bv.x, bv.y, bv.z, bv.w};
    unsigned oa[4], ob[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d0, d1, a0, a1, b0, b1;
      unpack_bf2(du[e], d0, d1);
      unpack_bf2(au[e], a0, a1);
      unpack_bf2(bu[e], b0, b1);
      oa[e] = pack_bf2(d0 * b0, d1 * b1);
      ob[e] = pack_bf2(d0 * a0, d1 * a1);
    }
    *reinterpret_cast<uint4*>(da + t * ldda + c) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
    *reinterpret_cast<uint4*>(db + t * lddb + c) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
  }
}

// depth-wise 3x3 (pad 1) weight gradient: partial[chunk][tap][c] = sum over the chunk's pixels of dy[pix][c] x[pix + off(tap)][c],
// partial[chunk][9][c] = sum dy (chunk = (sample, band of 8 image rows)).  LDS-tiled as dwconv3x3_tile_kernel of dat.hip: one workgroup
// per (sample, 8 image rows, 64 channels) walks the row band in 16-column tiles; the x halo tile and the dy tile arrive by LDS-DMA, a thread = (4-channel group, tile column) slides a 3 x 3 window
// down its column (three 8-byte LDS reads + one of dy per pixel) and keeps the ten sums of its four channels in registers.  (The register
// form before it issued its four loads per pixel inside the pixel loop: one memory round trip per pixel, 66 us per launch at DAT x4 size.)
constexpr int WT_H = 8, WT_W = 16, WT_CB = 64;
constexpr int WT_XP = (WT_H + 2) * (WT_W + 2) * 8, WT_YP = WT_H * WT_W * 8;          // 16-byte pieces of the two tiles
constexpr int WT_XI = (WT_XP + 255) / 256, WT_YI = (WT_YP + 255) / 256;
__device__ uint4 g_wt_zero[1];

__global__ __launch_bounds__(256) void dwconv3x3_wgrad_tile_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                                   float* __restrict__ partial, int H, int W, int C8) {
  __shared__ __attribute__((aligned(16))) unsigned char xt[WT_XI * 256 * 16];
  __shared__ __attribute__((aligned(16))) unsigned char yt[WT_YI * 256 * 16];
  __shared__ float red[4][10][WT_CB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, y0 = blockIdx.x * WT_H, c0 = blockIdx.z * WT_CB;
  const int CP = C8 * 8;
  const unsigned xbase = (unsigned)(size_t)xt, ybase = (unsigned)(size_t)yt;
  const bf16_t* zero16 = reinterpret_cast<const bf16_t*>(g_wt_zero);
  const int cg = tid & 15, col = tid >> 4;
  srk_f32x2_t acc[10][2];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t][0] = acc[t][1] = srk_f32x2_t{0.f, 0.f};
  const int tilesx = (W + WT_W - 1) / WT_W;
  for (int tx = 0; tx < tilesx; ++tx) {
    const int x0 = tx * WT_W;
    if (tx > 0) __syncthreads();                       // the previous tile's readers are done
#pragma unroll
    for (int it = 0; it < WT_XI; ++it) {
      const int p = it * 256 + wave * 64 + lane;
      const int pix = p >> 3, ch = p & 7;
      const int py = pix / (WT_W + 2), px = pix - py * (WT_W + 2);
      const int yy = y0 + py - 1, xx = x0 + px - 1;
      const bool ok = p < WT_XP && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && c0 + ch * 8 < CP;
      const bf16_t* src = ok ? x + (((long long)b * H + yy) * W + xx) * ldx + c0 + ch * 8 : zero16;
      srk_glds16(src, __builtin_amdgcn_readfirstlane(xbase + (unsigned)((it * 256 + wave * 64) * 16)));
    }
#pragma unroll
    for (int it = 0; it < WT_YI; ++it) {
      const int p = it * 256 + wave * 64 + lane;
      const int pix = p >> 3, ch = p & 7;
      const int py = pix / WT_W, px = pix - py * WT_W;
      const int yy = y0 + py, xx = x0 + px;
      const bool ok = p < WT_YP && yy < H && xx < W && c0 + ch * 8 < CP;
      const bf16_t* src = ok ? dy + (((long long)b * H + yy) * W + xx) * lddy + c0 + ch * 8 : zero16;
      srk_glds16(src, __builtin_amdgcn_readfirstlane(ybase + (unsigned)((it * 256 + wave * 64) * 16)));
    }
    srk_wait_vmcnt<0>();
    __syncthreads();
    const unsigned char* xl = xt + col * 128 + cg * 8;            // halo pixel (r, col + dx) at + (r * (WT_W + 2) + dx) * 128
    const unsigned char* yl = yt + col * 128 + cg * 8;            // dy pixel (r, col) at + r * WT_W * 128
    srk_f32x2_t win[3][3][2];
    auto take_row = [&](int slot, int r) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const uint2 v = *reinterpret_cast<const uint2*>(xl + (r * (WT_W + 2) + dx) * 128);
        win[slot][dx][0] = srk_f32x2_t{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
        win[slot][dx][1] = srk_f32x2_t{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
      }
    };
    take_row(0, 0);
    take_row(1, 1);
#pragma unroll
    for (int it = 0; it < WT_H; ++it) {
      take_row((it + 2) % 3, it + 2);
      const uint2 dv = *reinterpret_cast<const uint2*>(yl + it * WT_W * 128);
      const srk_f32x2_t d[2] = {srk_f32x2_t{__uint_as_float(dv.x << 16), __uint_as_float(dv.x & 0xffff0000u)},
                                srk_f32x2_t{__uint_as_float(dv.y << 16), __uint_as_float(dv.y & 0xffff0000u)}};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc[9][h] += d[h];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) acc[r * 3 + dx][h] = __builtin_elementwise_fma(d[h], win[(it + r) % 3][dx][h], acc[r * 3 + dx][h]);
      }
    }
  }
  // sum over the 16 tile columns: the wave's four (lanes l, l + 16, l + 32, l + 48), then the four waves through LDS
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float v = acc[t][h][e];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wave][t][cg * 4 + 2 * h + e] = v;
      }
  __syncthreads();
  float* dst = partial + ((long long)b * gridDim.x + blockIdx.x) * 10 * CP;
  for (int i = tid; i < 10 * WT_CB; i += 256) {
    const int t = i / WT_CB, ch = i - t * WT_CB;
    if (c0 + ch < CP) dst[t * CP + c0 + ch] = (red[0][t][ch] + red[1][t][ch]) + (red[2][t][ch] + red[3][t][ch]);
  }
}

// comb = a_chan * cg[b][c] + a_tok * tg[t]  (dual_gate_combine).  Backward: d_chan = dcomb * cg, d_tok = dcomb * tg,
// dsmap[t] = (sum_c dcomb * a_tok) * tg (1 - tg)   [gradient w.r.t. the spatial map BEFORE its sigmoid],
// dcg_partial[b][chunk][c] = sum over the chunk's tokens of dcomb * a_chan   [gradient w.r.t. the channel gate AFTER its sigmoid]
__global__ __launch_bounds__(256) void dual_gate_bwd_kernel(const bf16_t* __restrict__ dcomb, const bf16_t* __restrict__ a_chan,
                                                            const bf16_t* __restrict__ a_tok, const float* __restrict__ cgate,
                                                            const float* __restrict__ tgate, bf16_t* __restrict__ d_chan,
                                                            bf16_t* __restrict__ d_tok, float* __restrict__ dcg_partial,
                                                            float* __restrict__ dsmap, int HW, int CA) {
  // one workgroup per (sample, 64-token chunk); a wave handles 16 tokens, 4 lanes per token?  Simple form: thread = 8-channel piece
  // x token lane; CA / 8 <= 32 pieces, 8 token lanes
  __shared__ float red[8][256];
  __shared__ float trow[64];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int C8 = CA / 8;
  const int piece = tid % 32, tl = tid / 32;
  const int t0 = chunk * 64;
  float cg[8], accg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    cg[e] = piece < C8 ? cgate[(long long)b * CA + piece * 8 + e] : 0.f;
    accg[e] = 0.f;
  }
  if (tid < 64) trow[tid] = 0.f;
  __syncthreads();
  for (int tt = tl; tt < 64 && t0 + tt < HW; tt += 8) {
    const long long t = (long long)b * HW + t0 + tt;
    float part = 0.f;
    if (piece < C8) {
      const long long o = t * CA + piece * 8;
      const uint4 dv = *reinterpret_cast<const uint4*>(dcomb + o);
      const uint4 cv = *reinterpret_cast<const uint4*>(a_chan + o);
      const uint4 tv = *reinterpret_cast<const uint4*>(a_tok + o);
      const float tg = tgate[t];
      const unsigned du[4] = {dv.x, dv.y, dv.z, dv.w}, cu[4] = {cv.x, cv.y, cv.z, cv.w}, tu[4] = {tv.x, tv.y, tv.z, tv.w};
      unsigned oc[4], ot[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d0, d1, c0, c1, k0, k1;
        unpack_bf2(du[e], d0, d1);
        unpack_bf2(cu[e], c0, c1);
        unpack_bf2(tu[e], k0, k1);
        oc[e] = pack_bf2(d0 * cg[2 * e], d1 * cg[2 * e + 1]);
        ot[e] = pack_bf2(d0 * tg, d1 * tg);
        accg[2 * e] += d0 * c0;
        accg[2 * e + 1] += d1 * c1;
        part += d0 * k0 + d1 * k1;
      }
      *reinterpret_cast<uint4*>(d_chan + o) = make_uint4(oc[0], oc[1], oc[2], oc[3]);
      *reinterpret_cast<uint4*>(d_tok + o) = make_uint4(ot[0], ot[1], ot[2], ot[3]);
    }
    // sum over the 32 pieces of this token: the 32 lanes tid % 32 of token lane tl are half a wave
    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
    part += __shfl_xor(part, 8, 64); part += __shfl_xor(part, 16, 64);
    if (piece == 0) trow[tt] = part;
  }
  __syncthreads();
  if (tid < 64 && t0 + tid < HW) {
    const long long t = (long long)b * HW + t0 + tid;
    const float tg = tgate[t];
    dsmap[t] = trow[tid] * tg * (1.0f - tg);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __syncthreads();
    red[tl][piece * 8 + e] = accg[e];
  }
  __syncthreads();
  if (tid < CA) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][tid];
    dcg_partial[((long long)b * gridDim.x + chunk) * CA + tid] = s;
  }
}

// ---- spatial interaction: y1 = W0 x + b0 (S <= 16 outputs), z = y1 * s + t (BatchNorm), a = gelu(z), smap = w3 . a + b3 ---------------
// 16 lanes per token (lane j holds channels 64 i + 4 j .. + 3, the LayerNorm layout: a token row is read as contiguous 128-byte runs),
// 16 tokens per workgroup step, 16 steps per workgroup (256 tokens).  y1[s] by a 16-lane all-reduce per s; W0 and the per-s vectors in
// LDS.  WHAT 0: statistics of y1 (sum, sum sq).  WHAT 1: backward statistics (sum dz, sum dz y1, sum dsmap a, sum dsmap) with
// dz = dsmap * w3 * gelu'(z).  WHAT 2: dy1 = A dz + B y1 + C; dx (+)= W0^T dy1 per lane (coalesced stores); d W0 [s][c] += dy1[s] x[c]
// through LDS (thread c owns column c and keeps its 16 sums in registers across the steps); d b0 = sum dy1.
template <int WHAT, int NV>
__global__ __launch_bounds__(256) void spatial_gate_train_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ W0,
                                                                 const float* __restrict__ b0, const float* __restrict__ sc,
                                                                 const float* __restrict__ sh, const float* __restrict__ w3,
                                                                 const float* __restrict__ dsmap, const float* __restrict__ cA,
                                                                 const float* __restrict__ cB, const float* __restrict__ cC,
                                                                 bf16_t* __restrict__ dx, int lddx, int accumulate,
                                                                 float* __restrict__ partial, long long rows, int S) {
  constexpr int C = NV * 64;
  constexpr int NVAL = WHAT == 0 ? 2 : 4;
  extern __shared__ float sm[];
  float* Wl = sm;                        // [16][C], rows >= S zero
  float* vec = Wl + 16 * C;              // [7][16]: b0, sc, sh, w3, cA, cB, cC (zero beyond S)
  float* red = vec + 7 * 16;             // WHAT 0 / 1: [16 groups][NVAL * 16]; WHAT 2: dyl [16][16] + xs [16][C]
  const int tid = threadIdx.x, grp = tid >> 4, j = tid & 15;
  for (int i = tid; i < 16 * C; i += 256) Wl[i] = i < S * C ? W0[i] : 0.f;
  if (tid < 7 * 16) {
    const int which = tid >> 4, s = tid & 15;
    const float* src = which == 0 ? b0 : which == 1 ? sc : which == 2 ? sh : which == 3 ? w3 : which == 4 ? cA : which == 5 ? cB : cC;
    vec[tid] = (src != nullptr && s < S) ? src[s] : 0.f;
  }
  __syncthreads();
  float acc[NVAL][16];                   // WHAT 0 / 1: per-token sums (identical in the 16 lanes of a token)
  float dwa[16];                         // WHAT 2: d W0[s][tid]
  float dba[16];                         // WHAT 2: d b0 (per group)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
#pragma unroll
    for (int k = 0; k < NVAL; ++k) acc[k][s] = 0.f;
    dwa[s] = dba[s] = 0.f;
  }
  float* dyl = red;                      // [16 tokens][16]
  float* xs = red + 256;                 // [16 tokens][C]
  // the token row (and its dsmap) of step it + 1 is requested while step it computes: loaded at the top of its own step, every one of
  // the 16 steps began with an exposed memory round trip (52 % of the wave-cycles parked)
  uint2 un[NV];
  float dsn = 0.f;
  auto fetch = [&](int it) {
    const long long tn = (long long)blockIdx.x * 256 + it * 16 + grp;
#pragma unroll
    for (int i = 0; i < NV; ++i) un[i] = tn < rows ? *reinterpret_cast<const uint2*>(x + tn * ldx + 64 * i + 4 * j) : make_uint2(0u, 0u);
    if constexpr (WHAT != 0) dsn = tn < rows ? dsmap[tn] : 0.f;
  };
  fetch(0);
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
    const long long t = (long long)blockIdx.x * 256 + it * 16 + grp;
    const bool ok = t < rows;
    float xv[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      unpack_bf2(un[i].x, xv[i][0], xv[i][1]);
      unpack_bf2(un[i].y, xv[i][2], xv[i][3]);
    }
    const float ds_cur = dsn;
    if (it + 1 < 16) fetch(it + 1);
    float y1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(Wl + s * C + 64 * i + 4 * j);
        d += xv[i][0] * wv.x + xv[i][1] * wv.y + xv[i][2] * wv.z + xv[i][3] * wv.w;
      }
      y1[s] = wave_sum16(d) + vec[s];
    }
    if constexpr (WHAT == 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc[0][s] += ok ? y1[s] : 0.f;
        acc[1][s] += ok ? y1[s] * y1[s] : 0.f;
      }
    } else {
      const float ds = ds_cur;
      float dy1[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float z = y1[s] * vec[16 + s] + vec[32 + s];
        const float dz = ds * vec[48 + s] * dgelu_shared_exp(z);
        if constexpr (WHAT == 1) {
          acc[0][s] += dz;
          acc[1][s] += dz * y1[s];
          acc[2][s] += s < S ? ds * gelu_f(z) : 0.f;
          acc[3][s] += s == 0 ? ds : 0.f;
        } else {
          dy1[s] = (ok && s < S) ? vec[64 + s] * dz + vec[80 + s] * y1[s] + vec[96 + s] : 0.f;
          dba[s] += dy1[s];
        }
      }
      if constexpr (WHAT == 2) {
        if (ok) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) {
              const uint2 u = *reinterpret_cast<const uint2*>(dx + t * lddx + 64 * i + 4 * j);
              unpack_bf2(u.x, o[0], o[1]);
              unpack_bf2(u.y, o[2], o[3]);
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
              const float4 wv = *reinterpret_cast<const float4*>(Wl + s * C + 64 * i + 4 * j);
              o[0] = fmaf(wv.x, dy1[s], o[0]);
              o[1] = fmaf(wv.y, dy1[s], o[1]);
              o[2] = fmaf(wv.z, dy1[s], o[2]);
              o[3] = fmaf(wv.w, dy1[s], o[3]);
            }
            *reinterpret_cast<uint2*>(dx + t * lddx + 64 * i + 4 * j) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
          }
        }
        // d W0: the 16 tokens' dy1 and x rows through LDS, thread c accumulates column c
        __syncthreads();                 // the previous step's readers are done
        if (j == 0) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            *reinterpret_cast<float4*>(dyl + grp * 16 + 4 * s4) = make_float4(dy1[4 * s4], dy1[4 * s4 + 1], dy1[4 * s4 + 2], dy1[4 * s4 + 3]);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *reinterpret_cast<float4*>(xs + grp * C + 64 * i + 4 * j) = make_float4(xv[i][0], xv[i][1], xv[i][2], xv[i][3]);
        __syncthreads();
        if (tid < C) {
#pragma unroll 4
          for (int k = 0; k < 16; ++k) {
            const float xk = xs[k * C + tid];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const float4 d = *reinterpret_cast<const float4*>(dyl + k * 16 + 4 * s4);
              dwa[4 * s4] = fmaf(d.x, xk, dwa[4 * s4]);
              dwa[4 * s4 + 1] = fmaf(d.y, xk, dwa[4 * s4 + 1]);
              dwa[4 * s4 + 2] = fmaf(d.z, xk, dwa[4 * s4 + 2]);
              dwa[4 * s4 + 3] = fmaf(d.w, xk, dwa[4 * s4 + 3]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (WHAT != 2) {
    if (j == 0) {
#pragma unroll
      for (int k = 0; k < NVAL; ++k)
#pragma unroll
        for (int s = 0; s < 16; ++s) red[grp * (NVAL * 16) + k * 16 + s] = acc[k][s];
    }
    __syncthreads();
    if (tid < NVAL * 16) {
      float t = 0.f;
#pragma unroll
      for (int gq = 0; gq < 16; ++gq) t += red[gq * (NVAL * 16) + tid];
      partial[(long long)blockIdx.x * (NVAL * 16) + tid] = t;
    }
  } else {
    float* dst = partial + (long long)blockIdx.x * 16 * (C + 1);
    if (tid < C) {
#pragma unroll
      for (int s = 0; s < 16; ++s) dst[s * C + tid] = dwa[s];
    }
    if (j == 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) red[grp * 16 + s] = dba[s];
    }
    __syncthreads();
    if (tid < 16) {
      float t = 0.f;
#pragma unroll
      for (int gq = 0; gq < 16; ++gq) t += red[gq * 16 + tid];
      dst[16 * C + tid] = t;
    }
  }
}

// LayerNorm backward on bf16 rows: x [rows][ldx] (first C columns normalised), dy [rows][lddy]; dx bf16; dgamma / dbeta partials per
// workgroup [block][2][C].  16 lanes per row (4 rows per wave); a lane owns the 8-column pieces j, j + 16, ... of every row it visits
// (16-byte loads and stores), so its share of d gamma / d beta stays in registers (K pieces of 8) and the 16 lane groups of the
// workgroup are summed once, at the end, through LDS (no LDS float atomics: they are the slow path on this part).
template <int K>
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, bf16_t* __restrict__ dx, int lddx,
                                                        float* __restrict__ partial, long long rows, int C, int CPo) {
  extern __shared__ float sm[];          // [16 lane groups][2][128 K]
  constexpr int W = 128 * K;             // columns covered by the K pieces of the 16 lanes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, sub = lane >> 4;
  const float invC = 1.0f / (float)C;
  float dg[K][8], db[K][8], gm[K][8];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (j + 16 * k) * 8 + e;
      dg[k][e] = db[k][e] = 0.f;
      gm[k][e] = c < C ? gamma[c] : 0.f;
    }
  for (long long m = ((long long)blockIdx.x * 4 + wave) * 4 + sub; m < rows; m += (long long)gridDim.x * 16) {
    float xv[K][8], dv[K][8];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c0 = (j + 16 * k) * 8;
      uint4 xu = make_uint4(0, 0, 0, 0), du = make_uint4(0, 0, 0, 0);
      if (c0 < C) {                      // a piece that straddles C reads up to 7 columns of the rows' padding (< CP_out <= row stride)
        xu = *reinterpret_cast<const uint4*>(x + m * ldx + c0);
        du = *reinterpret_cast<const uint4*>(dy + m * lddy + c0);
      }
      unpack_bf2(xu.x, xv[k][0], xv[k][1]); unpack_bf2(xu.y, xv[k][2], xv[k][3]); unpack_bf2(xu.z, xv[k][4], xv[k][5]); unpack_bf2(xu.w, xv[k][6], xv[k][7]);
      unpack_bf2(du.x, dv[k][0], dv[k][1]); unpack_bf2(du.y, dv[k][2], dv[k][3]); unpack_bf2(du.z, dv[k][4], dv[k][5]); unpack_bf2(du.w, dv[k][6], dv[k][7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (c0 + e >= C) xv[k][e] = dv[k][e] = 0.f;
        s += xv[k][e];
        q += xv[k][e] * xv[k][e];
      }
    }
    const float mean = wave_sum16(s) * invC;
    const float var = fmaxf(wave_sum16(q) * invC - mean * mean, 0.f);
    const float rstd = rsqrtf(var + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[k][e] = (j + 16 * k) * 8 + e < C ? (xv[k][e] - mean) * rstd : 0.f;         // x hat
        const float t = dv[k][e] * gm[k][e];
        s1 += t;
        s2 += t * xv[k][e];
      }
    s1 = wave_sum16(s1) * invC;
    s2 = wave_sum16(s2) * invC;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int c0 = (j + 16 * k) * 8;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = c0 + e < C ? rstd * (dv[k][e] * gm[k][e] - s1 - xv[k][e] * s2) : 0.f;
        dg[k][e] += dv[k][e] * xv[k][e];
        db[k][e] += dv[k][e];
      }
      if (c0 < CPo)
        *reinterpret_cast<uint4*>(dx + m * lddx + c0) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
  }
  const int grp = wave * 4 + sub;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[(grp * 2 + 0) * W + (j + 16 * k) * 8 + e] = dg[k][e];
      sm[(grp * 2 + 1) * W + (j + 16 * k) * 8 + e] = db[k][e];
    }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float t = 0.f;
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) t += sm[(gq * 2 + which) * W + c];
    partial[(long long)blockIdx.x * 2 * C + i] = t;
  }
}

// ---- channel attention ------------------------------------------------------------------------------------------------------------------
constexpr int TG_CH = 256;            // tokens per Gram chunk
constexpr int TG_SZ = 32 * 32 + 64;   // partial: G[32][32] = sum_n x[n][i] y[n][j], sum_n x[n][i]^2, sum_n y[n][j]^2

// out[n][32 h + i] (+)= sum_j M[b][h][i][j] src[n][32 h + j]  (+ dg[b][h][i] * src2[n][32 h + i]) : fp32 matrices, fp32 accumulation
__global__ __launch_bounds__(256) void chan_apply_mat_kernel(const float* __restrict__ M, const bf16_t* __restrict__ src, int lds_,
                                                             const float* __restrict__ dg, const bf16_t* __restrict__ src2, int lds2,
                                                             bf16_t* __restrict__ out, int ldo, int nH, int N, int accumulate) {
  __shared__ float Ml[32][33];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) Ml[i >> 5][i & 31] = M[((long long)b * nH + h) * 1024 + i];
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= N) return;
  const long long row = (long long)b * N + n;
  float v[32];
#pragma unroll
  for (int c = 0; c < 32; c += 8) {
    const uint4 sv = *reinterpret_cast<const uint4*>(src + row * lds_ + h * 32 + c);
    unpack_bf2(sv.x, v[c], v[c + 1]); unpack_bf2(sv.y, v[c + 2], v[c + 3]); unpack_bf2(sv.z, v[c + 4], v[c + 5]); unpack_bf2(sv.w, v[c + 6], v[c + 7]);
  }
#pragma unroll
  for (int c = 0; c < 32; c += 8) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) s = fmaf(Ml[c + e][j], v[j], s);
      o[e] = s;
    }
    if (dg) {
      const uint4 s2 = *reinterpret_cast<const uint4*>(src2 + row * lds2 + h * 32 + c);
      float w[8];
      unpack_bf2(s2.x, w[0], w[1]); unpack_bf2(s2.y, w[2], w[3]); unpack_bf2(s2.z, w[4], w[5]); unpack_bf2(s2.w, w[6], w[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += dg[((long long)b * nH + h) * 32 + c + e] * w[e];
    }
    if (accumulate) {
      const uint4 ov = *reinterpret_cast<const uint4*>(out + row * ldo + h * 32 + c);
      float w[8];
      unpack_bf2(ov.x, w[0], w[1]); unpack_bf2(ov.y, w[2], w[3]); unpack_bf2(ov.z, w[4], w[5]); unpack_bf2(ov.w, w[6], w[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += w[e];
    }
    *reinterpret_cast<uint4*>(out + row * ldo + h * 32 + c) =
        make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
  }
}

}  // namespace

int srk_launch_win_attn_bwd_padded(const bf16_t* qkv, int ldq, int CA, const float* bias, const bf16_t* dout, int ldo, bf16_t* dqkv,
                                   float* dbias, float* tiles, int B, int H, int W, int Hp, int Wp, int wh, int ww, int sy, int sx, int nH,
                                   float scale, hipStream_t stream);

extern "C" {

#define REQP(c, ...) SRK_REQUIRE(c, SRK_E_SHAPE, __VA_ARGS__)

int srk_sum_rows_f32(const float* in, int outer, int R, int n, float* out, srk_stream_t stream) {
  SRK_REQUIRE(in && out, SRK_E_NULL, "sum_rows: null pointer");
  REQP(outer > 0 && outer <= 65535 && R > 0 && n > 0, "sum_rows: bad shape");
  hipLaunchKernelGGL(sum_rows_kernel, dim3((n + 63) / 64, outer), dim3(256), 0, (hipStream_t)stream, in, R, n, out);
  return srk_check_launch("sum_rows");
}

int srk_bn_train_coeffs(const float* partial, int R, int row_stride, int ld, int C, float n, const float* gamma, const float* beta, float eps, float* coef,
                        float* running_mean, float* running_var, float momentum, const int* real_of, srk_stream_t stream) {
  SRK_REQUIRE(partial && gamma && beta && coef, SRK_E_NULL, "bn_train_coeffs: null pointer");
  REQP(R > 0 && C > 0 && C <= ld && row_stride >= 2 * ld && n > 0.f && (running_mean == nullptr) == (running_var == nullptr),
       "bn_train_coeffs: bad shape");
  hipLaunchKernelGGL(bn_train_coeffs_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, R, row_stride, ld, C, n, gamma, beta, eps, coef,
                     running_mean, running_var, momentum, real_of);
  return srk_check_launch("bn_train_coeffs");
}

int srk_bn_train_bwd_coeffs(const float* partial, int R, int row_stride, int ld, int C, float n, const float* fwd_coef, float* coef,
                            srk_stream_t stream) {
  SRK_REQUIRE(partial && fwd_coef && coef, SRK_E_NULL, "bn_train_bwd_coeffs: null pointer");
  REQP(R > 0 && C > 0 && C <= ld && row_stride >= 2 * ld && n > 0.f, "bn_train_bwd_coeffs: bad shape");
  hipLaunchKernelGGL(bn_train_bwd_coeffs_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, R, row_stride, ld, C, n, fwd_coef,
                     coef);
  return srk_check_launch("bn_train_bwd_coeffs");
}

int64_t srk_chan_stats_chunks(int64_t rows) { return rows <= 0 ? 0 : (rows + ST_ROWS - 1) / ST_ROWS; }

int srk_chan_stats(const uint16_t* p, int ldp, const uint16_t* q, int ldq, float* partial, int samples, int64_t rows_per_sample, int C8,
                   srk_stream_t stream) {
  SRK_REQUIRE(p && q && partial, SRK_E_NULL, "chan_stats: null pointer");
  REQP(samples > 0 && samples <= 65535 && rows_per_sample > 0 && C8 > 0 && C8 <= 64 && ldp % 8 == 0 && ldq % 8 == 0, "chan_stats: bad shape");
  hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)srk_chan_stats_chunks(rows_per_sample), samples), dim3(256), 0, (hipStream_t)stream, p, ldp,
                     q, ldq, partial, (long long)rows_per_sample, C8);
  return srk_check_launch("chan_stats");
}

int srk_affine_act_bf16(const uint16_t* x, int ldx, const float* scale, const float* shift, uint16_t* out, int ldo, int64_t rows, int C8,
                        int rows_per_sample, int act, srk_stream_t stream) {
  SRK_REQUIRE(x && scale && shift && out, SRK_E_NULL, "affine_act: null pointer");
  REQP(rows > 0 && C8 > 0 && rows * C8 < (1LL << 31) && ldx % 8 == 0 && ldo % 8 == 0, "affine_act: bad shape");
  const EwGeom eg = ew_geom(rows, C8, 4);
  hipLaunchKernelGGL(token_elementwise_kernel<0>, eg.grid, dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16_t*)nullptr, 0, scale, shift,
                     (const float*)nullptr, out, ldo, (long long)rows, C8, rows_per_sample, act, eg.cw, eg.lanes);
  return srk_check_launch("affine_act");
}

int srk_dgelu_affine_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const float* scale, const float* shift, uint16_t* out,
                          int ldo, int64_t rows, int C8, srk_stream_t stream) {
  SRK_REQUIRE(dy && x && scale && shift && out, SRK_E_NULL, "dgelu_affine: null pointer");
  REQP(rows > 0 && C8 > 0 && rows * C8 < (1LL << 31) && lddy % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "dgelu_affine: bad shape");
  hipLaunchKernelGGL(token_elementwise_flat_kernel<1>, dim3(grid_cap(rows * C8)), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, scale, shift,
                     (const float*)nullptr, out, ldo, (long long)rows, C8, 0, 0);
  return srk_check_launch("dgelu_affine");
}

int srk_lincomb2_bf16(const uint16_t* p, int ldp, const uint16_t* q, int ldq, const float* A, const float* Bc, const float* Cc, uint16_t* out,
                      int ldo, int64_t rows, int C8, int rows_per_sample, int accumulate, srk_stream_t stream) {
  SRK_REQUIRE(out, SRK_E_NULL, "lincomb2: null pointer");
  REQP(rows > 0 && C8 > 0 && rows * C8 < (1LL << 31) && ldo % 8 == 0 && (p == nullptr || ldp % 8 == 0) && (q == nullptr || ldq % 8 == 0) && (A == nullptr || p != nullptr) &&
           (Bc == nullptr || q != nullptr),
       "lincomb2: bad shape");
  const EwGeom eg = ew_geom(rows, C8, 4);
  hipLaunchKernelGGL(token_elementwise_kernel<2>, eg.grid, dim3(256), 0, (hipStream_t)stream, p, ldp, q, ldq, A, Bc, Cc, out, ldo, (long long)rows,
                     C8, rows_per_sample, accumulate, eg.cw, eg.lanes);
  return srk_check_launch("lincomb2");
}

int srk_mul_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* a, int lda, const uint16_t* b, int ldb, uint16_t* da, int ldda, uint16_t* db,
                     int lddb, int64_t rows, int C8, srk_stream_t stream) {
  SRK_REQUIRE(dy && a && b && da && db, SRK_E_NULL, "mul_bwd: null pointer");
  REQP(rows > 0 && C8 > 0 && rows * C8 < (1LL << 31) && (lddy | lda | ldb | ldda | lddb) % 8 == 0, "mul_bwd: bad shape");
  hipLaunchKernelGGL(mul_bwd_kernel, dim3(grid_cap(rows * C8)), dim3(256), 0, (hipStream_t)stream, dy, lddy, a, lda, b, ldb, da, ldda, db, lddb,
                     (long long)rows, C8);
  return srk_check_launch("mul_bwd");
}

int srk_dwconv3x3_wgrad_chunks(int H) { return H <= 0 ? 0 : (H + WT_H - 1) / WT_H; }

int srk_dwconv3x3_wgrad(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, float* partial, int B, int H, int W, int C8,
                        srk_stream_t stream) {
  SRK_REQUIRE(dy && x && partial, SRK_E_NULL, "dwconv3x3_wgrad: null pointer");
  REQP(B > 0 && H > 0 && W > 0 && C8 > 0 && C8 <= 64 && lddy % 8 == 0 && ldx % 8 == 0, "dwconv3x3_wgrad: bad shape");
  REQP(B <= 65535, "dwconv3x3_wgrad: at most 65535 samples");
  hipLaunchKernelGGL(dwconv3x3_wgrad_tile_kernel, dim3((H + WT_H - 1) / WT_H, B, (8 * C8 + WT_CB - 1) / WT_CB), dim3(256), 0, (hipStream_t)stream, dy,
                     lddy, x, ldx, partial, H, W, C8);
  return srk_check_launch("dwconv3x3_wgrad");
}

int srk_dual_gate_bwd(const uint16_t* dcomb, const uint16_t* a_chan, const uint16_t* a_tok, const float* cgate, const float* tgate,
                      uint16_t* d_chan, uint16_t* d_tok, float* dcg_partial, float* dsmap, int B, int HW, int CA, srk_stream_t stream) {
  SRK_REQUIRE(dcomb && a_chan && a_tok && cgate && tgate && d_chan && d_tok && dcg_partial && dsmap, SRK_E_NULL, "dual_gate_bwd: null pointer");
  REQP(B > 0 && HW > 0 && CA > 0 && CA <= 256 && CA % 8 == 0, "dual_gate_bwd: bad shape");
  hipLaunchKernelGGL(dual_gate_bwd_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, dcomb, a_chan, a_tok, cgate, tgate, d_chan,
                     d_tok, dcg_partial, dsmap, HW, CA);
  return srk_check_launch("dual_gate_bwd");
}

// what: 0 forward statistics, 1 backward statistics, 2 backward apply (see spatial_gate_train_kernel)
int srk_spatial_gate_train(int what, const uint16_t* x, int ldx, const float* W0, const float* b0, const float* bn_scale, const float* bn_shift,
                           const float* w3, const float* dsmap, const float* cA, const float* cB, const float* cC, uint16_t* dx, int lddx,
                           int accumulate, float* partial, int64_t rows, int C, int S, srk_stream_t stream) {
  SRK_REQUIRE(x && W0 && b0 && partial, SRK_E_NULL, "spatial_gate_train: null pointer");
  REQP(rows > 0 && C > 0 && C % 8 == 0 && C <= 256 && S > 0 && S <= 16 && ldx % 8 == 0 && what >= 0 && what <= 2, "spatial_gate_train: bad shape");
  SRK_REQUIRE(what == 0 || (bn_scale && bn_shift && w3 && dsmap), SRK_E_NULL, "spatial_gate_train: backward operands missing");
  SRK_REQUIRE(what != 2 || (cA && cB && cC && dx && lddx % 8 == 0), SRK_E_NULL, "spatial_gate_train: apply operands missing");
  REQP(C % 64 == 0, "spatial_gate_train: C must be a multiple of 64 (the head-padded width)");
  const unsigned grid = (unsigned)((rows + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define SGT_LAUNCH(WHAT_, NV_)                                                                                                              \
  {                                                                                                                                         \
    constexpr size_t lds = (size_t)(16 * NV_ * 64 + 7 * 16 + (WHAT_ == 2 ? 256 + 16 * NV_ * 64 : 16 * 64)) * sizeof(float);                 \
    static SrkPerDevice<bool> configured_pd;                                                                                                \
    bool& configured = configured_pd.here();                                                                                                \
    if (!configured) {                                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_gate_train_kernel<WHAT_, NV_>),                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                                        \
        srk_set_error("spatial_gate_train: cannot reserve LDS");                                                                            \
        return SRK_E_LAUNCH;                                                                                                                \
      }                                                                                                                                     \
      configured = true;                                                                                                                    \
    }                                                                                                                                       \
    hipLaunchKernelGGL((spatial_gate_train_kernel<WHAT_, NV_>), dim3(grid), dim3(256), lds, st, x, ldx, W0, b0, bn_scale, bn_shift, w3,     \
                       dsmap, cA, cB, cC, dx, lddx, accumulate, partial, (long long)rows, S);                                               \
  }
#define SGT_NV(WHAT_)                                  \
  switch (C / 64) {                                    \
    case 1: SGT_LAUNCH(WHAT_, 1) break;                \
    case 2: SGT_LAUNCH(WHAT_, 2) break;                \
    case 3: SGT_LAUNCH(WHAT_, 3) break;                \
    default: SGT_LAUNCH(WHAT_, 4) break;               \
  }
  if (what == 0) SGT_NV(0) else if (what == 1) SGT_NV(1) else SGT_NV(2)
#undef SGT_NV
#undef SGT_LAUNCH
  return srk_check_launch("spatial_gate_train");
}

int64_t srk_rowln_bwd_blocks(int64_t rows) {
  const int64_t g = (rows + 15) / 16;
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

int srk_rowln_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const float* gamma, uint16_t* dx, int lddx, float* partial,
                       int64_t rows, int C, int CP_out, srk_stream_t stream) {
  SRK_REQUIRE(dy && x && gamma && dx && partial, SRK_E_NULL, "rowln_bwd: null pointer");
  REQP(rows > 0 && C > 0 && C <= CP_out && CP_out % 8 == 0 && CP_out <= 512 && (lddy | ldx | lddx) % 8 == 0 && lddy >= CP_out && ldx >= CP_out &&
           lddx >= CP_out,
       "rowln_bwd: bad shape (CP_out and the row strides multiples of 8, C <= CP_out <= 512 <= strides)");
  const dim3 grid((unsigned)srk_rowln_bwd_blocks(rows));
  hipStream_t st = (hipStream_t)stream;
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rowln_bwd_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 2 * 384 * 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rowln_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 2 * 512 * 4) != hipSuccess) {
      srk_set_error("rowln_bwd: cannot reserve LDS");
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  if (CP_out <= 128)
    hipLaunchKernelGGL(rowln_bwd_kernel<1>, grid, dim3(256), 16 * 2 * 128 * sizeof(float), st, dy, lddy, x, ldx, gamma, dx, lddx, partial,
                       (long long)rows, C, CP_out);
  else if (CP_out <= 384)
    hipLaunchKernelGGL(rowln_bwd_kernel<3>, grid, dim3(256), 16 * 2 * 384 * sizeof(float), st, dy, lddy, x, ldx, gamma, dx, lddx, partial,
                       (long long)rows, C, CP_out);
  else
    hipLaunchKernelGGL(rowln_bwd_kernel<4>, grid, dim3(256), 16 * 2 * 512 * sizeof(float), st, dy, lddy, x, ldx, gamma, dx, lddx, partial,
                       (long long)rows, C, CP_out);
  return srk_check_launch("rowln_bwd");
}

int64_t srk_chan_gram_floats(int B, int N, int num_heads) { return (int64_t)B * num_heads * ((N + TG_CH - 1) / TG_CH) * TG_SZ; }

// partial [B][heads][chunks][1088]: G = sum_n x[n][i] y[n][j], sum x^2, sum y^2 per 256-token chunk (the caller sums the chunks)
int srk_chan_gram(const uint16_t* x, int ldx, const uint16_t* y, int ldy, float* partial, int B, int N, int num_heads, srk_stream_t stream) {
  return srk_launch_chan_gram(x, ldx, y, ldy, partial, B, N, num_heads, (hipStream_t)stream);      // csrc/dat.hip (matrix cores)
}

// out[n][32 h + i] (+)= sum_j M[b][h][i][j] src[n][32 h + j] + diag[b][h][i] src2[n][32 h + i]
int srk_chan_apply_mat(const float* M, const uint16_t* src, int ldsrc, const float* diag, const uint16_t* src2, int ldsrc2, uint16_t* out, int ldo,
                       int B, int N, int num_heads, int accumulate, srk_stream_t stream) {
  SRK_REQUIRE(M && src && out && (diag == nullptr || src2 != nullptr), SRK_E_NULL, "chan_apply_mat: null pointer");
  REQP(B > 0 && N > 0 && num_heads > 0 && ldsrc % 8 == 0 && ldo % 8 == 0 && (src2 == nullptr || ldsrc2 % 8 == 0), "chan_apply_mat: bad shape");
  hipLaunchKernelGGL(chan_apply_mat_kernel, dim3((N + 255) / 256, num_heads, B), dim3(256), 0, (hipStream_t)stream, M, src, ldsrc, diag, src2, ldsrc2,
                     out, ldo, num_heads, N, accumulate);
  return srk_check_launch("chan_apply_mat");
}

size_t srk_win_attention_bwd_padded_scratch(int B, int Hp, int Wp, int wh, int ww, int num_heads) {
  if (B <= 0 || wh <= 0 || ww <= 0 || num_heads <= 0 || Hp % wh || Wp % ww) return 0;
  // one dS tile per (window, head) + the transposed bias [heads][N][N]
  return ((size_t)B * (Hp / wh) * (Wp / ww) + 1) * num_heads * wh * ww * wh * ww * sizeof(float);
}

int srk_win_attention_bwd_padded(const uint16_t* qkv, int ldq, int CA, const float* bias, const uint16_t* d_out, int ldo, uint16_t* d_qkv,
                                 float* d_bias, void* scratch, int B, int H, int W, int Hp, int Wp, int wh, int ww, int shift_y, int shift_x,
                                 int num_heads, float scale, srk_stream_t stream) {
  return srk_launch_win_attn_bwd_padded(qkv, ldq, CA, bias, d_out, ldo, d_qkv, d_bias, (float*)scratch, B, H, W, Hp, Wp, wh, ww, shift_y, shift_x,
                                        num_heads, scale, (hipStream_t)stream);
}

}  // extern "C"
