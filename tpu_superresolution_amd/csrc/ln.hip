// LayerNorm over the channel dimension of the token stream, forward and backward
// (reference nn.LayerNorm uses: norm1/norm2 network_swinir.py:199,205, patch_embed.norm :519-528,
// final norm :725,800; eps 1e-5, affine).
//
// Token rows are fp32 [rows][CP] with CP = channels padded to a multiple of 64 (pad columns are
// zero and stay zero: gamma/beta are zero there).  A 16-lane group owns one row (4 rows per wave):
// lane j of the group holds columns 64*i + 4*j .. +3 (float4 loads, 256 B contiguous per group).
// Statistics are two-pass in registers over the C real columns.  `gather` folds the cyclic shift +
// window partition into the row index: output row m (window order) reads token win_row_to_token(m).
#include "common.h"

namespace {

constexpr int MAXV = 4;  // CP <= 256

template <int NV>  // NV = CP / 64
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ yb,
                                                     float* __restrict__ yf, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int C, WinGeom geom,
                                                     int gather, float eps) {
  constexpr int CP = NV * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, sub = lane >> 4;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 64 * i + 4 * j;
    gm[i] = make_float4(c < C ? gamma[c] : 0.f, c + 1 < C ? gamma[c + 1] : 0.f, c + 2 < C ? gamma[c + 2] : 0.f,
                        c + 3 < C ? gamma[c + 3] : 0.f);
    bt[i] = make_float4(c < C ? beta[c] : 0.f, c + 1 < C ? beta[c + 1] : 0.f, c + 2 < C ? beta[c + 2] : 0.f,
                        c + 3 < C ? beta[c + 3] : 0.f);
  }
  const float invC = 1.0f / (float)C;
  for (int m = (blockIdx.x * 4 + wave) * 4 + sub; m < rows; m += gridDim.x * 16) {
    const long long src = gather ? win_row_to_token(geom, m) : m;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = *reinterpret_cast<const float4*>(x + src * CP + 64 * i + 4 * j);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);   // pad columns are zero
    }
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
    const float rstd = rsqrtf(wave_sum16(q) * invC + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float o0 = v[i].x * rstd * gm[i].x + bt[i].x, o1 = v[i].y * rstd * gm[i].y + bt[i].y;
      const float o2 = v[i].z * rstd * gm[i].z + bt[i].z, o3 = v[i].w * rstd * gm[i].w + bt[i].w;
      if (yb) *reinterpret_cast<uint2*>(yb + (long long)m * CP + 64 * i + 4 * j) = pack_bf4(o0, o1, o2, o3);
      if (yf) *reinterpret_cast<float4*>(yf + (long long)m * CP + 64 * i + 4 * j) = make_float4(o0, o1, o2, o3);
    }
    if (j == 0 && mean_out) {
      mean_out[m] = mean;
      rstd_out[m] = rstd;
    }
  }
}

// Backward.  For row m (iteration order) with token t = gather ? map(m) : m:
//   dy     = dyb[dy_by_m ? m : t]          (bf16, grad w.r.t. the LN output)
//   xhat   = (x[t] - mean[s]) * rstd[s],   s = stats_by_m ? m : t
//   dx     = rstd * (dy*g - mean_c(dy*g) - xhat * mean_c(dy*g*xhat))
//   gx[t]  = (accumulate ? gx[t] : 0) + dx
//   gxb[out_by_m ? m : t] = bf16(gx[t] * rowscale[sample(t)])    (optional; rowscale = DropPath factor of
//                                                                 the branch that consumes gxb, or null)
//   dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy     (per-workgroup partials -> atomics)
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dyb, const float* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, float* __restrict__ gx,
                                                     bf16_t* __restrict__ gxb, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int C, WinGeom geom, int gather,
                                                     int dy_by_m, int stats_by_m, int out_by_m, int accumulate,
                                                     const float* __restrict__ rowscale, int rows_per_sample) {
  constexpr int CP = NV * 64;
  __shared__ float red[4][2][CP];          // per-wave partial rows (plain stores: LDS float atomics here cost 13 us of a 66-us launch)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, sub = lane >> 4;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 64 * i + 4 * j;
    gm[i] = make_float4(c < C ? gamma[c] : 0.f, c + 1 < C ? gamma[c + 1] : 0.f, c + 2 < C ? gamma[c + 2] : 0.f,
                        c + 3 < C ? gamma[c + 3] : 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invC = 1.0f / (float)C;
  for (int m = (blockIdx.x * 4 + wave) * 4 + sub; m < rows; m += gridDim.x * 16) {
    const long long t = gather ? win_row_to_token(geom, m) : m;
    const long long rdy = dy_by_m ? m : t, rst = stats_by_m ? m : t, rout = out_by_m ? m : t;
    const float mean = mean_in[rst], rstd = rstd_in[rst];
    float4 xh[NV], dyv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 64 * i + 4 * j;
      const float4 xv = *reinterpret_cast<const float4*>(x + t * CP + c);
      const uint2 u = *reinterpret_cast<const uint2*>(dyb + rdy * CP + c);
      float y0, y1, y2, y3;
      unpack_bf2(u.x, y0, y1);
      unpack_bf2(u.y, y2, y3);
      dyv[i] = make_float4(y0, y1, y2, y3);
      xh[i] = make_float4(c < C ? (xv.x - mean) * rstd : 0.f, c + 1 < C ? (xv.y - mean) * rstd : 0.f,
                          c + 2 < C ? (xv.z - mean) * rstd : 0.f, c + 3 < C ? (xv.w - mean) * rstd : 0.f);
      const float g0 = y0 * gm[i].x, g1 = y1 * gm[i].y, g2 = y2 * gm[i].z, g3 = y3 * gm[i].w;
      s1 += (g0 + g1) + (g2 + g3);
      s2 += (g0 * xh[i].x + g1 * xh[i].y) + (g2 * xh[i].z + g3 * xh[i].w);
      dg[i].x += y0 * xh[i].x; dg[i].y += y1 * xh[i].y; dg[i].z += y2 * xh[i].z; dg[i].w += y3 * xh[i].w;
      db[i].x += y0; db[i].y += y1; db[i].z += y2; db[i].w += y3;
    }
    s1 = wave_sum16(s1) * invC;
    s2 = wave_sum16(s2) * invC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 64 * i + 4 * j;
      float4 o;
      o.x = c < C ? rstd * (dyv[i].x * gm[i].x - s1 - xh[i].x * s2) : 0.f;
      o.y = c + 1 < C ? rstd * (dyv[i].y * gm[i].y - s1 - xh[i].y * s2) : 0.f;
      o.z = c + 2 < C ? rstd * (dyv[i].z * gm[i].z - s1 - xh[i].z * s2) : 0.f;
      o.w = c + 3 < C ? rstd * (dyv[i].w * gm[i].w - s1 - xh[i].w * s2) : 0.f;
      float* gp = gx + t * CP + c;
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(gp);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(gp) = o;
      if (gxb) {
        // the bf16 copy feeds the NEXT residual branch's backward; fold that branch's DropPath factor in
        const float f = rowscale ? rowscale[t / rows_per_sample] : 1.0f;
        *reinterpret_cast<uint2*>(gxb + rout * CP + c) = pack_bf4(o.x * f, o.y * f, o.z * f, o.w * f);
      }
    }
  }
  // reduce dgamma/dbeta: the wave's 4 sub-rows by lane exchange -> one LDS row per wave -> one global atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 64 * i + 4 * j;
    float v[8] = {dg[i].x, dg[i].y, dg[i].z, dg[i].w, db[i].x, db[i].y, db[i].z, db[i].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] += __shfl_xor(v[e], 16);
      v[e] += __shfl_xor(v[e], 32);
    }
    if (sub == 0) {
      *reinterpret_cast<float4*>(&red[wave][0][c]) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(&red[wave][1][c]) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]));
    atomicAdd(dbeta + c, (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]));
  }
}

}  // namespace

int srk_launch_ln_fwd(const float* x, const float* gamma, const float* beta, bf16_t* yb, float* yf, float* mean,
                      float* rstd, int rows, int C, int CP, const WinGeom* geom, hipStream_t stream) {
  SRK_REQUIRE(CP % 64 == 0 && CP <= 64 * MAXV && C <= CP && C > 0, SRK_E_SHAPE, "layernorm: bad C/CP %d/%d", C, CP);
  SRK_REQUIRE(x && gamma && beta && (yb || yf), SRK_E_NULL, "layernorm: null pointer");
  WinGeom g0 = {0, 0, 0, 0, 0};
  const WinGeom g = geom ? *geom : g0;
  const int gather = geom != nullptr;
  const int grid = cdiv(rows, 16) < 4096 ? cdiv(rows, 16) : 4096;
#define LN_CASE(NV)                                                                                              \
  case NV:                                                                                                       \
    hipLaunchKernelGGL(ln_fwd_kernel<NV>, dim3(grid), dim3(256), 0, stream, x, gamma, beta, yb, yf, mean, rstd, \
                       rows, C, g, gather, 1e-5f);                                                               \
    break;
  switch (CP / 64) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4)
  }
#undef LN_CASE
  return srk_check_launch("ln_fwd");
}

int srk_launch_ln_bwd(const bf16_t* dyb, const float* x, const float* mean, const float* rstd, const float* gamma,
                      float* gx, bf16_t* gxb, float* dgamma, float* dbeta, int rows, int C, int CP, const WinGeom* geom,
                      int dy_by_m, int stats_by_m, int out_by_m, int accumulate, const float* rowscale,
                      int rows_per_sample, hipStream_t stream) {
  SRK_REQUIRE(CP % 64 == 0 && CP <= 64 * MAXV && C <= CP && C > 0, SRK_E_SHAPE, "layernorm_bwd: bad C/CP %d/%d", C, CP);
  SRK_REQUIRE(dyb && x && mean && rstd && gamma && gx && dgamma && dbeta, SRK_E_NULL, "layernorm_bwd: null pointer");
  WinGeom g0 = {0, 0, 0, 0, 0};
  const WinGeom g = geom ? *geom : g0;
  const int gather = geom != nullptr;
  const int grid = cdiv(rows, 16) < 1024 ? cdiv(rows, 16) : 1024;
#define LN_CASE(NV)                                                                                                  \
  case NV:                                                                                                           \
    hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3(grid), dim3(256), 0, stream, dyb, x, mean, rstd, gamma, gx, gxb,     \
                       dgamma, dbeta, rows, C, g, gather, dy_by_m, stats_by_m, out_by_m, accumulate, rowscale,       \
                       rows_per_sample > 0 ? rows_per_sample : 1);                                                   \
    break;
  switch (CP / 64) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4)
  }
#undef LN_CASE
  return srk_check_launch("ln_bwd");
}
