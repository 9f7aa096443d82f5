// Row-major epilogue of gemm_kernel (all epilogues except the narrow image heads).
//
// The MFMA accumulators hold the output tile with 16 different rows per wave-instruction (64-byte
// pieces), which makes per-lane global accesses in the epilogue poorly coalesced.  Instead the
// 128 x BN fp32 tile goes through LDS (two 64-row halves; the K-loop staging buffers are free by
// then) and is re-read ROW-major: a 16-lane group owns one row, lane j holds columns 64c + 4j .. +3
// (c < BN/64) -- the same layout as the LayerNorm kernels -- so that every global access of the
// epilogue is a 256-byte (fp32) / 128-byte (bf16) contiguous run per 16 lanes, and row-wise
// reductions (fused LayerNorm backward) are 4 shuffle steps.
#pragma once
#include "gemm.h"

// LayerNorm (eps 1e-5) of a row held as NC float4 per lane by a 16-lane group -> bf16 row + stats
template <int NC>
__device__ __forceinline__ void fused_ln_row_at(const GemmParams& p, const float4 (&o)[NC], long long ro, int j16,
                                                const float (&lg)[NC][4], const float (&lb)[NC][4]);

template <int NC>
__device__ __forceinline__ void fused_ln_row(const GemmParams& p, const float4 (&o)[NC], long long t, int j16,
                                             const float (&lg)[NC][4], const float (&lb)[NC][4]) {
  fused_ln_row_at<NC>(p, o, p.xn_window ? token_to_win_row(p.xn_geom, (int)t) : t, j16, lg, lb);
}

// same, with the output row already resolved
template <int NC>
__device__ __forceinline__ void fused_ln_row_at(const GemmParams& p, const float4 (&o)[NC], long long ro, int j16,
                                                const float (&lg)[NC][4], const float (&lb)[NC][4]) {
  const float invC = 1.0f / (float)p.xn_C;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += (o[c].x + o[c].y) + (o[c].z + o[c].w);     // pad columns are zero
  const float mean = wave_sum16(s) * invC;
  float d[NC][4];
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float ov[4] = {o[c].x, o[c].y, o[c].z, o[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[c][e] = 64 * c + 4 * j16 + e < p.xn_C ? ov[e] - mean : 0.f;
      q += d[c][e] * d[c][e];
    }
  }
  const float rstd = rsqrtf(wave_sum16(q) * invC + 1e-5f);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = d[c][e] * rstd * lg[c][e] + lb[c][e];     // gamma/beta are zero in the pad columns
    }
    *reinterpret_cast<uint2*>(p.xn_out + ro * p.ldo + 64 * c + 4 * j16) = pack_bf4(y[0], y[1], y[2], y[3]);
  }
  if (j16 == 0) {
    p.xn_mean[ro] = mean;
    p.xn_rstd[ro] = rstd;
  }
}

template <int EP, int NTT>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams& p, f32x4_t (&acc)[4][NTT], unsigned char* smem, int m0,
                                                   int n0, int tid) {
  constexpr int BN = NTT * 32, NC = BN / 64, BNP = BN + 4;
  float* T = reinterpret_cast<float*>(smem);   // [64][BNP]
  float* colred = T + 64 * BNP;                 // [2][BN]  (EP_LNBWD)
  const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int sub = g, j16 = r16;                 // row-major phase: group -> row, lane-in-group -> column quad

  float4 bias[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    bias[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EP != EP_DGELU && EP != EP_DLRELU && EP != EP_LNBWD) {
      const int n = n0 + 64 * c + 4 * j16;
      if (p.bias && n < p.N) bias[c] = *reinterpret_cast<const float4*>(p.bias + n);
    }
  }
  // EP_LNBWD per-lane column state
  float gm[NC][4], cg[NC][4], cb[NC][4];
  if constexpr (EP == EP_LNBWD) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 64 * c + 4 * j16 + e;
        gm[c][e] = n < p.ln_C ? p.ln_gamma[n] : 0.f;
        cg[c][e] = 0.f;
        cb[c][e] = 0.f;
      }
  }
  const float invC = EP == EP_LNBWD ? 1.0f / (float)p.ln_C : 0.f;
  // fused forward LayerNorm (EP_PROJ_RES / EP_RES): per-lane gamma / beta of this lane's columns
  float lg[NC][4], lb[NC][4];
  if constexpr (EP == EP_PROJ_RES || EP == EP_RES) {
    if (p.xn_out) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 64 * c + 4 * j16 + e;
          lg[c][e] = n < p.xn_C ? p.xn_gamma[n] : 0.f;
          lb[c][e] = n < p.xn_C ? p.xn_beta[n] : 0.f;
        }
    }
  }

#ifndef SRK_ROWEP_EARLY_RES
#define SRK_ROWEP_EARLY_RES 1
#endif
  // EP_RES (the 3x3 convs of the RSTBs): the residual rows of a half are requested BEFORE its accumulators go through LDS, so that
  // their round trip runs under the tile write, the barrier and the LayerNorm of the rows ahead instead of in a per-row
  // load -> use -> store chain
  constexpr bool EARLY = SRK_ROWEP_EARLY_RES != 0 && (EP == EP_RES || EP == EP_RES_BF16);
  float4 er[EARLY ? 4 : 1][NC];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if constexpr (EARLY) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int lr = wave * 16 + u * 4 + sub;
        const int m = m0 + (lr >> 5) * 64 + half * 32 + (lr & 31);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          er[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < p.M && n < p.N) er[u][c] = *reinterpret_cast<const float4*>(p.res + (long long)m * p.ldo + n);
        }
      }
    }
    if (half) srk_lds_barrier();      // LDS hazard only: a __syncthreads() would also wait for the first half's global stores
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NTT; ++j) {
        const f32x4_t a = acc[2 * half + ii][j];
        *reinterpret_cast<float4*>(T + (wm * 32 + ii * 16 + r16) * BNP + wn * (NTT * 16) + j * 16 + 4 * g) =
            make_float4(a[0], a[1], a[2], a[3]);
      }
    if constexpr (EP == EP_LNBWD) {
      if (half == 0)
        for (int i = tid; i < 2 * BN; i += 256) colred[i] = 0.f;
    }
    srk_lds_barrier();
    // Global operands of the row epilogue are fetched for a batch of rows BEFORE any of them is consumed: the
    // compiler cannot hoist these loads above the previous row's stores (possible aliasing), and at 2 workgroups
    // per CU a dependent load-use-store chain per row would serialise ~8 memory round trips per tile.
    constexpr int PFB = (EP == EP_LNBWD || EP == EP_PROJ_RES || EP == EP_RES) ? 1 : 4;   // measured: batching only pays for the bf16-aux epilogues
#pragma unroll
    for (int it0 = 0; it0 < 4; it0 += PFB) {
      float4 pf_a[PFB][NC], pf_b[PFB][NC];
      uint2 pf_u[PFB][NC];
      float pf_mean[PFB], pf_rstd[PFB];
#pragma unroll
      for (int u = 0; u < PFB; ++u) {
        const int lr = wave * 16 + (it0 + u) * 4 + sub;
        const int m = m0 + (lr >> 5) * 64 + half * 32 + (lr & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          if constexpr (EP == EP_PROJ_RES) {
            pf_a[u][c] = *reinterpret_cast<const float4*>(p.res + (long long)win_row_to_token(p.geom, m) * p.ldo + n);
          } else if constexpr (EP == EP_RES || EP == EP_RES_BF16) {
            if constexpr (!EARLY) pf_a[u][c] = *reinterpret_cast<const float4*>(p.res + (long long)m * p.ldo + n);
          } else if constexpr (EP == EP_DGELU || EP == EP_DLRELU) {
            pf_u[u][c] = *reinterpret_cast<const uint2*>(p.aux + (long long)m * p.ldo + n);
          } else if constexpr (EP == EP_LNBWD) {
            const long long t = p.ln_rows_window ? win_row_to_token(p.geom, m) : m;
            pf_a[u][c] = *reinterpret_cast<const float4*>(p.ln_x + t * p.ldo + n);
            pf_b[u][c] = *reinterpret_cast<const float4*>(p.outf + t * p.ldo + n);
          }
        }
        if constexpr (EP == EP_LNBWD) {
          const long long t = p.ln_rows_window ? win_row_to_token(p.geom, m) : m;
          const long long st = p.ln_stats_by_m ? m : t;
          pf_mean[u] = p.ln_mean[st];
          pf_rstd[u] = p.ln_rstd[st];
        }
      }
#pragma unroll
    for (int u = 0; u < PFB; ++u) {
      const int it = it0 + u;
      const int lr = wave * 16 + it * 4 + sub;
      const int m = m0 + (lr >> 5) * 64 + half * 32 + (lr & 31);
      if (m >= p.M) continue;
      float4 v[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        v[c] = *reinterpret_cast<const float4*>(T + lr * BNP + 64 * c + 4 * j16);
        v[c].x += bias[c].x; v[c].y += bias[c].y; v[c].z += bias[c].z; v[c].w += bias[c].w;
      }
      // ---------------------------------------------------------------------------------------
      if constexpr (EP == EP_BF16) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n < p.N) *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v[c].x, v[c].y, v[c].z, v[c].w);
        }
      } else if constexpr (EP == EP_F32_BF16) {
        const float f = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          *reinterpret_cast<float4*>(p.outf + (long long)m * p.ldo + n) = v[c];
          if (p.outb) *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v[c].x * f, v[c].y * f, v[c].z * f, v[c].w * f);
        }
      } else if constexpr (EP == EP_QKV) {
        const long long b_ = m >> 6;
        const int tok = m & 63;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          const int which = n / p.CA, rem = n - which * p.CA;
          const int h = rem >> 5, d = rem & 31;
          const float s = which == 0 ? p.scale : 1.0f;
          bf16_t* dst = p.outb + ((((long long)which * p.B_ + b_) * p.nH + h) * 64 + tok) * 32 + d;
          *reinterpret_cast<uint2*>(dst) = pack_bf4(v[c].x * s, v[c].y * s, v[c].z * s, v[c].w * s);
        }
      } else if constexpr (EP == EP_PROJ_RES) {
        const long long t = win_row_to_token(p.geom, m);
        const float f = p.rowscale ? p.rowscale[t / p.rows_per_sample] : 1.0f;
        float4 o[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n >= p.N) continue;
          const float4 rv = pf_a[u][c];
          o[c] = make_float4(rv.x + v[c].x * f, rv.y + v[c].y * f, rv.z + v[c].z * f, rv.w + v[c].w * f);
          *reinterpret_cast<float4*>(p.outf + t * p.ldo + n) = o[c];
        }
        if (p.xn_out) fused_ln_row<NC>(p, o, t, j16, lg, lb);
      } else if constexpr (EP == EP_GELU) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          if (p.outb) *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v[c].x, v[c].y, v[c].z, v[c].w);   // u: training only
          *reinterpret_cast<uint2*>(p.outb2 + (long long)m * p.ldo + n) =
              gelu_pack4(v[c].x, v[c].y, v[c].z, v[c].w);
        }
      } else if constexpr (EP == EP_RES || EP == EP_RES_BF16) {
        const float f = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.0f;
        float4 o[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n >= p.N) continue;
          float4 rv;
          if constexpr (EARLY) rv = er[it][c]; else rv = pf_a[u][c];
          o[c] = make_float4(rv.x + v[c].x * f, rv.y + v[c].y * f, rv.z + v[c].z * f, rv.w + v[c].w * f);
          if constexpr (EP == EP_RES) *reinterpret_cast<float4*>(p.outf + (long long)m * p.ldo + n) = o[c];
          if (EP == EP_RES_BF16 || p.outb)
            *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(o[c].x, o[c].y, o[c].z, o[c].w);
        }
        if constexpr (EP == EP_RES) {
          if (p.xn_out) fused_ln_row<NC>(p, o, m, j16, lg, lb);
        }
      } else if constexpr (EP == EP_DGELU || EP == EP_DLRELU) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          const uint2 ua = pf_u[u][c];
          float u0, u1, u2, u3;
          unpack_bf2(ua.x, u0, u1);
          unpack_bf2(ua.y, u2, u3);
          uint2 o;
          if constexpr (EP == EP_DGELU) {
            o = dgelu_mul_pack4(v[c].x, v[c].y, v[c].z, v[c].w, u0, u1, u2, u3);
          } else {
            const float s = p.scale;
            o = pack_bf4(u0 > 0.f ? v[c].x : v[c].x * s, u1 > 0.f ? v[c].y : v[c].y * s, u2 > 0.f ? v[c].z : v[c].z * s,
                         u3 > 0.f ? v[c].w : v[c].w * s);
          }
          *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = o;
        }
      } else if constexpr (EP == EP_LRELU) {
        const float s = p.scale;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) =
              pack_bf4(v[c].x > 0.f ? v[c].x : v[c].x * s, v[c].y > 0.f ? v[c].y : v[c].y * s, v[c].z > 0.f ? v[c].z : v[c].z * s,
                       v[c].w > 0.f ? v[c].w : v[c].w * s);
        }
      } else if constexpr (EP == EP_PS) {
        const int hw = p.H * p.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int n = n0 + 64 * c + 4 * j16;
          if (n >= p.N) continue;
          const int ij = n / p.Cs, cc = n - ij * p.Cs;
          const int si = ij / p.r, sj = ij - si * p.r;
          bf16_t* dst = p.outb + (((long long)(b * p.H * p.r + y * p.r + si)) * (p.W * p.r) + x * p.r + sj) * p.Cs + cc;
          *reinterpret_cast<uint2*>(dst) = pack_bf4(v[c].x, v[c].y, v[c].z, v[c].w);
        }
      } else if constexpr (EP == EP_LNBWD) {
        // v = dL/d(LN output) of row m (all BN columns live in this 16-lane group)
        const long long t = p.ln_rows_window ? win_row_to_token(p.geom, m) : m;
        const float mean = pf_mean[u], rstd = pf_rstd[u];
        float xh[NC][4], dy[NC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 xv = pf_a[u][c];
          const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
          const float dv[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dy[c][e] = dv[e];
            xh[c][e] = 64 * c + 4 * j16 + e < p.ln_C ? (xs[e] - mean) * rstd : 0.f;
            const float dg = dv[e] * gm[c][e];
            s1 += dg;
            s2 += dg * xh[c][e];
          }
        }
        s1 = wave_sum16(s1) * invC;
        s2 = wave_sum16(s2) * invC;
        const float f = p.rowscale ? p.rowscale[t / p.rows_per_sample] : 1.0f;
        const long long ro = p.ln_out_window ? token_to_win_row(p.geom, (int)t) : t;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          float* gp = (p.ln_skip ? p.ln_skip : p.outf) + t * p.ldo + 64 * c + 4 * j16;
          const float4 old = pf_b[u][c];
          float o[4] = {old.x, old.y, old.z, old.w};
          if (p.ln_skip) {
            const float4 sk = *reinterpret_cast<const float4*>(gp);
            o[0] += sk.x; o[1] += sk.y; o[2] += sk.z; o[3] += sk.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (64 * c + 4 * j16 + e < p.ln_C) {
              o[e] += rstd * (dy[c][e] * gm[c][e] - s1 - xh[c][e] * s2);
              cg[c][e] += dy[c][e] * xh[c][e];
              cb[c][e] += dy[c][e];
            }
          }
          *reinterpret_cast<float4*>(gp) = make_float4(o[0], o[1], o[2], o[3]);
          if (p.outb)
            *reinterpret_cast<uint2*>(p.outb + ro * p.ldo + 64 * c + 4 * j16) = pack_bf4(o[0] * f, o[1] * f, o[2] * f, o[3] * f);
        }
      }
    }
    }
  }
  if constexpr (EP == EP_LNBWD) {
    // dgamma / dbeta: lanes of the 4 groups x 4 waves hold partials of the same columns
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&colred[64 * c + 4 * j16 + e], cg[c][e]);
        atomicAdd(&colred[BN + 64 * c + 4 * j16 + e], cb[c][e]);
      }
    __syncthreads();
    for (int n = tid; n < p.ln_C; n += 256) {
      atomicAdd(p.ln_dgamma + n, colred[n]);
      atomicAdd(p.ln_dbeta + n, colred[BN + n]);
    }
  }
}
