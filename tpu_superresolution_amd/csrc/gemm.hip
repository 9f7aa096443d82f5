// bf16 MFMA GEMM / implicit-GEMM 3x3 conv for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//   D[m][n] = sum_k A[m][k] * Wt[n][k]   (+ epilogue)
//
// One 256-thread workgroup (4 waves as 2(M) x 2(N)) owns a 128 x BN output tile, BN = 32*NT
// (NT = n-tiles of 16 per wave: 6/4/2 -> BN 192/128/64); a narrow variant (4 waves x 32 rows x 16
// cols) serves N == 16 (3-channel image heads).  K is consumed in 64-wide chunks staged through
// LDS with a register-staged double buffer (global->VGPR issued before the MFMA block of the
// current chunk, VGPR->LDS after it, one barrier per chunk).  LDS tiles are [rows][64] bf16 with
// the 16-byte chunk index XOR-swizzled by (row & 7), which makes every ds_read_b128 fragment read
// conflict-free.  The MFMA is issued as D^T = Wt . A^T so that each lane ends up with four
// consecutive n for one m: epilogues then store 8 B (bf16) / 16 B (fp32) per lane.
#include "gemm.h"
#include "gemm_rowep.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

#ifndef SRK_GEMM_DMA
#define SRK_GEMM_DMA 1
#endif
__device__ uint4 g_gemm_zero[1];     // 16 zero bytes: DMA source for halo pixels / rows beyond M or N

template <int LD>
struct RowCtx {  // per-thread staging context for the 4 A passes
  long long base[4];  // element offset of (row, k=0) or conv pixel base
  int yx[4];          // conv: (y << 16) | x ; rows: valid flag
};

template <int LD, int EP, int NT, bool NARROW>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BN = NARROW ? 16 : NT * 32;
  constexpr int WPASS = NARROW ? 1 : BN / 32;  // W staging passes (32 rows per pass)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);           // [2][BM][64]
  bf16_t* Ws = As + 2 * BM * BK;                          // [2][BN or 32][64]
  constexpr int WS_ROWS = NARROW ? 32 : BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = NARROW ? wave : (wave >> 1);
  const int wn = NARROW ? 0 : (wave & 1);
  // 1-D grid; logical tile order is n-chunk fastest and contiguous per XCD, so the N-chunks of one M-tile (which
  // re-read the same A rows) run back to back on one XCD and share its L2
  const int ntn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;
  const int srow = tid >> 3, schunk = tid & 7;  // staging: 32 rows x 8 chunks per pass

  // ---- per-thread A row contexts ------------------------------------------------------------
  RowCtx<LD> ctx;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int m = m0 + srow + 32 * ps;
    if constexpr (LD == LD_ROWS) {
      ctx.base[ps] = (long long)m * p.lda;
      ctx.yx[ps] = m < p.M;
    } else {
      const int hw = p.H * p.W;
      const int mm = m < p.M ? m : 0;
      const int b = mm / hw, rem = mm - b * hw;
      const int y = rem / p.W, x = rem - y * p.W;
      ctx.yx[ps] = m < p.M ? ((y << 16) | x) : (0x7fff << 16);  // invalid rows: y out of range
      if constexpr (LD == LD_CONV3) {
        ctx.base[ps] = ((long long)(b * p.H + y) * p.W + x) * p.CinP;
      } else {
        ctx.base[ps] = ((long long)(b * p.H * p.r + y * p.r) * (p.W * p.r) + x * p.r) * p.Cs;
      }
    }
  }

  uint4 ra[4];
  uint4 rw[WPASS];

  // Pixel-shuffled source (the 256 x 256 x 64 gradient of the upsampler): the tiles of one XCD sweep the taps in lock step, and a
  // source row pair comes back as tap row dy = +1 of tile y - 1, dy = 0 of tile y and dy = -1 of tile y + 1 -- a third of the K loop
  // apart, by when ~4 MB of other rows have passed through the XCD's 4-MB L2 (measured: 1.94 GB fetched for a 268-MB tensor).
  // Rotating the tap-row order by the tile's image row makes the three users of a row pair read it at the same time.
  int krot = 0;
  if constexpr (LD == LD_CONV3_PS) {
    const int rows_per_tile = BM > p.W ? BM / p.W : 1;
    const int ty = (m0 / p.W) / rows_per_tile;
    krot = ((3 - ty % 3) % 3) * (p.K / BK / 3);
  }
  const int nk_all = p.K / BK;

  auto load_stage = [&](int kc) {
    kc = kc + krot < nk_all ? kc + krot : kc + krot - nk_all;
    const int k0 = kc * BK;
    if constexpr (LD == LD_ROWS) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        ra[ps] = ctx.yx[ps] ? *reinterpret_cast<const uint4*>(p.A + ctx.base[ps] + k0 + schunk * 8)
                            : make_uint4(0, 0, 0, 0);
      }
    } else {
      const int tap = k0 / p.CinP;
      const int ci0 = k0 - tap * p.CinP;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      long long koff;
      if constexpr (LD == LD_CONV3) {
        koff = (long long)(dy * p.W + dx) * p.CinP + ci0 + schunk * 8;
      } else {
        // logical channel block ci0..ci0+63 == sub-pixel ij = ci0 / Cs (Cs is a multiple of 64)
        const int ij = ci0 / p.Cs, cc = ci0 - ij * p.Cs;
        const int si = ij / p.r, sj = ij - si * p.r;
        koff = ((long long)(dy * p.r + si) * (p.W * p.r) + (dx * p.r + sj)) * p.Cs + cc + schunk * 8;
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int y = (ctx.yx[ps] >> 16) + dy, x = (ctx.yx[ps] & 0xffff) + dx;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        ra[ps] = ok ? *reinterpret_cast<const uint4*>(p.A + ctx.base[ps] + koff) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
      const int n = n0 + srow + 32 * ps;
      rw[ps] = (n < p.N) ? *reinterpret_cast<const uint4*>(p.Wt + (long long)n * p.K + k0 + schunk * 8)
                         : make_uint4(0, 0, 0, 0);
    }
  };

  auto store_stage = [&](int buf) {
    bf16_t* a = As + buf * BM * BK;
    bf16_t* w = Ws + buf * WS_ROWS * BK;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      *reinterpret_cast<uint4*>(a + swz_off(srow + 32 * ps, schunk)) = ra[ps];
    }
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
      *reinterpret_cast<uint4*>(w + swz_off(srow + 32 * ps, schunk)) = rw[ps];
    }
  };

  // LDS-DMA form of load_stage + store_stage: the 16-byte piece a thread used to carry through a VGPR goes straight to its
  // swizzled LDS position.  A DMA instruction writes lane-linear (lane l -> row l / 8, physical chunk l % 8 of the wave's
  // eight rows), so the lane fetches the LOGICAL chunk that swz_off() keeps there: schunk ^ (row & 7).  Issued at the top
  // of step kc for step kc + 1 (the other buffer was last read in step kc - 1, behind that step's barrier); no register
  // staging, no ds_write pass, and the only wait is vmcnt(0) in front of the step's closing barrier.
  const unsigned smem_base = (unsigned)(size_t)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int csrc = schunk ^ (srow & 7);
  const bf16_t* zero16 = reinterpret_cast<const bf16_t*>(g_gemm_zero);
  auto issue_stage = [&](int kc, int buf) {
    kc = kc + krot < nk_all ? kc + krot : kc + krot - nk_all;
    const int k0 = kc * BK;
    const unsigned abase = smem_base + (unsigned)(buf * BM * BK * 2);
    const unsigned wbase = smem_base + (unsigned)((2 * BM * BK + buf * WS_ROWS * BK) * 2);
    if constexpr (LD == LD_ROWS) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const bf16_t* src = ctx.yx[ps] ? p.A + ctx.base[ps] + k0 + csrc * 8 : zero16;
        srk_glds16(src, __builtin_amdgcn_readfirstlane(abase + (unsigned)((wave_u * 8 + 32 * ps) * BK * 2)));
      }
    } else {
      const int tap = k0 / p.CinP;
      const int ci0 = k0 - tap * p.CinP;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      long long koff;
      if constexpr (LD == LD_CONV3) {
        koff = (long long)(dy * p.W + dx) * p.CinP + ci0 + csrc * 8;
      } else {
        const int ij = ci0 / p.Cs, cc = ci0 - ij * p.Cs;
        const int si = ij / p.r, sj = ij - si * p.r;
        koff = ((long long)(dy * p.r + si) * (p.W * p.r) + (dx * p.r + sj)) * p.Cs + cc + csrc * 8;
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int y = (ctx.yx[ps] >> 16) + dy, x = (ctx.yx[ps] & 0xffff) + dx;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const bf16_t* src = ok ? p.A + ctx.base[ps] + koff : zero16;
        srk_glds16(src, __builtin_amdgcn_readfirstlane(abase + (unsigned)((wave_u * 8 + 32 * ps) * BK * 2)));
      }
    }
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
      const int n = n0 + srow + 32 * ps;
      const bf16_t* src = (n < p.N) ? p.Wt + (long long)n * p.K + k0 + csrc * 8 : zero16;
      srk_glds16(src, __builtin_amdgcn_readfirstlane(wbase + (unsigned)((wave_u * 8 + 32 * ps) * BK * 2)));
    }
  };

  constexpr int MT = NARROW ? 2 : 4;   // m-tiles (16 rows) per wave
  constexpr int NTT = NARROW ? 1 : NT; // n-tiles per wave
  f32x4_t acc[MT][NTT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
#if SRK_GEMM_DMA
  issue_stage(0, 0);
  srk_wait_vmcnt<0>();
  srk_lds_barrier();
#else
  load_stage(0);
  store_stage(0);
  __syncthreads();
#endif

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
#if SRK_GEMM_DMA
    if (kc + 1 < nk) issue_stage(kc + 1, buf ^ 1);
#else
    if (kc + 1 < nk) load_stage(kc + 1);
#endif
    const bf16_t* a = As + buf * BM * BK;
    const bf16_t* w = Ws + buf * WS_ROWS * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t xf[MT], wf[NTT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * (MT * 16) + i * 16 + r16;
        xf[i] = *reinterpret_cast<const bf16x8_t*>(a + swz_off(row, ks * 4 + g));
      }
#pragma unroll
      for (int j = 0; j < NTT; ++j) {
        const int row = wn * (NTT * 16) + j * 16 + r16;
        wf[j] = *reinterpret_cast<const bf16x8_t*>(w + swz_off(row, ks * 4 + g));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
#if SRK_GEMM_DMA
    srk_wait_vmcnt<0>();                 // this wave's pieces of the next stage are in LDS ...
    srk_lds_barrier();                   // ... everyone's are; and every wave is done reading this stage
#else
    if (kc + 1 < nk) store_stage(buf ^ 1);
    __syncthreads();
#endif
  }

  // ---- row-major epilogue through LDS (everything except the narrow image heads) ----------------
  if constexpr (!NARROW) {
    gemm_epilogue_rows<EP, NTT>(p, acc, smem, m0, n0, tid);
    return;
  }

  // ---- epilogue: lane holds D[n = nb + 4g + e][m = mb + r16], e = 0..3 ----------------------------
#pragma unroll
  for (int j = 0; j < NTT; ++j) {
    const int n = n0 + wn * (NTT * 16) + j * 16 + 4 * g;
    if (n >= p.N) continue;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EP != EP_DGELU && EP != EP_DLRELU) {
      if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * (MT * 16) + i * 16 + r16;
      if (m >= p.M) continue;
      float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
      if constexpr (EP == EP_BF16) {
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v0, v1, v2, v3);
      } else if constexpr (EP == EP_F32_BF16) {
        *reinterpret_cast<float4*>(p.outf + (long long)m * p.ldo + n) = make_float4(v0, v1, v2, v3);
        const float f = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.0f;
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v0 * f, v1 * f, v2 * f, v3 * f);
      } else if constexpr (EP == EP_QKV) {
        const int which = n / p.CA, rem = n - which * p.CA;
        const int h = rem >> 5, d = rem & 31;
        const float s = which == 0 ? p.scale : 1.0f;
        const long long b_ = m >> 6;
        const int tok = m & 63;
        bf16_t* dst = p.outb + ((((long long)which * p.B_ + b_) * p.nH + h) * 64 + tok) * 32 + d;
        *reinterpret_cast<uint2*>(dst) = pack_bf4(v0 * s, v1 * s, v2 * s, v3 * s);
      } else if constexpr (EP == EP_PROJ_RES) {
        const long long t = win_row_to_token(p.geom, m);
        if (p.rowscale) {
          const float f = p.rowscale[t / p.rows_per_sample];
          v0 *= f; v1 *= f; v2 *= f; v3 *= f;
        }
        const float4 rv = *reinterpret_cast<const float4*>(p.res + t * p.ldo + n);
        *reinterpret_cast<float4*>(p.outf + t * p.ldo + n) = make_float4(rv.x + v0, rv.y + v1, rv.z + v2, rv.w + v3);
      } else if constexpr (EP == EP_GELU) {
        if (p.outb) *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(v0, v1, v2, v3);
        *reinterpret_cast<uint2*>(p.outb2 + (long long)m * p.ldo + n) =
            gelu_pack4(v0, v1, v2, v3);
      } else if constexpr (EP == EP_RES) {
        if (p.rowscale) {
          const float f = p.rowscale[m / p.rows_per_sample];
          v0 *= f; v1 *= f; v2 *= f; v3 *= f;
        }
        const float4 rv = *reinterpret_cast<const float4*>(p.res + (long long)m * p.ldo + n);
        const float4 o = make_float4(rv.x + v0, rv.y + v1, rv.z + v2, rv.w + v3);
        *reinterpret_cast<float4*>(p.outf + (long long)m * p.ldo + n) = o;
        if (p.outb) *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(o.x, o.y, o.z, o.w);
      } else if constexpr (EP == EP_RES_BF16) {
        const float4 rv = *reinterpret_cast<const float4*>(p.res + (long long)m * p.ldo + n);
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) = pack_bf4(rv.x + v0, rv.y + v1, rv.z + v2, rv.w + v3);
      } else if constexpr (EP == EP_DGELU) {
        const uint2 u = *reinterpret_cast<const uint2*>(p.aux + (long long)m * p.ldo + n);
        float u0, u1, u2, u3;
        unpack_bf2(u.x, u0, u1);
        unpack_bf2(u.y, u2, u3);
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) =
            dgelu_mul_pack4(v0, v1, v2, v3, u0, u1, u2, u3);
      } else if constexpr (EP == EP_LRELU) {
        const float s = p.scale;
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) =
            pack_bf4(v0 > 0.f ? v0 : v0 * s, v1 > 0.f ? v1 : v1 * s, v2 > 0.f ? v2 : v2 * s, v3 > 0.f ? v3 : v3 * s);
      } else if constexpr (EP == EP_DLRELU) {
        const uint2 u = *reinterpret_cast<const uint2*>(p.aux + (long long)m * p.ldo + n);
        float u0, u1, u2, u3;
        unpack_bf2(u.x, u0, u1);
        unpack_bf2(u.y, u2, u3);
        const float s = p.scale;
        *reinterpret_cast<uint2*>(p.outb + (long long)m * p.ldo + n) =
            pack_bf4(u0 > 0.f ? v0 : v0 * s, u1 > 0.f ? v1 : v1 * s, u2 > 0.f ? v2 : v2 * s, u3 > 0.f ? v3 : v3 * s);
      } else if constexpr (EP == EP_PS) {
        const int hw = p.H * p.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.W, x = rem - y * p.W;
        const int ij = n / p.Cs, c = n - ij * p.Cs;
        const int si = ij / p.r, sj = ij - si * p.r;
        bf16_t* dst = p.outb + (((long long)(b * p.H * p.r + y * p.r + si)) * (p.W * p.r) + x * p.r + sj) * p.Cs + c;
        *reinterpret_cast<uint2*>(dst) = pack_bf4(v0, v1, v2, v3);
      } else if constexpr (EP == EP_IMG || EP == EP_PS_IMG) {
        const int hw = p.H * p.W;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.W, x = rem - y * p.W;
        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nn = n + e;
          if constexpr (EP == EP_IMG) {
            if (nn < p.Cimg && y < p.Hc && x < p.Wc)
              p.outf[(((long long)b * p.Cimg + nn) * p.Hc + y) * p.Wc + x] = vv[e] * p.inv_range + p.mean[nn];
          } else {
            const int rr = p.r * p.r;
            const int c = nn / rr, ij = nn - c * rr;
            const int oy = y * p.r + ij / p.r, ox = x * p.r + ij % p.r;
            // denoising head (network_swinir.py:836-838): x + conv_last(res), x = the normalised input image [M][4] (r == 1)
            const float add = (p.res != nullptr && c < 4) ? p.res[(long long)m * 4 + c] : 0.f;
            if (c < p.Cimg && oy < p.Hc && ox < p.Wc)
              p.outf[(((long long)b * p.Cimg + c) * p.Hc + oy) * p.Wc + ox] = (vv[e] + add) * p.inv_range + p.mean[c];
          }
        }
      }
    }
  }
}

template <int LD, int EP, int NT, bool NARROW>
int launch(const GemmParams& p, hipStream_t stream) {
  constexpr int BN = NARROW ? 16 : NT * 32;
  constexpr int WS_ROWS = NARROW ? 32 : BN;
  constexpr size_t lds = (size_t)(2 * BM * BK + 2 * WS_ROWS * BK) * sizeof(bf16_t);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<LD, EP, NT, NARROW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      srk_set_error("gemm: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN));
  const int fam = LD == LD_ROWS ? FAM_GEMM_LINEAR : FAM_GEMM_CONV;
  srk_probe_pre(fam, stream, p.flops, p.bytes);
  hipLaunchKernelGGL((gemm_kernel<LD, EP, NT, NARROW>), grid, dim3(256), lds, stream, p);
  srk_probe_post(fam, stream);
  return srk_check_launch("gemm");
}

template <int LD, int EP>
int dispatch_nt(const GemmParams& p, hipStream_t stream) {
  if (p.N % 192 == 0) return launch<LD, EP, 6, false>(p, stream);
  if (p.N % 128 == 0) return launch<LD, EP, 4, false>(p, stream);
  if (p.N % 64 == 0) return launch<LD, EP, 2, false>(p, stream);
  srk_set_error("gemm: N=%d is not a multiple of 64", p.N);
  return SRK_E_SHAPE;
}

}  // namespace

int srk_launch_gemm(int loader, int epilogue, const GemmParams& p, hipStream_t stream) {
  SRK_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.K % BK == 0, SRK_E_SHAPE, "gemm: bad M/N/K %d/%d/%d", p.M, p.N, p.K);
  SRK_REQUIRE(p.A && p.Wt, SRK_E_NULL, "gemm: null operand");
  if (loader != LD_ROWS) {
    SRK_REQUIRE(p.CinP % 64 == 0 && p.K == 9 * p.CinP && p.M == p.B * p.H * p.W, SRK_E_SHAPE,
                "conv: inconsistent geometry K=%d CinP=%d M=%d B*H*W=%d", p.K, p.CinP, p.M, p.B * p.H * p.W);
    SRK_REQUIRE(p.H < 32768 && p.W < 32768, SRK_E_SHAPE, "conv: image too large");
    if (loader == LD_CONV3_PS) SRK_REQUIRE(p.Cs % 64 == 0 && p.CinP == p.r * p.r * p.Cs, SRK_E_SHAPE, "conv(ps-in): bad Cs");
  } else {
    SRK_REQUIRE(p.lda >= p.K && p.lda % 8 == 0, SRK_E_SHAPE, "gemm: bad lda %d", p.lda);
  }
  if (p.xn_out) {
    SRK_REQUIRE((epilogue == EP_PROJ_RES || epilogue == EP_RES) && (p.N == 64 || p.N == 128 || p.N == 192), SRK_E_SHAPE,
                "gemm: fused LayerNorm output needs a residual epilogue and N in {64,128,192} (N=%d)", p.N);
    SRK_REQUIRE(p.xn_mean && p.xn_rstd && p.xn_gamma && p.xn_beta, SRK_E_NULL, "gemm: fused LayerNorm: null pointer");
  }
  if (epilogue == EP_LNBWD) {
    SRK_REQUIRE(loader == LD_ROWS && (p.N == 64 || p.N == 128 || p.N == 192), SRK_E_SHAPE, "gemm(ln-bwd epilogue): N=%d must be one tile (64/128/192)", p.N);
    SRK_REQUIRE(p.ln_x && p.ln_mean && p.ln_rstd && p.ln_gamma && p.ln_dgamma && p.ln_dbeta && p.outf, SRK_E_NULL,
                "gemm(ln-bwd epilogue): null pointer");
  }
  if (loader == LD_ROWS) {
    const int rc = srk_launch_gemm_stream(epilogue, p, stream);
    if (rc != SRK_NOT_COVERED) return rc;
  }
#define CASE(LD, EP) \
  if (loader == LD && epilogue == EP) return dispatch_nt<LD, EP>(p, stream);
  CASE(LD_ROWS, EP_BF16)
  CASE(LD_ROWS, EP_QKV)
  CASE(LD_ROWS, EP_PROJ_RES)
  CASE(LD_ROWS, EP_GELU)
  CASE(LD_ROWS, EP_RES)
  CASE(LD_ROWS, EP_DGELU)
  CASE(LD_ROWS, EP_LNBWD)
  CASE(LD_ROWS, EP_LRELU)       // 1x1 conv of the '3conv' residual connection (network_swinir.py:466-471)
  CASE(LD_ROWS, EP_DLRELU)
  CASE(LD_CONV3, EP_RES)
  CASE(LD_CONV3, EP_RES_BF16)
  CASE(LD_CONV3, EP_LRELU)
  CASE(LD_CONV3, EP_PS)
  CASE(LD_CONV3, EP_BF16)
  CASE(LD_CONV3, EP_DLRELU)
  CASE(LD_CONV3, EP_F32_BF16)
  CASE(LD_CONV3, EP_GELU)        // first conv of HAT's CAB (hat_arch.py:66-68)
  CASE(LD_CONV3, EP_DGELU)       // ... and the gradient through it (dgrad of the second conv times gelu'(u))
  CASE(LD_ROWS, EP_RES_BF16)
  CASE(LD_CONV3_PS, EP_BF16)
  CASE(LD_CONV3_PS, EP_DLRELU)
#undef CASE
  if (loader == LD_CONV3 && epilogue == EP_IMG && p.N == 16) return launch<LD_CONV3, EP_IMG, 1, true>(p, stream);
  if (loader == LD_CONV3 && epilogue == EP_PS_IMG && p.N == 16) return launch<LD_CONV3, EP_PS_IMG, 1, true>(p, stream);
  srk_set_error("gemm: unsupported loader/epilogue combination %d/%d (N=%d)", loader, epilogue, p.N);
  return SRK_E_UNSUPPORTED;
}
