// Window attention with 256 queries per window, forward, for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 softmax):
//   * HAT window self-attention, 16 x 16 windows, optional cyclic shift with the arithmetic shift mask
//     (reference hat_arch.py:129-197 WindowAttention, :281-325 HAB.forward);
//   * HAT overlapping cross-attention: 16 x 16 query windows against 24 x 24 key/value windows cut with stride 16 and zero
//     padding 4 (nn.Unfold in hat_arch.py:378, :403-439 OCAB.forward) -- padded keys are ZERO vectors that still take part in
//     the softmax with score = bias, exactly as the reference's unfold produces them.
//
// q, k, v are read straight from the qkv projection's output in raster token order ([T][3*CA] bf16, head h at columns
// which*CA + 32 h .. +31, head_dim padded to 32 with zeros): roll + window_partition (+ unfold) are folded into the row
// addresses of the loads, window_reverse + the inverse roll into the rows of the store.  The softmax scale multiplies the
// fp32 scores (q is stored unscaled).
//
// One 256-thread workgroup per (window, head).  K and V of the window are staged once in LDS (row pitch 40 elements: the
// 16-byte fragment reads of 16 consecutive rows then hit 64 distinct banks).  Each wave owns 64 queries = four 16-query tiles:
//   S^T = K Q^T   one MFMA per 16-key tile (head_dim 32 = one K step); a lane then holds, for ITS query (lane & 15), the keys
//                 16 j + 4 (lane >> 4) + 0..3 of every tile j: a softmax row lives in NT*4 registers x 4 lanes;
//   softmax       bias (dense fp32 [nH][256][NK], float4 per lane) + mask, row max / sum by two permlane swaps;
//   O^T = V^T P^T P goes from the accumulators straight into the B operand (keys of tiles 2jj and 2jj+1 interleaved as the
//                 MFMA's K slots), V^T comes from LDS with the transposing read (ds_read_b64_tr_b16), same key order.
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KP = 40;      // LDS row pitch (elements) of the K / V tiles

struct Win256Params {
  const bf16_t* qkv;   // [T][ldq]
  bf16_t* out;         // [T][ldo]
  const float* bias;   // dense: [nH][256][NK];  table: the relative_position_bias_table parameter [table_rows][nH]
  int table_rows;      // > 0: bias comes from the table through the closed-form relative position index (no dense tensor)
  int ldq, ldo, CA;
  int B, H, W;         // feature map (token addressing)
  int Hp, Wp;          // window frame: H, W zero-padded at the bottom / right to multiples of the window (DAT: dat_arch.py:376-384
                       // pads q, k, v to a multiple of the larger split; padded tokens are zero vectors that take part in the softmax
                       // of their window with score = bias, and their outputs are dropped).  == H, W when nothing is padded
  int wh, ww;          // query window (wh * ww == 64 * QT)
  int sy, sx;          // cyclic shift (self-attention), 0 = none
  int kh, kw, pad;     // key window: == (wh, ww, 0) for self-attention; (24, 24, 4) for the overlapping cross-attention
  int nWh, nWw, nH;
  float scale;
};

__device__ __forceinline__ int region_label(int v, int n, int w, int s) { return v < n - w ? 0 : (v < n - s ? 1 : 2); }

// TABLE: the bias of a (query, key) pair is looked up in the head's column of the bias table, staged once in LDS (3.8 KB / 6 KB),
// through the closed-form index -- rpi_sa[p][k] = (yp - yk + ws - 1)(2 ws - 1) + (xp - xk + ws - 1) (hat_arch.py:881-894),
// rpi_oca[p][k] = (yk - yp + ws - wse + 1)(ws + wse - 1) + (xk - xp + ws - wse + 1), wrapped by the table length when negative
// (:896-918 + torch's negative indexing) -- instead of streaming a dense [256][NK] fp32 slab per workgroup from L2.
// QT = 16-query tiles per wave: 4 (256-token windows) or 2 (128-token windows: DAT's split_size [8, 16])
template <int NT, bool OCA, bool TABLE, int QT = 4>        // NT = key tiles of 16 (8: 128 keys, 16: 256 keys, 36: 576 keys)
__global__ __launch_bounds__(256, (NT > 16 ? 1 : 3)) void win256_attn_fwd_kernel(const Win256Params p) {
  constexpr int NK = NT * 16;
  constexpr int NQ = 64 * QT;                            // queries per window
  constexpr int KW = OCA ? 24 : 16;                      // key-window width in table mode (16 x 16 / 24 x 24 windows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);          // [NK][KP]
  bf16_t* Vs = Ks + NK * KP;                             // [NK][KP]
  float* tab = reinterpret_cast<float*>(Vs + NK * KP);   // [table_rows] (TABLE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.nH;
  const int wflat = blockIdx.x / p.nH;
  const int nW = p.nWh * p.nWw;
  const int b = wflat / nW, w = wflat - b * nW;
  const int wy = w / p.nWw, wx = w - wy * p.nWw;
  const long long tok0 = (long long)b * p.H * p.W;

  // query fragment of a 16-query tile: requested one tile ahead (the first one before the K / V staging), so that its round trip
  // runs under the previous tile's work instead of in front of every tile's dependent chain
  auto load_q = [&](int qt) {
    const int ql = wave * (16 * QT) + qt * 16 + r16;
    const int qy = ql / p.ww, qx = ql - qy * p.ww;
    int y = wy * p.wh + qy + p.sy, x = wx * p.ww + qx + p.sx;
    if (y >= p.Hp) y -= p.Hp;
    if (x >= p.Wp) x -= p.Wp;
    if (y >= p.H || x >= p.W) return bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};          // zero padding of the window frame
    return *reinterpret_cast<const bf16x8_t*>(p.qkv + (tok0 + (long long)y * p.W + x) * p.ldq + h * 32 + 8 * g);
  };
  bf16x8_t qf_next = load_q(0);
  // ---- stage K, V of the window -------------------------------------------------------------------
  for (int kk = tid; kk < NK; kk += 256) {
    const int ky = kk / p.kw, kx = kk - ky * p.kw;
    int y, x;
    bool ok = true;
    if constexpr (OCA) {
      y = wy * p.wh - p.pad + ky;
      x = wx * p.ww - p.pad + kx;
      ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    } else {
      y = wy * p.wh + ky + p.sy;
      x = wx * p.ww + kx + p.sx;
      if (y >= p.Hp) y -= p.Hp;
      if (x >= p.Wp) x -= p.Wp;
      ok = y < p.H && x < p.W;
    }
    uint4 kv[4], vv[4];
    if (ok) {
      const bf16_t* row = p.qkv + (tok0 + (long long)y * p.W + x) * p.ldq + h * 32;
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
      *reinterpret_cast<uint4*>(Ks + kk * KP + 8 * c) = kv[c];
      *reinterpret_cast<uint4*>(Vs + kk * KP + 8 * c) = vv[c];
    }
  }
  if constexpr (TABLE) {
    for (int i = tid; i < p.table_rows; i += 256) tab[i] = p.bias[(long long)i * p.nH + h];
  }
  __syncthreads();

  const float* bias_h = p.bias + (long long)h * NQ * NK;
  const bool masked = !OCA && (p.sy > 0 || p.sx > 0);
  // interior windows of a shifted map have one region label throughout: only the last window row / column is masked
  const bool need_mask = masked && (wy == p.nWh - 1 || wx == p.nWw - 1);

  // 16 key tiles: the four query tiles are unrolled -- without a loop the compiler does not hoist the per-tile LDS addresses into
  // registers (236 -> ~150 VGPRs), which lets a third workgroup share the CU; 36 key tiles (overlapping windows) stay rolled
  constexpr int QT_UNROLL = NT <= 16 ? QT : 1;
#pragma unroll QT_UNROLL
  for (int qt = 0; qt < QT; ++qt) {
    // this lane's query: window-local index, raster token, region label
    const int ql = wave * (16 * QT) + qt * 16 + r16;
    const int qy = ql / p.ww, qx = ql - qy * p.ww;
    int y = wy * p.wh + qy + p.sy, x = wx * p.ww + qx + p.sx;
    if (y >= p.Hp) y -= p.Hp;
    if (x >= p.Wp) x -= p.Wp;
    const bool qreal = y < p.H && x < p.W;                 // a query in the zero padding has no token: its row is not stored
    const long long qtok = tok0 + (long long)y * p.W + x;
    const int qlab = need_mask ? region_label(wy * p.wh + qy, p.Hp, p.wh, p.sy) * 3 + region_label(wx * p.ww + qx, p.Wp, p.ww, p.sx) : 0;
    const bf16x8_t qf = qf_next;
    if (qt < QT - 1) qf_next = load_q(qt + 1);

    // ---- S^T tiles -----------------------------------------------------------------------------------
    f32x4_t s[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (16 * j + r16) * KP + 8 * g);
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // ---- scale, bias, mask, row max ----------------------------------------------------------------------
    // key kl = 16 j + 4 g + e of this lane: with c = 16 j + 4 g (a multiple of 4) the key row is c / KW and the key column
    // c % KW + e (no carry: KW is a multiple of 4) -- one division per tile, not per element
    const float* brow = bias_h + (long long)ql * NK + 4 * g;
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float4 bv;
      if constexpr (TABLE) {
        const int c = 16 * j + 4 * g;
        const int ky = c / KW, kx0 = c - ky * KW;
        float be[4];
        if constexpr (OCA) {
          const int base = (ky - qy - 7) * 39 + (kx0 - qx - 7);                 // ws 16, wse 24: off = -7, 16 + 24 - 1 = 39
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int idx = base + e;
            if (idx < 0) idx += p.table_rows;                                    // torch's negative-index wrap
            be[e] = tab[idx];
          }
        } else {
          const int base = (qy - ky + 15) * 31 + (qx - kx0 + 15);
#pragma unroll
          for (int e = 0; e < 4; ++e) be[e] = tab[base - e];
        }
        bv = make_float4(be[0], be[1], be[2], be[3]);
      } else {
        bv = *reinterpret_cast<const float4*>(brow + 16 * j);
      }
      float v0 = s[j][0] * p.scale + bv.x, v1 = s[j][1] * p.scale + bv.y, v2 = s[j][2] * p.scale + bv.z, v3 = s[j][3] * p.scale + bv.w;
      if (need_mask) {                                                  // border windows only; any window shape with ww % 4 == 0
        const int cm = 16 * j + 4 * g;
        const int kym = cm / p.ww, kxm = cm - kym * p.ww;
        const int lh = region_label(wy * p.wh + kym, p.Hp, p.wh, p.sy) * 3;
        const int xb = wx * p.ww + kxm;
        if (lh + region_label(xb, p.Wp, p.ww, p.sx) != qlab) v0 += -100.0f;          // hat_arch.py:939 (-100, not -inf)
        if (lh + region_label(xb + 1, p.Wp, p.ww, p.sx) != qlab) v1 += -100.0f;
        if (lh + region_label(xb + 2, p.Wp, p.ww, p.sx) != qlab) v2 += -100.0f;
        if (lh + region_label(xb + 3, p.Wp, p.ww, p.sx) != qlab) v3 += -100.0f;
      }
      s[j] = f32x4_t{v0, v1, v2, v3};
      mx = fmaxf(mx, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
    }
    mx = xrow_max4(mx);
    constexpr float L2E = 1.4426950408889634f;      // exp(x - m) = exp2(x log2e - m log2e): one fma + v_exp_f32 per score (as softmax_numerators)
    const float mxl = mx * L2E;
    f32x4_t a4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4_t t = s[j] * L2E - mxl;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[j][e] = __builtin_amdgcn_exp2f(t[e]);
      a4 += s[j];
    }
    const float inv = __builtin_amdgcn_rcpf(xrow_sum4((a4[0] + a4[1]) + (a4[2] + a4[3])));

    // ---- O^T = V^T P^T ---------------------------------------------------------------------------------
    f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int jj = 0; jj < NT / 2; ++jj) {
      // B operand: K slots (g, 0..7) = keys 32 jj + 4 g + 0..3 and 32 jj + 16 + 4 g + 0..3 of this lane's query
      const uint2 lo = pack_bf4(s[2 * jj][0], s[2 * jj][1], s[2 * jj][2], s[2 * jj][3]);
      const uint2 hi = pack_bf4(s[2 * jj + 1][0], s[2 * jj + 1][1], s[2 * jj + 1][2], s[2 * jj + 1][3]);
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, u32x4{lo.x, lo.y, hi.x, hi.y});
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x4_t a0 = lds_tr_read(tr_addr(Vs, KP, 32 * jj + 4 * g, 16 * dt, lane));
        const bf16x4_t a1 = lds_tr_read(tr_addr(Vs, KP, 32 * jj + 16 + 4 * g, 16 * dt, lane));
        const bf16x8_t vf = bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    // lane holds O[q = r16][d = 16 dt + 4 g + 0..3]
    if (qreal) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<uint2*>(p.out + qtok * p.ldo + h * 32 + 16 * dt + 4 * g) =
            pack_bf4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    }
  }
}

template <int NT, bool OCA, bool TABLE, int QT = 4>
int launch(const Win256Params& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * NT * 16 * KP * sizeof(bf16_t) + (TABLE ? 1536 * sizeof(float) : 0);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&win256_attn_fwd_kernel<NT, OCA, TABLE, QT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      srk_set_error("win256 attention: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const long long grid = (long long)p.B * p.nWh * p.nWw * p.nH;
  SRK_REQUIRE(grid > 0 && grid < (1LL << 31), SRK_E_SHAPE, "win256 attention: bad grid %lld", grid);
  hipLaunchKernelGGL((win256_attn_fwd_kernel<NT, OCA, TABLE, QT>), dim3((unsigned)grid), dim3(256), lds, stream, p);
  return srk_check_launch("win256_attn_fwd");
}

}  // namespace

int srk_launch_win256_attn_fwd(const bf16_t* qkv, int ldq, int CA, const float* bias, int table_rows, bf16_t* out, int ldo, int B, int H, int W,
                               int wh, int ww, int sy, int sx, int nH, float scale, int overlap, hipStream_t stream) {
  return srk_launch_win_attn_fwd_padded(qkv, ldq, CA, bias, table_rows, out, ldo, B, H, W, H, W, wh, ww, sy, sx, nH, scale, overlap, stream);
}

// Hp x Wp: the window frame -- the H x W map zero-padded at the bottom / right (Hp >= H, Wp >= W, multiples of the window)
int srk_launch_win_attn_fwd_padded(const bf16_t* qkv, int ldq, int CA, const float* bias, int table_rows, bf16_t* out, int ldo, int B, int H,
                                   int W, int Hp, int Wp, int wh, int ww, int sy, int sx, int nH, float scale, int overlap,
                                   hipStream_t stream) {
  SRK_REQUIRE(qkv && bias && out, SRK_E_NULL, "win256 attention: null pointer");
  SRK_REQUIRE(ww % 4 == 0, SRK_E_UNSUPPORTED, "win256 attention: window width %d must be a multiple of 4", ww);
  SRK_REQUIRE(wh > 0 && ww > 0 && (wh * ww == 256 || wh * ww == 128), SRK_E_UNSUPPORTED,
              "win256 attention: the window must hold 256 or 128 tokens (got %dx%d)", wh, ww);
  SRK_REQUIRE(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp % wh == 0 && Wp % ww == 0, SRK_E_SHAPE,
              "win256 attention: window frame %dx%d must cover the %dx%d map and be a multiple of the %dx%d window", Hp, Wp, H, W, wh, ww);
  SRK_REQUIRE(overlap == 0 || (Hp == H && Wp == W && wh * ww == 256), SRK_E_UNSUPPORTED,
              "overlapping cross-attention takes unpadded maps and 16x16 windows");
  // CA is the column distance between the q, k and v blocks of a row; a caller may run a SUBSET of the heads (DAT's two branches take
  // heads 0..nH/2-1 and nH/2..nH-1 with different window shapes) by offsetting qkv / out by 32 * first_head
  SRK_REQUIRE(nH > 0 && CA >= nH * 32 && CA % 32 == 0 && ldq >= 3 * CA && ldq % 8 == 0 && ldo >= nH * 32 && ldo % 4 == 0, SRK_E_SHAPE,
              "win256 attention: bad layout nH=%d CA=%d ldq=%d ldo=%d", nH, CA, ldq, ldo);
  SRK_REQUIRE(sy >= 0 && sy < wh && sx >= 0 && sx < ww, SRK_E_SHAPE, "shift_size must in 0-window_size");
  Win256Params p;
  p.table_rows = table_rows;
  if (table_rows > 0) {
    SRK_REQUIRE(wh == 16 && ww == 16 && table_rows == (overlap > 0 ? 39 * 39 : 31 * 31), SRK_E_UNSUPPORTED,
                "win256 attention: table-indexed bias is built for 16x16 windows (961 rows; 1521 for the overlapping form), got %d rows",
                table_rows);
  }
  p.qkv = qkv; p.out = out; p.bias = bias; p.ldq = ldq; p.ldo = ldo; p.CA = CA; p.B = B; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp;
  p.wh = wh; p.ww = ww;
  p.sy = sy; p.sx = sx; p.kh = wh; p.kw = ww; p.pad = 0; p.nWh = Hp / wh; p.nWw = Wp / ww; p.nH = nH; p.scale = scale;
  if (overlap > 0) {
    // overlapping cross-attention: key window = window + 2 * overlap per side-pair, i.e. wh + overlap: overlap = int(ratio * ws)
    SRK_REQUIRE(wh == 16 && ww == 16 && overlap == 8 && sy == 0 && sx == 0, SRK_E_UNSUPPORTED,
                "overlapping cross-attention is built for 16x16 windows with overlap 8 (24x24 keys), no shift");
    p.kh = wh + overlap; p.kw = ww + overlap; p.pad = overlap / 2;
    return table_rows > 0 ? launch<36, true, true>(p, stream) : launch<36, true, false>(p, stream);
  }
  if (wh * ww == 128) {
    SRK_REQUIRE(table_rows == 0, SRK_E_UNSUPPORTED, "win256 attention: 128-token windows take a dense bias");
    return launch<8, false, false, 2>(p, stream);
  }
  return table_rows > 0 ? launch<16, false, true>(p, stream) : launch<16, false, false>(p, stream);
}
