// All-taps weight gradient of a 3x3 / stride 1 / pad 1 conv on NHWC bf16 (gfx950, v_mfma_f32_16x16x32_bf16):
//
//   dW[n][tap][k] += sum_{b,y,x} Y[b][y][x][n] * X[b][y+dy][x+dx][k]      db[n] += sum Y[b][y][x][n]
//
// wgrad.hip treats the nine taps as nine independent GEMM tiles: every tap re-reads the whole Y tile and a
// shifted copy of the same X rows, and a workgroup gets 16-32 MFMAs per 24 KB it stages -- the kernel is bound
// by staging latency at a few percent of either roofline.  Here a workgroup owns 64 output channels x 64 input
// channels for ALL nine taps: per 64-pixel run of an image row it stages the Y rows once (8 KB) and the X halo
// (3 rows x 66 pixels, 25 KB) once, and issues 288 MFMAs on them (each wave: all four n-fragments x nine of
// the 36 (tap, k-fragment) column groups; 144 accumulator registers).  Operands are m-major in memory, so the
// fragments come from the transposing LDS read, with the same row padding as wgrad.hip.
#include "wgrad.h"

#ifndef SRK_NT_TAPS
#define SRK_NT_TAPS 0
#endif

namespace {

constexpr int CT = 64;                 // pixels per chunk: one run inside an image row (needs W % 64 == 0)
constexpr int SP = 64 + 16;            // LDS row stride in elements: 32 B x odd -> conflict-free transposing reads
constexpr int HALO = CT + 2;           // pixels per halo row
constexpr int XROWS = 3 * HALO;
constexpr int YPIECES = CT * 8 / 256;                 // 16-byte pieces per thread
constexpr int XPIECES = (XROWS * 8 + 255) / 256;

template <bool SHUF>
__global__ __launch_bounds__(256) void conv_wgrad_taps_kernel(const WgradParams p, int ntiles, int chunks_per) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ys = reinterpret_cast<bf16_t*>(smem);      // [2][CT][SP]
  bf16_t* Xs = Ys + 2 * CT * SP;                     // [2][XROWS][SP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // 1-D grid, (m-split, tile) with the tiles of one m-split contiguous on one XCD: they re-read the same rows
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / ntiles, btile = logical - bsplit * ntiles;
  const int ntn = p.N / 64;
  const int tn = btile % ntn, tk = btile / ntn;
  const int n0 = tn * 64, k0 = tk * 64;
  const int c_begin = bsplit * chunks_per;
  const int c_end = min(p.M / CT, c_begin + chunks_per);
  if (c_begin >= c_end) return;
  const int hw = p.H * p.W;

  uint4 ry[YPIECES], rx[XPIECES];
  auto load_stage = [&](int ch) {
    const int m0 = ch * CT;
    const int b = m0 / hw, rem = m0 - b * hw;
    const int y = rem / p.W, x0 = rem - y * p.W;
#pragma unroll
    for (int t = 0; t < YPIECES; ++t) {
      const int pid = tid + 256 * t;
      const int row = pid >> 3, c8 = pid & 7;
      if constexpr (SHUF) {
        // Y lives pixel-shuffled: channel n = (si*r + sj)*Cs + c of pixel (y, x) is stored at [r*y+si][r*x+sj][c]
        const int nn = n0 + c8 * 8;
        const int ij = nn / p.Cs, c = nn - ij * p.Cs;
        const int si = ij / p.r, sj = ij - si * p.r;
        ry[t] = *reinterpret_cast<const uint4*>(
            p.Y + (((long long)(b * p.H * p.r + y * p.r + si)) * (p.W * p.r) + (x0 + row) * p.r + sj) * p.Cs + c);
      } else {
        ry[t] = *reinterpret_cast<const uint4*>(p.Y + (long long)(m0 + row) * p.ldy + n0 + c8 * 8);
      }
    }
#pragma unroll
    for (int t = 0; t < XPIECES; ++t) {
      const int pid = tid + 256 * t;
      const int hr = pid >> 3, c8 = pid & 7;
      const int rr = hr / HALO, px = hr - rr * HALO;
      const int yy = y + rr - 1, xx = x0 + px - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (hr < XROWS && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
        v = *reinterpret_cast<const uint4*>(p.X + ((long long)(b * p.H + yy) * p.W + xx) * p.ldx + k0 + c8 * 8);
      rx[t] = v;
    }
  };
  auto store_stage = [&](int buf) {
    bf16_t* ys = Ys + buf * CT * SP;
    bf16_t* xs = Xs + buf * XROWS * SP;
#pragma unroll
    for (int t = 0; t < YPIECES; ++t) {
      const int pid = tid + 256 * t;
      *reinterpret_cast<uint4*>(ys + (pid >> 3) * SP + (pid & 7) * 8) = ry[t];
    }
#pragma unroll
    for (int t = 0; t < XPIECES; ++t) {
      const int pid = tid + 256 * t;
      if (pid < XROWS * 8) *reinterpret_cast<uint4*>(xs + (pid >> 3) * SP + (pid & 7) * 8) = rx[t];
    }
  };

  f32x4_t acc[4][9], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[i][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.db != nullptr && tk == 0 && wave == 0;
  const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  // this wave's nine column groups: cg = 9 wave + q -> (tap = cg / 4, k-fragment = cg % 4)

  load_stage(c_begin);
  store_stage(0);
  __syncthreads();
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) load_stage(ch + 1);
    const bf16_t* ys = Ys + buf * CT * SP;
    const bf16_t* xs = Xs + buf * XROWS * SP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // k order inside a fragment: jj < 4 -> pixel 32ks + 4g + jj, jj >= 4 -> pixel 32ks + 16 + 4g + jj - 4 (same for Y and X)
      bf16x8_t yf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4_t lo = lds_tr_read(tr_addr(ys, SP, 32 * ks + 4 * g, 16 * i, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(ys, SP, 32 * ks + 16 + 4 * g, 16 * i, lane));
        yf[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int cg = 9 * wave + q;
        const int tap = cg >> 2;
        const int xb = (tap / 3) * HALO + (tap % 3);        // halo row of pixel x0 shifted by this tap's (dy, dx)
        const bf16x4_t lo = lds_tr_read(tr_addr(xs, SP, xb + 32 * ks + 4 * g, 16 * (cg & 3), lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(xs, SP, xb + 32 * ks + 16 + 4 * g, 16 * (cg & 3), lane));
        const bf16x8_t xf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf, acc[i][q], 0, 0, 0);
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, accb[i], 0, 0, 0);
      }
    }
    if (ch + 1 < c_end) store_stage(buf ^ 1);
    __syncthreads();
  }

  // acc[i][q][e] = dW[n = n0 + 16i + 4g + e][tap][k = k0 + 16kf + r16]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 16 * i + 4 * g;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int cg = 9 * wave + q;
      const long long col = (long long)(cg >> 2) * p.K + k0 + 16 * (cg & 3) + r16;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + col, acc[i][q][e]);
    }
    if (do_bias && r16 == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.db + n + e, accb[i][e]);
    }
  }
}

SrkOpt g_taps_enabled{OPT_TAPS_ENABLED, 1};

template <bool SHUF>
int launch_taps(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (CT + XROWS) * SP * sizeof(bf16_t);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<SHUF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      srk_set_error("conv wgrad: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const int tiles = (p.N / 64) * (p.K / 64);
  const int nchunks = p.M / CT;
  // one workgroup per CU (LDS-limited): ~256 workgroups; every m-split costs one pass of fp32 atomics over the tile
  int splits = 256 / tiles;
  if (splits < 1) splits = 1;
  if (splits > nchunks) splits = nchunks;
  const int chunks_per = cdiv(nchunks, splits);
  splits = cdiv(nchunks, chunks_per);
  srk_probe_pre(FAM_WGRAD_CONV, stream, p.flops, p.bytes);
  hipLaunchKernelGGL((conv_wgrad_taps_kernel<SHUF>), dim3(tiles * splits), dim3(256), lds, stream, p, tiles, chunks_per);
  srk_probe_post(FAM_WGRAD_CONV, stream);
  return srk_check_launch("conv wgrad (all taps)");
}

// ---- LDS-DMA ring variant of the all-taps kernel -----------------------------------------------------------------------
// Same tile and MFMA schedule as conv_wgrad_taps_kernel, but the dY rows and the X halo of a 64-pixel run arrive by
// LDS-DMA into a 4-deep ring (33.5 KB per stage, three runs in flight) instead of one register-staged run: the
// register-staged loop was bound by its load latency (2.4 us per run for 0.5 us of MFMA work).  Out-of-image halo
// pixels are DMA'd from a zero page (the source address is per lane), rows are unpadded 128-byte pixel rows whose
// 32-byte column pairs are XOR-swizzled with (row >> 1) & 3 on the source address (conflict-free transposing reads for
// any 8 consecutive rows), no wave stores inside the loop (exact counted vmcnt), one raw barrier per run.
__device__ uint4 g_zero_page[8];     // 128 zero bytes: DMA source for out-of-image pixels

constexpr int TD_YROWS = CT, TD_STAGE_ROWS = CT + XROWS;       // 64 + 198 rows of 64 bf16
constexpr int TD_STAGE_BYTES = TD_STAGE_ROWS * 128;
constexpr int TD_RING = 4;
constexpr int TD_PIECES = TD_STAGE_ROWS * 8;                   // 16-byte pieces per stage (2096)
constexpr int TD_PPW = TD_PIECES / 4;                          // per wave (524)
constexpr int TD_NI = (TD_PPW + 63) / 64;                      // DMA instructions per wave per run (9)

__device__ __forceinline__ const bf16_t* tr_addr_swz64(const bf16_t* tile, int rbase, int c0, int lane) {
  const int ll = lane & 15;
  const int row = rbase + (ll >> 2);
  return tile + row * 64 + (((c0 >> 4) ^ ((row >> 1) & 3)) << 4) + ((ll & 3) << 2);
}

template <bool SHUF>
__global__ __launch_bounds__(256) void conv_wgrad_taps_dma_kernel(const WgradParams p, int ntiles, int chunks_per, float* partial,
                                                                  int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned ring_base = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / ntiles, btile = logical - bsplit * ntiles;
  const int ntn = p.N / 64;
  const int tn = btile % ntn, tk = btile / ntn;
  const int n0 = tn * 64, k0 = tk * 64;
  const int c_begin = bsplit * chunks_per;
  const int c_end = min(p.M / CT, c_begin + chunks_per);
  if (c_begin >= c_end) return;
  const int hw = p.H * p.W;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // stage image: rows 0..63 = dY of the run's pixels, rows 64.. = X halo (3 rows x 66 pixels); this wave moves pieces
  // [wave * TD_PPW, (wave + 1) * TD_PPW).  Which piece a lane moves in DMA instruction i never changes, so everything
  // that depends on the lane only is computed once: the element offset relative to the run's first pixel and, for halo
  // pieces, (halo row, halo pixel) for the per-run bounds test.  Per run and instruction this leaves two compares, a
  // select and a 64-bit add (the address math otherwise costs more issue slots than the 72 MFMAs of the run).
  int loff[TD_NI];          // element offset of the piece relative to the run base (dY or X)
  int lrp[TD_NI];           // dY piece: -1; halo piece: (rr << 8) | px
#pragma unroll
  for (int i = 0; i < TD_NI; ++i) {
    const int qq = wave * TD_PPW + i * 64 + lane;
    const int row = qq >> 3, pos = qq & 7;
    const int c = (((pos >> 1) ^ ((row >> 1) & 3)) << 1) | (pos & 1);           // logical 16-byte piece of this LDS position
    if (row < TD_YROWS) {
      lrp[i] = -1;
      if constexpr (SHUF) {
        const int nn = n0 + c * 8;
        const int ij = nn / p.Cs, cc = nn - ij * p.Cs;
        const int si = ij / p.r, sj = ij - si * p.r;
        loff[i] = (si * (p.W * p.r) + row * p.r + sj) * p.Cs + cc;
      } else {
        loff[i] = row * p.ldy + n0 + c * 8;
      }
    } else {
      const int hr = row - TD_YROWS;
      const int rr = hr / HALO, px = hr - rr * HALO;
      lrp[i] = (rr << 8) | px;
      loff[i] = ((rr - 1) * p.W + (px - 1)) * p.ldx + k0 + c * 8;
    }
  }
  auto issue = [&](int ch) {
    const int m0 = ch * CT;
    const int b = m0 / hw, rem = m0 - b * hw;
    const int y = rem / p.W, x0 = rem - y * p.W;
    const bf16_t* ybase = SHUF ? p.Y + (((long long)(b * p.H * p.r + y * p.r)) * (p.W * p.r) + (long long)x0 * p.r) * p.Cs
                               : p.Y + (long long)m0 * p.ldy;
    const bf16_t* xbase = p.X + (long long)m0 * p.ldx;
    const unsigned dst = ring_base + (unsigned)(((ch - c_begin) % TD_RING) * TD_STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < TD_NI; ++i) {
      if (i * 64 + lane < TD_PPW) {
        const bf16_t* src;
        if (lrp[i] < 0) {
          src = ybase + loff[i];
        } else {
          const int yy = y + (lrp[i] >> 8) - 1, xx = x0 + (lrp[i] & 255) - 1;
          src = ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) ? xbase + loff[i] : zero;
        }
        srk_glds16<SRK_NT_TAPS != 0>(src, __builtin_amdgcn_readfirstlane(dst + (wave * TD_PPW + i * 64) * 16));
      }
    }
  };

  f32x4_t acc[4][9], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[i][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.db != nullptr && tk == 0 && wave == 0;
  const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  const int nrun = c_end - c_begin;
  for (int s = 0; s < TD_RING - 1 && s < nrun; ++s) issue(c_begin + s);
  for (int it = 0; it < nrun; ++it) {
    // run `it` has landed once at most the (TD_RING - 2) runs issued after it are outstanding
    if (it + TD_RING - 2 < nrun) srk_wait_vmcnt<TD_NI*(TD_RING - 2)>(); else srk_wait_vmcnt<0>();
    srk_lds_barrier();
    if (it + TD_RING - 1 < nrun) issue(c_begin + it + TD_RING - 1);
    const bf16_t* ys = reinterpret_cast<const bf16_t*>(smem + (it % TD_RING) * TD_STAGE_BYTES);
    const bf16_t* xs = ys + TD_YROWS * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4_t lo = lds_tr_read(tr_addr_swz64(ys, 32 * ks + 4 * g, 16 * i, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr_swz64(ys, 32 * ks + 16 + 4 * g, 16 * i, lane));
        yf[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int cg = 9 * wave + q;
        const int tap = cg >> 2;
        const int xb = (tap / 3) * HALO + (tap % 3);
        // NOTE the swizzle is a function of the row inside the STAGE image (dY rows first), so the halo rows are offset
        const bf16x4_t lo = lds_tr_read(tr_addr_swz64(ys, TD_YROWS + xb + 32 * ks + 4 * g, 16 * (cg & 3), lane));
        const bf16x4_t hi = lds_tr_read(tr_addr_swz64(ys, TD_YROWS + xb + 32 * ks + 16 + 4 * g, 16 * (cg & 3), lane));
        const bf16x8_t xf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf, acc[i][q], 0, 0, 0);
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, accb[i], 0, 0, 0);
      }
    }
    (void)xs;
  }

  if (partial != nullptr) {
    // partial tile in accumulator order, summed over the splits by conv_wgrad_taps_reduce_kernel (see wgrad.hip)
    f32x4_t* slab = reinterpret_cast<f32x4_t*>(partial) + ((size_t)btile * nsplit + bsplit) * WS_SLAB_VEC + (size_t)wave * 36 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 9; ++q) slab[(i * 9 + q) * 64] = acc[i][q];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 16 * i + 4 * g;
    if (partial == nullptr) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int cg = 9 * wave + q;
        const long long col = (long long)(cg >> 2) * p.K + k0 + 16 * (cg & 3) + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + col, acc[i][q][e]);
      }
    }
    if (do_bias && r16 == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.db + n + e, accb[i][e]);
    }
  }
}

// sum of the nsplit partial tiles of conv_wgrad_taps_dma_kernel, added to dW (36 workgroups per 64 x 64 x 9-tap tile)
__global__ __launch_bounds__(256) void conv_wgrad_taps_reduce_kernel(const WgradParams p, const float* __restrict__ partial, int nsplit) {
  const int btile = blockIdx.x / 36;
  const int v = (blockIdx.x - btile * 36) * 256 + threadIdx.x;
  const int ntn = p.N / 64;
  const int tn = btile % ntn, tk = btile / ntn;
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(partial) + (size_t)btile * nsplit * WS_SLAB_VEC + v;
  f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= nsplit; s += 8) {
    f32x4_t t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(s + u) * WS_SLAB_VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += t[u];
  }
  for (; s < nsplit; ++s) sum += src[(size_t)s * WS_SLAB_VEC];
  const int lane = v & 63, t = v >> 6;
  const int q = t % 9, i = (t / 9) % 4, wave = t / 36;
  const int r16 = lane & 15, g = lane >> 4;
  const int n = tn * 64 + 16 * i + 4 * g;
  const int cg = 9 * wave + q;
  const long long col = (long long)(cg >> 2) * p.K + tk * 64 + 16 * (cg & 3) + r16;
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + col, sum[e]);
}

SrkOpt g_taps_dma{OPT_TAPS_DMA, 1};

template <bool SHUF>
int launch_taps_dma(const WgradParams& p, hipStream_t stream) {
  constexpr int lds = TD_RING * TD_STAGE_BYTES;
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_dma_kernel<SHUF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess) {
      srk_set_error("conv wgrad (dma): cannot reserve %d bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const int tiles = (p.N / 64) * (p.K / 64);
  const int nchunks = p.M / CT;
  int splits = 256 / tiles;
  if (splits < 1) splits = 1;
  if (splits > nchunks) splits = nchunks;
  const int chunks_per = cdiv(nchunks, splits);
  splits = cdiv(nchunks, chunks_per);
  float* partial =
      srk_wgrad_partials_enabled() && splits > 1 ? srk_wgrad_scratch(stream, (size_t)tiles * splits * WS_SLAB_VEC * 16) : nullptr;
  srk_probe_pre(FAM_WGRAD_CONV, stream, p.flops, p.bytes);
  hipLaunchKernelGGL((conv_wgrad_taps_dma_kernel<SHUF>), dim3(tiles * splits), dim3(256), lds, stream, p, tiles, chunks_per, partial,
                     splits);
  if (partial) hipLaunchKernelGGL(conv_wgrad_taps_reduce_kernel, dim3(tiles * 36), dim3(256), 0, stream, p, partial, splits);
  srk_probe_post(FAM_WGRAD_CONV, stream);
  return srk_check_launch("conv wgrad (all taps, dma)");
}

// ---- image-head variant: Cout <= 16, dY in fp32 ------------------------------------------------------------------
// Weight gradient of the convs that produce the image (conv_last 64 -> 3 at HR resolution, UpsampleOneStep): same
// all-taps structure with ONE 16-row n-fragment.  dY arrives as fp32 [pixels][COP] (the L1-loss gradient) and is split
// into bf16 hi + lo on the way into LDS (two MFMAs per fragment pair, ~2^-17 relative), so this path keeps fp32-grade
// precision while the 2 M-pixel reduction runs on the matrix cores instead of the VALU (0.9 ms -> see profiles/).
template <int COP>
__global__ __launch_bounds__(256) void smallconv_wgrad_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gy,
                                                                   float* __restrict__ dW, float* __restrict__ db, int B, int H, int W,
                                                                   int Cin, int CinP, int Co, int ntiles, int chunks_per) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Yh = reinterpret_cast<bf16_t*>(smem);      // [2][CT][16]
  bf16_t* Yl = Yh + 2 * CT * 16;                     // [2][CT][16]
  bf16_t* Xs = Yl + 2 * CT * 16;                     // [2][XROWS][SP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / ntiles, tk = logical - bsplit * ntiles;
  const int k0 = tk * 64;
  const int M = B * H * W;
  const int c_begin = bsplit * chunks_per;
  const int c_end = min(M / CT, c_begin + chunks_per);
  if (c_begin >= c_end) return;
  const int hw = H * W;
  for (int i = tid; i < 2 * 2 * CT * 16 / 2; i += 256) reinterpret_cast<unsigned*>(Yh)[i] = 0u;   // columns >= COP stay zero

  constexpr int YP = CT * COP / 4;          // float4 pieces of dY per chunk (64 or 256)
  float4 ry = make_float4(0.f, 0.f, 0.f, 0.f);
  uint4 rx[XPIECES];
  // chunk -> (image, column block, row): consecutive chunks of a workgroup walk DOWN a CT-pixel column, so two of the three
  // halo rows of a chunk were read by the previous one (a row-major walk re-read them W / CT chunks later, after they had left
  // the L2: 0.79 GB of fetches per launch for a 268 MB input)
  const int nxb = W / CT;
  auto load_stage = [&](int ch) {
    const int b = ch / (H * nxb), r = ch - b * (H * nxb);
    const int xb = r / H, y = r - xb * H, x0 = xb * CT;
    const int m0 = b * hw + y * W + x0;
    if (tid < YP) ry = *reinterpret_cast<const float4*>(gy + (long long)m0 * COP + tid * 4);
#pragma unroll
    for (int t = 0; t < XPIECES; ++t) {
      const int pid = tid + 256 * t;
      const int hr = pid >> 3, c8 = pid & 7;
      const int rr = hr / HALO, px = hr - rr * HALO;
      const int yy = y + rr - 1, xx = x0 + px - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (hr < XROWS && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
        v = *reinterpret_cast<const uint4*>(x + ((long long)(b * H + yy) * W + xx) * CinP + k0 + c8 * 8);
      rx[t] = v;
    }
  };
  auto store_stage = [&](int buf) {
    if (tid < YP) {
      const int row = tid / (COP / 4), f4 = tid % (COP / 4);
      const float gv[4] = {ry.x, ry.y, ry.z, ry.w};
      float hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = bf2f(f2bf(gv[e]));
        lo[e] = gv[e] - hi[e];
      }
      *reinterpret_cast<uint2*>(Yh + (buf * CT + row) * 16 + f4 * 4) = pack_bf4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint2*>(Yl + (buf * CT + row) * 16 + f4 * 4) = pack_bf4(lo[0], lo[1], lo[2], lo[3]);
    }
    bf16_t* xs = Xs + buf * XROWS * SP;
#pragma unroll
    for (int t = 0; t < XPIECES; ++t) {
      const int pid = tid + 256 * t;
      if (pid < XROWS * 8) *reinterpret_cast<uint4*>(xs + (pid >> 3) * SP + (pid & 7) * 8) = rx[t];
    }
  };

  f32x4_t acc[9], accb = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 9; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = db != nullptr && tk == 0 && wave == 0;
  const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  load_stage(c_begin);
  __syncthreads();                                   // the zero fill above is complete
  store_stage(0);
  __syncthreads();
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) load_stage(ch + 1);
    const bf16_t* yh = Yh + buf * CT * 16;
    const bf16_t* yl = Yl + buf * CT * 16;
    const bf16_t* xs = Xs + buf * XROWS * SP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yfh, yfl;
      {
        const bf16x4_t lo = lds_tr_read(tr_addr(yh, 16, 32 * ks + 4 * g, 0, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(yh, 16, 32 * ks + 16 + 4 * g, 0, lane));
        yfh = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x4_t lo2 = lds_tr_read(tr_addr(yl, 16, 32 * ks + 4 * g, 0, lane));
        const bf16x4_t hi2 = lds_tr_read(tr_addr(yl, 16, 32 * ks + 16 + 4 * g, 0, lane));
        yfl = bf16x8_t{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int cg = 9 * wave + q;
        const int tap = cg >> 2, kf = cg & 3;
        const int xb = (tap / 3) * HALO + (tap % 3);
        const bf16x4_t lo = lds_tr_read(tr_addr(xs, SP, xb + 32 * ks + 4 * g, 16 * kf, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(xs, SP, xb + 32 * ks + 16 + 4 * g, 16 * kf, lane));
        const bf16x8_t xf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yfh, xf, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yfl, xf, acc[q], 0, 0, 0);
      }
      if (do_bias) {
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yfh, ones, accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yfl, ones, accb, 0, 0, 0);
      }
    }
    if (ch + 1 < c_end) store_stage(buf ^ 1);
    __syncthreads();
  }

  // acc[q][e] = dW[n = 4g + e][k = k0 + 16 kf + r16][tap]   (state_dict layout [Cout][Cin][3][3])
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int cg = 9 * wave + q;
    const int tap = cg >> 2, k = k0 + 16 * (cg & 3) + r16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 4 * g + e;
      if (n < Co && k < Cin) atomicAdd(dW + ((long long)(n * Cin) + k) * 9 + tap, acc[q][e]);
    }
  }
  if (do_bias && r16 == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * g + e < Co) atomicAdd(db + 4 * g + e, accb[e]);
  }
}

template <int COP>
int launch_smallconv_mfma(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int CinP, int Co,
                          hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * 2 * CT * 16 + 2 * XROWS * SP) * sizeof(bf16_t);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_wgrad_mfma_kernel<COP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      srk_set_error("smallconv wgrad: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  const int tiles = CinP / 64;
  const int nchunks = (B * H * W) / CT;
  int splits = 512 / tiles;                       // two workgroups per CU
  if (splits > nchunks) splits = nchunks;
  const int chunks_per = cdiv(nchunks, splits);
  splits = cdiv(nchunks, chunks_per);
  hipLaunchKernelGGL((smallconv_wgrad_mfma_kernel<COP>), dim3(tiles * splits), dim3(256), lds, stream, x, gy, dW, db, B, H, W, Cin, CinP, Co,
                     tiles, chunks_per);
  return srk_check_launch("smallconv wgrad (mfma)");
}

// ---- image-head dgrad on the matrix cores ---------------------------------------------------------------------------
//   dX[pix][ci] = sum_{tap,co} gy[pix - off(tap)][co] * W[co][ci][tap]      (Cout <= 4, Cin <= 64; conv_last of the x2/x4 heads)
// A GEMM with K = 9 taps x 4 channels = 36 (padded to 64): the A fragment of a pixel is gathered straight from the fp32
// dY (two taps = 8 consecutive k per lane), W sits in registers, both operands are split into bf16 hi + lo (hi*hi + lo*hi
// + hi*lo, ~2^-16 relative: the result is then rounded to bf16 once, like the fp32 VALU kernel it replaces), and the
// 64 x 64 output tile goes through LDS so that every store is a full 128-byte pixel row.
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16_t h = f2bf(v[i]);
    hi[i] = (short)h;
    lo[i] = (short)f2bf(v[i] - bf2f(h));
  }
}

__global__ __launch_bounds__(256) void imghead_dgrad_mfma_kernel(const float* __restrict__ gy, const float* __restrict__ wgt,
                                                                 bf16_t* __restrict__ dx, int B, int H, int W, int Cin, int Co, int nchunks) {
  constexpr int TP = 64 + 8;
  __shared__ __attribute__((aligned(16))) bf16_t T[64 * TP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;

  // W fragments: row n = ci = 16 j + r16, k = 32 ks + 8 g + jj  ->  (tap, co) = (k / 4, k % 4)
  bf16x8_t wh[4][2], wl[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = 32 * ks + 8 * g + jj;
        const int tap = k >> 2, co = k & 3, ci = 16 * j + r16;
        v[jj] = (tap < 9 && co < Co && ci < Cin) ? wgt[((co * Cin) + ci) * 9 + tap] : 0.f;
      }
      split_bf16x8(v, wh[j][ks], wl[j][ks]);
    }

  const int hw = H * W;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int pix = c * 64 + 16 * wave + r16;
    const int b = pix / hw, rem = pix - b * hw;
    const int y = rem / W, x = rem - y * W;
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int tap = 8 * ks + 2 * g + half;
        float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tap < 9) {
          // output pixel that used input pixel (y, x) through tap (ky, kx) is (y - (ky - 1), x - (kx - 1))
          const int yy = y - (tap / 3 - 1), xs = x - (tap % 3 - 1);
          if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W)
            gv = *reinterpret_cast<const float4*>(gy + ((long long)(b * H + yy) * W + xs) * 4);
        }
        v[4 * half] = gv.x; v[4 * half + 1] = gv.y; v[4 * half + 2] = gv.z; v[4 * half + 3] = gv.w;
      }
      bf16x8_t xh, xl;
      split_bf16x8(v, xh, xl);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j][ks], xh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j][ks], xh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j][ks], xl, acc[j], 0, 0, 0);
      }
    }
    // acc[j][e] = dX[pixel 16 wave + r16][ci = 16 j + 4 g + e]
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint2*>(T + (16 * wave + r16) * TP + 16 * j + 4 * g) = pack_bf4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    srk_lds_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pid = tid + 256 * t;
      const int row = pid >> 3, c8 = pid & 7;
      *reinterpret_cast<uint4*>(dx + ((long long)c * 64 + row) * 64 + c8 * 8) = *reinterpret_cast<const uint4*>(T + row * TP + c8 * 8);
    }
    srk_lds_barrier();
  }
}

}  // namespace

void srk_conv_wgrad_taps_enable(int on) {
  g_taps_enabled = on ? 1 : 0;    // 0: per-tap tiles / VALU image head; 1: all-taps register-staged; 2: all-taps LDS-DMA ring (default)
  g_taps_dma = on >= 2 ? 1 : 0;
}
int srk_conv_wgrad_taps_mode() { return g_taps_enabled ? (g_taps_dma ? 2 : 1) : 0; }

// SRK_WGRAD_NOT_COVERED when the all-taps kernel does not apply (the caller then uses the per-tap tiles of wgrad.hip)
int srk_launch_conv_wgrad_taps(const WgradParams& p, hipStream_t stream) {
  if (!g_taps_enabled || !p.conv || p.W % CT != 0 || p.N % 64 != 0 || p.K % 64 != 0 || p.ldx % 8 != 0) return SRK_WGRAD_NOT_COVERED;
  if (p.r > 1) {
    if (p.Cs % 8 != 0 || p.N != p.r * p.r * p.Cs) return SRK_WGRAD_NOT_COVERED;
    return g_taps_dma ? launch_taps_dma<true>(p, stream) : launch_taps<true>(p, stream);
  }
  if (p.ldy % 8 != 0) return SRK_WGRAD_NOT_COVERED;
  return g_taps_dma ? launch_taps_dma<false>(p, stream) : launch_taps<false>(p, stream);
}

// image-head convs (Cout <= 16, fp32 dY): SRK_WGRAD_NOT_COVERED -> the VALU kernel of misc.hip
int srk_launch_smallconv_wgrad_mfma(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int CinP,
                                    int Co, int CoP, hipStream_t stream) {
  if (!g_taps_enabled || W % CT != 0 || CinP % 64 != 0 || Co > CoP || ((long long)B * H * W) % CT != 0) return SRK_WGRAD_NOT_COVERED;
  if (CoP == 4) return launch_smallconv_mfma<4>(x, gy, dW, db, B, H, W, Cin, CinP, Co, stream);
  if (CoP == 16) return launch_smallconv_mfma<16>(x, gy, dW, db, B, H, W, Cin, CinP, Co, stream);
  return SRK_WGRAD_NOT_COVERED;
}

int srk_launch_imghead_dgrad_mfma(const float* gy, const float* wgt, bf16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                                  hipStream_t stream) {
  const long long npix = (long long)B * H * W;
  if (!g_taps_enabled || CoP != 4 || CinP != 64 || Co > 4 || Cin > 64 || npix % 64 != 0 || npix > (1ll << 30)) return SRK_WGRAD_NOT_COVERED;
  const int nchunks = (int)(npix / 64);
  const int grid = nchunks < 2048 ? nchunks : 2048;
  hipLaunchKernelGGL(imghead_dgrad_mfma_kernel, dim3(grid), dim3(256), 0, stream, gy, wgt, dx, B, H, W, Cin, Co, nchunks);
  return srk_check_launch("image-head dgrad (mfma)");
}
