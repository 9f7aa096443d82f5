// Weight-gradient GEMM for linear layers and 3x3 convs (gfx950, v_mfma_f32_16x16x32_bf16):
//
//   dW[n][k] += sum_m Y[m][n] * X[m][k]          (Y = grad of the layer output, X = layer input)
//   db[n]    += sum_m Y[m][n]
//
// Both operands are stored with the reduction index m as the slow (row) dimension, which is the
// transpose of what the MFMA A/B fragments want (8 consecutive k per lane).  Row tiles of 64 m are
// staged into LDS as they lie in memory (coalesced 16-B pieces) and the fragments are fetched with
// the hardware transposing read ds_read_b64_tr_b16.  A workgroup (4 waves, 2x2) owns a
// (64*TA) x (64*TB) tile of dW and a contiguous slice of m; partial sums are added to the fp32
// staging gradient (packed [NP][KP] layout) with float atomics (contiguous 64-B runs per 16 lanes).
// The bias gradient comes for free from one extra MFMA column against an all-ones B fragment.
#include "wgrad.h"

namespace {

template <int TA, int TB, bool CONV>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradMulti mp) {
  // several independent problems (same M) share one launch: blockIdx.x -> (problem, tile)
  // 1-D grid, logical order = (m-split, tile) with the tiles of one m-split contiguous on one XCD: they re-read the
  // same X / Y rows, which then come from that XCD's L2
  const int ntiles = mp.tile_begin[mp.nprob];
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / ntiles, btile = logical - bsplit * ntiles;
  int pi = 0;
  while (pi + 1 < mp.nprob && btile >= mp.tile_begin[pi + 1]) ++pi;
  const WgradParams& p = mp.p[pi];
  constexpr int TN = 64 * TA, TK = 64 * TB;
  constexpr int SY = TN + 16, SX = TK + 16;        // LDS row strides: 32 B x odd -> the 8 rows a half-wave
                                                   // tr-reads (below) fall in 8 disjoint bank windows: conflict-free
  constexpr int PY = (64 * TN / 8) / 256;          // 16-B pieces per thread
  constexpr int PX = (64 * TK / 8) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ys = reinterpret_cast<bf16_t*>(smem);    // [2][64][SY]
  bf16_t* Xs = Ys + 2 * 64 * SY;                   // [2][64][SX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int ntn = p.N / TN, ntk = p.K / TK;
  int bx = btile - mp.tile_begin[pi];
  const int tn = bx % ntn; bx /= ntn;
  const int tk = bx % ntk; bx /= ntk;
  const int tap = bx;                               // 0 unless CONV
  const int n0 = tn * TN, k0 = tk * TK;
  const int dy = CONV ? tap / 3 - 1 : 0, dx = CONV ? tap % 3 - 1 : 0;
  const int m_begin = bsplit * mp.m_per;
  const int m_end = min(p.M, m_begin + mp.m_per);
  const int nchunk = (m_end - m_begin + 63) / 64;
  if (nchunk <= 0) return;

  uint4 ry[PY], rx[PX];
  auto load_stage = [&](int ch) {
    const int mb = m_begin + ch * 64;
#pragma unroll
    for (int t = 0; t < PY; ++t) {
      const int pid = tid + 256 * t;
      const int row = pid / (TN / 8), c8 = pid % (TN / 8);
      const int m = mb + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (CONV && p.r > 1) {
          const int hw = p.H * p.W;
          const int b = m / hw, rem = m - b * hw;
          const int y = rem / p.W, x = rem - y * p.W;
          const int nn = n0 + c8 * 8;
          const int ij = nn / p.Cs, c = nn - ij * p.Cs;
          const int si = ij / p.r, sj = ij - si * p.r;
          v = *reinterpret_cast<const uint4*>(
              p.Y + (((long long)(b * p.H * p.r + y * p.r + si)) * (p.W * p.r) + x * p.r + sj) * p.Cs + c);
        } else {
          v = *reinterpret_cast<const uint4*>(p.Y + (long long)m * p.ldy + n0 + c8 * 8);
        }
      }
      ry[t] = v;
    }
#pragma unroll
    for (int t = 0; t < PX; ++t) {
      const int pid = tid + 256 * t;
      const int row = pid / (TK / 8), c8 = pid % (TK / 8);
      const int m = mb + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (CONV) {
          const int hw = p.H * p.W;
          const int b = m / hw, rem = m - b * hw;
          const int y = rem / p.W + dy, x = rem % p.W + dx;
          if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
            v = *reinterpret_cast<const uint4*>(p.X + ((long long)(b * p.H + y) * p.W + x) * p.ldx + k0 + c8 * 8);
        } else {
          v = *reinterpret_cast<const uint4*>(p.X + (long long)m * p.ldx + k0 + c8 * 8);
        }
      }
      rx[t] = v;
    }
  };
  auto store_stage = [&](int buf) {
    bf16_t* ys = Ys + buf * 64 * SY;
    bf16_t* xs = Xs + buf * 64 * SX;
#pragma unroll
    for (int t = 0; t < PY; ++t) {
      const int pid = tid + 256 * t;
      *reinterpret_cast<uint4*>(ys + (pid / (TN / 8)) * SY + (pid % (TN / 8)) * 8) = ry[t];
    }
#pragma unroll
    for (int t = 0; t < PX; ++t) {
      const int pid = tid + 256 * t;
      *reinterpret_cast<uint4*>(xs + (pid / (TK / 8)) * SX + (pid % (TK / 8)) * 8) = rx[t];
    }
  };

  constexpr int NTW = 2 * TA, KTW = 2 * TB;       // 16-tiles per wave
  f32x4_t acc[NTW][KTW], accb[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.db != nullptr && tk == 0 && wk == 0 && tap == 0;
  const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) load_stage(ch + 1);
    const bf16_t* ys = Ys + buf * 64 * SY;
    const bf16_t* xs = Xs + buf * 64 * SX;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[NTW], xf[KTW];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int c0 = wn * (32 * TA) + 16 * i;
        // k order inside the fragment: jj<4 -> m = 32ks + 4g + jj, jj>=4 -> m = 32ks + 16 + 4g + jj-4 (same for Y and X,
        // so the product is unchanged); a 32-lane half then reads 8 CONSECUTIVE rows per instruction
        const bf16x4_t lo = lds_tr_read(tr_addr(ys, SY, 32 * ks + 4 * g, c0, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(ys, SY, 32 * ks + 16 + 4 * g, c0, lane));
        yf[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < KTW; ++j) {
        const int c0 = wk * (32 * TB) + 16 * j;
        const bf16x4_t lo = lds_tr_read(tr_addr(xs, SX, 32 * ks + 4 * g, c0, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr(xs, SX, 32 * ks + 16 + 4 * g, c0, lane));
        xf[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, accb[i], 0, 0, 0);
      }
    }
    if (ch + 1 < nchunk) store_stage(buf ^ 1);
    __syncthreads();
  }

  // acc[i][j][e] = dW[n = n0 + wn*32TA + 16i + 4g + e][k = k0 + wk*32TB + 16j + r16]
  const long long koff = CONV ? (long long)tap * p.K : 0;
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int n = n0 + wn * (32 * TA) + 16 * i + 4 * g;
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int k = k0 + wk * (32 * TB) + 16 * j + r16;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + koff + k, acc[i][j][e]);
    }
    if (do_bias && r16 == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.db + n + e, accb[i][e]);
    }
  }
}

// ---- streaming variant of the 192 x 192 linear tile -------------------------------------------------------------
// Same tile, same MFMA schedule, but the Y / X chunks arrive by LDS-DMA into a ring that fills the LDS (144 KB): 6 stages
// of 32 rows (default; up to 120 KB per CU in flight while one stage is consumed) or 3 stages of 64 rows, instead of
// the single register-staged chunk of wgrad_kernel, whose load latency (~2 us under load) exceeded the 0.5 us of MFMA
// work per chunk.  No wave stores inside the loop, so every wave's vmcnt sees only its own DMAs and the counted wait
// is exact; one raw barrier per chunk.  A DMA writes lane-linear, which rules out the padded rows: rows are 384 B and
// the 32-byte column pairs are XOR-swizzled with (row >> 1) & 3 on the SOURCE address -- the 8 consecutive rows a
// half-wave transposing read touches then fall in 8 different 32-byte bank windows (row stride 384 B = 1.5 x 256 B).
// The operands are read once per launch, so the DMAs carry the nt (streaming) cache policy: inside the train step
// that is worth 35 us per launch (the reads no longer evict what the neighbouring kernels hand to each other through
// the L2 / Infinity Cache); on its own the loader streams at 6.0 TB/s default and 6.8 TB/s nt (tools/read_bw.hip).
// The split partials go to scratch slabs in accumulator order (1-KiB store instructions) and wgrad_reduce_kernel adds
// their sum to dW: 9.4 M fp32 atomics per launch in 4 x 64-B segments cost ~27 us more than the slab stores, and the
// fixed summation order makes dW reproducible run to run.
constexpr int WS_LDS_BYTES = 147456;              // the whole ring: 3 stages of 64 rows or 6 stages of 32 rows

__device__ __forceinline__ const bf16_t* tr_addr_swz(const bf16_t* tile, int rbase, int c0, int lane) {
  const int ll = lane & 15;
  const int row = rbase + (ll >> 2);
  return tile + row * 192 + (((c0 >> 4) ^ ((row >> 1) & 3)) << 4) + ((ll & 3) << 2);
}

// W8: eight waves (two per SIMD) share the 192 x 192 tile as 4 x 2 blocks of 48 x 96 instead of four waves with 96 x 96 each:
// with one wave per SIMD nothing covers that wave's own LDS reads, DMA issue and waits (MFMA pipe 32 % busy, 40 % issue stalls);
// waves 0..3 still issue the whole DMA ring.
template <int ROWS, bool NT, bool W8>
__global__ __launch_bounds__(W8 ? 512 : 256) void wgrad_stream_kernel(const WgradMulti mp) {
  constexpr int FI = W8 ? 3 : 6;                       // 16-row n-fragments per wave
  constexpr int STAGE_ELEMS = 2 * ROWS * 192;          // Y chunk + X chunk, bf16 elements
  constexpr int RING = WS_LDS_BYTES / (STAGE_ELEMS * 2);
  constexpr int NI = ROWS * 24 / 256;                  // DMA instructions per wave, chunk and operand (6 or 3)
  constexpr int AHEAD = RING - 1;                      // chunks issued ahead of the one being consumed
  const int ntiles = mp.tile_begin[mp.nprob];
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / ntiles, btile = logical - bsplit * ntiles;
  int pi = 0;
  while (pi + 1 < mp.nprob && btile >= mp.tile_begin[pi + 1]) ++pi;
  const WgradParams& p = mp.p[pi];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* ring = reinterpret_cast<bf16_t*>(smem);
  const unsigned ring_base = (unsigned)(size_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;             // n block (16 FI rows), k half (96 columns)
  const bool loader = wave < 4;                        // the DMA ring is issued by four waves in either shape
  const int ntn = p.N / 192;
  const int bx = btile - mp.tile_begin[pi];
  const int tn = bx % ntn, tk = bx / ntn;
  const int n0 = tn * 192, k0 = tk * 192;
  const int m_begin = bsplit * mp.m_per;
  const int m_end = min(p.M, m_begin + mp.m_per);
  const int nchunk = (m_end - m_begin) / ROWS;
  if (nchunk <= 0) return;

  // this wave's quarter of a stage image: ROWS * 6 16-byte pieces of Y and of X (NI + NI DMA instructions).  Which
  // piece a lane moves in instruction i never changes: its element offset relative to the chunk's first row is computed
  // once (with one wave per SIMD the per-chunk address arithmetic otherwise competes with the MFMAs for issue slots).
  int yoff[NI], xoff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = (wave & 3) * (NI * 64) + i * 64 + lane;
    const int row = q / 24, pos = q - row * 24;
    const int c = (((pos >> 1) ^ ((row >> 1) & 3)) << 1) | (pos & 1);
    yoff[i] = row * p.ldy + n0 + c * 8;
    xoff[i] = row * p.ldx + k0 + c * 8;
  }
  auto issue = [&](int ch) {
    if (!loader) return;
    const int m0 = m_begin + ch * ROWS;
    const bf16_t* yb = p.Y + (long long)m0 * p.ldy;
    const bf16_t* xb = p.X + (long long)m0 * p.ldx;
    const unsigned dst = ring_base + (unsigned)((ch % RING) * STAGE_ELEMS * 2);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      srk_glds16<NT>(yb + yoff[i], __builtin_amdgcn_readfirstlane(dst + (wave * (NI * 64) + i * 64) * 16));
#pragma unroll
    for (int i = 0; i < NI; ++i)
      srk_glds16<NT>(xb + xoff[i], __builtin_amdgcn_readfirstlane(dst + ROWS * 192 * 2 + (wave * (NI * 64) + i * 64) * 16));
  };

  f32x4_t acc[FI][6], accb[FI];
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.db != nullptr && tk == 0 && wk == 0;
  const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  for (int c = 0; c < AHEAD && c < nchunk; ++c) issue(c);
  for (int ch = 0; ch < nchunk; ++ch) {
    // chunk ch has landed once at most the AHEAD - 1 chunks issued after it are outstanding
    if (ch + AHEAD - 1 < nchunk) {
      srk_wait_vmcnt<2 * NI * (AHEAD - 1)>();
    } else {
      srk_wait_vmcnt<0>();
    }
    srk_lds_barrier();                                                       // ... for every wave; MFMA(ch-1) done everywhere
    if (ch + AHEAD < nchunk) issue(ch + AHEAD);
    const bf16_t* ys = ring + (ch % RING) * STAGE_ELEMS;
    const bf16_t* xs = ys + ROWS * 192;
#pragma unroll
    for (int ks = 0; ks < ROWS / 32; ++ks) {
      bf16x8_t yf[FI], xf[6];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int c0 = wn * (16 * FI) + 16 * i;
        const bf16x4_t lo = lds_tr_read(tr_addr_swz(ys, 32 * ks + 4 * g, c0, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr_swz(ys, 32 * ks + 16 + 4 * g, c0, lane));
        yf[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c0 = wk * 96 + 16 * j;
        const bf16x4_t lo = lds_tr_read(tr_addr_swz(xs, 32 * ks + 4 * g, c0, lane));
        const bf16x4_t hi = lds_tr_read(tr_addr_swz(xs, 32 * ks + 16 + 4 * g, c0, lane));
        xf[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < FI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, accb[i], 0, 0, 0);
      }
    }
  }

  if (mp.partial != nullptr) {
    // partial tile in accumulator order ([wave][i][j][lane] x 4 floats: every store instruction writes 1 KiB
    // contiguous); wgrad_reduce_kernel sums the splits.  Float atomics of this shape (4 x 64-B segments per
    // instruction, 9.4 M of them per launch) cost ~60 us per launch.
    f32x4_t* slab = reinterpret_cast<f32x4_t*>(mp.partial) + (((size_t)btile * 144 + wave * (FI * 6)) * mp.nsplit + bsplit) * 64 + lane;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) slab[(size_t)(i * 6 + j) * mp.nsplit * 64] = acc[i][j];
  }
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int n = n0 + wn * (16 * FI) + 16 * i + 4 * g;
    if (mp.partial == nullptr) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int k = k0 + wk * 96 + 16 * j + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + k, acc[i][j][e]);
      }
    }
    if (do_bias && r16 == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.db + n + e, accb[i][e]);
    }
  }
}

// Sum of the nsplit partial tiles of wgrad_stream_kernel, added to dW.  One thread per accumulator vector
// (9216 per tile), 36 workgroups per tile; the partials are read as 1-KiB wave rows, 8 splits in flight.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradMulti mp) {
  if ((int)blockIdx.x >= 36 * mp.tile_begin[mp.nprob]) {        // riding job: 64 nH workgroups after the tile workgroups
    __shared__ float part[4][64];
    const int b = blockIdx.x - 36 * mp.tile_begin[mp.nprob];
    rpb_reduce_block(mp.rpb.slab, mp.rpb.dtable, mp.rpb.nslab, mp.rpb.nH, b & 63, b >> 6, part);
    return;
  }
  const int btile = blockIdx.x / 36;
  const int q = (blockIdx.x - btile * 36) * 256 + threadIdx.x;
  int pi = 0;
  while (pi + 1 < mp.nprob && btile >= mp.tile_begin[pi + 1]) ++pi;
  const WgradParams& p = mp.p[pi];
  const int ntn = p.N / 192;
  const int bx = btile - mp.tile_begin[pi];
  const int tn = bx % ntn, tk = bx / ntn;
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(mp.partial) + ((size_t)btile * 144 + (q >> 6)) * mp.nsplit * 64 + (q & 63);
  f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= mp.nsplit; s += 8) {
    f32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(s + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; s < mp.nsplit; ++s) sum += src[s * 64];
  const int lane = q & 63, t = q >> 6;
  const int fi = mp.w8 ? 3 : 6;                                  // n-fragments per wave of the kernel that wrote the slabs
  const int j = t % 6, i = (t / 6) % fi, wave = t / (6 * fi);
  const int wn = wave >> 1, wk = wave & 1, r16 = lane & 15, g = lane >> 4;
  const int n = tn * 192 + wn * (16 * fi) + 16 * i + 4 * g;
  const int k = tk * 192 + wk * 96 + 16 * j + r16;
#pragma unroll
  for (int e = 0; e < 4; ++e) atomicAdd(p.dW + (long long)(n + e) * p.ldw + k, sum[e]);
}

thread_local const RpbJob* t_rpb = nullptr;     // job offered by srk_launch_wgrad_multi_rpb to the next reduce launch of this thread
thread_local bool t_rpb_done = false;

SrkOpt g_wgrad_stream{OPT_WGRAD_STREAM, 1};
SrkOpt g_wgrad_rows{OPT_WGRAD_ROWS, 32};      // rows per ring stage of the streaming kernel: 32 (6 stages) or 64 (3 stages)
SrkOpt g_wgrad_nt{OPT_WGRAD_NT, 1};         // nt (streaming) cache policy on its operand DMAs
#ifndef SRK_WGRAD_W8_DEFAULT
#define SRK_WGRAD_W8_DEFAULT 1
#endif
SrkOpt g_wgrad_w8{OPT_WGRAD_W8, SRK_WGRAD_W8_DEFAULT};   // eight waves per workgroup (two per SIMD) in the streaming kernel
SrkOpt g_wgrad_partials{OPT_WGRAD_PARTIALS, 1};   // split partials to scratch slabs + reduce kernel (1) or fp32 atomics straight into dW (0)

template <int TA, int TB, bool CONV>
int launch(const WgradParams* ps, int nprob, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 64 * ((64 * TA + 16) + (64 * TB + 16)) * sizeof(bf16_t);
  static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TA, TB, CONV>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      srk_set_error("wgrad: cannot reserve %zu bytes of LDS", lds);
      return SRK_E_LAUNCH;
    }
    configured = true;
  }
  WgradMulti mp;
  mp.nprob = nprob;
  mp.partial = nullptr;
  mp.nsplit = 0;
  mp.rpb = RpbJob{nullptr, nullptr, 0, 0};
  mp.w8 = 0;
  int tiles = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < nprob; ++i) {
    mp.p[i] = ps[i];
    mp.tile_begin[i] = tiles;
    tiles += (ps[i].N / (64 * TA)) * (ps[i].K / (64 * TB)) * (CONV ? 9 : 1);
    flops += ps[i].flops;
    bytes += ps[i].bytes;
  }
  mp.tile_begin[nprob] = tiles;
  // one workgroup per CU (LDS-limited): aim at ~256 workgroups; every split costs one pass of fp32 atomics over dW
  const int M = ps[0].M;
  int splits = 256 / tiles;
  if (splits < 1) splits = 1;
  int m_per = round_up(cdiv(M, splits), 64);
  if (m_per < 256) m_per = 256;
  mp.m_per = m_per;
  splits = cdiv(M, m_per);
  const int fam = CONV ? FAM_WGRAD_CONV : FAM_WGRAD_LINEAR;
  if constexpr (TA == 3 && TB == 3 && !CONV) {
    bool ok = g_wgrad_stream && M % 64 == 0;
    for (int i = 0; i < nprob; ++i) ok = ok && ps[i].ldy % 8 == 0 && ps[i].ldx % 8 == 0;
    if (ok) {
      constexpr int slds = WS_LDS_BYTES;
      using KernelFn = void (*)(const WgradMulti);
      static const KernelFn fns[8] = {&wgrad_stream_kernel<64, false, false>, &wgrad_stream_kernel<64, true, false>,
                                      &wgrad_stream_kernel<32, false, false>, &wgrad_stream_kernel<32, true, false>,
                                      &wgrad_stream_kernel<64, false, true>,  &wgrad_stream_kernel<64, true, true>,
                                      &wgrad_stream_kernel<32, false, true>,  &wgrad_stream_kernel<32, true, true>};
      static SrkPerDevice<bool> sconf_pd; bool& sconf = sconf_pd.here();
      if (!sconf) {
        for (KernelFn f : fns)
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, slds) != hipSuccess) {
            srk_set_error("wgrad(stream): cannot reserve %d bytes of LDS", slds);
            return SRK_E_LAUNCH;
          }
        sconf = true;
      }
      const KernelFn fn = fns[(g_wgrad_w8 ? 4 : 0) + (g_wgrad_rows == 32 ? 2 : 0) + (g_wgrad_nt ? 1 : 0)];
      mp.nsplit = splits;
      mp.w8 = g_wgrad_w8;
      mp.partial = g_wgrad_partials && splits > 1 ? srk_wgrad_scratch(stream, (size_t)tiles * splits * WS_SLAB_VEC * 16) : nullptr;
      srk_probe_pre(fam, stream, flops, bytes);
      hipLaunchKernelGGL(fn, dim3(tiles * splits), dim3(g_wgrad_w8 ? 512 : 256), slds, stream, mp);
      if (mp.partial) {
        int extra = 0;
        if (t_rpb && t_rpb->slab) {
          mp.rpb = *t_rpb;
          extra = 64 * t_rpb->nH;
          t_rpb_done = true;
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(tiles * 36 + extra), dim3(256), 0, stream, mp);
      }
      srk_probe_post(fam, stream);
      return srk_check_launch("wgrad(stream)");
    }
  }
  srk_probe_pre(fam, stream, flops, bytes);
  hipLaunchKernelGGL((wgrad_kernel<TA, TB, CONV>), dim3(tiles * splits), dim3(256), lds, stream, mp);
  srk_probe_post(fam, stream);
  return srk_check_launch("wgrad");
}

inline void tile_class(const WgradParams& p, int& a, int& b) {
  a = p.N % 192 == 0 ? 3 : (p.N % 128 == 0 ? 2 : 1);
  b = p.K % 192 == 0 ? 3 : (p.K % 128 == 0 ? 2 : 1);
}

template <bool CONV>
int dispatch(const WgradParams* ps, int nprob, hipStream_t stream) {
  int a, b;
  tile_class(ps[0], a, b);
#define WCASE(A, B) \
  if (a == A && b == B) return launch<A, B, CONV>(ps, nprob, stream);
  WCASE(3, 3) WCASE(3, 1) WCASE(1, 3) WCASE(2, 2) WCASE(2, 1) WCASE(1, 2) WCASE(1, 1) WCASE(3, 2) WCASE(2, 3)
#undef WCASE
  srk_set_error("wgrad: no tile for N=%d K=%d", ps[0].N, ps[0].K);
  return SRK_E_UNSUPPORTED;
}

int validate(const WgradParams& p) {
  SRK_REQUIRE(p.M > 0 && p.N % 64 == 0 && p.K % 64 == 0, SRK_E_SHAPE, "wgrad: bad M/N/K %d/%d/%d", p.M, p.N, p.K);
  SRK_REQUIRE(p.Y && p.X && p.dW, SRK_E_NULL, "wgrad: null operand");
  if (p.conv) {
    SRK_REQUIRE(p.M == p.B * p.H * p.W, SRK_E_SHAPE, "wgrad(conv): M != B*H*W");
    if (p.r > 1) SRK_REQUIRE(p.Cs % 8 == 0 && p.N == p.r * p.r * p.Cs, SRK_E_SHAPE, "wgrad(conv,ps): bad Cs");
  }
  return SRK_OK;
}

}  // namespace

void srk_wgrad_stream_enable(int on) { g_wgrad_stream = on ? 1 : 0; }
void srk_wgrad_partials_enable(int on) { g_wgrad_partials = on ? 1 : 0; }
void srk_wgrad_w8_enable(int on) { g_wgrad_w8 = on ? 1 : 0; }
void srk_wgrad_stream_tune(int rows, int nt) {
  if (rows == 32 || rows == 64) g_wgrad_rows = rows;
  if (nt >= 0) g_wgrad_nt = nt ? 1 : 0;
}
int srk_wgrad_partials_enabled() { return g_wgrad_partials; }
int srk_wgrad_stream_enabled() { return g_wgrad_stream; }
int srk_wgrad_w8_enabled() { return g_wgrad_w8; }
void srk_wgrad_stream_tune_get(int* rows, int* nt) { *rows = g_wgrad_rows; *nt = g_wgrad_nt; }

// Workspace for the split partials of the streaming weight-gradient kernels.  The caller owns it (kernels never
// allocate): the model executor binds a region of its arena around the backward pass, the stand-alone entry points use
// what srk_set_wgrad_workspace() registered for the calling thread.  Without a (large enough) workspace the launchers
// fall back to fp32 atomics straight into dW.
namespace {
thread_local float* t_wgrad_ws = nullptr;
thread_local size_t t_wgrad_ws_bytes = 0;
}  // namespace
void srk_wgrad_bind_workspace(void* ptr, size_t bytes, void** prev_ptr, size_t* prev_bytes) {
  if (prev_ptr) *prev_ptr = t_wgrad_ws;
  if (prev_bytes) *prev_bytes = t_wgrad_ws_bytes;
  t_wgrad_ws = static_cast<float*>(ptr);
  t_wgrad_ws_bytes = ptr ? bytes : 0;
}
float* srk_wgrad_scratch(hipStream_t, size_t bytes) { return bytes <= t_wgrad_ws_bytes ? t_wgrad_ws : nullptr; }

int srk_launch_wgrad(const WgradParams& p, hipStream_t stream) {
  int rc = validate(p);
  if (rc) return rc;
  if (p.conv) {
    rc = srk_launch_conv_wgrad_taps(p, stream);
    if (rc != SRK_WGRAD_NOT_COVERED) return rc;
  }
  return p.conv ? dispatch<true>(&p, 1, stream) : dispatch<false>(&p, 1, stream);
}

// Up to 4 linear problems with the same M.  Problems of the same tile class go out as one launch (so that the
// launch fills the chip with few m-splits, i.e. few atomic passes); the rest are launched one by one.
int srk_launch_wgrad_multi_rpb(const WgradParams* ps, int nprob, const RpbJob* rpb, hipStream_t stream) {
  t_rpb = rpb;
  t_rpb_done = false;
  int rc = srk_launch_wgrad_multi(ps, nprob, stream);
  t_rpb = nullptr;
  if (rc == SRK_OK && rpb && rpb->slab && !t_rpb_done) rc = srk_launch_rpb_reduce(rpb->slab, rpb->dtable, rpb->nslab, rpb->nH, stream);
  return rc;
}

int srk_launch_wgrad_multi(const WgradParams* ps, int nprob, hipStream_t stream) {
  SRK_REQUIRE(nprob >= 1 && nprob <= 4, SRK_E_SHAPE, "wgrad_multi: nprob=%d", nprob);
  bool same = true;
  int a0, b0;
  tile_class(ps[0], a0, b0);
  for (int i = 0; i < nprob; ++i) {
    int rc = validate(ps[i]);
    if (rc) return rc;
    int a, b;
    tile_class(ps[i], a, b);
    same = same && a == a0 && b == b0 && ps[i].M == ps[0].M && !ps[i].conv;
  }
  if (same) return dispatch<false>(ps, nprob, stream);
  for (int i = 0; i < nprob; ++i) {
    int rc = srk_launch_wgrad(ps[i], stream);
    if (rc) return rc;
  }
  return SRK_OK;
}
