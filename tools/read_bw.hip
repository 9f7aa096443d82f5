// Developer probe: pure-read HBM streaming ceilings on gfx950 (register loads and LDS-DMA, default and nt policy).
//   hipcc --offload-arch=gfx950 -O3 -o tools/read_bw tools/read_bw.hip && tools/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// U independent 16-byte loads per lane and iteration. CONTIG: each workgroup sweeps its own contiguous range;
// otherwise the whole grid sweeps the buffer front to back.
template <bool NT, int U, bool CONTIG>
__global__ void read_regs(const f4* __restrict__ src, size_t n16, float* sink) {
  const size_t per_it = (size_t)blockDim.x * U;
  const size_t iters = n16 / (per_it * gridDim.x);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t it = 0; it < iters; ++it) {
    const size_t base = CONTIG ? ((size_t)blockIdx.x * iters + it) * per_it : (it * gridDim.x + blockIdx.x) * per_it;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f4* p = src + base + (size_t)u * blockDim.x + threadIdx.x;
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;
}

template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// every wave streams 1-KiB pieces into its own LDS ring of D pieces and keeps D-1 in flight; nobody reads the LDS.
template <bool NT, int D, bool CONTIG>
__global__ void read_dma(const char* __restrict__ src, size_t bytes) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const size_t piece = 1024;
  const size_t per_it = piece * nw;
  const size_t iters = bytes / (per_it * gridDim.x);
  const unsigned ring = (unsigned)(size_t)lds + wave * D * 1024;
  for (size_t it = 0; it < iters; ++it) {
    const size_t base = CONTIG ? ((size_t)blockIdx.x * iters + it) * per_it : (it * gridDim.x + blockIdx.x) * per_it;
    glds16<NT>(src + base + wave * piece + lane * 16, __builtin_amdgcn_readfirstlane(ring + (unsigned)(it % D) * 1024));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, slot = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// The loader of wgrad_stream_kernel on its own: 4 waves, 64-row chunks of a 192-column Y tile and a 192-column X tile
// (12 DMA instructions per wave and chunk), 3-slot ring, chunk c+2 issued behind the barrier that publishes chunk c.
// swz: the 32-byte pair XOR on the source address; barrier: workgroup barrier per chunk (else each wave runs free).
template <bool NT>
__global__ __launch_bounds__(256) void read_wgrad_like(const char* __restrict__ Y, const char* __restrict__ X, int ldy, int ldx,
                                                        int ntn, int ntk, int m_per, int swz, int barrier, int remap) {
  extern __shared__ char lds[];
  const unsigned ring_base = (unsigned)(size_t)lds;
  const int ntiles = ntn * ntk;
  const int logical = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bsplit = logical / ntiles, bx = logical - bsplit * ntiles;
  const int tn = bx % ntn, tk = bx / ntn;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int yoff[6], xoff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave * 384 + i * 64 + lane;
    const int row = q / 24, pos = q - row * 24;
    const int c = swz ? ((((pos >> 1) ^ ((row >> 1) & 3)) << 1) | (pos & 1)) : pos;
    yoff[i] = (row * ldy + tn * 192 + c * 8) * 2;
    xoff[i] = (row * ldx + tk * 192 + c * 8) * 2;
  }
  const int nchunk = m_per / 64;
  auto issue = [&](int ch) {
    const size_t m0 = (size_t)bsplit * m_per + (size_t)ch * 64;
    const char* yb = Y + m0 * ldy * 2;
    const char* xb = X + m0 * ldx * 2;
    const unsigned dst = ring_base + (unsigned)((ch % 3) * 49152);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16<NT>(yb + yoff[i], __builtin_amdgcn_readfirstlane(dst + (wave * 384 + i * 64) * 16));
#pragma unroll
    for (int i = 0; i < 6; ++i)
      glds16<NT>(xb + xoff[i], __builtin_amdgcn_readfirstlane(dst + 24576 + (wave * 384 + i * 64) * 16));
  };
  issue(0);
  if (nchunk > 1) issue(1);
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk)
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ch + 2 < nchunk) issue(ch + 2);
  }
}


struct MixProb { const char* Y; const char* X; int ldy, ldx, ntn, tile_begin; };
struct MixParams { MixProb p[4]; int nprob, ntiles, m_per; };

// the loader of wgrad_stream_kernel with its real per-block problem mix (qkv 3 tiles, proj 1, fc1 2, fc2 2; 32 row splits)
template <bool NT>
__global__ __launch_bounds__(256) void read_wgrad_mix(const MixParams mp) {
  extern __shared__ char lds[];
  const unsigned ring_base = (unsigned)(size_t)lds;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bsplit = logical / mp.ntiles, btile = logical - bsplit * mp.ntiles;
  int pi = 0;
  while (pi + 1 < mp.nprob && btile >= mp.p[pi + 1].tile_begin) ++pi;
  const MixProb& p = mp.p[pi];
  const int bx = btile - p.tile_begin;
  const int tn = bx % p.ntn, tk = bx / p.ntn;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int yoff[6], xoff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave * 384 + i * 64 + lane;
    const int row = q / 24, pos = q - row * 24;
    const int c = (((pos >> 1) ^ ((row >> 1) & 3)) << 1) | (pos & 1);
    yoff[i] = (row * p.ldy + tn * 192 + c * 8) * 2;
    xoff[i] = (row * p.ldx + tk * 192 + c * 8) * 2;
  }
  const int nchunk = mp.m_per / 64;
  auto issue = [&](int ch) {
    const size_t m0 = (size_t)bsplit * mp.m_per + (size_t)ch * 64;
    const char* yb = p.Y + m0 * p.ldy * 2;
    const char* xb = p.X + m0 * p.ldx * 2;
    const unsigned dst = ring_base + (unsigned)((ch % 3) * 49152);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16<NT>(yb + yoff[i], __builtin_amdgcn_readfirstlane(dst + (wave * 384 + i * 64) * 16));
#pragma unroll
    for (int i = 0; i < 6; ++i)
      glds16<NT>(xb + xoff[i], __builtin_amdgcn_readfirstlane(dst + 24576 + (wave * 384 + i * 64) * 16));
  };
  issue(0);
  if (nchunk > 1) issue(1);
  for (int ch = 0; ch < nchunk; ++ch) {
    if (ch + 1 < nchunk)
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ch + 2 < nchunk) issue(ch + 2);
  }
}

__global__ void fill_random(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (x & 0x7fff7fffu) | 0x30003000u;      // two finite bf16 values of mixed magnitude
  }
}

template <typename F>
static double time_us(F launch, int reps = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3 / reps;
}

template <bool NT, int U, bool CONTIG>
static void run_regs(const void* buf, size_t bytes, float* sink, int grid, int block) {
  size_t n16 = bytes / 16;
  size_t per = (size_t)block * U * grid;
  size_t used = n16 / per * per * 16;
  double us = time_us([&] { read_regs<NT, U, CONTIG><<<grid, block>>>((const f4*)buf, n16, sink); });
  printf("regs  %-3s U=%-2d %-7s grid %5d x %4d : %8.1f us  %5.2f TB/s\n", NT ? "nt" : "def", U, CONTIG ? "contig" : "sweep", grid,
         block, us, used / us * 1e-6);
  fflush(stdout);
}

template <bool NT, int D, bool CONTIG>
static void run_dma(const void* buf, size_t bytes, int grid, int block) {
  size_t per = (size_t)(block / 64) * 1024 * grid;
  size_t used = bytes / per * per;
  size_t lds = (size_t)(block / 64) * D * 1024;
  CK(hipFuncSetAttribute((const void*)read_dma<NT, D, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double us = time_us([&] { read_dma<NT, D, CONTIG><<<grid, block, lds>>>((const char*)buf, bytes); });
  printf("dma   %-3s D=%-2d %-7s grid %5d x %4d (%3zu KiB in flight/WG) : %8.1f us  %5.2f TB/s\n", NT ? "nt" : "def", D,
         CONTIG ? "contig" : "sweep", grid, block, lds / 1024, us, used / us * 1e-6);
  fflush(stdout);
}

template <bool NT>
static void run_wgrad_like(const void* buf, size_t bytes, int ldy, int ldx, int swz, int barrier, int remap) {
  const int ntn = ldy / 192, ntk = ldx / 192, ntiles = ntn * ntk;
  const int splits = 256 / ntiles, grid = splits * ntiles;
  // rows per split so that Y and X (laid back to back) fill the buffer
  size_t rows = bytes / ((size_t)(ldy + ldx) * 2);
  int m_per = (int)(rows / splits) / 64 * 64;
  if (m_per > 8192) m_per = 8192;
  const size_t M = (size_t)m_per * splits;
  const char* Y = (const char*)buf;
  const char* X = Y + M * ldy * 2;
  const size_t hbm = M * (size_t)(ldy + ldx) * 2, req = (size_t)grid * m_per * 768;
  CK(hipFuncSetAttribute((const void*)read_wgrad_like<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
  double us = time_us([&] { read_wgrad_like<NT><<<grid, 256, 3 * 49152>>>(Y, X, ldy, ldx, ntn, ntk, m_per, swz, barrier, remap); });
  printf("wgrad-like %-3s ldy %4d ldx %4d tiles %d x %d m_per %5d swz %d barrier %d remap %d : %8.1f us  unique %6.1f MB %5.2f TB/s  requested %6.1f MB %5.2f TB/s\n",
         NT ? "nt" : "def", ldy, ldx, ntn, ntk, m_per, swz, barrier, remap, us, hbm * 1e-6, hbm / us * 1e-6, req * 1e-6, req / us * 1e-6);
  fflush(stdout);
}

template <bool NT>
static void run_wgrad_mix(const void* buf, size_t bytes, int copies) {
  const size_t M = 131072;
  const int lds_[4][2] = {{576, 192}, {192, 192}, {384, 192}, {192, 384}};
  const int ntn_[4] = {3, 1, 2, 1}, nt_[4] = {3, 1, 2, 2};
  size_t per_copy = 0;
  for (int i = 0; i < 4; ++i) per_copy += M * (size_t)(lds_[i][0] + lds_[i][1]) * 2;
  if (per_copy * copies > bytes) { printf("buffer too small\n"); return; }
  MixParams mp[8];
  for (int c = 0; c < copies; ++c) {
    const char* q = (const char*)buf + per_copy * c;
    int tb = 0;
    for (int i = 0; i < 4; ++i) {
      mp[c].p[i].Y = q; q += M * (size_t)lds_[i][0] * 2;
      mp[c].p[i].X = q; q += M * (size_t)lds_[i][1] * 2;
      mp[c].p[i].ldy = lds_[i][0]; mp[c].p[i].ldx = lds_[i][1]; mp[c].p[i].ntn = ntn_[i]; mp[c].p[i].tile_begin = tb;
      tb += nt_[i];
    }
    mp[c].nprob = 4; mp[c].ntiles = tb; mp[c].m_per = 4096;
  }
  CK(hipFuncSetAttribute((const void*)read_wgrad_mix<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
  int k = 0;
  double us = time_us([&] { read_wgrad_mix<NT><<<256, 256, 3 * 49152>>>(mp[k++ % copies]); }, 12);
  printf("wgrad-mix  %-3s real problem mix, M 131072, 32 splits x 8 tiles, rotating over %d copies : %8.1f us  unique %6.1f MB %5.2f TB/s  requested %6.1f MB\n",
         NT ? "nt" : "def", copies, us, per_copy * 1e-6, per_copy / us * 1e-6, 256 * 4096 * 768e-6);
  fflush(stdout);
}

// A streaming-GEMM-shaped pipeline without the math: 4 loader waves DMA `slot_bytes` per tile into an R-deep LDS ring
// (R - 1 tiles in flight, one barrier per tile), 4 consumer waves store `store_bytes` per tile and, with reg_bytes > 0,
// also load that many bytes per tile into registers one tile ahead (operands that bypass the LDS ring).
template <bool NT>
__global__ __launch_bounds__(512) void ring_like(const char* __restrict__ src, const char* __restrict__ src2, char* __restrict__ dst,
                                                 int slot_bytes, int R, int store_bytes, int reg_bytes, int ntiles_total) {
  extern __shared__ char lds[];
  const unsigned ring_base = (unsigned)(size_t)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, gi = blockIdx.x;
  const int nt = (ntiles_total - gi + G - 1) / G;
  const int ni = slot_bytes / 4096;                 // DMA instructions per loader wave per tile (4 waves x 1 KiB)
  if (wave >= 4) {
    const int lw = wave - 4;
    auto issue = [&](int t) {
      const char* base = src + (size_t)(gi + (size_t)t * G) * slot_bytes;
      const unsigned slot = ring_base + (unsigned)((t % R) * slot_bytes);
      for (int i = 0; i < ni; ++i)
        glds16<NT>(base + (size_t)(i * 4 + lw) * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(slot + (i * 4 + lw) * 1024));
    };
    for (int s = 0; s < R - 1 && s < nt; ++s) issue(s);
    for (int t = 0; t < nt; ++t) {
      // conservative: wait until only the tiles issued after t are outstanding (ni * (R - 2) instructions)
      const int keep = (t + R - 2 < nt) ? ni * (R - 2) : 0;
      if (keep >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else if (keep >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (keep >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (keep >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
      else if (keep >= 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
      else if (keep >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (keep >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if (keep >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (keep >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      else if (keep >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (keep >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (t + R - 1 < nt) issue(t + R - 1);
    }
  } else {
    const int nst = store_bytes / 4096, nrg = reg_bytes / 4096;   // per consumer wave: 1 KiB instructions per tile
    f4 pre[8];
    for (int i = 0; i < 8; ++i) pre[i] = f4{0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](int t) {
      const f4* base = (const f4*)(src2 + (size_t)(gi + (size_t)t * G) * reg_bytes);
      for (int i = 0; i < nrg && i < 8; ++i) pre[i] = NT ? __builtin_nontemporal_load(base + (i * 4 + wave) * 64 + lane) : base[(i * 4 + wave) * 64 + lane];
    };
    if (nrg) prefetch(0);
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      f4 cur[8];
      for (int i = 0; i < 8; ++i) cur[i] = pre[i];
      if (nrg && t + 1 < nt) prefetch(t + 1);
      f4* out = (f4*)(dst + (size_t)(gi + (size_t)t * G) * store_bytes);
      for (int i = 0; i < nst; ++i) out[(i * 4 + wave) * 64 + lane] = cur[i & 7];
    }
  }
}

template <bool NT>
static void run_ring_like(const void* buf, size_t bytes, int slot_bytes, int R, int store_bytes, int reg_bytes) {
  // buffer split: [DMA source | register source | store destination]
  const size_t per_tile = (size_t)slot_bytes + reg_bytes + store_bytes;
  int ntiles = (int)(bytes / per_tile);
  if (ntiles > 8192 * 2) ntiles = 8192 * 2;
  const char* src = (const char*)buf;
  const char* src2 = src + (size_t)ntiles * slot_bytes;
  char* dst = (char*)src2 + (size_t)ntiles * reg_bytes;
  const size_t lds = (size_t)R * slot_bytes;
  CK(hipFuncSetAttribute((const void*)ring_like<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double us = time_us([&] { ring_like<NT><<<256, 512, lds>>>(src, src2, dst, slot_bytes, R, store_bytes, reg_bytes, ntiles); });
  const double mb = (double)ntiles * per_tile * 1e-6;
  printf("ring-like %-3s slot %5d B x R %d (%3zu KiB LDS) + regs %5d B/tile, stores %5d B/tile, %5d tiles : %8.1f us  %7.1f MB  %5.2f TB/s\n",
         NT ? "nt" : "def", slot_bytes, R, lds / 1024, reg_bytes, store_bytes, ntiles, us, mb, mb / us);
  fflush(stdout);
}

int main(int argc, char** argv) {
  size_t mb = argc > 1 ? atoi(argv[1]) : 1536;
  size_t bytes = mb << 20;
  void* buf;
  float* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, bytes));
  const bool randfill = argc > 3;
  if (randfill) {
    fill_random<<<4096, 256>>>((unsigned*)buf, bytes / 4);
    CK(hipDeviceSynchronize());
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, buffer %zu MiB, %s data\n", prop.name, cus, mb, randfill ? "pseudo-random" : "all-zero");

  if (argc > 2 && argv[2][0] == 'r') {  // streaming-GEMM-shaped pipelines
    // LNBWD K=576, 16-row tiles: A 18 KB + x 12 KB + dx_in 12 KB per tile (44 KB slots -> R 3), 12 KB stored
    run_ring_like<true>(buf, bytes, 45056, 3, 12288, 0);
    run_ring_like<false>(buf, bytes, 45056, 3, 12288, 0);
    // ... with the fp32 row operands through registers: 20 KB slots -> R 7
    run_ring_like<true>(buf, bytes, 20480, 7, 12288, 24576);
    run_ring_like<true>(buf, bytes, 20480, 5, 12288, 24576);
    run_ring_like<true>(buf, bytes, 20480, 3, 12288, 24576);
    // PROJ_RES: A 6 KB + x 12 KB (20 KB slots, R 7), 12 KB stored
    run_ring_like<true>(buf, bytes, 20480, 7, 12288, 0);
    run_ring_like<true>(buf, bytes, 20480, 4, 12288, 0);
    // GELU: A 6 KB (8 KB slots), 12 + 12 KB stored per 16 rows... per 32-row tile: 12 KB in, 48 KB out
    run_ring_like<true>(buf, bytes, 12288, 8, 49152, 0);
    // pure read / pure write shapes
    run_ring_like<true>(buf, bytes, 45056, 3, 0, 0);
    run_ring_like<true>(buf, bytes, 20480, 7, 0, 0);
    run_ring_like<true>(buf, bytes, 4096, 3, 32768, 0);
    return 0;
  }
  if (argc > 2) {  // wgrad-like loader patterns only
    run_wgrad_mix<false>(buf, bytes, 1);
    run_wgrad_mix<true>(buf, bytes, 1);
    run_wgrad_mix<false>(buf, bytes, 2);
    run_wgrad_mix<true>(buf, bytes, 2);
    for (int nt = 0; nt < 2; ++nt) {
      auto run = [&](int ldy, int ldx, int swz, int bar, int remap) {
        if (nt) run_wgrad_like<true>(buf, bytes, ldy, ldx, swz, bar, remap);
        else run_wgrad_like<false>(buf, bytes, ldy, ldx, swz, bar, remap);
      };
      run(192, 192, 0, 0, 0);
      run(192, 192, 1, 0, 0);
      run(192, 192, 0, 1, 0);
      run(192, 192, 1, 1, 0);
      run(576, 192, 1, 1, 0);
      run(576, 192, 1, 1, 1);
      run(576, 192, 0, 1, 1);
      run(384, 192, 1, 1, 1);
      run(768, 384, 1, 1, 1);
      run(768, 384, 0, 0, 1);
    }
    return 0;
  }

  for (int wg : {cus, 2 * cus, 4 * cus, 8 * cus}) {
    run_regs<false, 4, false>(buf, bytes, sink, wg, 256);
    run_regs<false, 8, false>(buf, bytes, sink, wg, 256);
    run_regs<true, 8, false>(buf, bytes, sink, wg, 256);
    run_regs<false, 8, true>(buf, bytes, sink, wg, 256);
    run_regs<true, 8, true>(buf, bytes, sink, wg, 256);
  }
  run_regs<false, 8, false>(buf, bytes, sink, cus, 1024);
  run_regs<true, 8, false>(buf, bytes, sink, cus, 1024);
  run_regs<false, 16, false>(buf, bytes, sink, 2 * cus, 512);
  run_regs<true, 16, false>(buf, bytes, sink, 2 * cus, 512);

  for (int block : {64, 256, 512}) {
    run_dma<false, 8, false>(buf, bytes, cus, block);
    run_dma<true, 8, false>(buf, bytes, cus, block);
    run_dma<false, 16, false>(buf, bytes, cus, block);
    run_dma<true, 16, false>(buf, bytes, cus, block);
    run_dma<false, 16, true>(buf, bytes, cus, block);
    run_dma<true, 16, true>(buf, bytes, cus, block);
  }
  run_dma<false, 32, false>(buf, bytes, cus, 256);
  run_dma<true, 32, false>(buf, bytes, cus, 256);
  run_dma<false, 16, false>(buf, bytes, 2 * cus, 256);
  run_dma<true, 16, false>(buf, bytes, 2 * cus, 256);
  return 0;
}
