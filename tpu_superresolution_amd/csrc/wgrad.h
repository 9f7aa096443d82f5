// Internal interface of the weight-gradient GEMM (wgrad.hip).
#pragma once
#include "common.h"

struct WgradParams {
  const bf16_t* Y;   // [M][ldy] bf16: gradient of the layer output (conv+PixelShuffle: stored shuffled, see r/Cs)
  int ldy;
  const bf16_t* X;   // [M][ldx] bf16: layer input (conv: NHWC, ldx = CinP)
  int ldx;
  int M, N, K;       // conv: K = CinP (per tap)
  float* dW;         // fp32 staging gradient, packed layout [N][ldw]
  int ldw;           // linear: K ; conv: 9*CinP
  float* db;         // fp32 [N] or null
  int m_per;         // rows per workgroup (set by the launcher)
  int conv;          // 0 linear, 1 conv3x3
  int B, H, W;       // conv geometry
  double flops;      // algorithmic (un-padded) FLOPs of this launch, for the timing probe
  double bytes;      // algorithmic HBM bytes of this launch
  int r, Cs;         // conv: Y stored pixel-shuffled with factor r, Cs stored channels (r <= 1: plain)
};

// A small reduction that rides in the reduce launch of the block's weight gradients (its own launch between two large
// kernels costs ~10 us of drain + ramp for 12 MB of traffic): the relative-position-bias table gradient of attn.hip.
struct RpbJob {
  const float* slab;    // [nslab][nH][64][64] partial d(bias), or null
  float* dtable;        // [(2 ws - 1)^2][nH]
  int nslab, nH;
};

struct WgradMulti {
  int nprob;
  int m_per;            // rows per workgroup (set by the launcher)
  int tile_begin[5];    // prefix sum of tiles per problem
  float* partial;       // streaming kernels: scratch slabs [tile][split][WS_SLAB_VEC] of 4 floats, or null (atomics)
  int nsplit;
  int w8;               // the slabs were written by the eight-wave shape of wgrad_stream_kernel (4 x 2 blocks of 48 x 96)
  RpbJob rpb;           // extra workgroups of wgrad_reduce_kernel (blockIdx >= 36 tiles), or slab == null
  WgradParams p[4];
};

// One workgroup per (query token i, head h): the 4 waves sum row i of every slab (256-B coalesced reads, lane = key token j),
// combine through LDS, and scatter the 64 row sums into the table with rpi(i, j).
__device__ __forceinline__ void rpb_reduce_block(const float* __restrict__ slab, float* __restrict__ dtable, int nslab, int nH, int i,
                                                 int h, float (*part)[64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // eight independent loads in flight per wave: one dependent load per iteration made this a chain of ~32 memory round trips
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long long sstride = (long long)nH * 4096;
  const float* base = slab + ((long long)h * 64 + i) * 64 + lane;
  int sidx = wave;
  for (; sidx + 28 < nslab; sidx += 32) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += base[(sidx + 4 * u) * sstride];
  }
  for (; sidx < nslab; sidx += 4) a8[0] += base[sidx * sstride];
  part[wave][lane] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  __syncthreads();
  if (wave == 0) {
    const float v = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    // rpi(i,j) = (yi-yj+7)*15 + (xi-xj+7)
    const int t = ((i >> 3) - (lane >> 3) + 7) * 15 + ((i & 7) - (lane & 7) + 7);
    atomicAdd(dtable + t * nH + h, v);
  }
}

constexpr int WS_SLAB_VEC = 9216;   // accumulator vectors (4 floats) of one workgroup tile: 192 x 192 or 64 x 64 x 9 taps
constexpr size_t WS_WORKSPACE_BYTES = (size_t)256 * WS_SLAB_VEC * 16;   // a launch has at most 256 (tile, split) workgroups
float* srk_wgrad_scratch(hipStream_t stream, size_t bytes);   // the bound workspace if it is large enough, else null
void srk_wgrad_bind_workspace(void* ptr, size_t bytes, void** prev_ptr, size_t* prev_bytes);
void srk_wgrad_partials_enable(int on);
void srk_wgrad_w8_enable(int on);
void srk_wgrad_stream_tune(int rows, int nt);   // rows 32/64 (0 = keep), nt 0/1 (-1 = keep)
int srk_wgrad_partials_enabled();
int srk_wgrad_stream_enabled();
int srk_wgrad_w8_enabled();
void srk_wgrad_stream_tune_get(int* rows, int* nt);

int srk_launch_wgrad(const WgradParams& p, hipStream_t stream);

// convwgrad.hip: all-taps conv weight gradient (W % 64 == 0, N % 64 == 0, K % 64 == 0); SRK_WGRAD_NOT_COVERED otherwise
#define SRK_WGRAD_NOT_COVERED 1
int srk_launch_conv_wgrad_taps(const WgradParams& p, hipStream_t stream);
void srk_conv_wgrad_taps_enable(int on);
int srk_conv_wgrad_taps_mode();
int srk_launch_smallconv_wgrad_mfma(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int CinP, int Co,
                                    int CoP, hipStream_t stream);
int srk_launch_imghead_dgrad_mfma(const float* gy, const float* wgt, bf16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                                  hipStream_t stream);
void srk_wgrad_stream_enable(int on);   // LDS-DMA ring variant of the 192x192 linear tile (wgrad.hip)
int srk_launch_wgrad_multi(const WgradParams* ps, int nprob, hipStream_t stream);
// as above; *rpb (may be null) is carried out by the same launches when the streaming path with split partials runs, else by
// srk_launch_rpb_reduce
int srk_launch_wgrad_multi_rpb(const WgradParams* ps, int nprob, const RpbJob* rpb, hipStream_t stream);
int srk_launch_rpb_reduce(const float* slab, float* dtable, int nslab, int nH, hipStream_t stream);   // attn.hip
