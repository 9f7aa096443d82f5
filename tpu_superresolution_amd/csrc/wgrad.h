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

struct WgradMulti {
  int nprob;
  int m_per;            // rows per workgroup (set by the launcher)
  int tile_begin[5];    // prefix sum of tiles per problem
  float* partial;       // streaming kernels: scratch slabs [tile][split][WS_SLAB_VEC] of 4 floats, or null (atomics)
  int nsplit;
  WgradParams p[4];
};

constexpr int WS_SLAB_VEC = 9216;   // accumulator vectors (4 floats) of one workgroup tile: 192 x 192 or 64 x 64 x 9 taps
constexpr size_t WS_WORKSPACE_BYTES = (size_t)256 * WS_SLAB_VEC * 16;   // a launch has at most 256 (tile, split) workgroups
float* srk_wgrad_scratch(hipStream_t stream, size_t bytes);   // the bound workspace if it is large enough, else null
void srk_wgrad_bind_workspace(void* ptr, size_t bytes, void** prev_ptr, size_t* prev_bytes);
void srk_wgrad_partials_enable(int on);
void srk_wgrad_stream_tune(int rows, int nt);   // rows 32/64 (0 = keep), nt 0/1 (-1 = keep)
int srk_wgrad_partials_enabled();

int srk_launch_wgrad(const WgradParams& p, hipStream_t stream);

// convwgrad.hip: all-taps conv weight gradient (W % 64 == 0, N % 64 == 0, K % 64 == 0); SRK_WGRAD_NOT_COVERED otherwise
#define SRK_WGRAD_NOT_COVERED 1
int srk_launch_conv_wgrad_taps(const WgradParams& p, hipStream_t stream);
void srk_conv_wgrad_taps_enable(int on);
int srk_launch_smallconv_wgrad_mfma(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int CinP, int Co,
                                    int CoP, hipStream_t stream);
int srk_launch_imghead_dgrad_mfma(const float* gy, const float* wgt, bf16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                                  hipStream_t stream);
void srk_wgrad_stream_enable(int on);   // LDS-DMA ring variant of the 192x192 linear tile (wgrad.hip)
int srk_launch_wgrad_multi(const WgradParams* ps, int nprob, hipStream_t stream);
