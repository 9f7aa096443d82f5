// Internal interface of the MFMA GEMM / implicit-GEMM conv kernel family (gemm.hip).
#pragma once
#include "common.h"

// A-operand loaders
enum SrkLoader {
  LD_ROWS = 0,     // A[M][lda] bf16 row-major, K contiguous
  LD_CONV3 = 1,    // 3x3/pad1/stride1 im2col over NHWC bf16 [B][H][W][CinP]; K = 9*CinP (tap-major)
  LD_CONV3_PS = 2, // same, but the NHWC source is stored pixel-shuffled: logical channel
                   // n' = (i*r+j)*Cs + c of pixel (y,x) lives at src[b][r*y+i][r*x+j][c]  (dgrad of conv+PixelShuffle)
};

// epilogues.  v = acc + bias[n] unless noted.
enum SrkEpilogue {
  EP_BF16 = 0,       // outb[m*ldo+n] = bf16(v)
  EP_QKV = 1,        // n=(which,h,d): qkv[which][b_][h][p][d] = bf16(v * (which==0 ? scale : 1))
  EP_PROJ_RES = 2,   // t=token(m): outf[t][n] = res[t][n] + v     (window_reverse + roll(+shift) + residual)
  EP_GELU = 3,       // outb[m][n] = bf16(v) (pre-activation u), outb2[m][n] = bf16(gelu(v))
  EP_RES = 4,        // outf[m][n] = res[m][n] + v ; optional outb[m][n] = bf16(same)
  EP_DGELU = 5,      // outb[m][n] = bf16(v * gelu'(aux[m][n]))          (no bias)
  EP_LRELU = 6,      // outb = bf16(leaky_relu(v, slope))
  EP_PS = 7,         // pixel-shuffle store: n'=(ij)*Cs+c -> outb[b][r*y+i][r*x+j][c] = bf16(v)
  EP_IMG = 8,        // outf NCHW image: out[b][n][y][x] = v/range + mean[n]  (n < Cimg, crop to Hc x Wc)
  EP_PS_IMG = 9,     // n=c*r*r+i*r+j: out[b][c][r*y+i][r*x+j] = v/range + mean[c]  (crop)
  EP_RES_BF16 = 10,  // outb[m][n] = bf16(res[m][n] + v)
  EP_DLRELU = 11,    // outb = bf16(v * (aux[m][n] > 0 ? 1 : slope))     (no bias)
  EP_F32_BF16 = 12,  // outf[m][n] = v ; outb[m][n] = bf16(v)             (no bias unless given)
  EP_LNBWD = 13,     // acc = dL/d(LN output) for a whole row (needs N == one tile): LayerNorm backward fused in.
                     //   t = ln_rows_window ? token(geom, m) : m ; stats at (ln_stats_by_m ? m : t)
                     //   dx = rstd*(dy*g - mean_c(dy*g) - xhat*mean_c(dy*g*xhat)) ; outf[t] += dx (gradient stream)
                     //   outb[ln_out_window ? winrow(geom, t) : t] = bf16(outf[t] * rowscale[sample]) ; dgamma/dbeta atomics
};

struct GemmParams {
  // operands
  const bf16_t* A;     // LD_ROWS: [M][lda];  conv: NHWC source
  int lda;
  const bf16_t* Wt;    // packed weights [N][K] bf16 (K contiguous)
  int M, N, K;
  // conv geometry (output pixel grid == logical input grid)
  int B, H, W, CinP;   // CinP: logical input channels per tap (multiple of 64)
  int r, Cs;           // LD_CONV3_PS / EP_PS / EP_PS_IMG: shuffle factor and stored channel count
  // epilogue
  const float* bias;   // [N] fp32 or null
  float* outf;
  bf16_t* outb;
  bf16_t* outb2;
  const float* res;    // fp32 residual
  const bf16_t* aux;   // bf16 auxiliary (pre-activation / activation sign)
  double flops;        // algorithmic (un-padded) FLOPs of this launch, for the timing probe
  double bytes;        // algorithmic HBM bytes of this launch (every operand read / result written once, un-padded)
  int ldo;             // row stride (elements) of outf/outb/res/aux
  float scale;         // EP_QKV q scale; EP_LRELU/EP_DLRELU slope
  int nH, CA;          // EP_QKV: heads and nH*32
  WinGeom geom;        // EP_PROJ_RES
  float inv_range;     // EP_IMG / EP_PS_IMG
  float mean[4];
  int Cimg, Hc, Wc;    // valid image channels and crop size (output pixels)
  long long B_;        // EP_QKV: number of windows
  const float* rowscale;  // per-sample DropPath factor (EP_PROJ_RES / EP_RES: scales v; EP_F32_BF16: scales the bf16 copy)
  int rows_per_sample;    // tokens per sample for rowscale indexing
  // EP_LNBWD
  const float* ln_x;      // fp32 LN input [T][ldo]
  const float* ln_mean;
  const float* ln_rstd;
  const float* ln_gamma;  // [C] (un-padded parameter)
  float* ln_dgamma;       // [C] accumulated
  float* ln_dbeta;
  int ln_C;               // real channel count
  int ln_rows_window, ln_stats_by_m, ln_out_window;
  float* ln_skip;         // optional: a second gradient stream of the same rows; the row result becomes outf[t] + dx + ln_skip[t] and is
                          //   stored to ln_skip[t] (outf untouched), outb = its bf16 copy  (the RSTB skip add folded into the layer's last LN backward)
  // EP_PROJ_RES / EP_RES: optional fused forward LayerNorm of the freshly written residual row (the norm that
  // consumes it next: norm2 after proj, the next block's norm1 / the final norm after fc2 / the RSTB conv).
  // Needs N == one tile.  xn_out row = xn_window ? winrow(xn_geom, token) : token; stats are stored at that row.
  bf16_t* xn_out;
  float* xn_mean;
  float* xn_rstd;
  const float* xn_gamma;
  const float* xn_beta;
  int xn_C, xn_window;
  WinGeom xn_geom;
  // fused MLP (srk_launch_mlp_fused): A = xn2 [M][lda], Wt = fc1 weights [HP][K], bias = fc1 bias [HP]; second layer below;
  // res / outf / outb / rowscale / xn_* as for EP_RES (the fc2 + residual epilogue)
  const bf16_t* W2;       // fc2 weights [N][HP] bf16
  const float* bias2;     // [N]
  bf16_t* u_out;          // [M][HP] pre-activation (training: read by the backward pass) or null
  int u_dgelu;            // fused MLP pair only: u_out (forward) / aux (backward) holds gelu'(u) instead of u -- the backward front waves
                          // then multiply instead of evaluating erf + exp per element (they are VALU-bound on it)
  bf16_t* h_out;          // [M][HP] gelu(u)         (training: operand of the fc2 weight gradient) or null
  int HP;                 // hidden width (padded), 384
};

int srk_launch_gemm(int loader, int epilogue, const GemmParams& p, hipStream_t stream);

// gemm_stream.hip: persistent LDS-DMA variant for the LD_ROWS block GEMMs; SRK_NOT_COVERED -> use the tile kernel
#define SRK_NOT_COVERED 1
int srk_launch_gemm_stream(int epilogue, const GemmParams& p, hipStream_t stream);
void srk_gemm_stream_enable(int on);
int srk_gemm_stream_enabled();
void srk_gemm_stream_tune_get(int* bm, int* ks2, int* split, int* nb);
void srk_gemm_stream_tune(int bm, int ks2, int split, int nb);   // 0 / -1 / -1 / 0: defaults
// Mlp.forward + residual (+ the next LayerNorm) as ONE persistent kernel: out = res + rowscale * (gelu(A W1^T + b1) W2^T + b2),
// the 16 x 384 hidden tile stays in LDS.  C = 180 (192 padded), hidden 360 (384).  SRK_NOT_COVERED -> run fc1 / fc2 separately.
int srk_launch_mlp_fused(const GemmParams& p, hipStream_t stream);
void srk_mlp_fused_enable(int on);
int srk_mlp_fused_enabled();
// ... and its backward: d u = (d x2 . W2) * gelu'(u) stays in LDS between the two dgrads, the LayerNorm (norm2) backward rides in the
// second one's epilogue.  A = d x2, Wt = W2^T, aux = u, u_out = d u (output), W2 = W1^T, HP = 384 + the EP_LNBWD fields.
int srk_launch_mlp_fused_bwd(const GemmParams& p, hipStream_t stream);
void srk_mlp_dgelu_store_enable(int on);
int srk_mlp_dgelu_store_enabled();
void srk_mlp_bwd_fused_enable(int on);
int srk_mlp_bwd_fused_enabled();
