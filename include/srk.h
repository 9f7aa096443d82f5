/* srk -- super-resolution kernels for AMD MI355X (gfx950): the C ABI of libsrk.so.
 *
 * This is the drop-in boundary of the MI355X-native SwinIR path.  The reference
 * (ViacheslavTimofeev/tpu_superresolution) has no FFI: its boundary is the Python module
 * modules/network_swinir.py (constructor + forward + state_dict) and the training step in
 * modules/finetune_swinir.py.  Each entry point below names the reference code it replaces.
 * The Python side (tpu_superresolution_amd/) binds these with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a raw DEVICE address (tensor.data_ptr()); the caller owns all memory,
 *     kernels never allocate; `stream` is a hipStream_t (0 = default stream);
 *   - all work is enqueued asynchronously on `stream`; no entry point synchronises;
 *   - return value 0 (SRK_OK) or a negative SRK_E_* code; the message of the last error of the
 *     calling thread is returned by srk_last_error();
 *   - "bf16" is raw bfloat16 bits (uint16_t); token tensors use channels padded to a multiple of
 *     64 ("CP"); pad columns are zero.
 */
#ifndef SRK_H_
#define SRK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRK_OK 0
#define SRK_E_SHAPE (-1)
#define SRK_E_NULL (-2)
#define SRK_E_UNSUPPORTED (-3)
#define SRK_E_LAUNCH (-4)
#define SRK_E_ALIGN (-5)
#define SRK_E_STATE (-6)

typedef void* srk_stream_t;

const char* srk_version(void);
const char* srk_last_error(void);

/* ---- bit-exact index operations (stand-alone; elem_bytes in {2,4,8}) ------------------------- */
/* window_partition  network_swinir.py:33-45   x [B,H,W,C] -> out [B*nW, ws, ws, C] */
int srk_window_partition(const void* x, void* out, int B, int H, int W, int C, int ws, int elem_bytes, srk_stream_t stream);
/* window_reverse    network_swinir.py:48-62   windows [B*nW, ws, ws, C] -> out [B,H,W,C] */
int srk_window_reverse(const void* windows, void* out, int B, int H, int W, int C, int ws, int elem_bytes, srk_stream_t stream);
/* torch.roll(x, shifts=(sh,sw), dims=(1,2))  network_swinir.py:249-252, :269-272 */
int srk_roll2d(const void* x, void* out, int B, int H, int W, int C, int sh, int sw, int elem_bytes, srk_stream_t stream);
/* nn.PixelShuffle(r) on NCHW   network_swinir.py:585,588,609   x [B,C*r*r,H,W] -> out [B,C,H*r,W*r] */
int srk_pixel_shuffle(const void* x, void* out, int B, int C, int H, int W, int r, int elem_bytes, srk_stream_t stream);
/* SwinTransformerBlock.calculate_mask  network_swinir.py:216-237   mask fp32 [nW, ws*ws, ws*ws] in {0,-100} */
int srk_shift_mask(float* mask, int H, int W, int ws, int shift, srk_stream_t stream);
/* relative_position_index  network_swinir.py:92-103   out int64 [ws*ws, ws*ws] */
int srk_relative_position_index(int64_t* out, int ws, srk_stream_t stream);

/* ---- building-block kernels (internal layouts; used by the executor and by the parity tests) -- */
typedef struct {
  int H, W;     /* token grid (multiples of 8) */
  int shift;    /* 0 or 4 (window 8) */
} srk_win_geom;

/* nn.LayerNorm over C (eps 1e-5)  network_swinir.py:199,205,519-528,725.  x fp32 [rows][CP]; outputs
 * y_bf16 / y_f32 [rows][CP] (either may be null), mean/rstd fp32 [rows] (may be null).  If geom is
 * non-null the output is in WINDOW order: row m reads token roll+partition(m) (:249-256). */
int srk_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y_bf16, float* y_f32,
                      float* mean, float* rstd, int rows, int C, int CP, const srk_win_geom* geom, srk_stream_t stream);
/* softmax(q k^T + rel-pos-bias + shift-mask) v   network_swinir.py:124-142.
 * qkv bf16 [3][B_][nH][64][32] (q pre-scaled), bias_dense fp32 [nH][64][64], out bf16 [B_*64][nH*32]. */
int srk_window_attention_fwd(const uint16_t* qkv, const float* bias_dense, uint16_t* out, int64_t B_, int nH,
                             const srk_win_geom* geom, srk_stream_t stream);
/* gradient of the above.  d_out bf16 [B_*64][nH*32]; d_qkv bf16 [B_*64][3*nH*32] (columns which,h,d);
 * d_table fp32 [225][nH] is ACCUMULATED; slab = scratch of srk_window_attention_bwd_scratch() bytes. */
int srk_window_attention_bwd(const uint16_t* qkv, const float* bias_dense, const uint16_t* d_out, uint16_t* d_qkv,
                             float* d_table, void* slab, int64_t B_, int nH, float scale, const srk_win_geom* geom,
                             srk_stream_t stream);
size_t srk_window_attention_bwd_scratch(int64_t B_, int nH);
/* The same gradient with q/k/v RE-PROJECTED from the saved LayerNorm output and the gradient of the output projection folded in
 * (csrc/attn_bwd_fused.hip; classical width only: nH = 6, head_dim padded to 32, C padded to 192, B_ >= number of CUs; anything
 * else returns SRK_E_UNSUPPORTED).  network_swinir.py:121-143 backwards: xn bf16 [B_*64][lda] = norm1 output in window order,
 * w_qkv bf16 [576][192] / b_qkv fp32 [576] (packed, q rows first, heads padded 30 -> 32), d_x1 bf16 [B_*64][ldg] = gradient of
 * the attention branch output in window order, w_proj_t bf16 [192 attention channel][192 channel] (d attn_out = d_x1 . w_proj_t^T),
 * d_qkv bf16 [B_*64][576]; d_table fp32 [225][6] is ACCUMULATED; slab = scratch of ..._bwd_fused_scratch() bytes. */
int srk_window_attention_bwd_fused(const uint16_t* xn, int lda, const uint16_t* w_qkv, const float* b_qkv, float scale,
                                   const uint16_t* d_x1, int ldg, const uint16_t* w_proj_t, const float* bias_dense, uint16_t* d_qkv,
                                   float* d_table, void* slab, int64_t B_, int nH, const srk_win_geom* geom, srk_stream_t stream);
size_t srk_window_attention_bwd_fused_scratch(int64_t B_, int nH);
/* dense bias [nH][64][64] from table [225][nH]   network_swinir.py:127-129 */
int srk_rel_pos_bias_expand(const float* table, float* bias_dense, int nH, srk_stream_t stream);
/* y[M][N] = a[M][K] . w[N][K]^T + bias  (bf16 in, fp32 accumulate, bf16 out); K % 64 == 0, N % 64 == 0 */
int srk_linear_bf16(const uint16_t* a, const uint16_t* w, const float* bias, uint16_t* y, int M, int N, int K, srk_stream_t stream);
/* dw[N][K] += y[M][N]^T . x[M][K] ; db[N] += colsum(y)   (bf16 in, fp32 out, accumulating; db may be null) */
int srk_linear_wgrad_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int M, int N, int K, srk_stream_t stream);
/* Up to four of these with the same M as ONE launch (the four linear layers of a transformer block: the launch then fills the chip with
 * few row splits per tile).  Problems whose N and K are all multiples of 192 go out together; any other mix is launched one by one.
 * ldy / ldx: row strides in elements (0: N / K).  dw [N][K] and db [N] (or null) are ACCUMULATED, as srk_linear_wgrad_bf16. */
typedef struct {
  const void* y; int ldy;      /* bf16 [M][ldy]: gradient of the layer output */
  const void* x; int ldx;      /* bf16 [M][ldx]: layer input */
  float* dw; float* db;
  int N, K;
} srk_wgrad_problem;
int srk_linear_wgrad_multi_bf16(const srk_wgrad_problem* problems, int count, int M, srk_stream_t stream);
/* 3x3/s1/p1 conv on NHWC bf16 [B][H][W][CinP] with packed weights [N][9*CinP] (tap-major), + bias -> bf16 NHWC [..][N] */
int srk_conv3x3_bf16(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y, int B, int H, int W, int CinP, int N,
                     srk_stream_t stream);
/* dw[N][9*CinP] += conv weight gradient (y = d output NHWC bf16 [..][N], x = input NHWC bf16 [..][CinP]); db += sum y */
int srk_conv3x3_wgrad_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int B, int H, int W, int CinP, int N,
                           srk_stream_t stream);
int srk_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, srk_stream_t stream);
/* on-device check of the ds_read_b64_tr_b16 contract the kernels rely on: in u16 [64][16], out u16 [64][8] */
int srk_probe_trread(const uint16_t* in, uint16_t* out, srk_stream_t stream);

/* Timing probe for the roofline leg of bench.py: while a probe is active every launch of the chosen kernel
 * family (1 linear GEMM, 2 conv GEMM, 3 linear wgrad, 4 conv wgrad, 5 attention fwd, 6 attention bwd) is bracketed
 * by HIP events on its own stream.  srk_probe_end synchronises those events and returns the summed kernel time,
 * the summed ALGORITHMIC (un-padded) FLOPs and HBM bytes (each operand read / result written once) and the
 * launch count.  Not thread-safe; one probe at a time.  srk_set_option("probe_stride", s) brackets only every s-th
 * launch of the family (a uniform sample when s is coprime with the per-block launch pattern): the two event records
 * per launch otherwise cost a few percent of a step. */
int srk_probe_begin(int family, int capacity);
int srk_probe_end(double* total_ms, double* flops, double* bytes, int* launches);

/* Kernel-selection switches (A/B testing; results are equivalent up to fp32 summation order).
 *   "gemm_stream" 1 (default) / 0: use the persistent LDS-DMA GEMM (csrc/gemm_stream.hip) for the block GEMMs it
 *   covers, or always the tile-per-workgroup GEMM (csrc/gemm.hip).  Env SRK_GEMM_STREAM=0 sets the initial value.
 *   "gemm_stream_bm" 0/16/32/64, "gemm_stream_ks2" -1/0/1, "gemm_stream_split" -1/0/1, "gemm_stream_nb" 0/4/8:
 *   tile-shape overrides of that kernel (0 / -1 = the measured per-epilogue defaults); used by tools/stream_sweep.py.
 *   "probe_stride" 1..1024: see srk_probe_begin.
 *   "wgrad_stream_w8" 1 (default) / 0: the streaming weight-gradient kernel runs eight waves per workgroup (4 x 2 blocks of
 *   48 x 96 of the 192 x 192 tile, two waves per SIMD) or four (2 x 2 blocks of 96 x 96); same sums in the same order.
 *   "block_light" 1 (default) / 0: SwinIR-light width (C <= 64, 6 heads x d <= 16, hidden <= 128), inference: each Swin block is
 *   ONE kernel (csrc/block_light.hip: LayerNorms, qkv, window attention, proj, MLP and both residuals of a window in LDS and
 *   registers) or the layer-per-launch path.
 *   "mlp_bwd_fused" 1 (default) / 0: the MLP half of a Swin block's backward as one kernel (csrc/gemm_stream.hip: fc2 dgrad, GELU',
 *   fc1 dgrad and the norm2 backward; d u stays on the CU between the two GEMMs) or the two streaming GEMMs.
 *   "mlp_dgelu_store" 1 (default) / 0: between the fused MLP forward and the fused MLP backward of a training plan the u buffer
 *   carries bf16(gelu'(u)) instead of bf16(u) (the backward's only use of u; its front waves then multiply instead of evaluating
 *   erf + exp per element).  Read by the training forward; changing "mlp_bwd_fused" / "gemm_stream" between that forward and its
 *   backward is then an error (SRK_E_UNSUPPORTED).  The forward's outputs do not depend on it.
 *   "attn_bwd_fused" 1 (default) / 0: attention backward with q/k/v re-projected from the saved norm1 output and the output-
 *   projection dgrad folded in (csrc/attn_bwd_fused.hip; classical width; the training forward then stores no q/k/v), or the
 *   dgrad GEMM + csrc/attn.hip on q/k/v saved by the forward.  Read when a training forward lays out its workspace.
 *   "attn_fused" 2 (default) / 1 / 0: qkv projection + window attention forward in one kernel per window
 *   (csrc/attn_fused.hip; classical width: 6 heads x 32, C padded to 192) as three 4-wave workgroups per CU (2) or one 8-wave
 *   workgroup per CU (1), or the projection GEMM + attention kernel (0).
 *   "wgrad_stream" 1 (default) / 0: LDS-DMA ring variant of the 192x192 linear weight-gradient tile or the
 *   register-staged one (both in csrc/wgrad.hip).
 *   "conv_wgrad_taps" 2 (default) / 1 / 0: all-taps conv weight gradient with the LDS-DMA ring / register-staged
 *   (csrc/convwgrad.hip, incl. the MFMA image-head kernels), or the per-tap tiles of wgrad.hip + the VALU image head.
 *   "wgrad_stream_rows" 32 (default) / 64: rows per ring stage of that kernel (6 or 3 stages in the 144 KB ring);
 *   "wgrad_stream_nt" 1 (default) / 0: streaming cache policy on its operand DMAs.
 *   "wgrad_partials" 1 (default) / 0: the streaming weight-gradient kernels write their per-split partial tiles to the
 *   caller's workspace (srk_set_wgrad_workspace; the model executor uses a region of its own workspace) and a reduce
 *   kernel adds their sum to dW in a fixed order, or every split adds into dW with fp32 atomics (order-dependent
 *   rounding, ~20 us slower per launch; also what happens when no workspace is registered).
 * Unknown names return SRK_E_UNSUPPORTED.
 * Scope: srk_set_option / srk_get_option act on ONE process-wide value per option (every thread reads it, including the autograd
 * engine's backward thread).  A plan carries its own values -- srk_swinir_plan_set_option -- which hold, in a thread-private copy of
 * the option set, for the duration of each call on that plan: two plans with different options can live in one process and run on
 * two threads.  "probe_stride" belongs to the one-per-process timing probe.  The cached device properties and the "LDS limit raised"
 * flags of the kernels are keyed by device id. */
int srk_set_option(const char* name, int value);
int srk_get_option(const char* name, int* value);

/* ---- training-step pieces  (finetune_swinir.py:148-179) ------------------------------------------ */
/* F.l1_loss(pred, target) (:66-67, :163) forward + backward in one pass; also counts non-finite pred
 * values (assert_finite :133-143, :164).  loss (fp32 scalar) and nonfinite (uint32) are ACCUMULATED
 * (zero them first); d_pred may be null; grad_scale multiplies d_pred (1.0 for a plain backward). */
int srk_l1_loss_fwd_bwd(const float* pred, const float* target, float* d_pred, float* loss, uint32_t* nonfinite,
                        int64_t n, float grad_scale, srk_stream_t stream);
/* Optional workspace of the stand-alone weight-gradient entry points (srk_linear_wgrad_bf16, srk_conv3x3_wgrad_bf16):
 * srk_wgrad_workspace_bytes() bytes of device memory owned by the caller, registered for the CALLING THREAD until replaced
 * (null / 0 unregisters).  With it the row-splits of a weight-gradient tile are summed in a fixed order (reproducible dW);
 * without it they are added with fp32 atomics.  The library never allocates device memory itself.  srk_swinir_backward
 * does not need this: it carves the region out of its own workspace. */
int64_t srk_wgrad_workspace_bytes(void);
int srk_set_wgrad_workspace(void* workspace, int64_t bytes);

/* Data path on the device (first slice): the paired transform of the training set -- ToImage +
 * ToDtype(scale=True), _ensure_3ch and paired_random_crop (finetune_swinir.py:80-110) -- from a pool of pre-decoded 8-bit
 * images in device memory.  pool: the images back to back, each [H][W][C] uint8 with C = 1 or 3.  lr_desc / hr_desc: B
 * descriptors of six int64 {byte offset in pool, H, W, C | (wide << 8), top, left} in DEVICE memory (wide = 1: little-endian
 * uint16 samples at an even byte offset, value = u16 / 65535); the caller draws (top, left) for the LR
 * side (the reference's random.randint pair, :101-102), the HR descriptor carries (top * scale, left * scale) and must lie
 * inside its image (the caller checks; the kernel does not).  lr_out: fp32 [B][3][P][P], hr_out: fp32 [B][3][P*scale][P*scale];
 * value = u8 / 255 in IEEE fp32, a gray image is repeated into three channels: bit-identical to the host transform. */
int srk_paired_crop_u8(const uint8_t* pool, const int64_t* lr_desc, const int64_t* hr_desc, float* lr_out, float* hr_out, int B,
                       int lr_patch, int scale, srk_stream_t stream);
/* Validation metrics of one batch in one pass (first slice): per-image PSNR of the images clamped to [0, 1]
 * (batch_psnr, finetune_swinir.py:69-74: 20 log10(max_val / sqrt(mse + 1e-8)), mse over the per_image = C*H*W elements of an
 * image) and the sum of |pred - target| over the batch for the validation L1 (F.l1_loss, :66-67, used by validate :181-207).
 * pred / target: fp32 [B][per_image].  psnr: fp32 [B] or null.  psnr_sum / abs_sum: fp32 scalars or null, ACCUMULATED (zero
 * them before the first batch; one host read at the end of the validation loop replaces the reference's per-batch .item()).
 * workspace: srk_batch_psnr_workspace(per_image, B) bytes of device memory owned by the caller.  Sums are formed in a fixed
 * order (no atomics): results are reproducible.  B <= 1024. */
int64_t srk_batch_psnr_workspace(int64_t per_image, int B);
int srk_batch_psnr(const float* pred, const float* target, void* workspace, int B, int64_t per_image, float max_val, float* psnr,
                   float* psnr_sum, float* abs_sum, srk_stream_t stream);
/* evaluate.py:24-29 on the device: psnr[b] = 20 log10(max_val / sqrt(max(mse_b, 1e-10))) of fp32 [B][per_image] images, NO
 * clamp; mean = their batch mean (what the reference's psnr() returns).  psnr / mean may be null.  workspace:
 * srk_eval_psnr_workspace bytes.  Fixed-order sums. */
int64_t srk_eval_psnr_workspace(int64_t per_image, int B);
int srk_eval_psnr(const float* x, const float* y, void* workspace, int B, int64_t per_image, float max_val, float* psnr, float* mean,
                  srk_stream_t stream);
/* pytorch_msssim.ssim(X, Y, data_range, size_average) as called at train.py:169 / evaluate.py:127,195: fp32 NCHW images, 11-tap
 * Gaussian (sigma 1.5) VALID separable filter, K = (0.01, 0.03); per_image[b] = mean over channels of the per-channel map mean
 * (size_average=False), mean = batch mean (size_average=True); either may be null.  H, W >= 11.  PARITY UNPINNED: the package
 * is a third-party dependency absent from the reference tree (sr_environment.yml:165); restated from its published algorithm. */
int64_t srk_ssim_workspace(int B, int C, int H, int W);
int srk_ssim(const float* x, const float* y, void* workspace, int B, int C, int H, int W, float data_range, float* per_image, float* mean,
             srk_stream_t stream);
/* sum of squares of a flat fp32 gradient, ACCUMULATED into sumsq[0] (for clip_grad_norm_ :170) */
int srk_grad_sumsq(const float* grads, int64_t n, float* sumsq, srk_stream_t stream);
/* clip_grad_norm_(max_norm) + AdamW step (:168-171, :303) on flat fp32 buffers.  The clip coefficient is
 * computed on the device from sumsq[0] (no host sync); grads are first divided by grad_div (world size).
 * max_norm <= 0 disables clipping.  step is the 1-based step count.  nonfinite: optional DEVICE counter (the one
 * srk_l1_loss_fwd_bwd fills, :133-143); when it is non-zero, or when the gradient norm itself is NaN/Inf, the call leaves
 * params and both moments untouched -- the reference raises before backward/step (:159-165), so the weights survive the raise. */
int srk_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const float* sumsq, float max_norm, float grad_div, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, const int32_t* nonfinite, srk_stream_t stream);

/* ---- generic GEMM / 3x3 conv with the fused epilogues, for host-orchestrated models (HAT: tpu_superresolution_amd/hat_arch.py)
 * D[m][n] = sum_k A[m][k] W[n][k] (+ epilogue).  A bf16: SRK_LD_ROWS [M][lda]; SRK_LD_CONV3 NHWC [B][H][Wd][CinP] (3x3, pad 1,
 * K = 9 * CinP tap-major).  W bf16 [N][K].  N % 64 == 0 (N == 16 for the image head), K % 64 == 0. */
enum { SRK_LD_ROWS = 0, SRK_LD_CONV3 = 1,
       SRK_LD_CONV3_PS = 2 /* 3x3 conv whose NHWC source is stored pixel-shuffled by r (Cs stored channels): dgrad of conv + PixelShuffle */ };
enum { SRK_EP_BF16 = 0,       /* outb = bf16(v + bias) */
       SRK_EP_GELU = 3,       /* outb2 = bf16(gelu(u)), u = v + bias; outb = bf16(u) if not null (kept for a backward pass) */
       SRK_EP_RES = 4,        /* outf = res + v + bias (fp32) [+ outb bf16 copy] [+ LayerNorm of the new row -> xn_out, N == 64/128/192] */
       SRK_EP_LRELU = 6,      /* outb = bf16(leaky_relu(v + bias, scale)) */
       SRK_EP_PS = 7,         /* conv + PixelShuffle(r): W rows permuted to (i*r + j)*Cs + c; store is the shuffled NHWC tensor */
       SRK_EP_IMG = 8,        /* N == 16: outf NCHW image [B][Cimg][Hc][Wc] = v * inv_range + mean[c] (crop) */
       SRK_EP_PS_IMG = 9,     /* N == 16: conv + PixelShuffle(r) straight into the NCHW image (UpsampleOneStep), n = c*r*r + i*r + j */
       SRK_EP_RES_BF16 = 10,  /* outb = bf16(res + v + bias) */
       SRK_EP_DGELU = 5,      /* outb = bf16(v * gelu'(aux))            (backward through an exact-erf GELU; aux = pre-activation) */
       SRK_EP_DLRELU = 11,    /* outb = bf16(v * (aux > 0 ? 1 : scale)) (backward through LeakyReLU; aux = its output) */
       SRK_EP_F32_BF16 = 12,  /* outf = v (fp32) [+ outb = bf16(v)] */
       SRK_EP_LNBWD = 13,     /* v = gradient of a LayerNorm OUTPUT row (N = the padded width, <= 192): LayerNorm backward through (ln_x, ln_mean,
                                 ln_rstd, ln_gamma) fused in: outf += d x in place (fp32 gradient stream), outb = bf16(outf * rowscale) or null,
                                 ln_dgamma / ln_dbeta ACCUMULATED.  A dgrad GEMM whose consumer is a LayerNorm backward (no bias) */
       SRK_EP_MLP_FUSED = 100 /* (reserved; see srk_mlp_fused_fwd) */ };
typedef struct {
  int loader, epilogue;
  const void* A; int lda;
  const void* W;
  int M, N, K;
  int B, H, Wd, CinP;          /* conv geometry (M == B*H*Wd) */
  int r, Cs;                   /* SRK_EP_PS */
  const float* bias;
  float* outf; void* outb; void* outb2;
  const float* res; const void* aux;
  int ldo;                     /* row stride (elements) of outf / outb / res */
  float scale;                 /* SRK_EP_LRELU slope */
  float inv_range; float mean[4]; int Cimg, Hc, Wc;     /* SRK_EP_IMG */
  void* xn_out; float* xn_mean; float* xn_rstd; const float* xn_gamma; const float* xn_beta; int xn_C;   /* SRK_EP_RES fused LayerNorm */
  const float* rowscale; int rows_per_sample;   /* SRK_EP_RES: outf = res + rowscale[m / rows_per_sample] * (v + bias)  (DropPath factor per sample) or null */
  const float* ln_x; const float* ln_mean; const float* ln_rstd; const float* ln_gamma; float* ln_dgamma; float* ln_dbeta; int ln_C;   /* SRK_EP_LNBWD */
} srk_gemm_args;
int srk_gemm_ex(const srk_gemm_args* args, srk_stream_t stream);
/* Mlp.forward + residual (+ next LayerNorm) in one kernel (csrc/gemm_stream.hip): out = res + gelu(xn W1^T + b1) W2^T + b2.
 * xn bf16 [M][192], W1 bf16 [384][192], W2 bf16 [192][384], res / out fp32 [M][192]; C 180 / hidden 360 zero-padded. */
int srk_mlp_fused_fwd(const uint16_t* xn, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* res,
                      float* out, uint16_t* out_bf16, uint16_t* xn_next, float* xn_mean, float* xn_rstd, const float* xn_gamma,
                      const float* xn_beta, int xn_C, int M, srk_stream_t stream);
/* SwinTransformerBlock.forward (network_swinir.py:239-279: LN1, roll + window partition, WindowAttention :114-145, reverse, residual,
 * LN2, Mlp :25-28, residual) as ONE kernel at the SwinIR-light width (csrc/block_light.hip; inference, no DropPath).
 * x, y: fp32 [B*H*W][64] token-major, channels >= C zero (y may alias x); y_bf16: optional bf16 copy of y.  Packed bf16 weights:
 * wqkv [3*192][64] (row = which*192 + head*32 + d), wproj [64][192] (column = head*32 + d), w1 [128][64], w2 [64][128]; fp32
 * biases in the same padded order; bias_dense [6][64][64] = table[rpi]; C <= 64, num_heads == 6, head_dim <= 16, hidden <= 128,
 * H and W multiples of 8, shift 0 or 4.  SRK_E_UNSUPPORTED for any other width. */
int srk_swin_block_fwd(const float* x, float* y, uint16_t* y_bf16, const float* norm1_w, const float* norm1_b, const float* norm2_w,
                       const float* norm2_b, const uint16_t* wqkv, const float* bqkv, const uint16_t* wproj, const float* bproj,
                       const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* bias_dense, float scale,
                       int C, int num_heads, int head_dim, int hidden, int B, int H, int W, int shift, srk_stream_t stream);
/* image -> padded, normalised NHWC4 (check_image_size + (x - mean) * range, hat_arch.py:963-975) and conv_first (fp32 VALU) */
int srk_img_prep(const float* x, float* out, int B, int Cimg, int H0, int W0, int H, int W, float range, const float* mean3, srk_stream_t stream);
int srk_stem_conv(const float* img4, const float* weight, const float* bias, float* out, int B, int H, int W, int Cin, int C, int CP,
                  srk_stream_t stream);

/* ---- HAT (reference hat_arch.py) ------------------------------------------------------------------------------------------------
 * Window attention with 256 queries per window, forward.  qkv bf16 [T][ldq] in RASTER token order straight from the qkv linear
 * (q | k | v at columns 0 / CA / 2 CA, head h at +32 h, head_dim zero-padded to 32, q NOT pre-scaled); out bf16 [T][ldo] raster.
 * bias: table_rows == 0: dense fp32 [num_heads][256][NK]; table_rows > 0 (16 x 16 windows): the relative_position_bias_table
 * parameter itself, fp32 [table_rows][num_heads] (961 rows, 1521 for the overlapping form) -- the kernel stages the head's
 * column in LDS and evaluates relative_position_index_SA / _OCA (:881-918) in closed form, negative OCA indices wrapped.  overlap == 0: (shifted) window self-attention, wh x ww windows with
 * wh * ww == 256 (WindowAttention.forward :163-197 + the roll / partition / reverse of HAB.forward :298-319; the shift mask of
 * calculate_mask :921-941 is evaluated arithmetically).  overlap == 8: overlapping cross-attention of OCAB.forward :403-432
 * (16 x 16 queries, 24 x 24 zero-padded keys, NK = 576). */
int srk_win256_attention_fwd(const uint16_t* qkv, int ldq, int CA, const float* bias, int table_rows, uint16_t* out, int ldo, int B, int H,
                             int W, int wh, int ww, int shift_y, int shift_x, int num_heads, float scale, int overlap, srk_stream_t stream);
/* The same with the H x W map zero-padded at the bottom / right to an Hp x Wp window frame (DAT: Adaptive_Spatial_Attention.forward
 * dat_arch.py:376-384 pads q, k, v to a multiple of the larger split; windows, cyclic shift and shift mask live on the frame, a padded
 * token is a zero vector that takes part in its window's softmax with score = bias, padded output rows are dropped) and with
 * windows of 256 OR 128 tokens (wh * ww; 128: split_size [8, 16], dense bias [num_heads][128][128]).  Hp, Wp multiples of wh, ww. */
int srk_win_attention_fwd_padded(const uint16_t* qkv, int ldq, int CA, const float* bias, int table_rows, uint16_t* out, int ldo, int B, int H,
                                 int W, int Hp, int Wp, int wh, int ww, int shift_y, int shift_x, int num_heads, float scale, int overlap,
                                 srk_stream_t stream);
/* ChannelAttention gate of CAB (:41-57): gate[b][c] = out_scale * sigmoid(W2 relu(W1 mean_b + b1) + b2), mean over the HW tokens
 * of x bf16 [B*HW][CP]; w1 [S][C], w2 [C][S] fp32 (the 1x1 convs).  workspace: srk_channel_gate_workspace bytes. */
size_t srk_channel_gate_workspace(int B, int HW, int CP);
int srk_channel_gate(const uint16_t* x, void* workspace, const float* w1, const float* b1, const float* w2, const float* b2, float out_scale,
                     float* gate, int B, int HW, int C, int CP, int S, srk_stream_t stream);
/* same with the hidden activation selectable: act 0 ReLU (HAT), 1 GELU (DAT's channel_interaction, dat_arch.py:315-321, BatchNorm folded) */
int srk_channel_gate_act(const uint16_t* x, void* workspace, const float* w1, const float* b1, const float* w2, const float* b2, float out_scale,
                         float* gate, int B, int HW, int C, int CP, int S, int act, srk_stream_t stream);
/* HAB.forward :322-323: x += conv * gate[sample] in place (fp32 [rows][CP]) and, if xn != null, xn = bf16(LayerNorm(x)) */
int srk_cab_add_ln(float* x, const uint16_t* conv, const float* gate, const float* gamma, const float* beta, uint16_t* xn, int64_t rows,
                   int rows_per_sample, int C, int CP, srk_stream_t stream);

/* ---- training pieces for the host-orchestrated models (csrc/hat_train.hip, csrc/attn256_bwd.hip) ----------------------------------
 * Gradient of srk_win256_attention_fwd (table-indexed bias, 16 x 16 windows; WindowAttention.forward hat_arch.py:163-197 and
 * OCAB.forward :403-439 backwards).  d_out bf16 [T][ldo] raster = gradient of the attention output; d_qkv bf16 [T][ldq] in the
 * layout of qkv (q gradient w.r.t. the UNSCALED q, as qkv holds it); d_table fp32 [table_rows][num_heads] is ACCUMULATED (the
 * negative relative_position_index_OCA entries wrap, :911-918); scratch: srk_win256_attention_bwd_scratch bytes.  In the
 * overlapping form the key / value gradients of a token are summed over the (up to four) key windows that hold it. */
size_t srk_win256_attention_bwd_scratch(int B, int H, int W, int num_heads, int CA, int table_rows, int overlap);
int srk_win256_attention_bwd(const uint16_t* qkv, int ldq, int CA, const float* table, int table_rows, const uint16_t* d_out, int ldo,
                             uint16_t* d_qkv, float* d_table, void* scratch, int B, int H, int W, int shift_y, int shift_x, int num_heads,
                             float scale, int overlap, srk_stream_t stream);
/* Gradient of `x += conv * gate[sample]`, gate = srk_channel_gate(conv) (HAB.forward :322 with CAB :41-75): g fp32 [B*HW][CP] is the
 * gradient of the sum, conv bf16 the second CAB conv's output, gate fp32 [B][CP] as the forward produced it.  d_conv bf16 [B*HW][CP] =
 * g * gate + (gradient through the average pool); dw1 [S][C], db1 [S], dw2 [C][S], db2 [C] are ACCUMULATED; dmean fp32 [B][CP] scratch. */
size_t srk_cab_bwd_workspace(int B, int HW, int CP);
int srk_cab_bwd(const uint16_t* conv, const float* g, const float* gate, void* workspace, const float* w1, const float* b1, const float* w2,
                const float* b2, float out_scale, float* dw1, float* db1, float* dw2, float* db2, float* dmean, uint16_t* d_conv, int B, int HW,
                int C, int CP, int S, srk_stream_t stream);
/* LayerNorm backward over C of CP columns: dy bf16 [rows][CP]; x, mean, rstd as srk_layernorm_fwd saw / produced them;
 * gx fp32 [rows][CP] = (accumulate ? gx : 0) + dx; gx_bf16 (optional) = bf16(gx); dgamma / dbeta fp32 [C] ACCUMULATED. */
int srk_layernorm_bwd(const uint16_t* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* gx,
                      uint16_t* gx_bf16, float* dgamma, float* dbeta, int rows, int C, int CP, int accumulate, srk_stream_t stream);
/* element-wise helpers of a backward pass: a += b and ab_bf16 = bf16(a); a += float(b); out = a + b   (n % 4 == 0) */
int srk_add_f32_bf16(float* a, const float* b, uint16_t* ab_bf16, int64_t n, srk_stream_t stream);
int srk_add_bf16_into_f32(float* a, const uint16_t* b, int64_t n, srk_stream_t stream);
int srk_add_f32(float* out, const float* a, const float* b, int64_t n, srk_stream_t stream);
/* image head backwards: d_pred fp32 NCHW [B][Cimg][Hc][Wc] -> gy fp32 [B*H*W][CoP] (times inv_range; r > 1: un-pixel-shuffled);
 * weight / input gradients of a 3x3 conv with few output channels (conv_last), fp32 parameters [Co][Cin][3][3] */
int srk_img_grad_prep(const float* d_pred, float* gy, int B, int Cimg, int Hc, int Wc, int H, int W, int r, int CoP, float inv_range,
                      srk_stream_t stream);
int srk_smallconv_wgrad(const uint16_t* x, const float* gy, float* dw, float* db, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                        srk_stream_t stream);
int srk_smallconv_dgrad(const float* gy, const float* weight, uint16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP,
                        srk_stream_t stream);
/* conv_first backwards: dw [C][Cin][3][3], db [C] ACCUMULATED from the padded NHWC4 image and gy fp32 [B*H*W][CP] */
int srk_stem_wgrad(const float* img4, const float* gy, float* dw, float* db, int B, int H, int W, int Cin, int C, int CP, srk_stream_t stream);
/* srk_conv3x3_wgrad_bf16 with y stored pixel-shuffled by r (the gradient of a conv + PixelShuffle(r) output, Cs stored channels) */
int srk_conv3x3_wgrad_ps_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int B, int H, int W, int CinP, int N, int r, int Cs,
                              srk_stream_t stream);
/* srk_mlp_fused_fwd that also stores u = xn W1^T + b1 and h = gelu(u) (bf16 [M][384]) for a backward pass */
int srk_mlp_fused_fwd_train(const uint16_t* xn, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* res,
                            float* out, uint16_t* out_bf16, uint16_t* u_out, uint16_t* h_out, uint16_t* xn_next, float* xn_mean,
                            float* xn_rstd, const float* xn_gamma, const float* xn_beta, int xn_C, const float* rowscale, int rows_per_sample,
                            int M, srk_stream_t stream);
/* dst[m][c] = bf16(float(src[m][c]) * f[m / rows_per_sample])  (a DropPath factor on a bf16 gradient copy; dst may alias src) */
int srk_rowscale_bf16(const uint16_t* src, uint16_t* dst, const float* f, int64_t rows, int rows_per_sample, int CP, srk_stream_t stream);

/* ---- DAT (reference dat_arch.py), inference pieces; token-major bf16 [T][ld], channels padded per head to 32 -------------------------
 * Depth-wise 3x3 conv, pad 1, over C8*8 channels of x (column 0 of the slice given): out = act((conv(x)) * scale + shift) * mul.
 * w fp32 [C8*8][9]; scale / shift fp32 [C8*8] carry the conv bias and an inference BatchNorm folded by the caller
 * (dat_arch.py:310-314, :463-467); act 0 none / 1 GELU; mul (optional) is SGFN's x1 in x1 * DWconv(LN(x2)) (:48-54). */
int srk_dwconv3x3(const uint16_t* x, int ldx, const float* w, const float* scale, const float* shift, const uint16_t* mul, int ldm, uint16_t* out,
                  int ldo, int B, int H, int W, int C8, int act, srk_stream_t stream);
/* LayerNorm (eps 1e-5) over C channels of a bf16 row slice -> bf16 [rows][ldo], columns C..CP_out written as zero (SpatialGate.norm :46) */
int srk_rowln_bf16(const uint16_t* x, int ldx, const float* gamma, const float* beta, uint16_t* out, int ldo, int64_t rows, int C, int CP_out,
                   srk_stream_t stream);
/* spatial_interaction (:322-327, :475-480): gate[t] = sigmoid(b3 + w3 . gelu(W0 x_t + b0)); W0 fp32 [S][CP] (BatchNorm folded, zero at pads) */
int srk_spatial_gate(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, float b3, int S, float* gate, int64_t rows,
                     int CP, srk_stream_t stream);
/* the same with b3 read from device memory (a parameter that changes every training step: no host round trip) */
int srk_spatial_gate_dev(const uint16_t* x, int ldx, const float* W0, const float* b0, const float* w3, const float* b3, int S, float* gate,
                         int64_t rows, int CP, srk_stream_t stream);
/* out = a * ga + b * gb with one gate per token (tgate [rows]) and one per (sample, channel) (cgate [B][CP]); tok_gate_on_a selects
 * which operand takes the token gate (:430-436 spatial block: 0; :518-524 channel block: 1).  Gates are post-sigmoid. */
int srk_dual_gate_combine(const uint16_t* a, const uint16_t* b, const float* cgate, const float* tgate, uint16_t* out, int64_t rows,
                          int rows_per_sample, int CP, int tok_gate_on_a, srk_stream_t stream);
/* Adaptive_Channel_Attention core (:481-505): per sample and head, q / k columns L2-normalised over the N tokens, logits
 * (d x d) * temperature[h], softmax, applied to v.  qkv bf16 [B*N][ldq] (q | k | v at 0 / CA / 2 CA, head h at +32 h); out bf16
 * [B*N][ldo] (head-padded channels).  Fixed-order reductions. */
size_t srk_channel_attention_workspace(int B, int N, int num_heads);
int srk_channel_attention_fwd(const uint16_t* qkv, int ldq, int CA, const float* temperature, void* workspace, uint16_t* out, int ldo, int B,
                              int N, int num_heads, int head_dim, srk_stream_t stream);

/* Backward of Mlp + residual + the LayerNorm in front of it as ONE kernel (autograd of hat_arch.py:86-92 + :324 / network_swinir.py:25-28
 * + :277): d u = (g W2) * gelu'(u) -> du_out, d xn = d u W1, LayerNorm backward through (ln_x, mean, rstd, gamma): gx += d x in place
 * (fp32), gxb = bf16(gx * rowscale[row / rows_per_sample]) (or null), d_gamma / d_beta ACCUMULATED.  Layouts as srk_mlp_fused_fwd with
 * the weights transposed: w2t [384][192], w1t [192][384]; C = 180, M % 64 == 0, M >= 64 * #CUs, else SRK_E_UNSUPPORTED. */
int srk_mlp_fused_bwd(const uint16_t* g, const uint16_t* w2t, const uint16_t* u, uint16_t* du_out, const uint16_t* w1t, const float* ln_x,
                      const float* ln_mean, const float* ln_rstd, const float* ln_gamma, float* gx, uint16_t* gxb, const float* rowscale,
                      int rows_per_sample, float* d_gamma, float* d_beta, int C, int M, srk_stream_t stream);

/* ---- DAT training pieces (csrc/dat_train.hip, csrc/attn_rect_bwd.hip) -----------------------------------------------------------
 * Token-sized work and token reductions only: the per-channel / per-sample functions in between (train-mode BatchNorm coefficients
 * dat_arch.py:301-313 / :464-476, channel_interaction on the pooled [B][C] vector, the d x d channel-attention matrices :497-503,
 * the DynamicPosBias MLP :93-130) are evaluated by the caller on the reductions these return (tpu_superresolution_amd/dat_train.py).
 * All bf16 operands are row-major [rows][ld] with 8-element (16-byte) aligned rows; C8 = channels / 8. */
/* backward of srk_win_attention_fwd_padded with a dense bias: d_qkv (q | k | v slices of the heads of this launch) and d_bias
 * [heads][N][N] ACCUMULATED (zero it first) over the windows.  scratch: null (d_bias by float atomics) or
 * srk_win_attention_bwd_padded_scratch bytes (per-window dS tiles + a reduction kernel, and a transposed copy of the bias for the
 * key-major pass: the fast path) */
size_t srk_win_attention_bwd_padded_scratch(int B, int Hp, int Wp, int wh, int ww, int num_heads);
int srk_win_attention_bwd_padded(const uint16_t* qkv, int ldq, int CA, const float* bias, const uint16_t* d_out, int ldo, uint16_t* d_qkv,
                                 float* d_bias, void* scratch, int B, int H, int W, int Hp, int Wp, int wh, int ww, int shift_y, int shift_x,
                                 int num_heads, float scale, srk_stream_t stream);
/* partial [samples][chunks][2][8 C8]: per 256-row chunk of a sample, sum_t p[t][c] and sum_t p[t][c] q[t][c] (fixed order; the caller
 * sums the chunks).  BatchNorm batch statistics (q = p), its backward sums (p = dz, q = x), the pooled mean (samples = B). */
int64_t srk_chan_stats_chunks(int64_t rows_per_sample);
int srk_chan_stats(const uint16_t* p, int ldp, const uint16_t* q, int ldq, float* partial, int samples, int64_t rows_per_sample, int C8,
                   srk_stream_t stream);
/* nn.BatchNorm2d in training mode between two token passes (dat_arch.py:301-313, :464-476), one launch each:
 *   srk_bn_train_coeffs: the R partial rows (row_stride floats apart; sum x at + 0, sum x^2 at + ld; from srk_chan_stats or
 *     srk_spatial_gate_train what 0) summed in a fixed order -> coef [4][ld] = scale (gamma rstd), shift (beta - mean scale), mean, rstd over n values per channel; running_mean / running_var
 *     (or null) move in place by `momentum` with the unbiased variance; real_of[c] = index of channel c in the module's un-padded buffers,
 *     -1 for padding (null: identity).
 *   srk_bn_train_bwd_coeffs: partial rows (sum dz, sum dz x) + the forward's coef -> coef [5][ld] = A, B, C of d x = A dz + B x + C,
 *     d gamma, d beta. */
/* out[o][i] = sum over r < R of in[o][r][i] (in: fp32 [outer][R][n] contiguous), rows added in a fixed order: the finishing sum of the
 * per-chunk partial rows the token passes above leave behind */
int srk_sum_rows_f32(const float* in, int outer, int R, int n, float* out, srk_stream_t stream);
int srk_bn_train_coeffs(const float* partial, int R, int row_stride, int ld, int C, float n, const float* gamma, const float* beta, float eps, float* coef,
                        float* running_mean, float* running_var, float momentum, const int* real_of, srk_stream_t stream);
int srk_bn_train_bwd_coeffs(const float* partial, int R, int row_stride, int ld, int C, float n, const float* fwd_coef, float* coef,
                            srk_stream_t stream);
/* out = act(x * scale[i][c] + shift[i][c]); i = row / rows_per_sample (rows_per_sample 0: one vector for all rows); act 1 = GELU */
int srk_affine_act_bf16(const uint16_t* x, int ldx, const float* scale, const float* shift, uint16_t* out, int ldo, int64_t rows, int C8,
                        int rows_per_sample, int act, srk_stream_t stream);
/* out = dy * gelu'(x * scale[c] + shift[c]) */
int srk_dgelu_affine_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const float* scale, const float* shift, uint16_t* out,
                          int ldo, int64_t rows, int C8, srk_stream_t stream);
/* out (+)= A[i][c] p + B[i][c] q + C[i][c]   (null A / B: coefficient 1; null p / q: no such term; accumulate: out is read first) */
int srk_lincomb2_bf16(const uint16_t* p, int ldp, const uint16_t* q, int ldq, const float* A, const float* Bc, const float* Cc, uint16_t* out,
                      int ldo, int64_t rows, int C8, int rows_per_sample, int accumulate, srk_stream_t stream);
/* d a = dy * b, d b = dy * a  (SpatialGate's x1 * x2, dat_arch.py:54) */
int srk_mul_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* a, int lda, const uint16_t* b, int ldb, uint16_t* da, int ldda, uint16_t* db,
                     int lddb, int64_t rows, int C8, srk_stream_t stream);
/* depth-wise 3x3 (pad 1): partial [B][srk_dwconv3x3_wgrad_chunks(H)][10][8 C8]; rows 0..8 the taps' weight gradient, row 9 the bias gradient */
int srk_dwconv3x3_wgrad_chunks(int H);
int srk_dwconv3x3_wgrad(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, float* partial, int B, int H, int W, int C8,
                        srk_stream_t stream);
/* backward of srk_dual_gate_combine, out = a_chan * cgate[b][c] + a_tok * tgate[t]:  d_chan = d * cgate, d_tok = d * tgate,
 * dcg_partial [B][ceil(HW / 64)][CA] = chunk sums of d * a_chan  (gradient w.r.t. the post-sigmoid channel gate),
 * dsmap [B * HW] = (sum_c d * a_tok) * tgate (1 - tgate)   (gradient w.r.t. the PRE-sigmoid spatial map) */
int srk_dual_gate_bwd(const uint16_t* dcomb, const uint16_t* a_chan, const uint16_t* a_tok, const float* cgate, const float* tgate,
                      uint16_t* d_chan, uint16_t* d_tok, float* dcg_partial, float* dsmap, int B, int HW, int CA, srk_stream_t stream);
/* spatial_interaction in training (:318-323 / :475-480): y1 = W0 x + b0 (S <= 16), z = y1 * bn_scale + bn_shift, smap = w3 . gelu(z) + b3.
 *   what 0: partial [blocks][2][16]  = sums of y1, y1^2 over each 256-row block (BatchNorm batch statistics)
 *   what 1: partial [blocks][4][16]  = sums of dz, dz * y1, dsmap * gelu(z), dsmap (in slot 0) with dz = dsmap * w3 * gelu'(z)
 *   what 2: dy1 = cA * dz + cB * y1 + cC (the BatchNorm backward, coefficients from the caller); dx (+)= W0^T dy1;
 *           partial [blocks][16][C + 1] = the block's d W0 [s][c] (first 16 C floats) and d b0 [s] (last 16) */
int srk_spatial_gate_train(int what, const uint16_t* x, int ldx, const float* W0, const float* b0, const float* bn_scale, const float* bn_shift,
                           const float* w3, const float* dsmap, const float* cA, const float* cB, const float* cC, uint16_t* dx, int lddx,
                           int accumulate, float* partial, int64_t rows, int C, int S, srk_stream_t stream);
/* LayerNorm (eps 1e-5) backward on bf16 rows (SpatialGate.norm): dx bf16 (columns C..CP_out zero); partial [blocks][2][C] with the
 * workgroups' d gamma / d beta sums, blocks = srk_rowln_bwd_blocks(rows); CP_out <= 512 */
int64_t srk_rowln_bwd_blocks(int64_t rows);
int srk_rowln_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const float* gamma, uint16_t* dx, int lddx, float* partial,
                       int64_t rows, int C, int CP_out, srk_stream_t stream);
/* channel attention (:497-508): partial [B][heads][ceil(N / 256)][1088] = per chunk G[i][j] = sum_n x[n][32 h + i] y[n][32 h + j] (1024),
 * sum_n x[n][i]^2 (32), sum_n y[n][j]^2 (32); srk_chan_gram_floats = the partial's size in floats */
int64_t srk_chan_gram_floats(int B, int N, int num_heads);
int srk_chan_gram(const uint16_t* x, int ldx, const uint16_t* y, int ldy, float* partial, int B, int N, int num_heads, srk_stream_t stream);
/* out[n][32 h + i] (+)= sum_j M[b][h][i][j] src[n][32 h + j] + diag[b][h][i] src2[n][32 h + i]   (M fp32 [B][heads][32][32]; diag optional) */
int srk_chan_apply_mat(const float* M, const uint16_t* src, int ldsrc, const float* diag, const uint16_t* src2, int ldsrc2, uint16_t* out, int ldo,
                       int B, int N, int num_heads, int accumulate, srk_stream_t stream);

/* DAT training, the small functions between token passes as one launch each (csrc/dat_small.hip):
 *   srk_channel_interaction_fwd / _bwd: channel_interaction (dat_arch.py:315-321) on the pooled 1 x 1 map: pooled [B][ldp] = per-sample token
 *     sums in the head-padded channel order, pad_of[c] = padded position of real channel c; pm = pooled[pad_of] * inv_hw -> 1x1 conv W1 [S][C]
 *     -> BatchNorm2d over the BATCH (batch statistics; running buffers, if given, move as nn.BatchNorm2d's do) -> GELU -> 1x1 conv W2 [C][S]
 *     -> sigmoid -> cgate [B][CA] (padding 0); pm_out [B][C] is what the backward needs.  _bwd: d cgate [B][ldg] (head-padded) -> the six
 *     parameter gradients and dpool [B][CA] = d pm * inv_hw scattered back (padding 0).  One workgroup with both weight matrices and every
 *     [B][C] / [B][S] array in LDS: srk_channel_interaction_covered(B, C, S) says whether a shape fits (B > 1, S <= 64, 150 KB); SRK_E_SHAPE
 *     beyond (the caller keeps a torch path for larger batches).
 *   srk_chan_attn_matrix_fwd / _bwd: Adaptive_Channel_Attention's d x d matrix (:497-503) from the chunk partials of srk_chan_gram(q, k):
 *     gram [B][heads][1088] = their sum (G | sum q^2 | sum k^2), A [B][heads][32][32] = softmax_j(temperature_h G_ij / (|q_i| |k_j|)) over the
 *     dh real channels (norms clamped at 1e-12 as F.normalize; padding rows / columns 0).  _bwd: dpartial = chunk partials of
 *     srk_chan_gram(d out, v) (= d A) -> dG, dGt (its transpose), dsq2 / dsk2 [B][heads][32] (the diagonal coefficients of d q / d k through
 *     the norms, as srk_chan_apply_mat takes them), dtemp [B][heads] (sum over B = d temperature). */
int srk_channel_interaction_covered(int B, int C, int S);
int srk_channel_interaction_fwd(const float* pooled, int ldp, float inv_hw, const int* pad_of, const float* W1, const float* b1,
                                const float* gamma, const float* beta, float eps, const float* W2, const float* b2, float* running_mean,
                                float* running_var, float momentum, float* pm_out, float* cgate, int B, int C, int S, int CA,
                                srk_stream_t stream);
int srk_channel_interaction_bwd(const float* pm, const float* dcgate, int ldg, float inv_hw, const int* pad_of, const float* W1,
                                const float* b1, const float* gamma, const float* beta, float eps, const float* W2, const float* b2,
                                float* dW1, float* db1, float* dgamma, float* dbeta, float* dW2, float* db2, float* dpool, int B, int C, int S,
                                int CA, srk_stream_t stream);
int srk_chan_attn_matrix_fwd(const float* partial, int nchunk, const float* temperature, float* gram, float* A, int B, int num_heads, int dh,
                             srk_stream_t stream);
int srk_chan_attn_matrix_bwd(const float* dpartial, int nchunk, const float* gram, const float* A, const float* temperature, float* dG, float* dGt,
                             float* dsq2, float* dsk2, float* dtemp, int B, int num_heads, int dh, srk_stream_t stream);

/* ---- whole-model executor: SwinIR.forward / backward  (network_swinir.py:805-840) ------------------- */
enum { SRK_UPSAMPLER_PIXELSHUFFLE = 1,          /* classical SR           (network_swinir.py:740-745, :813-817) */
       SRK_UPSAMPLER_PIXELSHUFFLEDIRECT = 2,    /* lightweight SR         (:746-749, :818-822) */
       SRK_UPSAMPLER_NEAREST_CONV = 3,          /* real-world SR, x2 / x4 (:750-759, :823-831) */
       SRK_UPSAMPLER_NONE = 4 };                /* denoising / JPEG artefact reduction, upscale 1 (:760-762, :832-836) */
enum { SRK_RESI_1CONV = 0, SRK_RESI_3CONV = 1 };   /* RSTB / conv_after_body residual connection (:464-471, :728-736) */

typedef struct {
  int img_size;          /* constructor img_size // patch_size: only its relation to window_size matters (:193-196) */
  int in_chans;          /* 1 or 3 */
  int embed_dim;         /* C, <= 256 */
  int num_layers;        /* len(depths), <= 16 */
  int depths[16];
  int num_heads[16];     /* C / num_heads <= 32 */
  int window_size;       /* must be 8 */
  int hidden_dim;        /* int(C * mlp_ratio) */
  int upscale;
  int upsampler;         /* SRK_UPSAMPLER_* */
  float img_range;
  float mean[3];         /* (0.4488, 0.4371, 0.4040) for 3-channel input, 0 otherwise (:658-662) */
  float qk_scale;        /* <= 0: head_dim ** -0.5 */
  int resi_connection;   /* SRK_RESI_* */
  int ape;               /* constructor ape: != 0 -> parameter absolute_pos_embed [1][img_size^2][C] (first in the parameter table, as in
                            named_parameters()), added to the tokens after patch_embed.norm (network_swinir.py:678-689, :793-795); the
                            input must then have exactly img_size x img_size tokens, as in the reference */
  int use_checkpoint;    /* constructor use_checkpoint (network_swinir.py:397-405 wraps every block in torch.utils.checkpoint): != 0 ->
                            a training forward keeps, per block, only what cannot be recomputed cheaply (norm1 / norm2 outputs, the
                            residual rows, statistics); the attention output and the MLP's u / h = gelu(u) live in ONE shared set of
                            buffers and the backward pass re-runs the block's fused attention-forward and MLP-forward kernels to refill
                            them (same kernels, same inputs: gradients are bit-identical to use_checkpoint = 0; -250 MB per block at
                            cfg3 bs 32, + two forward launches per block in backward) */
} srk_swinir_config;

typedef struct srk_swinir_plan srk_swinir_plan;

/* Unsupported configurations return SRK_E_UNSUPPORTED with a message (no CPU fallback exists). */
int srk_swinir_plan_create(const srk_swinir_config* cfg, srk_swinir_plan** plan);
void srk_swinir_plan_destroy(srk_swinir_plan* plan);
/* Per-plan option values (names and values as srk_set_option): they apply to every later call on this plan (pack, workspace_bytes,
 * forward, forward_features, backward) and to nothing else.  Set them before the first srk_swinir_workspace_bytes query: some options
 * change the workspace layout.  get: *is_set = 1 when the plan carries the value, 0 when the process-wide value applies. */
int srk_swinir_plan_set_option(srk_swinir_plan* plan, const char* name, int value);
int srk_swinir_plan_get_option(const srk_swinir_plan* plan, const char* name, int* value, int* is_set);

/* Flat parameter buffer: fp32, every tensor in the reference's state_dict layout at a 64-float aligned
 * offset, in named_parameters() order. */
int64_t srk_swinir_param_floats(const srk_swinir_plan* plan);
int srk_swinir_param_count(const srk_swinir_plan* plan);
/* name: reference state_dict key; shape: up to 4 dims (ndim returned) */
int srk_swinir_param_info(const srk_swinir_plan* plan, int index, const char** name, int64_t* offset, int64_t* numel,
                          int* ndim, int64_t shape[4]);

/* Device-resident constant tables the plan needs (pack descriptors): size, then upload into caller memory. */
size_t srk_swinir_const_bytes(const srk_swinir_plan* plan);
int srk_swinir_const_init(srk_swinir_plan* plan, void* const_dev, srk_stream_t stream);

/* Packed (padded bf16 + fp32 side tables) weights: refresh after every parameter update. */
size_t srk_swinir_packed_bytes(const srk_swinir_plan* plan);
int srk_swinir_pack(srk_swinir_plan* plan, const float* params, void* packed, srk_stream_t stream);

/* Workspace for one forward (+ backward if training != 0) at batch B and (unpadded) input size H0 x W0. */
size_t srk_swinir_workspace_bytes(const srk_swinir_plan* plan, int B, int H0, int W0, int training);

/* x fp32 NCHW [B][in_chans][H0][W0] in [0,1]  ->  y fp32 NCHW [B][in_chans][H0*s][W0*s].
 * training != 0 keeps the activations the backward needs in `workspace`.
 * drop_scale: null, or fp32 [n_blocks][2][B] per-sample DropPath factors (0 or 1/keep) for the
 * attention and MLP residual branches (timm DropPath in SwinTransformerBlock :276-277). */
int srk_swinir_forward(srk_swinir_plan* plan, const float* params, const void* packed, const float* x, float* y,
                       void* workspace, int B, int H0, int W0, int training, const float* drop_scale, srk_stream_t stream);

/* Backward in segments so the caller can overlap the gradient all-reduce of finished segments:
 * segment 0 = reconstruction tail + conv_after_body + final norm, 1..L = RSTB L-1..0, L+1 = patch-embed
 * norm + conv_first.  Segments must be run in increasing order after a training forward with the same
 * (B,H0,W0,workspace,drop_scale).  d_y fp32 NCHW like y.  grads: flat fp32 like params, ACCUMULATED.
 * After segment s, grads[begin,end) of srk_swinir_segment_range(s) are final. */
int srk_swinir_num_segments(const srk_swinir_plan* plan);
int srk_swinir_segment_range(const srk_swinir_plan* plan, int segment, int64_t* begin, int64_t* end);
int srk_swinir_backward(srk_swinir_plan* plan, const float* params, const void* packed, float* grads, const float* d_y,
                        void* workspace, int B, int H0, int W0, const float* drop_scale, int seg_begin, int seg_end,
                        srk_stream_t stream);

/* SwinIR.forward_features (network_swinir.py:790-803) as an inference entry of its own: f fp32 NCHW [B][embed_dim][H][W]
 * (what conv_first produced) -> patch_embed norm -> RSTBs -> final norm -> out fp32 NCHW [B][embed_dim][H][W].  H and W
 * must be multiples of window_size (the reference's window_partition needs the same).  workspace: as for a forward with
 * training == 0 at (B, H, W).  No gradient path (inference only). */
int srk_swinir_forward_features(srk_swinir_plan* plan, const float* params, const void* packed, const float* f, float* out,
                                void* workspace, int B, int H, int W, srk_stream_t stream);

/* Debug/parity access: byte offset and byte size of a named activation inside `workspace` for the
 * geometry of the last srk_swinir_workspace_bytes() query; returns SRK_E_STATE if unknown. */
int srk_swinir_workspace_lookup(const srk_swinir_plan* plan, const char* name, size_t* offset, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SRK_H_ */
