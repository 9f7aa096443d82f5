// Internal launcher prototypes shared by api.hip and swinir.hip.
#pragma once
#include "common.h"
#include "gemm.h"
#include "pack.h"
#include "wgrad.h"

// attn_fused.hip: projection + attention in one kernel (classical width); SRK_NOT_COVERED -> run them separately
int srk_launch_qkv_attn_fwd(const bf16_t* xn, int lda, const bf16_t* Wt, const float* bias, float scale, bf16_t* qkv, const float* biasd,
                            bf16_t* ao, long long B_, int nH, int CA, int K, WinGeom geom, hipStream_t stream);
void srk_attn_fused_enable(int on);
int srk_attn_fused_mode();
// block_light.hip: one whole Swin block of the light width per launch (inference); SRK_NOT_COVERED when the shape is another
void srk_block_light_enable(int on);
int srk_block_light_enabled();
int srk_launch_swin_block_light(const float* x, float* y, bf16_t* yb, const float* n1w, const float* n1b, const float* n2w, const float* n2b,
                                const bf16_t* Wqkv, const bf16_t* Wproj, const bf16_t* W1, const bf16_t* W2, const float* bqkv,
                                const float* bproj, const float* b1, const float* b2, const float* biasd, float scale, int C, int CP, int HP,
                                int nH, int dh, int HID, long long B_, WinGeom geom, hipStream_t stream);
int srk_launch_attn_fwd(const bf16_t* qkv, const float* biasd, bf16_t* ao, long long B_, int nH, WinGeom geom, hipStream_t stream);
int srk_attn_bwd_slabs(long long B_, int nH, int* wpw_out);
int srk_launch_attn_bwd(const bf16_t* qkv, const float* biasd, const bf16_t* dao, bf16_t* dqkv, float* dbias_slab,
                        float* dtable, long long B_, int nH, WinGeom geom, float scale, hipStream_t stream);
int srk_launch_rpb_expand(const float* table, float* biasd, int nH, hipStream_t stream);
// attn_bwd_fused.hip: attention backward of one window per workgroup pass with q/k/v re-projected from xn1 and the output-projection
// dgrad folded in (classical width); SRK_NOT_COVERED -> dproj GEMM + srk_launch_attn_bwd on saved q/k/v
void srk_attn_bwd_fused_enable(int on);
int srk_attn_bwd_fused_enabled();
int srk_qkv_attn_bwd_slabs(long long B_, int nH, int CA, int K);    // d(bias) slabs it writes; 0 = not covered
int srk_launch_qkv_attn_bwd(const bf16_t* xn, int lda, const bf16_t* Wqkv, const float* bqkv, float scale, const bf16_t* g, int ldg,
                            const bf16_t* WprojT, const float* biasd, bf16_t* dqkv, float* slab, long long B_, int nH, int CA, int K,
                            WinGeom geom, hipStream_t stream);

int srk_launch_ln_fwd(const float* x, const float* gamma, const float* beta, bf16_t* yb, float* yf, float* mean,
                      float* rstd, int rows, int C, int CP, const WinGeom* geom, hipStream_t stream);
int srk_launch_ln_bwd(const bf16_t* dyb, const float* x, const float* mean, const float* rstd, const float* gamma,
                      float* gx, bf16_t* gxb, float* dgamma, float* dbeta, int rows, int C, int CP, const WinGeom* geom,
                      int dy_by_m, int stats_by_m, int out_by_m, int accumulate, const float* rowscale, int rows_per_sample,
                      hipStream_t stream);

int srk_launch_window_partition(const void* x, void* out, int B, int H, int W, int C, int ws, int elem_bytes, int reverse, hipStream_t stream);
int srk_launch_roll2d(const void* x, void* out, int B, int H, int W, int C, int sh, int sw, int elem_bytes, hipStream_t stream);
int srk_launch_pixel_shuffle(const void* x, void* out, int B, int C, int H, int W, int r, int elem_bytes, hipStream_t stream);
int srk_launch_shift_mask(float* mask, int H, int W, int ws, int shift, hipStream_t stream);
int srk_launch_rel_pos_index(long long* out, int ws, hipStream_t stream);
int srk_launch_img_prep(const float* x, float* out, int B, int Cimg, int H0, int W0, int H, int W, float range, const float* mean, hipStream_t stream);
int srk_launch_stem_conv(const float* in, const float* wgt, const float* bias, float* out, int B, int H, int W, int Cin, int C, int CP, hipStream_t stream);
int srk_launch_stem_wgrad(const float* in, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int C, int CP, hipStream_t stream);
int srk_launch_img_grad_prep(const float* dpred, float* gy, int B, int Cimg, int Hc, int Wc, int H, int W, int r, int CoP, float inv_range, hipStream_t stream);
int srk_launch_smallconv_dgrad(const float* gy, const float* wgt, bf16_t* dx, int B, int H, int W, int Cin, int CinP, int Co, int CoP, hipStream_t stream);
int srk_launch_smallconv_wgrad(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int CinP, int Co, int CoP, hipStream_t stream);
// workgroups per image of the PSNR partial pass (8192 elements each, at most 64)
inline int srk_batch_psnr_chunks(long long per_image) {
  const long long c = (per_image + 8191) / 8192;
  return (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
}
int srk_launch_crop_u8(const unsigned char* pool, const long long* desc, float* out, int B, int patch, hipStream_t stream);
int srk_launch_batch_psnr(const float* pred, const float* target, float* partial, int B, long long per_image, float max_val,
                          float* psnr, float* psnr_sum, float* abs_sum, hipStream_t stream);
int srk_launch_zero_f32(float* p, long long n, hipStream_t stream);      // graph-safe zero fill (misc.hip)
int srk_launch_add_f32_bf16(float* a, const float* b, bf16_t* ab, long long n, hipStream_t stream);
int srk_launch_add_bf16_into_f32(float* a, const bf16_t* b, long long n, hipStream_t stream);
int srk_launch_cast_f32_bf16(const float* a, bf16_t* out, long long n, hipStream_t stream);
int srk_launch_ape_add(float* x, const float* ape, int B, int L, int C, int CP, hipStream_t stream);
int srk_launch_ape_grad(const float* gx, float* dape, int B, int L, int C, int CP, hipStream_t stream);
int srk_launch_dlrelu_bf16(bf16_t* g, const bf16_t* act, float slope, long long n, hipStream_t stream);
int srk_launch_nn2x_bf16(const bf16_t* in, bf16_t* out, int B, int h, int w, int C, hipStream_t stream);
int srk_launch_nn2x_sum_dlrelu(const bf16_t* g, const bf16_t* act, bf16_t* out, int B, int h, int w, int C, float slope, hipStream_t stream);
int srk_launch_nchw_tokens(const float* src, float* dst, int B, int C, int CP, int HW, int to_tokens, hipStream_t stream);
int srk_launch_l1_loss(const float* pred, const float* target, float* dpred, float* loss_sum, unsigned* nonfinite, long long n, float grad_scale, hipStream_t stream);
int srk_launch_sumsq(const float* g, long long n, float* out, hipStream_t stream);
int srk_launch_adamw(float* p, const float* g, float* m, float* v, long long n, const float* sumsq, const int* nonfinite, float max_norm, float grad_div, float lr, float beta1, float beta2, float eps, float wd, int step, hipStream_t stream);
int srk_launch_probe_trread(const bf16_t* in, bf16_t* out, hipStream_t stream);
int srk_launch_win256_attn_fwd(const bf16_t* qkv, int ldq, int CA, const float* bias, int table_rows, bf16_t* out, int ldo, int B, int H, int W, int wh, int ww, int sy, int sx, int nH, float scale, int overlap, hipStream_t stream);
int srk_launch_win_attn_fwd_padded(const bf16_t* qkv, int ldq, int CA, const float* bias, int table_rows, bf16_t* out, int ldo, int B, int H, int W, int Hp, int Wp, int wh, int ww, int sy, int sx, int nH, float scale, int overlap, hipStream_t stream);
// dat.hip: Gram partials of the channel attention on the matrix cores (also behind srk_chan_gram of dat_train.hip)
int srk_launch_chan_gram(const bf16_t* x, int ldx, const bf16_t* y, int ldy, float* partial, int B, int N, int nH, hipStream_t stream);
