// Device-side evaluation metrics:
//   srk_eval_psnr   evaluate.py:24-29: per image 20 log10(max / sqrt(max(mse, 1e-10))), NO clamp; optionally the batch mean
//   srk_ssim        pytorch_msssim.ssim as the reference calls it (train.py:169, evaluate.py:127,195; package pinned at 1.0.0 in
//                   sr_environment.yml:165 but absent from the reference tree and from this image): Wang et al. 2004 with an
//                   11-tap Gaussian (sigma 1.5) applied separably as a VALID depth-wise filter, K = (0.01, 0.03), per-channel
//                   mean of the SSIM map, then mean over channels.  Restated from the published algorithm: PARITY UNPINNED
//                   (no reference-produced fixture exists); tests compare with the torch-operator form in metrics.py and an
//                   independent scipy evaluation.
// Sums are formed in a fixed order (partials + one finishing workgroup): results are reproducible.
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void sqerr_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ partial,
                                                            long long per_image) {
  __shared__ float red[4];
  const long long base = (long long)blockIdx.y * per_image;
  float sq = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image; i += (long long)gridDim.x * blockDim.x) {
    const float d = x[base + i] - y[base + i];
    sq = fmaf(d, d, sq);
  }
  sq = wave_sum64(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void eval_psnr_finish_kernel(const float* __restrict__ partial, int chunks, int B, long long per_image, float max_val,
                                        float* __restrict__ psnr, float* __restrict__ mean_out) {
  __shared__ float sh[1024];
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float sq = 0.f;
    for (int c = 0; c < chunks; ++c) sq += partial[(long long)b * chunks + c];
    const float mse = fmaxf(sq / (float)per_image, 1e-10f);
    const float v = 20.0f * log10f(max_val / sqrtf(mse));
    if (psnr) psnr[b] = v;
    sh[b] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0 && mean_out) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sh[b];
    *mean_out = s / (float)B;
  }
}

// ---- SSIM ----------------------------------------------------------------------------------------------------------------
constexpr int ST = 32;            // output tile (valid positions) per workgroup: 32 x 32
constexpr int SI = ST + 10;       // input tile 42 x 42
struct GaussWin { float w[11]; };

__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ partial,
                                                        int H, int W, int tiles_x, int tiles_y, GaussWin gw, float C1, float C2) {
  __shared__ float xs[SI][SI + 1], ys[SI][SI + 1];
  __shared__ float hb[5][SI][ST + 1];        // horizontally filtered x, y, xx, yy, xy
  __shared__ float red[4];
  const int plane = blockIdx.z;                                 // b * C + c
  const int ty0 = blockIdx.y * ST, tx0 = blockIdx.x * ST;       // first valid output position of the tile
  const int OH = H - 10, OW = W - 10;
  const float* xp = X + (long long)plane * H * W;
  const float* yp = Y + (long long)plane * H * W;
  const int tid = threadIdx.x;
  for (int i = tid; i < SI * SI; i += 256) {
    const int r = i / SI, c = i - r * SI;
    const int gy = ty0 + r, gx = tx0 + c;
    const bool ok = gy < H && gx < W;
    xs[r][c] = ok ? xp[(long long)gy * W + gx] : 0.f;
    ys[r][c] = ok ? yp[(long long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < SI * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = xs[r][c + k], yv = ys[r][c + k], wk = gw.w[k];
      a = fmaf(wk, xv, a);
      b = fmaf(wk, yv, b);
      aa = fmaf(wk, xv * xv, aa);
      bb = fmaf(wk, yv * yv, bb);
      ab = fmaf(wk, xv * yv, ab);
    }
    hb[0][r][c] = a; hb[1][r][c] = b; hb[2][r][c] = aa; hb[3][r][c] = bb; hb[4][r][c] = ab;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = tid; i < ST * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    if (ty0 + r >= OH || tx0 + c >= OW) continue;
    float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float wk = gw.w[k];
      m1 = fmaf(wk, hb[0][r + k][c], m1);
      m2 = fmaf(wk, hb[1][r + k][c], m2);
      xx = fmaf(wk, hb[2][r + k][c], xx);
      yy = fmaf(wk, hb[3][r + k][c], yy);
      xy = fmaf(wk, hb[4][r + k][c], xy);
    }
    const float s11 = xx - m1 * m1, s22 = yy - m2 * m2, s12 = xy - m1 * m2;
    const float cs = (2.f * s12 + C2) / (s11 + s22 + C2);
    acc += ((2.f * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1)) * cs;
  }
  acc = wave_sum64(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) partial[((long long)plane * tiles_y + blockIdx.y) * tiles_x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void ssim_finish_kernel(const float* __restrict__ partial, int tiles, int B, int C, float inv_count, float* __restrict__ per_image,
                                   float* __restrict__ mean_out) {
  __shared__ float sh[1024];
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      float pc = 0.f;
      for (int t = 0; t < tiles; ++t) pc += partial[((long long)b * C + c) * tiles + t];
      s += pc * inv_count;                       // per-channel mean of the map
    }
    const float v = s / (float)C;
    if (per_image) per_image[b] = v;
    sh[b] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0 && mean_out) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sh[b];
    *mean_out = s / (float)B;
  }
}

}  // namespace

extern "C" {

int64_t srk_eval_psnr_workspace(int64_t per_image, int B) {
  if (per_image <= 0 || B <= 0) return 0;
  return (int64_t)sizeof(float) * B * srk_batch_psnr_chunks(per_image);
}

int srk_eval_psnr(const float* x, const float* y, void* workspace, int B, int64_t per_image, float max_val, float* psnr, float* mean,
                  srk_stream_t stream) {
  SRK_REQUIRE(x && y && workspace, SRK_E_NULL, "eval_psnr: null pointer");
  SRK_REQUIRE(B > 0 && B <= 1024 && per_image > 0 && max_val > 0.f, SRK_E_SHAPE, "eval_psnr: B=%d (1..1024) per_image=%lld", B, (long long)per_image);
  const int chunks = srk_batch_psnr_chunks(per_image);
  hipLaunchKernelGGL(sqerr_partial_kernel, dim3(chunks, B), dim3(256), 0, (hipStream_t)stream, x, y, static_cast<float*>(workspace), per_image);
  hipLaunchKernelGGL(eval_psnr_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace), chunks, B,
                     per_image, max_val, psnr, mean);
  return srk_check_launch("eval_psnr");
}

int64_t srk_ssim_workspace(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H < 11 || W < 11) return 0;
  return (int64_t)sizeof(float) * B * C * ((H - 10 + ST - 1) / ST) * ((W - 10 + ST - 1) / ST);
}

int srk_ssim(const float* x, const float* y, void* workspace, int B, int C, int H, int W, float data_range, float* per_image, float* mean,
             srk_stream_t stream) {
  SRK_REQUIRE(x && y && workspace, SRK_E_NULL, "ssim: null pointer");
  SRK_REQUIRE(B > 0 && B <= 1024 && C > 0 && (long long)B * C < 65536, SRK_E_SHAPE, "ssim: B=%d C=%d", B, C);
  SRK_REQUIRE(H >= 11 && W >= 11, SRK_E_UNSUPPORTED, "ssim: the 11-tap window needs H, W >= 11 (got %dx%d)", H, W);
  SRK_REQUIRE(data_range > 0.f, SRK_E_SHAPE, "ssim: data_range=%g", (double)data_range);
  GaussWin gw;
  float sum = 0.f;
  for (int i = 0; i < 11; ++i) {
    const float c = (float)(i - 5);
    gw.w[i] = expf(-(c * c) / (2.0f * 1.5f * 1.5f));
    sum += gw.w[i];
  }
  for (int i = 0; i < 11; ++i) gw.w[i] /= sum;
  const int tx = (W - 10 + ST - 1) / ST, ty = (H - 10 + ST - 1) / ST;
  const float C1 = (0.01f * data_range) * (0.01f * data_range), C2 = (0.03f * data_range) * (0.03f * data_range);
  hipLaunchKernelGGL(ssim_tile_kernel, dim3(tx, ty, B * C), dim3(256), 0, (hipStream_t)stream, x, y, static_cast<float*>(workspace), H, W, tx, ty,
                     gw, C1, C2);
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace), tx * ty, B, C,
                     1.0f / ((float)(H - 10) * (float)(W - 10)), per_image, mean);
  return srk_check_launch("ssim");
}

}  // extern "C"
