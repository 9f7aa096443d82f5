// Memory-bound helper kernels: bit-exact index ops (window partition / reverse / roll / pixel
// shuffle / mask / relative-position index), weight pack / gradient unpack, image pre/post
// processing, the 3-channel stem conv, small-Cout conv gradients, L1 loss, fused AdamW + global-norm
// clip, and the on-device probe of the transposing LDS read.
#include "kernels.h"
#include "pack.h"
#include "wgrad.h"

namespace {

// ------------------------------------------------------------------------------------------------
// index ops (reference network_swinir.py:33-62, :249-252, :269-272, nn.PixelShuffle) -- pure copies
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void window_partition_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C, int ws,
                                        int reverse) {
  const long long total = (long long)B * H * W * C;
  const int nWw = W / ws, nW = (H / ws) * nWw, N = ws * ws;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long long row = idx / C;                  // window-order row
    const int p = (int)(row % N);
    const long long b_ = row / N;
    const int w = (int)(b_ % nW);
    const long long b = b_ / nW;
    const int y = (w / nWw) * ws + p / ws, xx = (w % nWw) * ws + p % ws;
    const long long src = ((b * H + y) * W + xx) * C + c;
    if (reverse) out[src] = x[idx]; else out[idx] = x[src];
  }
}

template <typename T>
__global__ void roll2d_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C, int sh, int sw) {
  const long long total = (long long)B * H * W * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long b = t / H;
    const int hs = ((h - sh) % H + H) % H, wsrc = ((w - sw) % W + W) % W;   // out[h,w] = x[(h-sh)%H, (w-sw)%W]
    out[idx] = x[((b * H + hs) * W + wsrc) * C + c];
  }
}

template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int C, int H, int W, int r) {
  // out[b, c, h*r+i, w*r+j] = in[b, c*r*r + i*r + j, h, w]
  const long long total = (long long)B * C * r * r * H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % (W * r));
    long long t = idx / (W * r);
    const int oy = (int)(t % (H * r)); t /= (H * r);
    const int c = (int)(t % C);
    const long long b = t / C;
    const int h = oy / r, i = oy % r, w = ox / r, j = ox % r;
    out[idx] = x[((b * C * r * r + (c * r * r + i * r + j)) * H + h) * W + w];
  }
}

__global__ void shift_mask_kernel(float* __restrict__ mask, int H, int W, int ws, int shift) {
  const int nWw = W / ws, nW = (H / ws) * nWw, N = ws * ws;
  const long long total = (long long)nW * N * N;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % N), p = (int)((idx / N) % N), w = (int)(idx / ((long long)N * N));
    auto lab = [&](int t) {
      const int y = (w / nWw) * ws + t / ws, x = (w % nWw) * ws + t % ws;
      const int ly = y < H - ws ? 0 : (y < H - shift ? 1 : 2);
      const int lx = x < W - ws ? 0 : (x < W - shift ? 1 : 2);
      return ly * 3 + lx;
    };
    mask[idx] = lab(p) != lab(q) ? -100.0f : 0.0f;
  }
}

__global__ void rel_pos_index_kernel(long long* __restrict__ out, int ws) {
  const int N = ws * ws;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  const int p = idx / N, q = idx % N;
  out[idx] = (long long)((p / ws - q / ws + ws - 1) * (2 * ws - 1) + (p % ws - q % ws + ws - 1));
}

// ------------------------------------------------------------------------------------------------
// pack / unpack
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pack_map(const PackDesc& d, int n, int k, long long& idx) {
  int o;
  if (d.nmap == NM_DIRECT) {
    if (n >= d.Nreal) return false;
    o = n;
  } else if (d.nmap == NM_QKV) {
    const int which = n / d.CA, rem = n - which * d.CA;
    const int h = rem >> 5, dd = rem & 31;
    if (which >= 3 || dd >= d.dh) return false;
    o = which * (d.nH * d.dh) + h * d.dh + dd;
  } else {
    const int ij = n / d.Cs, c = n - ij * d.Cs;
    if (ij >= d.r * d.r) return false;
    o = c * d.r * d.r + ij;
    if (o >= d.Nreal) return false;
  }
  if (d.kind == PK_VEC) {
    idx = o;
    return true;
  }
  if (d.kind == PK_LINEAR) {
    int i;
    if (d.kmap == KM_DIRECT) {
      if (k >= d.Kreal) return false;
      i = k;
    } else {
      const int h = k >> 5, dd = k & 31;
      if (h >= d.nH || dd >= d.dh) return false;
      i = h * d.dh + dd;
    }
    idx = (long long)o * d.Kreal + i;
    return true;
  }
  const int tap = k / d.CinP, ci = k - tap * d.CinP;   // PK_CONV
  if (ci >= d.Kreal) return false;
  idx = ((long long)o * d.Kreal + ci) * 9 + tap;
  return true;
}

__device__ __forceinline__ int find_desc(const PackDesc* descs, int ndesc, int blk) {
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* __restrict__ descs, int ndesc,
                                                   const float* __restrict__ params, bf16_t* __restrict__ packed,
                                                   float* __restrict__ side) {
  const int di = find_desc(descs, ndesc, blockIdx.x);
  const PackDesc d = descs[di];
  const long long base = (long long)(blockIdx.x - d.blk0) * 1024 + threadIdx.x;
  if (d.kind == PK_RPB) {
    const int total = d.nH * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = base + 256 * u;
      if (e >= total) break;
      const int h = (int)(e >> 12), i = (int)((e >> 6) & 63), j = (int)(e & 63);
      const int t = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
      side[d.dst + e] = params[d.src + t * d.nH + h];
    }
    return;
  }
  if (d.kind == PK_VEC) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = base + 256 * u;
      if (e >= d.NP) break;
      long long idx;
      side[d.dst + e] = pack_map(d, (int)e, 0, idx) ? params[d.src + idx] : 0.f;
    }
    return;
  }
  const long long total = (long long)d.NP * d.KP;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long e = base + 256 * u;   // destination-linear index
    if (e >= total) break;
    int n, k;
    if (!d.transpose) {
      n = (int)(e / d.KP);
      k = (int)(e - (long long)n * d.KP);
    } else if (d.kind == PK_LINEAR) {
      k = (int)(e / d.NP);
      n = (int)(e - (long long)k * d.NP);
    } else {  // conv transposed: dst[ci][tap'][n], tap' = 8 - tap
      n = (int)(e % d.NP);
      const long long t2 = e / d.NP;
      const int tapf = (int)(t2 % 9), ci = (int)(t2 / 9);
      k = (8 - tapf) * d.CinP + ci;
    }
    long long idx;
    const float v = pack_map(d, n, k, idx) ? params[d.src + idx] : 0.f;
    packed[d.dst + e] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void unpack_kernel(const PackDesc* __restrict__ descs, int ndesc,
                                                     const float* __restrict__ gw, const float* __restrict__ gside,
                                                     float* __restrict__ grads) {
  const int di = find_desc(descs, ndesc, blockIdx.x);
  const PackDesc d = descs[di];
  if (d.transpose || d.kind == PK_RPB) return;
  const long long base = (long long)(blockIdx.x - d.blk0) * 1024 + threadIdx.x;
  const long long total = d.kind == PK_VEC ? d.NP : (long long)d.NP * d.KP;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long e = base + 256 * u;
    if (e >= total) break;
    long long idx;
    if (d.kind == PK_VEC) {
      if (pack_map(d, (int)e, 0, idx)) grads[d.src + idx] += gside[d.dst + e];
    } else {
      const int n = (int)(e / d.KP), k = (int)(e - (long long)n * d.KP);
      if (pack_map(d, n, k, idx)) grads[d.src + idx] += gw[d.dst + e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// image pre-processing and the stem conv (network_swinir.py:783-788, :809-810, :814)
// ------------------------------------------------------------------------------------------------
// x NCHW fp32 [B][Cimg][H0][W0] -> NHWC fp32 [B][H][W][4]: reflect pad bottom/right, (x-mean)*range
__global__ void img_prep_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int Cimg, int H0, int W0,
                                int H, int W, float range, float m0, float m1, float m2) {
  const long long total = (long long)B * H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const long long b = idx / ((long long)W * H);
    const int ys = y < H0 ? y : 2 * (H0 - 1) - y, xs = xx < W0 ? xx : 2 * (W0 - 1) - xx;   // 'reflect'
    const float mean[3] = {m0, m1, m2};
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cimg && c < 4; ++c) v[c] = (x[((b * Cimg + c) * H0 + ys) * W0 + xs] - mean[c < 3 ? c : 0]) * range;
    *reinterpret_cast<float4*>(out + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// conv_first: in NHWC fp32 [B][H][W][4], weight fp32 [C][Cin][3][3], out fp32 [B*H*W][CP] (pad cols 0)
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B,
                                                        int H, int W, int Cin, int C, int CP) {
  extern __shared__ float sm[];
  float* wl = sm;                 // [36][CP]  (tap*4 + ci)
  float* pl = sm + 36 * CP;       // [16][36]
  for (int i = threadIdx.x; i < 36 * CP; i += 256) {
    const int kk = i / CP, c = i % CP;
    const int tap = kk >> 2, ci = kk & 3;
    wl[i] = (c < C && ci < Cin) ? wgt[((c * Cin) + ci) * 9 + tap] : 0.f;
  }
  const long long pix0 = (long long)blockIdx.x * 16;
  const long long npix = (long long)B * H * W;
  for (int i = threadIdx.x; i < 16 * 9; i += 256) {
    const int pp = i / 9, tap = i % 9;
    const long long pix = pix0 + pp;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix < npix) {
      const int xx = (int)(pix % W), y = (int)((pix / W) % H);
      const long long b = pix / ((long long)W * H);
      const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
      if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W)
        v = *reinterpret_cast<const float4*>(in + ((b * H + yy) * W + xs) * 4);
    }
    *reinterpret_cast<float4*>(pl + pp * 36 + tap * 4) = v;
  }
  __syncthreads();
  const int q = CP / 4;
  for (int item = threadIdx.x; item < 16 * q; item += 256) {
    const int pp = item / q, c4 = (item % q) * 4;
    const long long pix = pix0 + pp;
    if (pix >= npix) continue;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int kk = 0; kk < 36; ++kk) {
      const float pv = pl[pp * 36 + kk];
      const float4 wv = *reinterpret_cast<const float4*>(wl + kk * CP + c4);
      a0 += pv * wv.x; a1 += pv * wv.y; a2 += pv * wv.z; a3 += pv * wv.w;
    }
    a0 += c4 < C ? bias[c4] : 0.f;
    a1 += c4 + 1 < C ? bias[c4 + 1] : 0.f;
    a2 += c4 + 2 < C ? bias[c4 + 2] : 0.f;
    a3 += c4 + 3 < C ? bias[c4 + 3] : 0.f;
    *reinterpret_cast<float4*>(out + pix * CP + c4) = make_float4(a0, a1, a2, a3);
  }
}

// conv_first weight/bias gradient: dW[c][ci][tap] += sum_pix gy[pix][c] * in[pix+off][ci]
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gy,
                                                         float* __restrict__ dW, float* __restrict__ db, int B, int H,
                                                         int W, int Cin, int C, int CP, int pix_per_block) {
  __shared__ float pl[64 * 36];
  const long long npix = (long long)B * H * W;
  const long long p_begin = (long long)blockIdx.x * pix_per_block;
  const long long p_end = p_begin + pix_per_block < npix ? p_begin + pix_per_block : npix;
  const int c = threadIdx.x;
  float acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.f;
  float accb = 0.f;
  for (long long pb = p_begin; pb < p_end; pb += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 9; i += 256) {
      const int pp = i / 9, tap = i % 9;
      const long long pix = pb + pp;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pix < p_end) {
        const int xx = (int)(pix % W), y = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W)
          v = *reinterpret_cast<const float4*>(in + ((b * H + yy) * W + xs) * 4);
      }
      *reinterpret_cast<float4*>(pl + pp * 36 + tap * 4) = v;
    }
    __syncthreads();
    if (c < C) {
      const int cnt = (int)(p_end - pb < 64 ? p_end - pb : 64);
      // eight gradient loads in flight (one dependent load per pixel made this a chain of memory round trips), patch
      // values read as float4 broadcasts
      for (int pp0 = 0; pp0 < cnt; pp0 += 8) {
        float g8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g8[u] = pp0 + u < cnt ? gy[(pb + pp0 + u) * CP + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float g = g8[u];
          accb += g;
          const float* pr = pl + (pp0 + u) * 36;       // rows beyond cnt hold zeros (staged above)
#pragma unroll
          for (int t4 = 0; t4 < 9; ++t4) {
            const float4 v = *reinterpret_cast<const float4*>(pr + 4 * t4);
            acc[4 * t4] += g * v.x; acc[4 * t4 + 1] += g * v.y; acc[4 * t4 + 2] += g * v.z; acc[4 * t4 + 3] += g * v.w;
          }
        }
      }
    }
  }
  if (c < C) {
    for (int tap = 0; tap < 9; ++tap)
      for (int ci = 0; ci < Cin; ++ci) atomicAdd(dW + (c * Cin + ci) * 9 + tap, acc[tap * 4 + ci]);
    atomicAdd(db + c, accb);
  }
}

// conv_first weight / bias gradient on the matrix cores.  dW[c][tap*4 + ci] = sum_pix gy[pix][c] * patch[pix][tap*4 + ci] is a
// (C x 48) x pixels GEMM: per 32-pixel chunk the gy rows (fp32, split into bf16 hi + lo) and the im2col patch rows (36 values +
// a column of ones that yields the bias gradient, hi + lo) are staged in LDS and read back as transposed fragments (k = pixel);
// three MFMAs per product (hi.hi + hi.lo + lo.hi) keep ~16 mantissa bits, as accurate as the fp32 VALU kernel above for these
// 131 072-term sums.  Every workgroup writes ONE partial [192][48] tile to the caller's scratch; stem_wgrad_reduce_kernel sums
// them in a fixed order (the VALU kernel ended in 5 040 contended atomics per workgroup and took 203 us for 100 MB of input).
constexpr int SW_GS = 200;                  // LDS row strides (elements): 192 + 8 and 48 + 8 keep the transposing reads conflict-free
constexpr int SW_PS = 56;
__device__ __forceinline__ bf16x8_t sw_frag(const bf16_t* tile, int stride, int c0, int lane) {   // T[k = 8 g + jj][c0 + r16]
  const int g = lane >> 4;
  const bf16x4_t lo = lds_tr_read(tr_addr(tile, stride, 8 * g, c0, lane)), hi = lds_tr_read(tr_addr(tile, stride, 8 * g + 4, c0, lane));
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void sw_split4(float4 v, uint2& hi, uint2& lo) {
  hi = pack_bf4(v.x, v.y, v.z, v.w);
  float h0, h1, h2, h3;
  unpack_bf2(hi.x, h0, h1);
  unpack_bf2(hi.y, h2, h3);
  lo = pack_bf4(v.x - h0, v.y - h1, v.z - h2, v.w - h3);
}
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ in, const float* __restrict__ gy,
                                                              float* __restrict__ partial, int B, int H, int W, int CP) {
  __shared__ __attribute__((aligned(16))) bf16_t Gh[32 * SW_GS], Gl[32 * SW_GS], Ph[32 * SW_PS], Pl[32 * SW_PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const long long npix = (long long)B * H * W;
  const int nchunk = (int)(npix / 32);
  // patch columns 36..47: a one in column 36 (bias gradient), zeros elsewhere; written once
  for (int i = tid; i < 32 * 12; i += 256) {
    const int px = i / 12, c = 36 + i % 12;
    Ph[px * SW_PS + c] = c == 36 ? (bf16_t)0x3F80 : (bf16_t)0;
    Pl[px * SW_PS + c] = 0;
  }
  f32x4_t acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const long long p0 = (long long)ch * 32;
    // gy rows: 32 x 192 fp32 = 1536 float4, six per thread
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int q = i * 256 + tid, px = q / 48, c4 = (q - px * 48) * 4;
      uint2 hi, lo;
      sw_split4(*reinterpret_cast<const float4*>(gy + (p0 + px) * CP + c4), hi, lo);
      *reinterpret_cast<uint2*>(Gh + px * SW_GS + c4) = hi;
      *reinterpret_cast<uint2*>(Gl + px * SW_GS + c4) = lo;
    }
    // patches: 32 pixels x 9 taps of one float4 (NHWC4 input), 288 pieces
    for (int q = tid; q < 288; q += 256) {
      const int px = q / 9, tap = q - px * 9;
      const long long pix = p0 + px;
      const int xx = (int)(pix % W), y = (int)((pix / W) % H);
      const long long b = pix / ((long long)W * H);
      const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W) v = *reinterpret_cast<const float4*>(in + ((b * H + yy) * W + xs) * 4);
      uint2 hi, lo;
      sw_split4(v, hi, lo);
      *reinterpret_cast<uint2*>(Ph + px * SW_PS + tap * 4) = hi;
      *reinterpret_cast<uint2*>(Pl + px * SW_PS + tap * 4) = lo;
    }
    __syncthreads();
    bf16x8_t bh[3], bl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bh[j] = sw_frag(Ph, SW_PS, 16 * j, lane);
      bl[j] = sw_frag(Pl, SW_PS, 16 * j, lane);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bf16x8_t ah = sw_frag(Gh, SW_GS, 16 * (3 * wave + i), lane), al = sw_frag(Gl, SW_GS, 16 * (3 * wave + i), lane);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // acc[i][j][e] = dW'[c = 16 (3 wave + i) + 4 g + e][n = 16 j + r16]
  float* dst = partial + (size_t)blockIdx.x * (192 * 48);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(16 * (3 * wave + i) + 4 * g + e) * 48 + 16 * j + r16] = acc[i][j][e];
}

// dW[(c*Cin + ci)*9 + tap] += sum_w partial[w][c][tap*4 + ci], db[c] += sum_w partial[w][c][36]; one workgroup per channel c,
// threads = 64 columns x 4 partial quarters, eight independent loads in flight
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int nw, float* __restrict__ dW,
                                                                float* __restrict__ db, int Cin) {
  __shared__ float part[4][64];
  const int c = blockIdx.x, n = threadIdx.x & 63, q = threadIdx.x >> 6;
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < 48) {
    const float* base = partial + (size_t)c * 48 + n;
    int w = q;
    for (; w + 28 < nw; w += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += base[(size_t)(w + 4 * u) * (192 * 48)];
    }
    for (; w < nw; w += 4) a8[0] += base[(size_t)w * (192 * 48)];
  }
  part[q][n] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  __syncthreads();
  if (q == 0 && n < 48) {
    const float v = (part[0][n] + part[1][n]) + (part[2][n] + part[3][n]);
    const int tap = n >> 2, ci = n & 3;
    if (n < 36 && ci < Cin) dW[(c * Cin + ci) * 9 + tap] += v;
    if (n == 36) db[c] += v;
  }
}

// ------------------------------------------------------------------------------------------------
// small-Cout convs (image heads: conv_last 64->3, light upsample C->r*r*3): gradients on the VALU
// ------------------------------------------------------------------------------------------------
// gy image gradient prep: dpred NCHW fp32 [B][Cimg][Hc][Wc] (cropped HR size) -> NHWC fp32
// [B][H][W][CoP] of the conv output (before PixelShuffle if r>1), scaled by inv_range; zero outside the crop.
__global__ void img_grad_prep_kernel(const float* __restrict__ dpred, float* __restrict__ gy, int B, int Cimg, int Hc,
                                     int Wc, int H, int W, int r, int CoP, float inv_range) {
  const long long total = (long long)B * H * W * CoP;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % CoP);
    long long t = idx / CoP;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const long long b = t / H;
    const int rr = r * r;
    const int c = n / rr, ij = n % rr;
    const int oy = y * r + ij / r, ox = x * r + ij % r;
    float v = 0.f;
    if (c < Cimg && oy < Hc && ox < Wc) v = dpred[((b * Cimg + c) * Hc + oy) * Wc + ox] * inv_range;
    gy[idx] = v;
  }
}

// dX[pix][ci] = sum_tap sum_co gy[pix - off(tap)][co] * W[co][ci][tap]  -> bf16 NHWC [pix][CinP]
__global__ __launch_bounds__(256) void smallconv_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ wgt,
                                                              bf16_t* __restrict__ dx, int B, int H, int W, int Cin,
                                                              int CinP, int Co, int CoP, int groups) {
  extern __shared__ float sm[];
  float* wl = sm;                       // [9][Co][CinP]
  float* gl = sm + 9 * Co * CinP;       // [16 pixels][9][CoP]
  for (int i = threadIdx.x; i < 9 * Co * CinP; i += 256) {
    const int ci = i % CinP, co = (i / CinP) % Co, tap = i / (CinP * Co);
    wl[i] = ci < Cin ? wgt[((co * Cin) + ci) * 9 + tap] : 0.f;
  }
  const long long npix = (long long)B * H * W;
  for (int grp = 0; grp < groups; ++grp) {     // the staged weights are reused for `groups` x 16 pixels
  const long long pix0 = ((long long)blockIdx.x * groups + grp) * 16;
  if (pix0 >= npix) break;
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * 9 * CoP; i += 256) {
    const int n = i % CoP, tap = (i / CoP) % 9, pp = i / (9 * CoP);
    const long long pix = pix0 + pp;
    float v = 0.f;
    if (pix < npix) {
      const int xx = (int)(pix % W), y = (int)((pix / W) % H);
      const long long b = pix / ((long long)W * H);
      // output pixel that used input pixel (y,xx) through tap (ky,kx) is (y - (ky-1), xx - (kx-1))
      const int yy = y - (tap / 3 - 1), xs = xx - (tap % 3 - 1);
      if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W) v = gy[((b * H + yy) * W + xs) * CoP + n];
    }
    gl[i] = v;
  }
  __syncthreads();
  const int q = CinP / 4;
  for (int item = threadIdx.x; item < 16 * q; item += 256) {
    const int pp = item / q, c4 = (item % q) * 4;
    const long long pix = pix0 + pp;
    if (pix >= npix) continue;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int tap = 0; tap < 9; ++tap)
      for (int co = 0; co < Co; ++co) {
        const float g = gl[(pp * 9 + tap) * CoP + co];
        const float4 wv = *reinterpret_cast<const float4*>(wl + (tap * Co + co) * CinP + c4);
        a0 += g * wv.x; a1 += g * wv.y; a2 += g * wv.z; a3 += g * wv.w;
      }
    *reinterpret_cast<uint2*>(dx + pix * CinP + c4) = pack_bf4(a0, a1, a2, a3);
  }
  }
}

// dW[co][ci][tap] += sum_q X[q][ci] * gy[q - off(tap)][co];  db[co] += sum gy.
// blockDim = CinP * PG: thread = (input channel ci, pixel group pg); a workgroup walks `pix_per_block` pixels in
// chunks of 64 whose 3x3 gy neighbourhoods are staged in LDS; each thread keeps 9*COP partial sums in registers,
// the PG groups are combined through LDS and one atomic per (co, ci, tap) per workgroup goes to memory.
template <int COP>
__global__ __launch_bounds__(256) void smallconv_wgrad_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gy,
                                                              float* __restrict__ dW, float* __restrict__ db, int B, int H,
                                                              int W, int Cin, int CinP, int Co, int pix_per_block) {
  extern __shared__ float gl[];         // [64 pixels][9][COP], later reused as [PG][9*COP][CinP] for the reduction
  const long long npix = (long long)B * H * W;
  const long long p_begin = (long long)blockIdx.x * pix_per_block;
  const long long p_end = p_begin + pix_per_block < npix ? p_begin + pix_per_block : npix;
  const int ci = threadIdx.x % CinP, pg = threadIdx.x / CinP, PG = blockDim.x / CinP;
  float acc[9 * COP];
#pragma unroll
  for (int i = 0; i < 9 * COP; ++i) acc[i] = 0.f;
  float accb = 0.f;
  for (long long pb = p_begin; pb < p_end; pb += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 9 * COP; i += blockDim.x) {
      const int n = i % COP, tap = (i / COP) % 9, pp = i / (9 * COP);
      const long long pix = pb + pp;
      float v = 0.f;
      if (pix < p_end) {
        const int xx = (int)(pix % W), y = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const int yy = y - (tap / 3 - 1), xs = xx - (tap % 3 - 1);
        if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W) v = gy[((b * H + yy) * W + xs) * COP + n];
      }
      gl[i] = v;
    }
    __syncthreads();
    const int cnt = (int)(p_end - pb < 64 ? p_end - pb : 64);
    for (int pp0 = pg; pp0 < cnt; pp0 += 4 * PG) {
      float xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = pp0 + u * PG;
        xv[u] = pp < cnt ? bf2f(x[(pb + pp) * CinP + ci]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = pp0 + u * PG < cnt;
        const int pp = ok ? pp0 + u * PG : 0;     // xv == 0 for the clamped ones
        const float* gp = gl + pp * 9 * COP;
        if (ok) accb += gp[4 * COP + (ci < COP ? ci : 0)];   // centre tap == gy at this pixel
#pragma unroll
        for (int t = 0; t < 9 * COP; t += 4) {
          const float4 gv = *reinterpret_cast<const float4*>(gp + t);
          acc[t] += xv[u] * gv.x; acc[t + 1] += xv[u] * gv.y; acc[t + 2] += xv[u] * gv.z; acc[t + 3] += xv[u] * gv.w;
        }
      }
    }
  }
  if (ci < Co) atomicAdd(db + ci, accb);
  __syncthreads();
  for (int t = 0; t < 9 * COP; ++t) gl[(pg * 9 * COP + t) * CinP + ci] = acc[t];
  __syncthreads();
  if (pg == 0 && ci < Cin) {
    for (int tap = 0; tap < 9; ++tap)
      for (int co = 0; co < Co; ++co) {
        float v = 0.f;
        for (int q = 0; q < PG; ++q) v += gl[(q * 9 * COP + tap * COP + co) * CinP + ci];
        atomicAdd(dW + ((co * Cin) + ci) * 9 + tap, v);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise helpers
// ------------------------------------------------------------------------------------------------
__global__ void add_f32_bf16_kernel(float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ ab,
                                    long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 x = reinterpret_cast<float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    reinterpret_cast<float4*>(a)[i] = x;
    if (ab) reinterpret_cast<uint2*>(ab)[i] = pack_bf4(x.x, x.y, x.z, x.w);
  }
}

__global__ void add_bf16_into_f32_kernel(float* __restrict__ a, const bf16_t* __restrict__ b, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 x = reinterpret_cast<float4*>(a)[i];
    const uint2 u = reinterpret_cast<const uint2*>(b)[i];
    float y0, y1, y2, y3;
    unpack_bf2(u.x, y0, y1);
    unpack_bf2(u.y, y2, y3);
    x.x += y0; x.y += y1; x.z += y2; x.w += y3;
    reinterpret_cast<float4*>(a)[i] = x;
  }
}

// absolute position embedding (network_swinir.py:793-795): x[b][l][c] += ape[l][c]; x rows are CP wide, ape rows C wide
__global__ void ape_add_kernel(float* __restrict__ x, const float* __restrict__ ape, long long rows, int L, int C, int CP) {
  const long long n = rows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / C;
    const int c = (int)(i - t * C);
    x[t * CP + c] += ape[(t % L) * C + c];
  }
}
// its gradient: d ape[l][c] += sum_b gx[b][l][c]  (one thread per (l, c), batch summed in order)
__global__ void ape_grad_kernel(const float* __restrict__ gx, float* __restrict__ dape, int B, int L, int C, int CP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * C) return;
  const int l = i / C, c = i - l * C;
  float a = 0.f;
  for (int b = 0; b < B; ++b) a += gx[((long long)b * L + l) * CP + c];
  dape[i] += a;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ a, bf16_t* __restrict__ out, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    reinterpret_cast<uint2*>(out)[i] = pack_bf4(x.x, x.y, x.z, x.w);
  }
}

// d(LeakyReLU) applied in place to a bf16 gradient: g *= (act > 0 ? 1 : slope)
__global__ void dlrelu_bf16_kernel(bf16_t* __restrict__ g, const bf16_t* __restrict__ act, float slope, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float a = bf2f(act[i]);
    if (!(a > 0.f)) g[i] = f2bf(bf2f(g[i]) * slope);
  }
}


// ------------------------------------------------------------------------------------------------
// 'nearest+conv' head (network_swinir.py:828-835): F.interpolate(scale_factor=2, mode='nearest') on NHWC bf16 and its
// backward (sum of the 2x2 children), the latter fused with the LeakyReLU derivative of the tensor that was upsampled
// (it is a conv + LeakyReLU output): gp = (g00 + g01 + g10 + g11) * (act > 0 ? 1 : slope), sums in fp32.
// One thread = 8 channels (16 bytes); C % 8 == 0.
// ------------------------------------------------------------------------------------------------
__global__ void nn2x_bf16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int B, int h, int w, int c8) {
  const long long total = (long long)B * 2 * h * 2 * w * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long px = i / c8;
    const int x = (int)(px % (2 * w));
    px /= 2 * w;
    const int y = (int)(px % (2 * h));
    const long long b = px / (2 * h);
    out[i] = in[((b * h + (y >> 1)) * w + (x >> 1)) * c8 + cc];
  }
}

__global__ void nn2x_sum_dlrelu_kernel(const uint4* __restrict__ g, const uint4* __restrict__ act, uint4* __restrict__ out, int B,
                                       int h, int w, int c8, float slope) {
  const long long total = (long long)B * h * w * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long px = i / c8;
    const int x = (int)(px % w);
    px /= w;
    const int y = (int)(px % h);
    const long long b = px / h;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint4 v = g[((b * 2 * h + 2 * y + (d >> 1)) * (2 * w) + 2 * x + (d & 1)) * c8 + cc];
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack_bf2(u[e], lo, hi);
        s[2 * e] += lo;
        s[2 * e + 1] += hi;
      }
    }
    const uint4 a = act[i];
    const unsigned au[4] = {a.x, a.y, a.z, a.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo, hi;
      unpack_bf2(au[e], lo, hi);
      o[e] = pack_bf2(lo > 0.f ? s[2 * e] : s[2 * e] * slope, hi > 0.f ? s[2 * e + 1] : s[2 * e + 1] * slope);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// forward_features as a stand-alone entry (network_swinir.py:790-803): NCHW fp32 [B][C][H][W] <-> token-major fp32 [B*H*W][CP]
// (pad columns zero).  32 x 32 (pixel, channel) tiles through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void nchw_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int CP,
                                                          int HW, int to_tokens) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  if (to_tokens) {
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, px = p0 + tx;
      tile[r][tx] = (c < C && px < HW) ? src[(b * C + c) * HW + px] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int px = p0 + r, c = c0 + tx;
      if (px < HW && c < CP) dst[(b * HW + px) * CP + c] = tile[tx][r];
    }
  } else {
    for (int r = ty; r < 32; r += 8) {
      const int px = p0 + r, c = c0 + tx;
      tile[r][tx] = (px < HW && c < C) ? src[(b * HW + px) * CP + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, px = p0 + tx;
      if (c < C && px < HW) dst[(b * C + c) * HW + px] = tile[tx][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// L1 loss (finetune_swinir.py:66-67) forward + backward, with a non-finite counter (:133-143)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      float* __restrict__ dpred, float* __restrict__ loss_sum,
                                                      unsigned* __restrict__ nonfinite, long long n, float inv_n,
                                                      float grad_scale) {
  __shared__ float red[4];
  float s = 0.f;
  unsigned bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float p = pred[i], d = p - target[i];
    s += fabsf(d);
    bad += !isfinite(p);
    if (dpred) dpred[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_n * grad_scale;
  }
  s = wave_sum64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  if (bad) atomicAdd(nonfinite, bad);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}

// ------------------------------------------------------------------------------------------------
// Validation metrics in one pass over (pred, target): per-image PSNR of the clamped images (batch_psnr,
// finetune_swinir.py:69-74: 20 log10(max / sqrt(mse + 1e-8)), mse over C*H*W of one image) and the sum of |pred - target|
// of the UNclamped values for the L1 loss (F.l1_loss, :66-67, :196-197).  Stage 1: workgroup (chunk, image) writes its
// partial sums; stage 2: one thread per image adds the partials in a fixed order (reproducible, no atomics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           float* __restrict__ partial, long long per_image) {
  __shared__ float red[2][4];
  const long long base = (long long)blockIdx.y * per_image;
  float sq = 0.f, ab = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image; i += (long long)gridDim.x * blockDim.x) {
    const float p = pred[base + i], t = target[base + i];
    const float d = fminf(fmaxf(p, 0.f), 1.f) - fminf(fmaxf(t, 0.f), 1.f);
    sq = fmaf(d, d, sq);
    ab += fabsf(p - t);
  }
  sq = wave_sum64(sq);
  ab = wave_sum64(ab);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sq;
    red[1][threadIdx.x >> 6] = ab;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = partial + 2 * ((long long)blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void psnr_finish_kernel(const float* __restrict__ partial, int chunks, int B, long long per_image, float max_val,
                                   float* __restrict__ psnr, float* __restrict__ psnr_sum, float* __restrict__ abs_sum) {
  // one workgroup, thread b = image b; the batch totals are added by thread 0 in image order
  __shared__ float sh_psnr[1024], sh_abs[1024];
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float sq = 0.f, ab = 0.f;
    for (int c = 0; c < chunks; ++c) {
      sq += partial[2 * ((long long)b * chunks + c)];
      ab += partial[2 * ((long long)b * chunks + c) + 1];
    }
    const float mse = sq / (float)per_image;
    const float v = 20.0f * log10f(max_val / sqrtf(mse + 1e-8f));
    if (psnr) psnr[b] = v;
    sh_psnr[b] = v;
    sh_abs[b] = ab;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ps = 0.f, as = 0.f;
    for (int b = 0; b < B; ++b) {
      ps += sh_psnr[b];
      as += sh_abs[b];
    }
    if (psnr_sum) *psnr_sum += ps;
    if (abs_sum) *abs_sum += as;
  }
}

// ------------------------------------------------------------------------------------------------
// Data path (first slice): paired crop + uint8 -> [0, 1] float + gray -> 3 channels on the device, from a
// pool of pre-decoded 8-bit images.  Restates pil_to_tensor01 / ensure_3ch / paired_random_crop
// (finetune_swinir.py:80-110): value = (float)u8 / 255 (IEEE division, bit-exact with the host form), C = 1 repeated
// three times, LR window (top, left) of size P, HR window (top * s, left * s) of size P * s.  One descriptor per sample
// and side: {byte offset of the image in the pool, H, W, C, top, left}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_u8_kernel(const unsigned char* __restrict__ pool, const long long* __restrict__ desc,
                                                      float* __restrict__ out, int patch) {
  const long long* d = desc + 6 * (long long)blockIdx.y;
  const unsigned char* img = pool + d[0];
  // channel field: low byte = channels (1 / 3); bit 8 set = 16-bit samples (little-endian uint16, value / 65535: what
  // torchvision's ToDtype(scale=True) does for uint16 -- pil_to_tensor01 in sr_datasets.py)
  const int W = (int)d[2], C = (int)(d[3] & 0xff), wide = (int)((d[3] >> 8) & 1), top = (int)d[4], left = (int)d[5];
  float* o = out + (long long)blockIdx.y * 3 * patch * patch;
  const int n = patch * patch;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int y = i / patch, x = i - y * patch;
    const long long e = ((long long)(top + y) * W + (left + x)) * C;
    float c0, c1, c2;
    if (wide) {
      const unsigned short* px = reinterpret_cast<const unsigned short*>(img) + e;
      c0 = (float)px[0] / 65535.0f;
      c1 = C == 1 ? c0 : (float)px[1] / 65535.0f;
      c2 = C == 1 ? c0 : (float)px[2] / 65535.0f;
    } else {
      const unsigned char* px = img + e;
      c0 = (float)px[0] / 255.0f;
      c1 = C == 1 ? c0 : (float)px[1] / 255.0f;
      c2 = C == 1 ? c0 : (float)px[2] / 255.0f;
    }
    o[i] = c0;
    o[n + i] = c1;
    o[2 * n + i] = c2;
  }
}

// ------------------------------------------------------------------------------------------------
// global-norm clip + AdamW (torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW semantics,
// finetune_swinir.py:168-171, :303)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = g[i];
    s += v * v;
  }
  s = wave_sum64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    const float* __restrict__ sumsq, float max_norm, float grad_div,
                                                    float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, const int* __restrict__ nonfinite) {
  // a non-finite forward (counter of the loss kernel) or a non-finite gradient norm leaves weights and moments untouched:
  // the reference raises before backward/step (finetune_swinir.py:159-165), so the model must survive for that raise
  if (nonfinite != nullptr && *nonfinite != 0) return;
  float coef = 1.0f / grad_div;
  if (sumsq != nullptr) {
    const float ss = *sumsq;
    if (!(ss == ss) || ss > 3.0e38f) return;
    if (max_norm > 0.f) {
      const float total = sqrtf(ss) / grad_div;
      const float c = max_norm / (total + 1e-6f);
      coef *= c < 1.0f ? c : 1.0f;
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// ------------------------------------------------------------------------------------------------
// probe: contract of ds_read_b64_tr_b16 as used by lds_tr_read/tr_addr (common.h)
// ------------------------------------------------------------------------------------------------
// in: [64][16] u16 tile.  out[lane*8 + jj] = fragment element jj of lane, expected = in[8*(lane>>4)+jj][lane&15]
__global__ void probe_trread_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 24];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 16; i += 64) tile[(i >> 4) * 24 + (i & 15)] = in[i];
  __syncthreads();
  const int g = lane >> 4;
  const bf16x4_t lo = lds_tr_read(tr_addr(tile, 24, 8 * g, 0, lane));
  const bf16x4_t hi = lds_tr_read(tr_addr(tile, 24, 8 * g + 4, 0, lane));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    out[lane * 8 + e] = (bf16_t)lo[e];
    out[lane * 8 + 4 + e] = (bf16_t)hi[e];
  }
}

inline int grid_for(long long n, int block = 256, int cap = 8192) {
  long long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define ELEM_DISPATCH(KERNEL, ...)                                                                          \
  if (elem_bytes == 4) hipLaunchKernelGGL(KERNEL<unsigned>, dim3(grid), dim3(256), 0, stream, (const unsigned*)x, (unsigned*)out, __VA_ARGS__); \
  else if (elem_bytes == 2) hipLaunchKernelGGL(KERNEL<unsigned short>, dim3(grid), dim3(256), 0, stream, (const unsigned short*)x, (unsigned short*)out, __VA_ARGS__); \
  else if (elem_bytes == 8) hipLaunchKernelGGL(KERNEL<unsigned long long>, dim3(grid), dim3(256), 0, stream, (const unsigned long long*)x, (unsigned long long*)out, __VA_ARGS__); \
  else { srk_set_error("index op: unsupported element size %d", elem_bytes); return SRK_E_UNSUPPORTED; }

int srk_launch_window_partition(const void* x, void* out, int B, int H, int W, int C, int ws, int elem_bytes,
                                int reverse, hipStream_t stream) {
  const int grid = grid_for((long long)B * H * W * C);
  ELEM_DISPATCH(window_partition_kernel, B, H, W, C, ws, reverse)
  return srk_check_launch("window_partition");
}

int srk_launch_roll2d(const void* x, void* out, int B, int H, int W, int C, int sh, int sw, int elem_bytes,
                      hipStream_t stream) {
  const int grid = grid_for((long long)B * H * W * C);
  ELEM_DISPATCH(roll2d_kernel, B, H, W, C, sh, sw)
  return srk_check_launch("roll2d");
}

int srk_launch_pixel_shuffle(const void* x, void* out, int B, int C, int H, int W, int r, int elem_bytes,
                             hipStream_t stream) {
  const int grid = grid_for((long long)B * C * r * r * H * W);
  ELEM_DISPATCH(pixel_shuffle_kernel, B, C, H, W, r)
  return srk_check_launch("pixel_shuffle");
}

int srk_launch_shift_mask(float* mask, int H, int W, int ws, int shift, hipStream_t stream) {
  const long long total = (long long)(H / ws) * (W / ws) * ws * ws * ws * ws;
  hipLaunchKernelGGL(shift_mask_kernel, dim3(grid_for(total)), dim3(256), 0, stream, mask, H, W, ws, shift);
  return srk_check_launch("shift_mask");
}

int srk_launch_rel_pos_index(long long* out, int ws, hipStream_t stream) {
  const int n = ws * ws * ws * ws;
  hipLaunchKernelGGL(rel_pos_index_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, out, ws);
  return srk_check_launch("rel_pos_index");
}

int srk_launch_pack(const PackDesc* d_descs, int ndesc, int total_blocks, const float* params, bf16_t* packed,
                    float* side, hipStream_t stream) {
  hipLaunchKernelGGL(pack_kernel, dim3(total_blocks), dim3(256), 0, stream, d_descs, ndesc, params, packed, side);
  return srk_check_launch("pack");
}

int srk_launch_unpack_grads(const PackDesc* d_descs, int ndesc, int total_blocks, const float* gstage_w,
                            const float* gstage_side, float* grads, hipStream_t stream) {
  hipLaunchKernelGGL(unpack_kernel, dim3(total_blocks), dim3(256), 0, stream, d_descs, ndesc, gstage_w, gstage_side, grads);
  return srk_check_launch("unpack_grads");
}

int srk_launch_img_prep(const float* x, float* out, int B, int Cimg, int H0, int W0, int H, int W, float range,
                        const float* mean, hipStream_t stream) {
  hipLaunchKernelGGL(img_prep_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, stream, x, out, B, Cimg, H0, W0,
                     H, W, range, mean[0], mean[1], mean[2]);
  return srk_check_launch("img_prep");
}

int srk_launch_stem_conv(const float* in, const float* wgt, const float* bias, float* out, int B, int H, int W, int Cin,
                         int C, int CP, hipStream_t stream) {
  SRK_REQUIRE(Cin <= 4 && CP % 4 == 0, SRK_E_SHAPE, "stem conv: Cin=%d > 4 unsupported", Cin);
  const size_t lds = (size_t)(36 * CP + 16 * 36) * sizeof(float);
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)((npix + 15) / 16)), dim3(256), lds, stream, in, wgt, bias, out, B, H,
                     W, Cin, C, CP);
  return srk_check_launch("stem_conv");
}

int srk_launch_stem_wgrad(const float* in, const float* gy, float* dW, float* db, int B, int H, int W, int Cin, int C,
                          int CP, hipStream_t stream) {
  SRK_REQUIRE(C <= 256 && Cin <= 4, SRK_E_SHAPE, "stem wgrad: C=%d > 256 or Cin=%d > 4 unsupported", C, Cin);
  const long long npix = (long long)B * H * W;
  if (CP == 192 && npix % 32 == 0 && npix >= 32 * 512) {     // matrix-core kernel + fixed-order reduction through the caller's scratch
    const int nw = 512;
    float* scratch = srk_wgrad_scratch(stream, (size_t)nw * 192 * 48 * sizeof(float));
    if (scratch) {
      hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(nw), dim3(256), 0, stream, in, gy, scratch, B, H, W, CP);
      hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(C), dim3(256), 0, stream, scratch, nw, dW, db, Cin);
      return srk_check_launch("stem_wgrad(mfma)");
    }
  }
  const int ppb = 512;       // measured at cfg3: 128 -> 468 us, 256 -> 258, 512 -> 200, 1024 -> 248 (the 5040 atomics per workgroup
                             // contend on the same 5040 addresses, so fewer, longer workgroups win until occupancy runs out)
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0, stream, in, gy, dW, db, B,
                     H, W, Cin, C, CP, ppb);
  return srk_check_launch("stem_wgrad");
}

int srk_launch_img_grad_prep(const float* dpred, float* gy, int B, int Cimg, int Hc, int Wc, int H, int W, int r, int CoP,
                             float inv_range, hipStream_t stream) {
  hipLaunchKernelGGL(img_grad_prep_kernel, dim3(grid_for((long long)B * H * W * CoP)), dim3(256), 0, stream, dpred, gy, B,
                     Cimg, Hc, Wc, H, W, r, CoP, inv_range);
  return srk_check_launch("img_grad_prep");
}

int srk_launch_smallconv_dgrad(const float* gy, const float* wgt, bf16_t* dx, int B, int H, int W, int Cin, int CinP,
                               int Co, int CoP, hipStream_t stream) {
  SRK_REQUIRE(Co <= 16 && Co <= CoP, SRK_E_SHAPE, "smallconv dgrad: Co=%d", Co);
  {
    const int rc = srk_launch_imghead_dgrad_mfma(gy, wgt, dx, B, H, W, Cin, CinP, Co, CoP, stream);   // Cout <= 4, Cin <= 64: matrix cores
    if (rc != SRK_WGRAD_NOT_COVERED) return rc;
  }
  const size_t lds = (size_t)(9 * Co * CinP + 16 * 9 * CoP) * sizeof(float);
  // UpsampleOneStep at embed_dim 180 (network_swinir.py:594-615: 12 or 16 output channels from 192 padded inputs) stages up to
  // 120 KB of weights: more than the default 64 KB dynamic-LDS limit, well inside the CU's 160 KB
  SRK_REQUIRE(lds <= 160 * 1024, SRK_E_SHAPE, "smallconv dgrad: LDS %zu too large", lds);
  if (lds > 64 * 1024) {
    static SrkPerDevice<size_t> reserved_pd; size_t& reserved = reserved_pd.here();
    if (lds > reserved) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
          hipSuccess) {
        srk_set_error("smallconv dgrad: cannot reserve %zu bytes of LDS", lds);
        return SRK_E_LAUNCH;
      }
      reserved = 160 * 1024;
    }
  }
  const long long npix = (long long)B * H * W;
  const int groups = npix >= (1 << 20) ? 32 : 4;
  hipLaunchKernelGGL(smallconv_dgrad_kernel, dim3((unsigned)((npix + 16 * groups - 1) / (16 * groups))), dim3(256), lds, stream,
                     gy, wgt, dx, B, H, W, Cin, CinP, Co, CoP, groups);
  return srk_check_launch("smallconv_dgrad");
}

int srk_launch_smallconv_wgrad(const bf16_t* x, const float* gy, float* dW, float* db, int B, int H, int W, int Cin,
                               int CinP, int Co, int CoP, hipStream_t stream) {
  SRK_REQUIRE(Co <= CoP && (CoP == 4 || CoP == 16) && CinP <= 256 && CinP % 64 == 0, SRK_E_SHAPE,
              "smallconv wgrad: Co=%d CoP=%d CinP=%d", Co, CoP, CinP);
  {
    const int rc = srk_launch_smallconv_wgrad_mfma(x, gy, dW, db, B, H, W, Cin, CinP, Co, CoP, stream);   // W % 64 == 0: matrix cores
    if (rc != SRK_WGRAD_NOT_COVERED) return rc;
  }
  const int PG = 256 / CinP;
  const int threads = CinP * PG;
  const size_t stage = (size_t)64 * 9 * CoP, red = (size_t)PG * 9 * CoP * CinP;
  const size_t lds = (stage > red ? stage : red) * sizeof(float);
  const long long npix = (long long)B * H * W;
  const int ppb = 4096;
  const unsigned grid = (unsigned)((npix + ppb - 1) / ppb);
  if (CoP == 4) {
    hipLaunchKernelGGL(smallconv_wgrad_kernel<4>, dim3(grid), dim3(threads), lds, stream, x, gy, dW, db, B, H, W, Cin, CinP, Co, ppb);
  } else {
    static SrkPerDevice<bool> configured_pd; bool& configured = configured_pd.here();
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_wgrad_kernel<16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        srk_set_error("smallconv wgrad: cannot reserve LDS");
        return SRK_E_LAUNCH;
      }
      configured = true;
    }
    hipLaunchKernelGGL(smallconv_wgrad_kernel<16>, dim3(grid), dim3(threads), lds, stream, x, gy, dW, db, B, H, W, Cin, CinP, Co, ppb);
  }
  return srk_check_launch("smallconv_wgrad");
}

// zero fill of n floats (any alignment).  A kernel of our own rather than hipMemsetAsync: under hipGraph capture the memset node of this
// ROCm build was not re-executed on replays (found through the overlapping-attention backward: a replayed training step accumulated onto
// the previous replay's sums), so nothing the library zeroes goes through the runtime's memset.
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long long n) {
  const long long head = (4 - (((size_t)p >> 2) & 3)) & 3;      // floats up to the first 16-byte boundary
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i0 < head && i0 < n) p[i0] = 0.f;
  const long long n4 = n > head ? (n - head) / 4 : 0;
  float4* q = reinterpret_cast<float4*>(p + head);
  for (long long i = i0; i < n4; i += (long long)gridDim.x * 256) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long tail0 = head + 4 * n4;
  if (i0 < n - tail0 && tail0 + i0 < n) p[tail0 + i0] = 0.f;
}

int srk_launch_zero_f32(float* p, long long n, hipStream_t stream) {
  if (n <= 0) return SRK_OK;
  SRK_REQUIRE(p != nullptr && ((size_t)p & 3) == 0, SRK_E_NULL, "zero_f32: null or unaligned pointer");
  const long long blocks = (n / 4 + 255) / 256 + 1;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p, n);
  return srk_check_launch("zero_f32");
}

int srk_launch_add_f32_bf16(float* a, const float* b, bf16_t* ab, long long n, hipStream_t stream) {
  SRK_REQUIRE(n % 4 == 0, SRK_E_SHAPE, "add: n %% 4 != 0");
  hipLaunchKernelGGL(add_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, b, ab, n / 4);
  return srk_check_launch("add");
}

int srk_launch_add_bf16_into_f32(float* a, const bf16_t* b, long long n, hipStream_t stream) {
  SRK_REQUIRE(n % 4 == 0, SRK_E_SHAPE, "add: n %% 4 != 0");
  hipLaunchKernelGGL(add_bf16_into_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, b, n / 4);
  return srk_check_launch("add_bf16");
}

int srk_launch_ape_add(float* x, const float* ape, int B, int L, int C, int CP, hipStream_t stream) {
  hipLaunchKernelGGL(ape_add_kernel, dim3(grid_for((long long)B * L * C)), dim3(256), 0, stream, x, ape, (long long)B * L, L, C, CP);
  return srk_check_launch("ape_add");
}

int srk_launch_ape_grad(const float* gx, float* dape, int B, int L, int C, int CP, hipStream_t stream) {
  hipLaunchKernelGGL(ape_grad_kernel, dim3((L * C + 255) / 256), dim3(256), 0, stream, gx, dape, B, L, C, CP);
  return srk_check_launch("ape_grad");
}

int srk_launch_cast_f32_bf16(const float* a, bf16_t* out, long long n, hipStream_t stream) {
  SRK_REQUIRE(n % 4 == 0, SRK_E_SHAPE, "cast: n %% 4 != 0");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, out, n / 4);
  return srk_check_launch("cast");
}

int srk_launch_dlrelu_bf16(bf16_t* g, const bf16_t* act, float slope, long long n, hipStream_t stream) {
  hipLaunchKernelGGL(dlrelu_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, g, act, slope, n);
  return srk_check_launch("dlrelu");
}


int srk_launch_nn2x_bf16(const bf16_t* in, bf16_t* out, int B, int h, int w, int C, hipStream_t stream) {
  SRK_REQUIRE(C % 8 == 0, SRK_E_SHAPE, "nearest 2x: C=%d must be a multiple of 8", C);
  hipLaunchKernelGGL(nn2x_bf16_kernel, dim3(grid_for((long long)B * 4 * h * w * (C / 8))), dim3(256), 0, stream,
                     reinterpret_cast<const uint4*>(in), reinterpret_cast<uint4*>(out), B, h, w, C / 8);
  return srk_check_launch("nn2x");
}

int srk_launch_nn2x_sum_dlrelu(const bf16_t* g, const bf16_t* act, bf16_t* out, int B, int h, int w, int C, float slope,
                               hipStream_t stream) {
  SRK_REQUIRE(C % 8 == 0, SRK_E_SHAPE, "nearest 2x backward: C=%d must be a multiple of 8", C);
  hipLaunchKernelGGL(nn2x_sum_dlrelu_kernel, dim3(grid_for((long long)B * h * w * (C / 8))), dim3(256), 0, stream,
                     reinterpret_cast<const uint4*>(g), reinterpret_cast<const uint4*>(act), reinterpret_cast<uint4*>(out), B, h, w,
                     C / 8, slope);
  return srk_check_launch("nn2x_sum_dlrelu");
}

int srk_launch_nchw_tokens(const float* src, float* dst, int B, int C, int CP, int HW, int to_tokens, hipStream_t stream) {
  SRK_REQUIRE(B > 0 && B < 65536 && CP >= C, SRK_E_SHAPE, "nchw<->tokens: bad shape B=%d C=%d CP=%d", B, C, CP);
  dim3 grid((HW + 31) / 32, (CP + 31) / 32, B);
  hipLaunchKernelGGL(nchw_tokens_kernel, grid, dim3(256), 0, stream, src, dst, C, CP, HW, to_tokens);
  return srk_check_launch("nchw_tokens");
}

int srk_launch_l1_loss(const float* pred, const float* target, float* dpred, float* loss_sum, unsigned* nonfinite,
                       long long n, float grad_scale, hipStream_t stream) {
  hipLaunchKernelGGL(l1_loss_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, stream, pred, target, dpred, loss_sum,
                     nonfinite, n, 1.0f / (float)n, grad_scale);
  return srk_check_launch("l1_loss");
}

int srk_launch_crop_u8(const unsigned char* pool, const long long* desc, float* out, int B, int patch, hipStream_t stream) {
  const int chunks = (patch * patch + 2047) / 2048;
  hipLaunchKernelGGL(crop_u8_kernel, dim3(chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks), B), dim3(256), 0, stream, pool, desc, out, patch);
  return srk_check_launch("crop_u8");
}

int srk_launch_batch_psnr(const float* pred, const float* target, float* partial, int B, long long per_image, float max_val,
                          float* psnr, float* psnr_sum, float* abs_sum, hipStream_t stream) {
  const int chunks = srk_batch_psnr_chunks(per_image);
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(chunks, B), dim3(256), 0, stream, pred, target, partial, per_image);
  hipLaunchKernelGGL(psnr_finish_kernel, dim3(1), dim3(256), 0, stream, partial, chunks, B, per_image, max_val, psnr, psnr_sum, abs_sum);
  return srk_check_launch("batch_psnr");
}

int srk_launch_sumsq(const float* g, long long n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, stream, g, n, out);
  return srk_check_launch("sumsq");
}

int srk_launch_adamw(float* p, const float* g, float* m, float* v, long long n, const float* sumsq, const int* nonfinite, float max_norm,
                     float grad_div, float lr, float beta1, float beta2, float eps, float wd, int step,
                     hipStream_t stream) {
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, p, g, m, v, n, sumsq, max_norm,
                     grad_div, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2), nonfinite);
  return srk_check_launch("adamw");
}

int srk_launch_probe_trread(const bf16_t* in, bf16_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(probe_trread_kernel, dim3(1), dim3(64), 0, stream, in, out);
  return srk_check_launch("probe_trread");
}
