// Shared device/host helpers for the srk (super-resolution kernels) library.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srk.h"

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // MFMA 16x16 C/D fragment

#define SRK_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void srk_set_error(const char* fmt, ...);
int srk_check_launch(const char* what);

// Host-side state that belongs to a DEVICE (the "this kernel's LDS limit has been raised" flags -- hipFuncSetAttribute acts on the
// current device's copy of the function -- and the cached CU count) is keyed by the current device id, so one process can drive
// several GPUs through the C ABI.
constexpr int SRK_MAX_DEVICES = 64;
int srk_current_device();     // hipGetDevice, clamped to [0, SRK_MAX_DEVICES)
int srk_device_cus();         // multiProcessorCount of the current device (cached), -1 when it cannot be read
// Kernel-variant options (srk_set_option).  Each has a process-wide value -- what srk_set_option writes and what every thread reads,
// including the autograd engine's backward thread -- and, while a plan call is running on a thread, a thread-private copy that
// carries the plan's own values (srk_swinir_plan_set_option): reads and writes go to the private copy while its bit is set.
enum SrkOptId {
  OPT_MLP_FUSED, OPT_MLP_BWD_FUSED, OPT_GEMM_STREAM, OPT_TUNE_BM, OPT_TUNE_KS2, OPT_TUNE_SPLIT, OPT_TUNE_NB, OPT_ATTN_BWD_FUSED,
  OPT_ATTN_FUSED, OPT_BLOCK_LIGHT, OPT_TAPS_ENABLED, OPT_TAPS_DMA, OPT_WGRAD_STREAM, OPT_WGRAD_ROWS, OPT_WGRAD_NT, OPT_WGRAD_W8,
  OPT_WGRAD_PARTIALS, OPT_MLP_DGELU_STORE, SRK_NUM_OPTS
};
struct SrkOptTls {
  int v[SRK_NUM_OPTS];
  unsigned on;
};
SrkOptTls& srk_opt_tls();                 // this thread's private copies (api.hip)
struct SrkOpt;
void srk_opt_register(SrkOpt* o);
void srk_opt_scope_begin(SrkOptTls* saved);   // private copies of ALL options for this thread (snapshot of the effective values)
void srk_opt_scope_end(const SrkOptTls& saved);
struct SrkOpt {
  int id;
  int value;                              // process-wide
  SrkOpt(int id_, int v) : id(id_), value(v) { srk_opt_register(this); }
  operator int() const {
    const SrkOptTls& t = srk_opt_tls();
    return ((t.on >> id) & 1u) ? t.v[id] : value;
  }
  SrkOpt& operator=(int v) {
    SrkOptTls& t = srk_opt_tls();
    if ((t.on >> id) & 1u) t.v[id] = v;
    else value = v;
    return *this;
  }
  SrkOpt(const SrkOpt&) = delete;
  SrkOpt& operator=(const SrkOpt&) = delete;
};

template <class T>
struct SrkPerDevice {
  T v[SRK_MAX_DEVICES] = {};
  T& here() { return v[srk_current_device()]; }
};

// optional per-launch timing probe (bench.py roofline leg): HIP events around launches of one kernel family
enum { FAM_GEMM_LINEAR = 1, FAM_GEMM_CONV = 2, FAM_WGRAD_LINEAR = 3, FAM_WGRAD_CONV = 4, FAM_ATTN_FWD = 5, FAM_ATTN_BWD = 6,
       FAM_LN = 7 };
void srk_probe_pre(int family, hipStream_t stream, double flops, double bytes = 0.0);
void srk_probe_post(int family, hipStream_t stream);

#define SRK_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      srk_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

// ---------------------------------------------------------------------------------------------
// bf16 helpers (device)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return __builtin_bit_cast(unsigned short, b);
}

// one v_cvt_pk_bf16_f32 (two scalar conversions + shift + or cost four VALU slots per pair)
typedef __bf16 srk_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float srk_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(srk_f32x2_t{lo, hi}, srk_bf16x2_t));
}

__device__ __forceinline__ uint2 pack_bf4(float a, float b, float c, float d) {
  return make_uint2(pack_bf2(a, b), pack_bf2(c, d));
}

__device__ __forceinline__ void unpack_bf2(unsigned u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}

// exact (erf) GELU and its derivative -- nn.GELU default (network_swinir.py:15, Mlp act_layer)
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. below fp32 rounding of the GELU result that is
// then stored as bf16): ~14 VALU instructions instead of libm erff's ~40 -- the GELU epilogues are VALU-heavy.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }

// bf16x4 of gelu(v) / of v * gelu'(u): what the GELU / dGELU epilogues store.  These epilogues are VALU-bound on this
// math; the derivative reuses erf's exp(-z^2) = exp(-x^2/2) for the Gaussian density instead of a second v_exp_f32.
// (Packed-fp32 float2 versions of the polynomial were measured: no gain for GELU -- the two transcendentals per
// element dominate -- so the scalar form stays.)
__device__ __forceinline__ float dgelu_shared_exp(float x) {
  const float z = x * 0.70710678118654752f;
  const float ax = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);                 // = exp(-x^2 / 2)
  const float erf = copysignf(1.0f - p * t * e, z);
  return fmaf(x * 0.39894228040143268f, e, fmaf(0.5f, erf, 0.5f));
}
// gelu(x) (the same operations as gelu_f: bit-identical) and gelu'(x) from one exp and one rcp
__device__ __forceinline__ void gelu_both(float x, float& gl, float& dg) {
  const float z = x * 0.70710678118654752f;
  const float ax = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float erf = copysignf(1.0f - p * t * e, z);
  gl = 0.5f * x * (1.0f + erf);
  dg = fmaf(x * 0.39894228040143268f, e, fmaf(0.5f, erf, 0.5f));
}
__device__ __forceinline__ uint2 gelu_pack4(float v0, float v1, float v2, float v3) {
  return pack_bf4(gelu_f(v0), gelu_f(v1), gelu_f(v2), gelu_f(v3));
}
__device__ __forceinline__ uint2 dgelu_mul_pack4(float v0, float v1, float v2, float v3, float u0, float u1, float u2, float u3) {
  return pack_bf4(v0 * dgelu_shared_exp(u0), v1 * dgelu_shared_exp(u1), v2 * dgelu_shared_exp(u2), v3 * dgelu_shared_exp(u3));
}

// ---------------------------------------------------------------------------------------------
// window index map: window-order row -> raster token (roll(-shift) + window_partition fused;
// reference network_swinir.py:249-256 and, as the scatter, :265-272).  ws == 8.
// ---------------------------------------------------------------------------------------------
struct WinGeom {
  int H, W;        // feature-map size (multiples of 8)
  int nWw, nW;     // windows per row, windows per image
  int shift;       // 0 or 4
};

__device__ __forceinline__ int win_row_to_token(const WinGeom& g, int m) {
  const int b_ = m >> 6, p = m & 63;
  const int b = b_ / g.nW, w = b_ - b * g.nW;
  const int wy = w / g.nWw, wx = w - wy * g.nWw;
  int y = wy * 8 + (p >> 3) + g.shift;
  int x = wx * 8 + (p & 7) + g.shift;
  if (y >= g.H) y -= g.H;
  if (x >= g.W) x -= g.W;
  return (b * g.H + y) * g.W + x;
}

// XCD-aware block remap (bijective for any grid size): hardware deals consecutive block ids round-robin over
// the 8 XCDs (each with a private L2), so blocks b and b+8 share an XCD.  Returning a LOGICAL id that is contiguous
// per XCD makes consecutive logical tiles (which share operand panels) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, slot = bid >> 3;
  const int q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// inverse of win_row_to_token: raster token -> window-order row
__device__ __forceinline__ int token_to_win_row(const WinGeom& g, int t) {
  const int hw = g.H * g.W;
  const int b = t / hw, rem = t - b * hw;
  int y = rem / g.W - g.shift, x = rem % g.W - g.shift;
  if (y < 0) y += g.H;
  if (x < 0) x += g.W;
  const int w = (y >> 3) * g.nWw + (x >> 3);
  return ((b * g.nW + w) << 6) | ((y & 7) << 3) | (x & 7);
}

// region label (0..8) of token p of window w in the shifted frame (network_swinir.py:219-230), ws 8 shift 4
__device__ __forceinline__ int win_region_label(const WinGeom& g, int w, int p) {
  const int wy = w / g.nWw, wx = w - wy * g.nWw;
  const int y = wy * 8 + (p >> 3), x = wx * 8 + (p & 7);
  const int ly = y < g.H - 8 ? 0 : (y < g.H - 4 ? 1 : 2);
  const int lx = x < g.W - 8 ? 0 : (x < g.W - 4 ? 1 : 2);
  return ly * 3 + lx;
}

// ---------------------------------------------------------------------------------------------
// LDS helpers
// ---------------------------------------------------------------------------------------------
// [rows][64] bf16 tile, 128-B rows, 16-B chunk index XOR-swizzled with (row & 7): conflict-free
// ds_read_b128 fragment reads (16 rows x one chunk per 16-lane group).
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) << 3); }

// hardware transposed read: per 16-lane group, a 4-row x 16-col block of 16-bit elements is
// delivered column-major (lane i gets column i of the 4 rows).  Lane 4q+p supplies the address
// of row q, columns 4p..4p+3.  See srk_probe_trread for the on-device check of this contract.
__device__ __forceinline__ bf16x4_t lds_tr_read(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) short s4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(p));
}

// address helper for lds_tr_read: tile is row-major with `stride` elements per row; the calling
// lane wants, for its 16-lane group, rows rbase..rbase+3 and columns c0..c0+15.
__device__ __forceinline__ const bf16_t* tr_addr(const bf16_t* tile, int stride, int rbase, int c0, int lane) {
  const int ll = lane & 15;
  return tile + (rbase + (ll >> 2)) * stride + c0 + ((ll & 3) << 2);
}

// value of another lane of the same 16-lane DPP row (cross-lane VALU operand, no LDS round trip):
// 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x141 row_half_mirror (i <-> 7-i), 0x140 row_mirror (i <-> 15-i)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over the 16 lanes sharing lane>>4, result in all of them (bitwise identical: every lane adds the same pairs).
// Same pairing tree as an xor-1/2/4/8 butterfly, but 4 DPP adds instead of 4 dependent ds_bpermute round trips.
__device__ __forceinline__ float wave_sum16(float v) {
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v);
  v += dpp_row<0x140>(v);
  return v;
}

// max / sum over lanes l, l^16, l^32, l^48 (the four 16-lane rows of a wave) without an LDS round trip: v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of the second, v_permlane32_swap the upper and lower
// halves; with both operands = v, {result0, result1} is {v, partner} or {partner, v} on every lane, so a commutative op of
// the two is the butterfly step.  (__shfl_xor with 16 / 32 compiles to ds_bpermute: two dependent LDS round trips.)
__device__ __forceinline__ float xrow_max4(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xrow_sum4(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Softmax numerators of one 16-query x 64-key score tile held as four MFMA accumulators (s[jt][e] = S[query r16][key
// 16 jt + 4 g + e], bias and mask already added).  Returns exp(s - rowmax) -- in [0, 1], UNNORMALISED -- as the two bf16
// B-operand fragments of P (k = key) and 1 / row sum: the caller scales its 8 outputs instead of the 64 probabilities.
// exp(x - m) = exp2(x log2e - m log2e): one packed fma per two scores + v_exp_f32, v_rcp_f32 for the sum (the phase is
// VALU-issue-bound: 4 cycles per wave-instruction, 8 for a transcendental).
__device__ __forceinline__ float softmax_numerators(const f32x4_t (&s)[4], bf16x8_t (&pf)[2]) {
  constexpr float L2E = 1.4426950408889634f;
  float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
  for (int jt = 1; jt < 4; ++jt) mx = fmaxf(fmaxf(mx, s[jt][0]), fmaxf(fmaxf(s[jt][1], s[jt][2]), s[jt][3]));
  mx = xrow_max4(mx);
  const float mxl = mx * L2E;
  f32x4_t e[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const f32x4_t t = s[jt] * L2E - mxl;
#pragma unroll
    for (int k = 0; k < 4; ++k) e[jt][k] = __builtin_amdgcn_exp2f(t[k]);
  }
  const f32x4_t a = (e[0] + e[1]) + (e[2] + e[3]);
  const float sum = xrow_sum4((a[0] + a[1]) + (a[2] + a[3]));
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) {
    const uint2 lo = pack_bf4(e[2 * ss][0], e[2 * ss][1], e[2 * ss][2], e[2 * ss][3]);
    const uint2 hi = pack_bf4(e[2 * ss + 1][0], e[2 * ss + 1][1], e[2 * ss + 1][2], e[2 * ss + 1][3]);
    pf[ss] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
  }
  return __builtin_amdgcn_rcpf(sum);
}

__device__ __forceinline__ float wave_sum64(float v) {
  v = wave_sum16(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA (global -> LDS without a VGPR destination) and the waits that go with it
// ---------------------------------------------------------------------------------------------
// One wave instruction writes 64 x 16 B (or 64 x 4 B) CONTIGUOUS bytes at the wave-uniform LDS byte address
// `lds_dst` (lane l lands at lds_dst + 16 l); the SOURCE address is per lane, which is where swizzles / gathers go.
// The instruction counts on vmcnt like any vector-memory operation and retires in issue order.
template <bool NT = false>
__device__ __forceinline__ void srk_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)   // streaming cache policy: for bytes this launch reads once
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
template <int N>
__device__ __forceinline__ void srk_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// this wave's LDS operations complete, then the workgroup barrier -- WITHOUT the vmcnt(0) drain that
// __syncthreads() implies while DMAs (or stores) are in flight
__device__ __forceinline__ void srk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
