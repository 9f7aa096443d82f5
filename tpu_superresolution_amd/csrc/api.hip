// extern "C" surface of libsrk.so (include/srk.h): argument validation + error reporting around the
// kernel launchers.  No torch types, no allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "kernels.h"
#include "wgrad.h"

static thread_local char g_err[512] = "";

SrkOptTls& srk_opt_tls() {
  static thread_local SrkOptTls t = {};
  return t;
}

static SrkOpt** opt_table() {
  static SrkOpt* table[SRK_NUM_OPTS] = {};
  return table;
}

void srk_opt_register(SrkOpt* o) {
  if (o->id >= 0 && o->id < SRK_NUM_OPTS) opt_table()[o->id] = o;
}

void srk_opt_scope_begin(SrkOptTls* saved) {
  SrkOptTls& t = srk_opt_tls();
  *saved = t;
  for (int i = 0; i < SRK_NUM_OPTS; ++i)
    if (!((t.on >> i) & 1u) && opt_table()[i]) t.v[i] = opt_table()[i]->value;
  t.on = (1u << SRK_NUM_OPTS) - 1u;
}

void srk_opt_scope_end(const SrkOptTls& saved) { srk_opt_tls() = saved; }

int srk_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SRK_MAX_DEVICES) return 0;
  return dev;
}

int srk_device_cus() {
  static SrkPerDevice<int> cus;
  int& c = cus.here();
  if (c == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    c = -1;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
  }
  return c;
}

void srk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int srk_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    srk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SRK_E_LAUNCH;
  }
  return SRK_OK;
}

// ---- timing probe ----------------------------------------------------------------------------------
#include <vector>
namespace {
struct Probe {
  int family = 0;
  bool active = false;
  size_t count = 0;
  size_t seen = 0;          // launches of the family since probe_begin
  int stride = 1;           // bracket every stride-th launch (option "probe_stride"): the two event records per launch
                            // cost ~2.5 us of stream time each, which matters at several hundred launches per step
  bool sampling = false;    // the launch between the current pre / post is a sampled one
  double flops = 0.0, bytes = 0.0;
  std::vector<hipEvent_t> ev0, ev1;
} g_probe;
}  // namespace

void srk_probe_pre(int family, hipStream_t stream, double flops, double bytes) {
  g_probe.sampling = false;
  if (!g_probe.active || family != g_probe.family || g_probe.count >= g_probe.ev0.size()) return;
  if ((g_probe.seen++ % (size_t)g_probe.stride) != 0) return;
  g_probe.sampling = true;
  hipEventRecord(g_probe.ev0[g_probe.count], stream);
  g_probe.flops += flops;
  g_probe.bytes += bytes;
}

void srk_probe_post(int family, hipStream_t stream) {
  if (!g_probe.sampling || !g_probe.active || family != g_probe.family) return;
  g_probe.sampling = false;
  hipEventRecord(g_probe.ev1[g_probe.count], stream);
  ++g_probe.count;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define REQ_PTR(p) SRK_REQUIRE((p) != nullptr, SRK_E_NULL, "%s: null pointer '%s'", __func__, #p)
#define REQ_ALIGN(p) SRK_REQUIRE(aligned16(p), SRK_E_ALIGN, "%s: '%s' is not 16-byte aligned", __func__, #p)

static int make_geom(const srk_win_geom* g, WinGeom* out, const char* fn) {
  SRK_REQUIRE(g->H > 0 && g->W > 0 && g->H % 8 == 0 && g->W % 8 == 0, SRK_E_SHAPE, "%s: H,W must be multiples of 8 (got %d,%d)",
              fn, g->H, g->W);
  SRK_REQUIRE(g->shift == 0 || g->shift == 4, SRK_E_SHAPE, "%s: shift must be 0 or 4 (got %d)", fn, g->shift);
  out->H = g->H;
  out->W = g->W;
  out->nWw = g->W / 8;
  out->nW = (g->H / 8) * (g->W / 8);
  out->shift = g->shift;
  return SRK_OK;
}

extern "C" {

const char* srk_version(void) { return "srk 0.1 (gfx950)"; }
const char* srk_last_error(void) { return g_err; }

int srk_window_partition(const void* x, void* out, int B, int H, int W, int C, int ws, int elem_bytes, srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(out);
  SRK_REQUIRE(B > 0 && C > 0 && ws > 0 && H > 0 && W > 0 && H % ws == 0 && W % ws == 0, SRK_E_SHAPE,
              "window_partition: H=%d,W=%d must be positive multiples of ws=%d", H, W, ws);
  return srk_launch_window_partition(x, out, B, H, W, C, ws, elem_bytes, 0, (hipStream_t)stream);
}

int srk_window_reverse(const void* windows, void* out, int B, int H, int W, int C, int ws, int elem_bytes, srk_stream_t stream) {
  REQ_PTR(windows); REQ_PTR(out);
  SRK_REQUIRE(B > 0 && C > 0 && ws > 0 && H > 0 && W > 0 && H % ws == 0 && W % ws == 0, SRK_E_SHAPE,
              "window_reverse: H=%d,W=%d must be positive multiples of ws=%d", H, W, ws);
  return srk_launch_window_partition(windows, out, B, H, W, C, ws, elem_bytes, 1, (hipStream_t)stream);
}

int srk_roll2d(const void* x, void* out, int B, int H, int W, int C, int sh, int sw, int elem_bytes, srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(out);
  SRK_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SRK_E_SHAPE, "roll2d: bad shape");
  return srk_launch_roll2d(x, out, B, H, W, C, sh, sw, elem_bytes, (hipStream_t)stream);
}

int srk_pixel_shuffle(const void* x, void* out, int B, int C, int H, int W, int r, int elem_bytes, srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(out);
  SRK_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && r > 0, SRK_E_SHAPE, "pixel_shuffle: bad shape");
  return srk_launch_pixel_shuffle(x, out, B, C, H, W, r, elem_bytes, (hipStream_t)stream);
}

int srk_shift_mask(float* mask, int H, int W, int ws, int shift, srk_stream_t stream) {
  REQ_PTR(mask);
  SRK_REQUIRE(ws > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws, SRK_E_SHAPE,
              "shift_mask: shift_size must in 0-window_size (H=%d W=%d ws=%d shift=%d)", H, W, ws, shift);
  return srk_launch_shift_mask(mask, H, W, ws, shift, (hipStream_t)stream);
}

int srk_relative_position_index(int64_t* out, int ws, srk_stream_t stream) {
  REQ_PTR(out);
  SRK_REQUIRE(ws > 0 && ws <= 64, SRK_E_SHAPE, "relative_position_index: bad ws %d", ws);
  return srk_launch_rel_pos_index((long long*)out, ws, (hipStream_t)stream);
}

int srk_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y_bf16, float* y_f32, float* mean,
                      float* rstd, int rows, int C, int CP, const srk_win_geom* geom, srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(gamma); REQ_PTR(beta); REQ_ALIGN(x);
  SRK_REQUIRE(rows > 0, SRK_E_SHAPE, "layernorm: rows=%d", rows);
  SRK_REQUIRE((mean == nullptr) == (rstd == nullptr), SRK_E_NULL, "layernorm: mean/rstd must both be given or both null");
  WinGeom g;
  if (geom) {
    int rc = make_geom(geom, &g, "layernorm");
    if (rc) return rc;
    SRK_REQUIRE(rows % (g.H * g.W) == 0, SRK_E_SHAPE, "layernorm: rows %d not a multiple of H*W", rows);
  }
  return srk_launch_ln_fwd(x, gamma, beta, y_bf16, y_f32, mean, rstd, rows, C, CP, geom ? &g : nullptr, (hipStream_t)stream);
}

int srk_window_attention_fwd(const uint16_t* qkv, const float* bias_dense, uint16_t* out, int64_t B_, int nH,
                             const srk_win_geom* geom, srk_stream_t stream) {
  REQ_PTR(qkv); REQ_PTR(bias_dense); REQ_PTR(out); REQ_PTR(geom); REQ_ALIGN(qkv); REQ_ALIGN(out); REQ_ALIGN(bias_dense);
  WinGeom g;
  int rc = make_geom(geom, &g, "window_attention_fwd");
  if (rc) return rc;
  SRK_REQUIRE(B_ > 0 && B_ % g.nW == 0 && nH > 0 && nH <= 64, SRK_E_SHAPE, "window_attention_fwd: B_=%lld nW=%d nH=%d",
              (long long)B_, g.nW, nH);
  return srk_launch_attn_fwd(qkv, bias_dense, out, B_, nH, g, (hipStream_t)stream);
}

size_t srk_window_attention_bwd_scratch(int64_t B_, int nH) {
  return (size_t)srk_attn_bwd_slabs(B_, nH, nullptr) * nH * 4096 * sizeof(float);
}

int srk_window_attention_bwd(const uint16_t* qkv, const float* bias_dense, const uint16_t* d_out, uint16_t* d_qkv,
                             float* d_table, void* slab, int64_t B_, int nH, float scale, const srk_win_geom* geom,
                             srk_stream_t stream) {
  REQ_PTR(qkv); REQ_PTR(bias_dense); REQ_PTR(d_out); REQ_PTR(d_qkv); REQ_PTR(d_table); REQ_PTR(slab); REQ_PTR(geom);
  REQ_ALIGN(qkv); REQ_ALIGN(d_out); REQ_ALIGN(d_qkv); REQ_ALIGN(slab);
  WinGeom g;
  int rc = make_geom(geom, &g, "window_attention_bwd");
  if (rc) return rc;
  SRK_REQUIRE(B_ > 0 && B_ % g.nW == 0 && nH > 0 && nH <= 64, SRK_E_SHAPE, "window_attention_bwd: B_=%lld nW=%d nH=%d",
              (long long)B_, g.nW, nH);
  return srk_launch_attn_bwd(qkv, bias_dense, d_out, d_qkv, (float*)slab, d_table, B_, nH, g, scale, (hipStream_t)stream);
}

size_t srk_window_attention_bwd_fused_scratch(int64_t B_, int nH) {
  return (size_t)srk_qkv_attn_bwd_slabs(B_, nH, nH * 32, 192) * nH * 4096 * sizeof(float);
}

int srk_window_attention_bwd_fused(const uint16_t* xn, int lda, const uint16_t* w_qkv, const float* b_qkv, float scale,
                                   const uint16_t* d_x1, int ldg, const uint16_t* w_proj_t, const float* bias_dense, uint16_t* d_qkv,
                                   float* d_table, void* slab, int64_t B_, int nH, const srk_win_geom* geom, srk_stream_t stream) {
  REQ_PTR(xn); REQ_PTR(w_qkv); REQ_PTR(d_x1); REQ_PTR(w_proj_t); REQ_PTR(bias_dense); REQ_PTR(d_qkv); REQ_PTR(d_table); REQ_PTR(slab);
  REQ_PTR(geom); REQ_ALIGN(xn); REQ_ALIGN(w_qkv); REQ_ALIGN(d_x1); REQ_ALIGN(w_proj_t); REQ_ALIGN(d_qkv); REQ_ALIGN(slab);
  WinGeom g;
  int rc = make_geom(geom, &g, "window_attention_bwd_fused");
  if (rc) return rc;
  SRK_REQUIRE(B_ > 0 && B_ % g.nW == 0, SRK_E_SHAPE, "window_attention_bwd_fused: B_=%lld nW=%d", (long long)B_, g.nW);
  const int nslab = srk_qkv_attn_bwd_slabs(B_, nH, nH * 32, 192);
  SRK_REQUIRE(nslab > 0, SRK_E_UNSUPPORTED,
              "window_attention_bwd_fused: covers 6 heads x 32, K = 192 and at least one window per CU (got nH=%d B_=%lld)", nH,
              (long long)B_);
  rc = srk_launch_qkv_attn_bwd(xn, lda, w_qkv, b_qkv, scale, d_x1, ldg, w_proj_t, bias_dense, d_qkv, (float*)slab, B_, nH, nH * 32, 192, g,
                               (hipStream_t)stream);
  SRK_REQUIRE(rc != SRK_NOT_COVERED, SRK_E_UNSUPPORTED, "window_attention_bwd_fused: the kernel cannot run on this build / shape");
  if (rc) return rc;
  return srk_launch_rpb_reduce((const float*)slab, d_table, nslab, nH, (hipStream_t)stream);
}

int srk_rel_pos_bias_expand(const float* table, float* bias_dense, int nH, srk_stream_t stream) {
  REQ_PTR(table); REQ_PTR(bias_dense);
  return srk_launch_rpb_expand(table, bias_dense, nH, (hipStream_t)stream);
}

int srk_linear_bf16(const uint16_t* a, const uint16_t* w, const float* bias, uint16_t* y, int M, int N, int K, srk_stream_t stream) {
  REQ_PTR(a); REQ_PTR(w); REQ_PTR(y); REQ_ALIGN(a); REQ_ALIGN(w); REQ_ALIGN(y);
  GemmParams p = {};
  p.A = a; p.lda = K; p.Wt = w; p.M = M; p.N = N; p.K = K; p.bias = bias; p.outb = y; p.ldo = N;
  return srk_launch_gemm(LD_ROWS, EP_BF16, p, (hipStream_t)stream);
}

int srk_linear_wgrad_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int M, int N, int K, srk_stream_t stream) {
  REQ_PTR(y); REQ_PTR(x); REQ_PTR(dw); REQ_ALIGN(y); REQ_ALIGN(x);
  WgradParams p = {};
  p.Y = y; p.ldy = N; p.X = x; p.ldx = K; p.M = M; p.N = N; p.K = K; p.dW = dw; p.ldw = K; p.db = db;
  return srk_launch_wgrad(p, (hipStream_t)stream);
}

int srk_linear_wgrad_multi_bf16(const srk_wgrad_problem* problems, int count, int M, srk_stream_t stream) {
  REQ_PTR(problems);
  SRK_REQUIRE(count >= 1 && count <= 4 && M > 0, SRK_E_SHAPE, "linear_wgrad_multi: count=%d (1..4), M=%d", count, M);
  WgradParams ps[4] = {};
  for (int i = 0; i < count; ++i) {
    const srk_wgrad_problem& q = problems[i];
    REQ_PTR(q.y); REQ_PTR(q.x); REQ_PTR(q.dw); REQ_ALIGN(q.y); REQ_ALIGN(q.x);
    ps[i].Y = static_cast<const bf16_t*>(q.y); ps[i].ldy = q.ldy > 0 ? q.ldy : q.N;
    ps[i].X = static_cast<const bf16_t*>(q.x); ps[i].ldx = q.ldx > 0 ? q.ldx : q.K;
    ps[i].M = M; ps[i].N = q.N; ps[i].K = q.K; ps[i].dW = q.dw; ps[i].ldw = q.K; ps[i].db = q.db;
  }
  return srk_launch_wgrad_multi(ps, count, (hipStream_t)stream);
}

int srk_conv3x3_bf16(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y, int B, int H, int W, int CinP, int N,
                     srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(w); REQ_PTR(y); REQ_ALIGN(x); REQ_ALIGN(w); REQ_ALIGN(y);
  GemmParams p = {};
  p.A = x; p.Wt = w; p.M = B * H * W; p.N = N; p.K = 9 * CinP; p.B = B; p.H = H; p.W = W; p.CinP = CinP;
  p.bias = bias; p.outb = y; p.ldo = N;
  return srk_launch_gemm(LD_CONV3, EP_BF16, p, (hipStream_t)stream);
}

int srk_conv3x3_wgrad_bf16(const uint16_t* y, const uint16_t* x, float* dw, float* db, int B, int H, int W, int CinP, int N,
                           srk_stream_t stream) {
  REQ_PTR(y); REQ_PTR(x); REQ_PTR(dw); REQ_ALIGN(y); REQ_ALIGN(x);
  WgradParams p = {};
  p.Y = y; p.ldy = N; p.X = x; p.ldx = CinP; p.M = B * H * W; p.N = N; p.K = CinP; p.dW = dw; p.ldw = 9 * CinP; p.db = db;
  p.conv = 1; p.B = B; p.H = H; p.W = W; p.r = 1;
  return srk_launch_wgrad(p, (hipStream_t)stream);
}

int srk_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, srk_stream_t stream) {
  REQ_PTR(x); REQ_PTR(y);
  return srk_launch_cast_f32_bf16(x, y, n, (hipStream_t)stream);
}

int srk_probe_begin(int family, int capacity) {
  SRK_REQUIRE(!g_probe.active, SRK_E_STATE, "probe_begin: a probe is already active");
  SRK_REQUIRE(family >= 1 && family <= 7 && capacity > 0 && capacity <= (1 << 20), SRK_E_SHAPE, "probe_begin: bad arguments");
  while ((int)g_probe.ev0.size() < capacity) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      srk_set_error("probe_begin: hipEventCreate failed");
      return SRK_E_LAUNCH;
    }
    g_probe.ev0.push_back(a);
    g_probe.ev1.push_back(b);
  }
  g_probe.family = family;
  g_probe.count = 0;
  g_probe.seen = 0;
  g_probe.flops = 0.0;
  g_probe.bytes = 0.0;
  g_probe.active = true;
  return SRK_OK;
}

int srk_probe_end(double* total_ms, double* flops, double* bytes, int* launches) {
  SRK_REQUIRE(g_probe.active, SRK_E_STATE, "probe_end: no active probe");
  g_probe.active = false;
  double ms = 0.0;
  for (size_t i = 0; i < g_probe.count; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_probe.ev1[i]) != hipSuccess || hipEventElapsedTime(&t, g_probe.ev0[i], g_probe.ev1[i]) != hipSuccess) {
      srk_set_error("probe_end: event query failed");
      return SRK_E_LAUNCH;
    }
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (flops) *flops = g_probe.flops;
  if (bytes) *bytes = g_probe.bytes;
  if (launches) *launches = (int)g_probe.count;
  return SRK_OK;
}

// ---- options ---------------------------------------------------------------------------------------------------------------------------
// Every option has ONE process-wide value (srk_set_option: seen by every thread, including the autograd engine's backward thread); a
// plan carries its own values (srk_swinir_plan_set_option), which live in a thread-private copy of the option set for the duration of
// each call on that plan (common.h: SrkOpt) -- two plans with different options can run in one process, also on two threads.
namespace {

int tune_get(int which) {
  int v[4];
  srk_gemm_stream_tune_get(&v[0], &v[1], &v[2], &v[3]);
  return v[which];
}
int tune_set(int which, int value) {
  int v[4];
  srk_gemm_stream_tune_get(&v[0], &v[1], &v[2], &v[3]);
  v[which] = value;
  srk_gemm_stream_tune(v[0], v[1], v[2], v[3]);
  return SRK_OK;
}
int wtune_get(int which) {
  int rows, nt;
  srk_wgrad_stream_tune_get(&rows, &nt);
  return which == 0 ? rows : nt;
}

struct OptEntry {
  const char* name;
  int (*get)();
  int (*set)(int);
};

const OptEntry kOptions[] = {
    {"gemm_stream", [] { return srk_gemm_stream_enabled(); }, [](int v) { srk_gemm_stream_enable(v); return (int)SRK_OK; }},
    {"attn_bwd_fused", [] { return srk_attn_bwd_fused_enabled(); }, [](int v) { srk_attn_bwd_fused_enable(v); return (int)SRK_OK; }},
    {"attn_fused", [] { return srk_attn_fused_mode(); }, [](int v) { srk_attn_fused_enable(v); return (int)SRK_OK; }},
    {"wgrad_stream_w8", [] { return srk_wgrad_w8_enabled(); }, [](int v) { srk_wgrad_w8_enable(v); return (int)SRK_OK; }},
    {"block_light", [] { return srk_block_light_enabled(); }, [](int v) { srk_block_light_enable(v); return (int)SRK_OK; }},
    {"mlp_bwd_fused", [] { return srk_mlp_bwd_fused_enabled(); }, [](int v) { srk_mlp_bwd_fused_enable(v); return (int)SRK_OK; }},
    {"mlp_dgelu_store", [] { return srk_mlp_dgelu_store_enabled(); }, [](int v) { srk_mlp_dgelu_store_enable(v); return (int)SRK_OK; }},
    {"mlp_fused", [] { return srk_mlp_fused_enabled(); }, [](int v) { srk_mlp_fused_enable(v); return (int)SRK_OK; }},
    {"wgrad_stream", [] { return srk_wgrad_stream_enabled(); }, [](int v) { srk_wgrad_stream_enable(v); return (int)SRK_OK; }},
    {"conv_wgrad_taps", [] { return srk_conv_wgrad_taps_mode(); }, [](int v) { srk_conv_wgrad_taps_enable(v); return (int)SRK_OK; }},
    {"wgrad_partials", [] { return srk_wgrad_partials_enabled(); }, [](int v) { srk_wgrad_partials_enable(v); return (int)SRK_OK; }},
    {"wgrad_stream_rows", [] { return wtune_get(0); },
     [](int v) {
       SRK_REQUIRE(v == 32 || v == 64, SRK_E_SHAPE, "wgrad_stream_rows: 32/64");
       srk_wgrad_stream_tune(v, -1);
       return (int)SRK_OK;
     }},
    {"wgrad_stream_nt", [] { return wtune_get(1); }, [](int v) { srk_wgrad_stream_tune(0, v != 0); return (int)SRK_OK; }},
    {"gemm_stream_bm", [] { return tune_get(0); },
     [](int v) {
       SRK_REQUIRE(v == 0 || v == 16 || v == 32 || v == 64, SRK_E_SHAPE, "gemm_stream_bm: 0/16/32/64");
       return tune_set(0, v);
     }},
    {"gemm_stream_ks2", [] { return tune_get(1); }, [](int v) { return tune_set(1, v < 0 ? -1 : (v != 0)); }},
    {"gemm_stream_split", [] { return tune_get(2); }, [](int v) { return tune_set(2, v < 0 ? -1 : (v != 0)); }},
    {"gemm_stream_nb", [] { return tune_get(3); },
     [](int v) {
       SRK_REQUIRE(v == 0 || v == 4 || v == 8, SRK_E_SHAPE, "gemm_stream_nb: 0/4/8");
       return tune_set(3, v);
     }},
    {"probe_stride", [] { return g_probe.stride; },       // the timing probe is one per process
     [](int v) {
       SRK_REQUIRE(v >= 1 && v <= 1024, SRK_E_SHAPE, "probe_stride: 1..1024");
       g_probe.stride = v;
       return (int)SRK_OK;
     }},
};

const OptEntry* find_option(const char* name) {
  for (const OptEntry& e : kOptions)
    if (strcmp(name, e.name) == 0) return &e;
  return nullptr;
}

}  // namespace

int srk_set_option(const char* name, int value) {
  REQ_PTR(name);
  const OptEntry* e = find_option(name);
  if (!e) {
    srk_set_error("srk_set_option: unknown option '%s'", name);
    return SRK_E_UNSUPPORTED;
  }
  return e->set(value);
}

int srk_get_option(const char* name, int* value) {
  REQ_PTR(name); REQ_PTR(value);
  const OptEntry* e = find_option(name);
  if (!e) {
    srk_set_error("srk_get_option: unknown option '%s'", name);
    return SRK_E_UNSUPPORTED;
  }
  *value = e->get();
  return SRK_OK;
}

int srk_probe_trread(const uint16_t* in, uint16_t* out, srk_stream_t stream) {
  REQ_PTR(in); REQ_PTR(out);
  return srk_launch_probe_trread(in, out, (hipStream_t)stream);
}

int srk_l1_loss_fwd_bwd(const float* pred, const float* target, float* d_pred, float* loss, uint32_t* nonfinite, int64_t n,
                        float grad_scale, srk_stream_t stream) {
  REQ_PTR(pred); REQ_PTR(target); REQ_PTR(loss); REQ_PTR(nonfinite);
  SRK_REQUIRE(n > 0, SRK_E_SHAPE, "l1_loss: n=%lld", (long long)n);
  return srk_launch_l1_loss(pred, target, d_pred, loss, nonfinite, n, grad_scale, (hipStream_t)stream);
}

int64_t srk_wgrad_workspace_bytes(void) { return (int64_t)WS_WORKSPACE_BYTES; }

int srk_set_wgrad_workspace(void* workspace, int64_t bytes) {
  SRK_REQUIRE(bytes >= 0 && (workspace != nullptr || bytes == 0), SRK_E_SHAPE, "set_wgrad_workspace: null pointer with %lld bytes",
              (long long)bytes);
  srk_wgrad_bind_workspace(workspace, (size_t)bytes, nullptr, nullptr);
  return SRK_OK;
}

int srk_paired_crop_u8(const uint8_t* pool, const int64_t* lr_desc, const int64_t* hr_desc, float* lr_out, float* hr_out, int B,
                       int lr_patch, int scale, srk_stream_t stream) {
  REQ_PTR(pool); REQ_PTR(lr_desc); REQ_PTR(hr_desc); REQ_PTR(lr_out); REQ_PTR(hr_out);
  SRK_REQUIRE(B > 0 && B <= 65535 && lr_patch > 0 && scale > 0 && (long long)lr_patch * scale <= 8192, SRK_E_SHAPE,
              "paired_crop: B=%d patch=%d scale=%d", B, lr_patch, scale);
  int rc = srk_launch_crop_u8(pool, reinterpret_cast<const long long*>(lr_desc), lr_out, B, lr_patch, (hipStream_t)stream);
  if (rc) return rc;
  return srk_launch_crop_u8(pool, reinterpret_cast<const long long*>(hr_desc), hr_out, B, lr_patch * scale, (hipStream_t)stream);
}

int64_t srk_batch_psnr_workspace(int64_t per_image, int B) {
  if (per_image <= 0 || B <= 0) return 0;
  return (int64_t)2 * sizeof(float) * B * srk_batch_psnr_chunks(per_image);
}

int srk_batch_psnr(const float* pred, const float* target, void* workspace, int B, int64_t per_image, float max_val, float* psnr,
                   float* psnr_sum, float* abs_sum, srk_stream_t stream) {
  REQ_PTR(pred); REQ_PTR(target); REQ_PTR(workspace);
  SRK_REQUIRE(B > 0 && B <= 1024 && per_image > 0, SRK_E_SHAPE, "batch_psnr: B=%d (1..1024) per_image=%lld", B, (long long)per_image);
  SRK_REQUIRE(max_val > 0.f, SRK_E_SHAPE, "batch_psnr: max_val=%g", (double)max_val);
  return srk_launch_batch_psnr(pred, target, static_cast<float*>(workspace), B, per_image, max_val, psnr, psnr_sum, abs_sum,
                               (hipStream_t)stream);
}

int srk_grad_sumsq(const float* grads, int64_t n, float* sumsq, srk_stream_t stream) {
  REQ_PTR(grads); REQ_PTR(sumsq);
  return srk_launch_sumsq(grads, n, sumsq, (hipStream_t)stream);
}

int srk_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* sumsq,
                        float max_norm, float grad_div, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, const int32_t* nonfinite, srk_stream_t stream) {
  REQ_PTR(params); REQ_PTR(grads); REQ_PTR(exp_avg); REQ_PTR(exp_avg_sq);
  SRK_REQUIRE(max_norm <= 0.f || sumsq != nullptr, SRK_E_NULL, "adamw: clipping needs sumsq");
  SRK_REQUIRE(step >= 1 && grad_div > 0.f, SRK_E_SHAPE, "adamw: step=%d grad_div=%f", step, grad_div);
  return srk_launch_adamw(params, grads, exp_avg, exp_avg_sq, n, sumsq, nonfinite, max_norm, grad_div, lr, beta1, beta2, eps,
                          weight_decay, step, (hipStream_t)stream);
}

}  // extern "C"
