// Whole-model executor for SwinIR (reference network_swinir.py:618-851): one C call enqueues the
// complete forward (or a backward segment) as a fixed sequence of kernel launches on one stream.
// The plan owns only host-side bookkeeping (parameter table, pack descriptors, workspace layout);
// every byte of device memory is provided by the caller.
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "wgrad.h"

namespace {

struct ParamEntry {
  std::string name;
  long long off, numel;
  int ndim;
  long long shape[4];
};

struct BlockW {
  long long n1w, n1b, rpb, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b;   // flat params
  long long Wqkv, WqkvT, Wproj, WprojT, Wfc1, Wfc1T, Wfc2, Wfc2T;                        // packed bf16
  long long bqkv, bproj, bfc1, bfc2, biasd;                                              // side fp32
  int nH, dh, CA, shift;
  float scale;
};

struct ConvW {
  long long w, b;        // flat params
  long long Wc, WcT;     // packed bf16 (WcT = -1 if unused)
  long long bc;          // side fp32
  int Cin, CinP, Cout, NP, r, Cs;
};

// residual-connection conv of an RSTB / conv_after_body: '1conv' = one 3x3 conv; '3conv' = conv3x3(C -> C/4) + LeakyReLU(0.2)
// + conv1x1(C/4 -> C/4) + LeakyReLU(0.2) + conv3x3(C/4 -> C)   (network_swinir.py:464-471, :728-736)
struct ResiW {
  int three = 0;
  ConvW c;               // '1conv'
  ConvW c0, c2;          // '3conv' outer convs (C/4 padded to 64 channels)
  long long w1 = 0, b1 = 0, W1 = 0, W1T = 0, b1s = 0;   // the 1x1 conv as a 64 x 64 linear layer: flat params, packed, side bias
  int C4 = 0;
};

struct ResiAct {
  size_t a1 = 0, a2 = 0;   // bf16 [T][64]: LeakyReLU outputs after conv0 / conv1x1 (saved for backward when training)
};

struct BlockAct {
  size_t x_in, xn1w, mean1, rstd1, qkv, ao, x1, xn2, mean2, rstd2, u, h, x_out;
};

struct Workspace {
  int B = 0, H0 = 0, W0 = 0, H = 0, W = 0, training = 0;
  int attn_recompute = 0;      // training: the attention backward re-projects q/k/v (attn_bwd_fused.hip); the forward stores none
  int ckpt = 0;                // training with use_checkpoint: ao / u / h of all blocks share one buffer set, refilled in backward
  int u_is_dgelu = 0;          // the last forward's fused MLP kernels stored gelu'(u) in the u buffers (GemmParams::u_dgelu)
  size_t ck_x = 0;             // ... scratch for the re-run MLP forward's residual output (discarded)
  int opt_sig = 0;             // option values the layout depends on
  long long T = 0;
  size_t img4, f0, x0, mean_pe, rstd_pe;
  std::vector<BlockAct> blk;
  std::vector<size_t> layer_in, layer_xb, layer_out;
  size_t meanf, rstdf, xnf, fb, t1;
  std::vector<size_t> up;      // pixel-shuffled activations per stage
  std::vector<ResiAct> resi;   // '3conv': per RSTB, last entry = conv_after_body
  std::vector<size_t> nc_u, nc_a;   // 'nearest+conv': nearest-upsampled inputs / LeakyReLU outputs of conv_up1, conv_up2
  size_t nc_hr = 0;                 // LeakyReLU(conv_hr)
  // backward
  size_t gx, gxb, gx2, gxb2, gxbw, du, dxn, dao, dqkv, slab, wgpart, gyimg, gt1, gfb, gfb32, gstage_w, gstage_side;
  size_t g3a = 0, g3b = 0;          // '3conv': bf16 [T][64] gradients of the two narrow activations
  size_t ncgA = 0, ncgB = 0, ncgU = 0;   // 'nearest+conv': bf16 gradient buffers at the output resolution
  std::vector<size_t> gup;
  size_t total = 0;
  std::map<std::string, std::pair<size_t, size_t>> names;
};

}  // namespace

struct srk_swinir_plan {
  srk_swinir_config cfg;
  int C, CP, HID, HP, L, nblk, Cimg;
  int nstage, stage_r;         // pixelshuffle: number of conv+PS stages and their factor
  bool no_shift;               // img_size <= window_size: shift disabled everywhere (:193-196)
  std::vector<ParamEntry> params;
  long long param_floats = 0;
  std::vector<std::pair<std::string, int>> options;   // the plan's own option values (srk_swinir_plan_set_option), applied around its calls
  std::vector<BlockW> blocks;
  std::vector<int> layer_first_blk;
  std::vector<ResiW> layer_conv;
  ResiW conv_after_body;
  ConvW conv_before_up, conv_last, up_direct, conv_hr;
  std::vector<ConvW> up_convs;       // 'pixelshuffle': conv + PixelShuffle stages; 'nearest+conv': conv_up1, conv_up2
  long long p_conv_first_w, p_conv_first_b, p_pe_w, p_pe_b, p_norm_w, p_norm_b;
  long long p_ape = -1;        // absolute_pos_embed [1][img_size^2][C] (cfg.ape) or -1
  // pack descriptors, grouped: group 0 = head, 1..L = layers, L+1 = tail
  std::vector<PackDesc> descs;
  std::vector<int> group_desc_begin, group_blocks;   // size L+3 / L+2
  std::vector<long long> group_param_begin;           // size L+3: flat offsets delimiting the groups
  long long packed_elems = 0, side_floats = 0;
  const PackDesc* d_descs = nullptr;
  Workspace ws;
};

namespace {

long long add_param(srk_swinir_plan* p, const std::string& name, std::initializer_list<long long> shape) {
  ParamEntry e;
  e.name = name;
  e.ndim = (int)shape.size();
  e.numel = 1;
  int i = 0;
  for (long long s : shape) {
    e.shape[i++] = s;
    e.numel *= s;
  }
  for (; i < 4; ++i) e.shape[i] = 1;
  e.off = p->param_floats;
  p->param_floats += (e.numel + 63) / 64 * 64;
  p->params.push_back(e);
  return e.off;
}

long long alloc_packed(srk_swinir_plan* p, long long elems) {
  const long long off = p->packed_elems;
  p->packed_elems += (elems + 127) / 128 * 128;   // 256-B aligned
  return off;
}
long long alloc_side(srk_swinir_plan* p, long long floats) {
  const long long off = p->side_floats;
  p->side_floats += (floats + 63) / 64 * 64;
  return off;
}

PackDesc base_desc(int kind, long long src, long long dst, int NP, int KP) {
  PackDesc d;
  memset(&d, 0, sizeof(d));
  d.kind = kind;
  d.src = src;
  d.dst = dst;
  d.NP = NP;
  d.KP = KP;
  d.r = 1;
  d.Cs = 64;
  d.CinP = 64;
  return d;
}

void push_desc(srk_swinir_plan* p, const PackDesc& d) { p->descs.push_back(d); }

void add_linear(srk_swinir_plan* p, long long src, long long dst, long long dstT, int NP, int KP, int Nreal, int Kreal, int nmap,
                int kmap, int nH, int dh) {
  PackDesc d = base_desc(PK_LINEAR, src, dst, NP, KP);
  d.Nreal = Nreal; d.Kreal = Kreal; d.nmap = nmap; d.kmap = kmap; d.nH = nH; d.dh = dh; d.CA = nH * 32;
  push_desc(p, d);
  d.transpose = 1;
  d.dst = dstT;
  push_desc(p, d);
}

void add_vec(srk_swinir_plan* p, long long src, long long dst, int NP, int Nreal, int nmap, int nH, int dh, int r, int Cs) {
  PackDesc d = base_desc(PK_VEC, src, dst, NP, 1);
  d.Nreal = Nreal; d.nmap = nmap; d.nH = nH; d.dh = dh; d.CA = nH * 32; d.r = r; d.Cs = Cs;
  push_desc(p, d);
}

// conv weights [Cout][Cin][3][3] -> packed [NP][9*CinP] (+ transposed/flipped copy for dgrad)
void add_conv(srk_swinir_plan* p, ConvW& c, const std::string& name, int Cout, int Cin, int NP, int CinP, int r, int Cs,
              bool want_T) {
  c.w = add_param(p, name + ".weight", {Cout, Cin, 3, 3});
  c.b = add_param(p, name + ".bias", {Cout});
  c.Cin = Cin; c.CinP = CinP; c.Cout = Cout; c.NP = NP; c.r = r; c.Cs = Cs;
  c.Wc = alloc_packed(p, (long long)NP * 9 * CinP);
  c.WcT = want_T ? alloc_packed(p, (long long)CinP * 9 * NP) : -1;
  c.bc = alloc_side(p, NP);
  PackDesc d = base_desc(PK_CONV, c.w, c.Wc, NP, 9 * CinP);
  d.Nreal = Cout; d.Kreal = Cin; d.nmap = r > 1 ? NM_PS : NM_DIRECT; d.r = r; d.Cs = Cs; d.CinP = CinP;
  push_desc(p, d);
  if (want_T) {
    d.transpose = 1;
    d.dst = c.WcT;
    push_desc(p, d);
  }
  add_vec(p, c.b, c.bc, NP, Cout, r > 1 ? NM_PS : NM_DIRECT, 1, 1, r, Cs);
}

void add_resi(srk_swinir_plan* p, ResiW& r, const std::string& name, int C, int CP) {
  r.three = p->cfg.resi_connection == SRK_RESI_3CONV;
  if (!r.three) {
    add_conv(p, r.c, name, C, C, CP, CP, 1, 64, true);
    return;
  }
  const int C4 = C / 4;
  r.C4 = C4;
  add_conv(p, r.c0, name + ".0", C4, C, 64, CP, 1, 64, true);
  r.w1 = add_param(p, name + ".2.weight", {C4, C4, 1, 1});
  r.b1 = add_param(p, name + ".2.bias", {C4});
  r.W1 = alloc_packed(p, 64 * 64);
  r.W1T = alloc_packed(p, 64 * 64);
  r.b1s = alloc_side(p, 64);
  add_linear(p, r.w1, r.W1, r.W1T, 64, 64, C4, C4, NM_DIRECT, KM_DIRECT, 1, 1);
  add_vec(p, r.b1, r.b1s, 64, C4, NM_DIRECT, 1, 1, 1, 64);
  add_conv(p, r.c2, name + ".4", C, C4, CP, 64, 1, 64, true);
}

void finish_group(srk_swinir_plan* p) {
  const int begin = p->group_desc_begin.back();
  int blocks = 0;
  for (size_t i = begin; i < p->descs.size(); ++i) {
    PackDesc& d = p->descs[i];
    d.blk0 = blocks;
    long long total;
    if (d.kind == PK_RPB) total = (long long)d.nH * 4096;
    else if (d.kind == PK_VEC) total = d.NP;
    else total = (long long)d.NP * d.KP;
    blocks += (int)((total + 1023) / 1024);
  }
  p->group_blocks.push_back(blocks);
  p->group_desc_begin.push_back((int)p->descs.size());
  p->group_param_begin.push_back(p->param_floats);
}

struct Arena {
  size_t off = 0;
  Workspace* ws;
  size_t get(const std::string& name, size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    ws->names[name] = std::make_pair(o, bytes);
    return o;
  }
};

// The plan's option values hold for the duration of one plan call on the calling thread: the thread gets a private copy of the option
// set (common.h: SrkOpt) into which the plan's values are written; the process-wide values are untouched.
struct PlanOptionScope {
  SrkOptTls saved;
  bool active = false;
  explicit PlanOptionScope(const srk_swinir_plan* p) {
    if (!p || p->options.empty()) return;
    srk_opt_scope_begin(&saved);
    active = true;
    for (const auto& o : p->options) srk_set_option(o.first.c_str(), o.second);
  }
  ~PlanOptionScope() {
    if (active) srk_opt_scope_end(saved);
  }
  PlanOptionScope(const PlanOptionScope&) = delete;
  PlanOptionScope& operator=(const PlanOptionScope&) = delete;
};

// option values that change the workspace layout: a forward re-lays the workspace out when they differ from the laid-out ones
int layout_option_sig() { return srk_attn_bwd_fused_enabled() | (srk_attn_fused_mode() << 1); }

void layout_workspace(const srk_swinir_plan* p, Workspace& w, int B, int H0, int W0, int training) {
  w = Workspace();
  w.B = B; w.H0 = H0; w.W0 = W0; w.training = training;
  w.H = (H0 + 7) / 8 * 8;
  w.W = (W0 + 7) / 8 * 8;
  w.T = (long long)B * w.H * w.W;
  w.opt_sig = layout_option_sig();
  w.ckpt = (training != 0 && p->cfg.use_checkpoint != 0) ? 1 : 0;
  w.attn_recompute = training != 0 && srk_attn_fused_mode() != 0;   // the forward that stores no q/k/v is the fused one
  for (const BlockW& b : p->blocks)
    if (srk_qkv_attn_bwd_slabs(w.T / 64, b.nH, b.CA, p->CP) == 0) w.attn_recompute = 0;
  const size_t T = (size_t)w.T;
  const size_t CP = p->CP, HP = p->HP;
  Arena a;
  a.ws = &w;
  w.img4 = a.get("img4", T * 4 * 4);
  w.f0 = a.get("f0", T * CP * 4);
  w.x0 = a.get("x0", T * CP * 4);
  w.mean_pe = a.get("mean_pe", T * 4);
  w.rstd_pe = a.get("rstd_pe", T * 4);
  w.blk.resize(p->nblk);
  w.layer_in.resize(p->L);
  w.layer_xb.resize(p->L);
  w.layer_out.resize(p->L);
  BlockAct shared;
  memset(&shared, 0, sizeof(shared));
  size_t pingP = 0, pingQ = 0, shared_xb = 0;
  size_t maxCA = 0;
  for (const BlockW& b : p->blocks) maxCA = b.CA > (int)maxCA ? b.CA : maxCA;
  size_t ck_ao = 0, ck_u = 0, ck_h = 0;
  if (w.ckpt) {
    ck_ao = a.get("ck.ao", T * maxCA * 2);
    ck_u = a.get("ck.u", T * HP * 2);
    ck_h = a.get("ck.h", T * HP * 2);
    w.ck_x = a.get("ck.x", T * CP * 4);
  }
  if (!training) {
    shared.xn1w = a.get("xn1w", T * CP * 2);
    shared.mean1 = a.get("mean1", T * 4);
    shared.rstd1 = a.get("rstd1", T * 4);
    shared.qkv = a.get("qkv", T * 3 * maxCA * 2);
    shared.ao = a.get("ao", T * maxCA * 2);
    shared.xn2 = a.get("xn2", T * CP * 2);
    shared.mean2 = a.get("mean2", T * 4);
    shared.rstd2 = a.get("rstd2", T * 4);
    shared.u = a.get("u", T * HP * 2);
    shared.h = a.get("h", T * HP * 2);
    pingP = a.get("pingP", T * CP * 4);
    pingQ = a.get("pingQ", T * CP * 4);
    shared_xb = a.get("xb", T * CP * 2);
  }
  w.resi.assign(p->L + 1, ResiAct());
  size_t cur = w.x0;
  for (int l = 0; l < p->L; ++l) {
    w.layer_in[l] = cur;
    const int first = p->layer_first_blk[l], depth = p->cfg.depths[l];
    for (int j = 0; j < depth; ++j) {
      const int bi = first + j;
      BlockAct& ba = w.blk[bi];
      const std::string pre = "blk" + std::to_string(bi) + ".";
      const BlockW& bw = p->blocks[bi];
      ba.x_in = cur;
      if (training) {
        ba.xn1w = a.get(pre + "xn1w", T * CP * 2);
        ba.mean1 = a.get(pre + "mean1", T * 4);
        ba.rstd1 = a.get(pre + "rstd1", T * 4);
        ba.qkv = w.attn_recompute ? 0 : a.get(pre + "qkv", T * 3 * bw.CA * 2);
        ba.ao = w.ckpt ? ck_ao : a.get(pre + "ao", T * bw.CA * 2);
        ba.x1 = a.get(pre + "x1", T * CP * 4);
        ba.xn2 = a.get(pre + "xn2", T * CP * 2);
        ba.mean2 = a.get(pre + "mean2", T * 4);
        ba.rstd2 = a.get(pre + "rstd2", T * 4);
        ba.u = w.ckpt ? ck_u : a.get(pre + "u", T * HP * 2);
        ba.h = w.ckpt ? ck_h : a.get(pre + "h", T * HP * 2);
        ba.x_out = a.get(pre + "x_out", T * CP * 4);
      } else {
        const size_t xin = ba.x_in;
        ba = shared;
        ba.x_in = xin;
        ba.x1 = pingP;
        ba.x_out = pingQ;
      }
      cur = ba.x_out;
    }
    if (p->cfg.resi_connection == SRK_RESI_3CONV) {
      const std::string pre = training ? "layer" + std::to_string(l) + "." : "shared.";
      w.resi[l].a1 = (training || l == 0) ? a.get(pre + "resi_a1", T * 64 * 2) : w.resi[0].a1;
      w.resi[l].a2 = (training || l == 0) ? a.get(pre + "resi_a2", T * 64 * 2) : w.resi[0].a2;
    }
    w.layer_xb[l] = training ? a.get("layer" + std::to_string(l) + ".xb", T * CP * 2) : shared_xb;
    // RSTB output: new buffer when training (the block input must survive), in place otherwise
    w.layer_out[l] = training ? a.get("layer" + std::to_string(l) + ".out", T * CP * 4) : w.layer_in[l];
    cur = w.layer_out[l];
  }
  w.meanf = a.get("meanf", T * 4);
  w.rstdf = a.get("rstdf", T * 4);
  w.xnf = a.get("xnf", T * CP * 2);
  w.fb = a.get("fb", T * CP * 2);
  if (p->cfg.resi_connection == SRK_RESI_3CONV) {
    w.resi[p->L].a1 = (training || p->L == 0) ? a.get("tail.resi_a1", T * 64 * 2) : w.resi[0].a1;
    w.resi[p->L].a2 = (training || p->L == 0) ? a.get("tail.resi_a2", T * 64 * 2) : w.resi[0].a2;
  }
  w.t1 = 0;
  size_t out_px = T;          // pixels at the output resolution
  if (p->cfg.upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
    w.t1 = a.get("t1", T * 64 * 2);
    size_t px = T;
    for (int k = 0; k < p->nstage; ++k) {
      px *= (size_t)p->stage_r * p->stage_r;
      w.up.push_back(a.get("up" + std::to_string(k), px * 64 * 2));
    }
    out_px = px;
  } else if (p->cfg.upsampler == SRK_UPSAMPLER_NEAREST_CONV) {
    w.t1 = a.get("t1", T * 64 * 2);
    size_t px = T;
    for (int k = 0; k < p->nstage; ++k) {
      px *= 4;
      w.nc_u.push_back(a.get("nc_u" + std::to_string(k), px * 64 * 2));
      w.nc_a.push_back(a.get("nc_a" + std::to_string(k), px * 64 * 2));
    }
    w.nc_hr = a.get("nc_hr", px * 64 * 2);
    out_px = px;
  }
  if (training) {
    w.gx = a.get("gx", T * CP * 4);
    w.gxb = a.get("gxb", T * CP * 2);
    w.gx2 = a.get("gx2", T * CP * 4);
    w.gxb2 = a.get("gxb2", T * CP * 2);
    w.gxbw = a.get("gxbw", T * CP * 2);
    w.du = a.get("du", T * HP * 2);
    w.dxn = a.get("dxn", T * CP * 2);
    w.dao = a.get("dao", T * maxCA * 2);
    w.dqkv = a.get("dqkv", T * 3 * maxCA * 2);
    int maxH = 1;
    for (const BlockW& b : p->blocks) maxH = b.nH > maxH ? b.nH : maxH;
    size_t nslab = (size_t)srk_attn_bwd_slabs(w.T / 64, 1, nullptr);
    if (w.attn_recompute) nslab = (size_t)srk_qkv_attn_bwd_slabs(w.T / 64, p->blocks[0].nH, p->blocks[0].CA, p->CP);
    w.slab = a.get("slab", nslab * maxH * 4096 * 4);
    w.wgpart = a.get("wgpart", WS_WORKSPACE_BYTES);      // split partials of the streaming weight-gradient kernels
    w.gfb = a.get("gfb", T * CP * 2);
    w.gfb32 = a.get("gfb32", T * CP * 4);
    if (p->cfg.resi_connection == SRK_RESI_3CONV) {
      w.g3a = a.get("g3a", T * 64 * 2);
      w.g3b = a.get("g3b", T * 64 * 2);
    }
    if (p->cfg.upsampler == SRK_UPSAMPLER_NEAREST_CONV) {
      w.gt1 = a.get("gt1", T * 64 * 2);
      w.ncgA = a.get("ncgA", out_px * 64 * 2);
      w.ncgB = a.get("ncgB", out_px * 64 * 2);
      w.ncgU = a.get("ncgU", out_px * 64 * 2);
      w.gyimg = a.get("gyimg", out_px * 4 * 4);
    } else if (p->cfg.upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
      w.gt1 = a.get("gt1", T * 64 * 2);
      size_t px = T;
      for (int k = 0; k < p->nstage; ++k) {
        px *= (size_t)p->stage_r * p->stage_r;
        w.gup.push_back(a.get("gup" + std::to_string(k), px * 64 * 2));
      }
      w.gyimg = a.get("gyimg", px * 4 * 4);
    } else {
      w.gyimg = a.get("gyimg", T * 16 * 4);
    }
    w.gstage_w = a.get("gstage_w", (size_t)p->packed_elems * 4);
    w.gstage_side = a.get("gstage_side", (size_t)p->side_floats * 4);
  }
  w.total = a.off;
}

struct Ctx {
  srk_swinir_plan* p;
  const float* params;
  const bf16_t* packed;
  const float* side;
  unsigned char* ws;
  hipStream_t stream;
  template <typename T>
  T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
};

inline const float* side_of(const srk_swinir_plan* p, const void* packed) {
  return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(packed) + (size_t)p->packed_elems * 2);
}

WinGeom make_wgeom(int H, int W, int shift) {
  WinGeom g;
  g.H = H; g.W = W; g.nWw = W / 8; g.nW = (H / 8) * (W / 8); g.shift = shift;
  return g;
}

#define RUN(expr)          \
  do {                     \
    int rc__ = (expr);     \
    if (rc__) return rc__; \
  } while (0)

int run_conv(const Ctx& c, const ConvW& cw, int loader, int ep, const bf16_t* in, int B, int H, int W, GemmParams extra) {
  GemmParams g = extra;
  g.A = in;
  g.Wt = c.packed + cw.Wc;
  g.M = B * H * W;
  g.N = cw.NP;
  g.K = 9 * cw.CinP;
  g.B = B; g.H = H; g.W = W; g.CinP = cw.CinP;
  g.bias = c.side + cw.bc;
  g.flops = 2.0 * g.M * cw.Cout * cw.Cin * 9;
  if (g.ldo == 0) g.ldo = cw.NP;
  return srk_launch_gemm(loader, ep, g, c.stream);
}

}  // namespace

extern "C" {

int srk_swinir_plan_create(const srk_swinir_config* cfg, srk_swinir_plan** out) {
  SRK_REQUIRE(cfg && out, SRK_E_NULL, "plan_create: null argument");
  SRK_REQUIRE(cfg->window_size == 8, SRK_E_UNSUPPORTED, "HIP path supports window_size == 8 only (got %d)", cfg->window_size);
  SRK_REQUIRE(cfg->img_size >= 8, SRK_E_UNSUPPORTED, "HIP path needs img_size >= window_size (got %d)", cfg->img_size);
  SRK_REQUIRE(cfg->in_chans == 1 || cfg->in_chans == 3, SRK_E_UNSUPPORTED, "in_chans must be 1 or 3 (got %d)", cfg->in_chans);
  SRK_REQUIRE(cfg->embed_dim > 0 && cfg->embed_dim <= 256, SRK_E_UNSUPPORTED, "embed_dim must be <= 256 (got %d)", cfg->embed_dim);
  SRK_REQUIRE(cfg->num_layers > 0 && cfg->num_layers <= 16, SRK_E_UNSUPPORTED, "num_layers must be 1..16");
  SRK_REQUIRE(cfg->hidden_dim > 0 && cfg->hidden_dim <= 1024, SRK_E_UNSUPPORTED, "hidden_dim must be <= 1024");
  SRK_REQUIRE(cfg->upsampler >= SRK_UPSAMPLER_PIXELSHUFFLE && cfg->upsampler <= SRK_UPSAMPLER_NONE, SRK_E_UNSUPPORTED,
              "unknown upsampler code %d", cfg->upsampler);
  SRK_REQUIRE(cfg->resi_connection == SRK_RESI_1CONV || cfg->resi_connection == SRK_RESI_3CONV, SRK_E_UNSUPPORTED,
              "unknown resi_connection code %d", cfg->resi_connection);
  SRK_REQUIRE(cfg->resi_connection == SRK_RESI_1CONV || (cfg->embed_dim / 4 >= 1 && cfg->embed_dim / 4 <= 64), SRK_E_UNSUPPORTED,
              "'3conv' needs embed_dim / 4 <= 64 (got %d)", cfg->embed_dim / 4);
  const int s = cfg->upscale;
  int nstage = 0, stage_r = 1;
  if (cfg->upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
    if (s >= 2 && (s & (s - 1)) == 0) {
      stage_r = 2;
      for (int t = s; t > 1; t >>= 1) ++nstage;
    } else if (s == 3) {
      stage_r = 3;
      nstage = 1;
    } else {
      srk_set_error("scale %d is not supported. Supported scales: 2^n and 3.", s);   // network_swinir.py:590
      return SRK_E_SHAPE;
    }
  } else if (cfg->upsampler == SRK_UPSAMPLER_PIXELSHUFFLEDIRECT) {
    SRK_REQUIRE(s >= 1 && s * s * cfg->in_chans <= 16, SRK_E_UNSUPPORTED,
                "HIP 'pixelshuffledirect' path supports upscale^2 * in_chans <= 16 (got %d)", s * s * cfg->in_chans);
  } else if (cfg->upsampler == SRK_UPSAMPLER_NEAREST_CONV) {
    // the reference applies conv_up1 always and conv_up2 only for upscale == 4 (network_swinir.py:831-834): x2 or x4
    SRK_REQUIRE(s == 2 || s == 4, SRK_E_UNSUPPORTED, "'nearest+conv' upsamples by 2 or 4 (got %d)", s);
    stage_r = 2;
    nstage = s == 4 ? 2 : 1;
  } else {
    SRK_REQUIRE(s == 1, SRK_E_UNSUPPORTED, "the denoising head (upsampler '') keeps the image size: upscale must be 1 (got %d)", s);
  }
  srk_swinir_plan* p = new srk_swinir_plan();
  p->cfg = *cfg;
  p->C = cfg->embed_dim;
  p->CP = round_up(p->C, 64);
  p->HID = cfg->hidden_dim;
  p->HP = round_up(p->HID, 64);
  p->L = cfg->num_layers;
  p->Cimg = cfg->in_chans;
  p->nstage = nstage;
  p->stage_r = stage_r;
  p->no_shift = cfg->img_size <= cfg->window_size;
  const int C = p->C, CP = p->CP, HID = p->HID, HP = p->HP;

  p->group_desc_begin.push_back(0);
  p->group_param_begin.push_back(0);
  // ---- group 0: head ----
  if (cfg->ape) p->p_ape = add_param(p, "absolute_pos_embed", {1, (long long)cfg->img_size * cfg->img_size, C});   // a direct parameter: first
  p->p_conv_first_w = add_param(p, "conv_first.weight", {C, cfg->in_chans, 3, 3});
  p->p_conv_first_b = add_param(p, "conv_first.bias", {C});
  p->p_pe_w = add_param(p, "patch_embed.norm.weight", {C});
  p->p_pe_b = add_param(p, "patch_embed.norm.bias", {C});
  finish_group(p);
  // ---- groups 1..L: RSTBs ----
  p->nblk = 0;
  for (int l = 0; l < p->L; ++l) {
    const int nH = cfg->num_heads[l], depth = cfg->depths[l];
    if (nH <= 0 || C % nH != 0 || C / nH > 32 || depth <= 0) {
      srk_set_error("layer %d: num_heads=%d depth=%d unsupported (need C %% nH == 0 and head_dim <= 32)", l, nH, depth);
      delete p;
      return SRK_E_UNSUPPORTED;
    }
    const int dh = C / nH, CA = nH * 32;
    p->layer_first_blk.push_back(p->nblk);
    for (int j = 0; j < depth; ++j) {
      const std::string pre = "layers." + std::to_string(l) + ".residual_group.blocks." + std::to_string(j) + ".";
      BlockW b;
      b.nH = nH; b.dh = dh; b.CA = CA;
      b.shift = (j % 2 == 0 || p->no_shift) ? 0 : 4;
      b.scale = cfg->qk_scale > 0.f ? cfg->qk_scale : 1.0f / sqrtf((float)dh);
      b.n1w = add_param(p, pre + "norm1.weight", {C});
      b.n1b = add_param(p, pre + "norm1.bias", {C});
      b.rpb = add_param(p, pre + "attn.relative_position_bias_table", {225, nH});
      b.qkvw = add_param(p, pre + "attn.qkv.weight", {3 * C, C});
      b.qkvb = add_param(p, pre + "attn.qkv.bias", {3 * C});
      b.projw = add_param(p, pre + "attn.proj.weight", {C, C});
      b.projb = add_param(p, pre + "attn.proj.bias", {C});
      b.n2w = add_param(p, pre + "norm2.weight", {C});
      b.n2b = add_param(p, pre + "norm2.bias", {C});
      b.fc1w = add_param(p, pre + "mlp.fc1.weight", {HID, C});
      b.fc1b = add_param(p, pre + "mlp.fc1.bias", {HID});
      b.fc2w = add_param(p, pre + "mlp.fc2.weight", {C, HID});
      b.fc2b = add_param(p, pre + "mlp.fc2.bias", {C});
      b.Wqkv = alloc_packed(p, 3LL * CA * CP);  b.WqkvT = alloc_packed(p, 3LL * CA * CP);
      b.Wproj = alloc_packed(p, (long long)CP * CA);  b.WprojT = alloc_packed(p, (long long)CP * CA);
      b.Wfc1 = alloc_packed(p, (long long)HP * CP);  b.Wfc1T = alloc_packed(p, (long long)HP * CP);
      b.Wfc2 = alloc_packed(p, (long long)HP * CP);  b.Wfc2T = alloc_packed(p, (long long)HP * CP);
      b.bqkv = alloc_side(p, 3 * CA); b.bproj = alloc_side(p, CP); b.bfc1 = alloc_side(p, HP); b.bfc2 = alloc_side(p, CP);
      b.biasd = alloc_side(p, nH * 4096);
      add_linear(p, b.qkvw, b.Wqkv, b.WqkvT, 3 * CA, CP, 3 * C, C, NM_QKV, KM_DIRECT, nH, dh);
      add_linear(p, b.projw, b.Wproj, b.WprojT, CP, CA, C, C, NM_DIRECT, KM_HEADS, nH, dh);
      add_linear(p, b.fc1w, b.Wfc1, b.Wfc1T, HP, CP, HID, C, NM_DIRECT, KM_DIRECT, nH, dh);
      add_linear(p, b.fc2w, b.Wfc2, b.Wfc2T, CP, HP, C, HID, NM_DIRECT, KM_DIRECT, nH, dh);
      add_vec(p, b.qkvb, b.bqkv, 3 * CA, 3 * C, NM_QKV, nH, dh, 1, 64);
      add_vec(p, b.projb, b.bproj, CP, C, NM_DIRECT, nH, dh, 1, 64);
      add_vec(p, b.fc1b, b.bfc1, HP, HID, NM_DIRECT, nH, dh, 1, 64);
      add_vec(p, b.fc2b, b.bfc2, CP, C, NM_DIRECT, nH, dh, 1, 64);
      PackDesc d = base_desc(PK_RPB, b.rpb, b.biasd, nH * 64, 64);
      d.nH = nH;
      push_desc(p, d);
      p->blocks.push_back(b);
      ++p->nblk;
    }
    ResiW rw;
    add_resi(p, rw, "layers." + std::to_string(l) + ".conv", C, CP);
    p->layer_conv.push_back(rw);
    finish_group(p);
  }
  // ---- group L+1: tail ----
  p->p_norm_w = add_param(p, "norm.weight", {C});
  p->p_norm_b = add_param(p, "norm.bias", {C});
  add_resi(p, p->conv_after_body, "conv_after_body", C, CP);
  if (cfg->upsampler == SRK_UPSAMPLER_NEAREST_CONV) {            // network_swinir.py:750-759 (registration order)
    add_conv(p, p->conv_before_up, "conv_before_upsample.0", 64, C, 64, CP, 1, 64, true);
    for (int k = 0; k < nstage; ++k) {
      ConvW cw;
      add_conv(p, cw, "conv_up" + std::to_string(k + 1), 64, 64, 64, 64, 1, 64, true);
      p->up_convs.push_back(cw);
    }
    add_conv(p, p->conv_hr, "conv_hr", 64, 64, 64, 64, 1, 64, true);
    add_conv(p, p->conv_last, "conv_last", cfg->in_chans, 64, 16, 64, 1, 64, false);
  } else if (cfg->upsampler == SRK_UPSAMPLER_NONE) {              // :760-762: conv_last C -> in_chans, added to the input
    add_conv(p, p->up_direct, "conv_last", cfg->in_chans, C, 16, CP, 1, 64, false);
  } else if (cfg->upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
    add_conv(p, p->conv_before_up, "conv_before_upsample.0", 64, C, 64, CP, 1, 64, true);
    for (int k = 0; k < nstage; ++k) {
      ConvW cw;
      add_conv(p, cw, "upsample." + std::to_string(2 * k), stage_r * stage_r * 64, 64, stage_r * stage_r * 64, 64, stage_r, 64, true);
      p->up_convs.push_back(cw);
    }
    add_conv(p, p->conv_last, "conv_last", cfg->in_chans, 64, 16, 64, 1, 64, false);
  } else {
    add_conv(p, p->up_direct, "upsample.0", s * s * cfg->in_chans, C, 16, CP, 1, 64, false);
  }
  finish_group(p);
  *out = p;
  return SRK_OK;
}

void srk_swinir_plan_destroy(srk_swinir_plan* plan) { delete plan; }

int64_t srk_swinir_param_floats(const srk_swinir_plan* plan) { return plan ? plan->param_floats : 0; }
int srk_swinir_param_count(const srk_swinir_plan* plan) { return plan ? (int)plan->params.size() : 0; }

int srk_swinir_param_info(const srk_swinir_plan* plan, int index, const char** name, int64_t* offset, int64_t* numel, int* ndim,
                          int64_t shape[4]) {
  SRK_REQUIRE(plan && index >= 0 && index < (int)plan->params.size(), SRK_E_SHAPE, "param_info: bad index %d", index);
  const ParamEntry& e = plan->params[index];
  if (name) *name = e.name.c_str();
  if (offset) *offset = e.off;
  if (numel) *numel = e.numel;
  if (ndim) *ndim = e.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return SRK_OK;
}

size_t srk_swinir_const_bytes(const srk_swinir_plan* plan) { return plan ? plan->descs.size() * sizeof(PackDesc) : 0; }

int srk_swinir_const_init(srk_swinir_plan* plan, void* const_dev, srk_stream_t stream) {
  SRK_REQUIRE(plan && const_dev, SRK_E_NULL, "const_init: null argument");
  const hipError_t e = hipMemcpyAsync(const_dev, plan->descs.data(), plan->descs.size() * sizeof(PackDesc),
                                      hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    srk_set_error("const_init: hipMemcpyAsync failed: %s", hipGetErrorString(e));
    return SRK_E_LAUNCH;
  }
  plan->d_descs = reinterpret_cast<const PackDesc*>(const_dev);
  return SRK_OK;
}

size_t srk_swinir_packed_bytes(const srk_swinir_plan* plan) {
  return plan ? (size_t)plan->packed_elems * 2 + (size_t)plan->side_floats * 4 : 0;
}

int srk_swinir_plan_set_option(srk_swinir_plan* plan, const char* name, int value) {
  SRK_REQUIRE(plan && name, SRK_E_NULL, "plan_set_option: null argument");
  int old = 0;
  RUN(srk_get_option(name, &old));              // unknown names fail here
  SrkOptTls saved;
  srk_opt_scope_begin(&saved);                  // validate (and normalise) the value on a private copy of the option set
  const int rc = srk_set_option(name, value);
  int stored = value;
  if (rc == SRK_OK) srk_get_option(name, &stored);
  srk_opt_scope_end(saved);
  if (rc != SRK_OK) return rc;
  for (auto& o : plan->options)
    if (o.first == name) {
      o.second = stored;
      return SRK_OK;
    }
  plan->options.emplace_back(name, stored);
  return SRK_OK;
}

int srk_swinir_plan_get_option(const srk_swinir_plan* plan, const char* name, int* value, int* is_set) {
  SRK_REQUIRE(plan && name && value, SRK_E_NULL, "plan_get_option: null argument");
  for (const auto& o : plan->options)
    if (o.first == name) {
      *value = o.second;
      if (is_set) *is_set = 1;
      return SRK_OK;
    }
  if (is_set) *is_set = 0;
  return srk_get_option(name, value);           // not set on the plan: the calling thread's default applies
}

int srk_swinir_pack(srk_swinir_plan* plan, const float* params, void* packed, srk_stream_t stream) {
  SRK_REQUIRE(plan && params && packed, SRK_E_NULL, "pack: null argument");
  PlanOptionScope opts(plan);
  SRK_REQUIRE(plan->d_descs, SRK_E_STATE, "pack: srk_swinir_const_init has not been called");
  bf16_t* pk = reinterpret_cast<bf16_t*>(packed);
  float* side = const_cast<float*>(side_of(plan, packed));
  for (int gi = 0; gi + 1 < (int)plan->group_desc_begin.size(); ++gi) {
    const int nb = plan->group_blocks[gi];
    const int d0 = plan->group_desc_begin[gi], d1 = plan->group_desc_begin[gi + 1];
    if (nb == 0 || d1 == d0) continue;
    RUN(srk_launch_pack(plan->d_descs + d0, d1 - d0, nb, params, pk, side, (hipStream_t)stream));
  }
  return SRK_OK;
}

size_t srk_swinir_workspace_bytes(const srk_swinir_plan* plan, int B, int H0, int W0, int training) {
  if (!plan || B <= 0 || H0 <= 0 || W0 <= 0) return 0;
  PlanOptionScope opts(plan);
  layout_workspace(plan, const_cast<srk_swinir_plan*>(plan)->ws, B, H0, W0, training);
  return plan->ws.total;
}

int srk_swinir_workspace_lookup(const srk_swinir_plan* plan, const char* name, size_t* offset, size_t* bytes) {
  SRK_REQUIRE(plan && name, SRK_E_NULL, "workspace_lookup: null argument");
  auto it = plan->ws.names.find(name);
  SRK_REQUIRE(it != plan->ws.names.end(), SRK_E_STATE, "workspace_lookup: unknown buffer '%s'", name);
  if (offset) *offset = it->second.first;
  if (bytes) *bytes = it->second.second;
  return SRK_OK;
}

int srk_swinir_num_segments(const srk_swinir_plan* plan) { return plan ? plan->L + 2 : 0; }

int srk_swinir_segment_range(const srk_swinir_plan* plan, int segment, int64_t* begin, int64_t* end) {
  SRK_REQUIRE(plan && segment >= 0 && segment < plan->L + 2, SRK_E_SHAPE, "segment_range: bad segment %d", segment);
  const int group = plan->L + 1 - segment;
  if (begin) *begin = plan->group_param_begin[group];
  if (end) *end = plan->group_param_begin[group + 1];
  return SRK_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// epilogue-side description of what follows a residual-connection conv: the LayerNorm that consumes its output
struct NextNorm {
  bf16_t* out = nullptr;
  float *mean = nullptr, *rstd = nullptr;
  const float *gamma = nullptr, *beta = nullptr;
  int window = 0;
  WinGeom geom = {};
};

void set_xn(GemmParams& g, const NextNorm& nn, int C) {
  if (!nn.out) return;
  g.xn_out = nn.out; g.xn_mean = nn.mean; g.xn_rstd = nn.rstd; g.xn_gamma = nn.gamma; g.xn_beta = nn.beta; g.xn_C = C;
  g.xn_window = nn.window; g.xn_geom = nn.geom;
}

// RSTB conv / conv_after_body forward: in bf16 [T][CP] -> (EP_RES: outf = res + conv, optional fused next LayerNorm) or
// (EP_RES_BF16: outb = bf16(res + conv)).  '3conv' runs conv3x3 -> LeakyReLU(0.2) -> conv1x1 -> LeakyReLU(0.2) -> conv3x3.
int resi_forward(const Ctx& c, const ResiW& rw, const ResiAct& ra, const bf16_t* in, int ep, GemmParams g, int B, int H, int W) {
  if (!rw.three) return run_conv(c, rw.c, LD_CONV3, ep, in, B, H, W, g);
  const int T = B * H * W;
  {
    GemmParams q = {};
    q.outb = c.at<bf16_t>(ra.a1); q.scale = 0.2f;
    RUN(run_conv(c, rw.c0, LD_CONV3, EP_LRELU, in, B, H, W, q));
  }
  {
    GemmParams q = {};
    q.A = c.at<bf16_t>(ra.a1); q.lda = 64; q.Wt = c.packed + rw.W1; q.M = T; q.N = 64; q.K = 64; q.bias = c.side + rw.b1s;
    q.outb = c.at<bf16_t>(ra.a2); q.ldo = 64; q.scale = 0.2f; q.flops = 2.0 * T * rw.C4 * rw.C4;
    RUN(srk_launch_gemm(LD_ROWS, EP_LRELU, q, c.stream));
  }
  return run_conv(c, rw.c2, LD_CONV3, ep, c.at<bf16_t>(ra.a2), B, H, W, g);
}

// patch_embed.norm -> RSTBs (-> final norm fused into the last RSTB conv when fuse_final).  f0 holds conv_first's output.
int forward_body(const Ctx& c, int B, const float* drop_scale, bool fuse_final) {
  srk_swinir_plan* p = c.p;
  const Workspace& w = p->ws;
  const float* params = c.params;
  const int H = w.H, W = w.W, C = p->C, CP = p->CP, HP = p->HP;
  const int T = (int)w.T;
  const int HW = H * W;
  hipStream_t st = c.stream;
  const double fl_qkv = 2.0 * T * 3 * C * C, fl_proj = 2.0 * T * C * C, fl_mlp = 2.0 * T * (double)p->HID * C;
  const bool fuse_ln = CP == 64 || CP == 128 || CP == 192;   // forward LayerNorms ride in the producing GEMM's epilogue
  bool ln1_done = false;

  RUN(srk_launch_ln_fwd(c.at<float>(w.f0), params + p->p_pe_w, params + p->p_pe_b, nullptr, c.at<float>(w.x0),
                        c.at<float>(w.mean_pe), c.at<float>(w.rstd_pe), T, C, CP, nullptr, st));
  if (p->p_ape >= 0) {       // x = x + absolute_pos_embed  (:793-795)
    const int L_ = p->cfg.img_size * p->cfg.img_size;
    SRK_REQUIRE(HW == L_, SRK_E_SHAPE, "The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1 "
                "(ape=True: absolute_pos_embed has img_size^2 rows)", HW, L_);
    RUN(srk_launch_ape_add(c.at<float>(w.x0), params + p->p_ape, B, L_, C, CP, st));
  }

  for (int l = 0; l < p->L; ++l) {
    const int first = p->layer_first_blk[l], depth = p->cfg.depths[l];
    for (int j = 0; j < depth; ++j) {
      const int bi = first + j;
      const BlockW& bw = p->blocks[bi];
      const BlockAct& ba = w.blk[bi];
      const WinGeom geom = make_wgeom(H, W, bw.shift);
      const float* ds_attn = drop_scale ? drop_scale + ((size_t)bi * 2 + 0) * B : nullptr;
      const float* ds_mlp = drop_scale ? drop_scale + ((size_t)bi * 2 + 1) * B : nullptr;
      if (!w.training) {   // light width, inference: the whole block is one kernel (block_light.hip); it normalises its own rows
        const int rc_light = srk_launch_swin_block_light(
            c.at<float>(ba.x_in), c.at<float>(ba.x_out), (j == depth - 1) ? c.at<bf16_t>(w.layer_xb[l]) : nullptr, params + bw.n1w,
            params + bw.n1b, params + bw.n2w, params + bw.n2b, c.packed + bw.Wqkv, c.packed + bw.Wproj, c.packed + bw.Wfc1,
            c.packed + bw.Wfc2, c.side + bw.bqkv, c.side + bw.bproj, c.side + bw.bfc1, c.side + bw.bfc2, c.side + bw.biasd, bw.scale, C,
            CP, HP, bw.nH, bw.dh, p->HID, T / 64, geom, st);
        if (rc_light != SRK_NOT_COVERED) {
          RUN(rc_light);
          ln1_done = false;
          continue;
        }
      }
      // LN1 (+ roll + window partition)            network_swinir.py:245-256
      // (normally already produced by the epilogue of the kernel that wrote x_in: fc2 of the previous block / RSTB conv)
      if (!ln1_done)
        RUN(srk_launch_ln_fwd(c.at<float>(ba.x_in), params + bw.n1w, params + bw.n1b, c.at<bf16_t>(ba.xn1w), nullptr,
                              c.at<float>(ba.mean1), c.at<float>(ba.rstd1), T, C, CP, &geom, st));
      ln1_done = false;
      // qkv projection (q scaled, :121-124) + softmax(qk^T + bias + mask) v (:125-142): one kernel per window where it applies
      const int rc_fused = srk_launch_qkv_attn_fwd(c.at<bf16_t>(ba.xn1w), CP, c.packed + bw.Wqkv, c.side + bw.bqkv, bw.scale,
                                                   (w.training && !w.attn_recompute) ? c.at<bf16_t>(ba.qkv) : nullptr, c.side + bw.biasd, c.at<bf16_t>(ba.ao), T / 64, bw.nH, bw.CA, CP,
                                                   geom, st);
      if (rc_fused != SRK_NOT_COVERED) {
        RUN(rc_fused);
      } else {
        SRK_REQUIRE(!w.attn_recompute, SRK_E_STATE, "forward: the fused qkv + attention kernel is off (option attn_fused) while the "
                    "workspace was laid out for the re-projecting attention backward; set attn_bwd_fused 0 as well");
        {
          GemmParams g = {};
          g.A = c.at<bf16_t>(ba.xn1w); g.lda = CP; g.Wt = c.packed + bw.Wqkv; g.M = T; g.N = 3 * bw.CA; g.K = CP;
          g.bias = c.side + bw.bqkv; g.outb = c.at<bf16_t>(ba.qkv); g.scale = bw.scale; g.nH = bw.nH; g.CA = bw.CA; g.B_ = T / 64;
          g.flops = fl_qkv;
          g.bytes = (double)T * (2.0 * C + 6.0 * C) + 6.0 * C * C;                 // xn1 in, q/k/v out, weights once
          RUN(srk_launch_gemm(LD_ROWS, EP_QKV, g, st));
        }
        RUN(srk_launch_attn_fwd(c.at<bf16_t>(ba.qkv), c.side + bw.biasd, c.at<bf16_t>(ba.ao), T / 64, bw.nH, geom, st));
      }
      {  // proj + window reverse + un-roll + residual :143, :265-276
        GemmParams g = {};
        g.A = c.at<bf16_t>(ba.ao); g.lda = bw.CA; g.Wt = c.packed + bw.Wproj; g.M = T; g.N = CP; g.K = bw.CA;
        g.bias = c.side + bw.bproj; g.res = c.at<float>(ba.x_in); g.outf = c.at<float>(ba.x1); g.ldo = CP; g.geom = geom;
        g.rowscale = ds_attn; g.rows_per_sample = HW; g.flops = fl_proj;
        g.bytes = (double)T * (2.0 * C + 4.0 * C + 4.0 * C + (fuse_ln ? 2.0 * C + 8 : 0)) + 2.0 * C * C;   // ao, x in; x1 (+xn2) out
        if (fuse_ln) {   // LN2 (:277) of the row just written, fused into the epilogue
          g.xn_out = c.at<bf16_t>(ba.xn2); g.xn_mean = c.at<float>(ba.mean2); g.xn_rstd = c.at<float>(ba.rstd2);
          g.xn_gamma = params + bw.n2w; g.xn_beta = params + bw.n2b; g.xn_C = C; g.xn_window = 0;
        }
        RUN(srk_launch_gemm(LD_ROWS, EP_PROJ_RES, g, st));
      }
      if (!fuse_ln)   // LN2                           :277
        RUN(srk_launch_ln_fwd(c.at<float>(ba.x1), params + bw.n2w, params + bw.n2b, c.at<bf16_t>(ba.xn2), nullptr,
                              c.at<float>(ba.mean2), c.at<float>(ba.rstd2), T, C, CP, nullptr, st));
      {  // Mlp (:25-28) + second residual (:277): fc1 + GELU and fc2 + residual, as one kernel where it applies
        GemmParams g1 = {};   // fc1 + GELU
        g1.A = c.at<bf16_t>(ba.xn2); g1.lda = CP; g1.Wt = c.packed + bw.Wfc1; g1.M = T; g1.N = HP; g1.K = CP;
        g1.bias = c.side + bw.bfc1; g1.outb = w.training ? c.at<bf16_t>(ba.u) : nullptr; g1.outb2 = c.at<bf16_t>(ba.h); g1.ldo = HP; g1.flops = fl_mlp;
        g1.bytes = (double)T * (2.0 * C + 4.0 * p->HID) + 2.0 * C * p->HID;           // xn2 in; u, h out
        GemmParams g = {};    // fc2 + residual
        g.A = c.at<bf16_t>(ba.h); g.lda = HP; g.Wt = c.packed + bw.Wfc2; g.M = T; g.N = CP; g.K = HP;
        g.bias = c.side + bw.bfc2; g.res = c.at<float>(ba.x1); g.outf = c.at<float>(ba.x_out); g.ldo = CP;
        g.outb = (j == depth - 1) ? c.at<bf16_t>(w.layer_xb[l]) : nullptr;
        g.rowscale = ds_mlp; g.rows_per_sample = HW; g.flops = fl_mlp;
        g.bytes = (double)T * (2.0 * p->HID + 4.0 * C + 4.0 * C + 2.0 * C) + 2.0 * C * p->HID;   // h, x1 in; x2 + (xb | next xn1) out
        if (fuse_ln && j + 1 < depth) {   // next block's norm1 (+ its roll + window partition) fused into this epilogue
          const BlockW& nb = p->blocks[bi + 1];
          const BlockAct& na = w.blk[bi + 1];
          g.xn_out = c.at<bf16_t>(na.xn1w); g.xn_mean = c.at<float>(na.mean1); g.xn_rstd = c.at<float>(na.rstd1);
          g.xn_gamma = params + nb.n1w; g.xn_beta = params + nb.n1b; g.xn_C = C; g.xn_window = 1;
          g.xn_geom = make_wgeom(H, W, nb.shift);
          ln1_done = true;
        }
        GemmParams gf = g;    // fused: the hidden tile never leaves the CU between the two layers
        gf.A = g1.A; gf.lda = CP; gf.Wt = g1.Wt; gf.K = CP; gf.bias = g1.bias; gf.W2 = g.Wt; gf.bias2 = g.bias; gf.HP = HP;
        gf.u_out = (w.training && !w.ckpt) ? c.at<bf16_t>(ba.u) : nullptr;      // use_checkpoint: refilled by the backward pass
        gf.h_out = (w.training && !w.ckpt) ? c.at<bf16_t>(ba.h) : nullptr;
        gf.flops = 2.0 * fl_mlp;
        gf.bytes = (double)T * (2.0 * C + 4.0 * C + 4.0 * C + 2.0 * C + (w.training ? 4.0 * p->HID : 0.0)) + 4.0 * C * p->HID;   // xn2, x1 in; x2, xb | xn1 (, u, h) out
        // the only reader of u is the backward's gelu'(u): when the fused backward kernel will run (same shape conditions as the
        // fused forward), the forward keeps gelu'(u) itself
        gf.u_dgelu = gf.u_out && fuse_ln && srk_mlp_bwd_fused_enabled() && srk_mlp_dgelu_store_enabled();
        const int rc_mlp = srk_launch_mlp_fused(gf, st);
        if (rc_mlp != SRK_NOT_COVERED) {
          RUN(rc_mlp);
          p->ws.u_is_dgelu = gf.u_dgelu;
        } else {
          p->ws.u_is_dgelu = 0;
          RUN(srk_launch_gemm(LD_ROWS, EP_GELU, g1, st));
          RUN(srk_launch_gemm(LD_ROWS, EP_RES, g, st));
        }
      }
    }
    {  // RSTB conv + residual                          :481-482
      GemmParams g = {};
      g.res = c.at<float>(w.layer_in[l]); g.outf = c.at<float>(w.layer_out[l]); g.ldo = CP;
      NextNorm nn;
      if (fuse_ln && l + 1 < p->L) {     // norm1 of the next RSTB's first block
        const int nbi = p->layer_first_blk[l + 1];
        const BlockW& nb = p->blocks[nbi];
        const BlockAct& na = w.blk[nbi];
        nn.out = c.at<bf16_t>(na.xn1w); nn.mean = c.at<float>(na.mean1); nn.rstd = c.at<float>(na.rstd1);
        nn.gamma = params + nb.n1w; nn.beta = params + nb.n1b; nn.window = 1; nn.geom = make_wgeom(H, W, nb.shift);
        ln1_done = true;
      } else if (fuse_ln && fuse_final) {              // final norm (:800)
        nn.out = c.at<bf16_t>(w.xnf); nn.mean = c.at<float>(w.meanf); nn.rstd = c.at<float>(w.rstdf);
        nn.gamma = params + p->p_norm_w; nn.beta = params + p->p_norm_b;
      }
      set_xn(g, nn, C);
      RUN(resi_forward(c, p->layer_conv[l], w.resi[l], c.at<bf16_t>(w.layer_xb[l]), EP_RES, g, B, H, W));
    }
  }
  if (!fuse_ln && fuse_final)   // final norm                          :800
    RUN(srk_launch_ln_fwd(c.at<float>(w.layer_out[p->L - 1]), params + p->p_norm_w, params + p->p_norm_b, c.at<bf16_t>(w.xnf), nullptr,
                          c.at<float>(w.meanf), c.at<float>(w.rstdf), T, C, CP, nullptr, st));
  return SRK_OK;
}

}  // namespace

extern "C" {

int srk_swinir_forward(srk_swinir_plan* plan, const float* params, const void* packed, const float* x, float* y,
                       void* workspace, int B, int H0, int W0, int training, const float* drop_scale, srk_stream_t stream_) {
  SRK_REQUIRE(plan && params && packed && x && y && workspace, SRK_E_NULL, "forward: null argument");
  PlanOptionScope opts(plan);
  SRK_REQUIRE(B > 0 && H0 > 0 && W0 > 0, SRK_E_SHAPE, "forward: bad shape B=%d H=%d W=%d", B, H0, W0);
  srk_swinir_plan* p = plan;
  Workspace& w = p->ws;
  if (w.B != B || w.H0 != H0 || w.W0 != W0 || w.training != training || w.opt_sig != layout_option_sig())
    layout_workspace(p, w, B, H0, W0, training);
  SRK_REQUIRE((H0 % 8 == 0 || H0 >= 2) && (W0 % 8 == 0 || W0 >= 2), SRK_E_SHAPE, "forward: reflect padding needs size >= 2");
  SRK_REQUIRE(w.H - H0 < H0 && w.W - W0 < W0, SRK_E_SHAPE,
              "forward: reflect padding %dx%d -> %dx%d needs pad < size (as torch 'reflect')", H0, W0, w.H, w.W);
  SRK_REQUIRE(w.T < (1LL << 31) / 4, SRK_E_SHAPE, "forward: too many tokens (%lld)", w.T);
  Ctx c = {p, params, reinterpret_cast<const bf16_t*>(packed), side_of(p, packed), reinterpret_cast<unsigned char*>(workspace),
           (hipStream_t)stream_};
  const int H = w.H, W = w.W, C = p->C, CP = p->CP;
  hipStream_t st = c.stream;

  RUN(srk_launch_img_prep(x, c.at<float>(w.img4), B, p->Cimg, H0, W0, H, W, p->cfg.img_range, p->cfg.mean, st));
  RUN(srk_launch_stem_conv(c.at<float>(w.img4), params + p->p_conv_first_w, params + p->p_conv_first_b, c.at<float>(w.f0), B, H,
                           W, p->Cimg, C, CP, st));
  RUN(forward_body(c, B, drop_scale, true));
  {  // conv_after_body + long skip                      :815
    GemmParams g = {};
    g.res = c.at<float>(w.f0); g.outb = c.at<bf16_t>(w.fb); g.ldo = CP;
    RUN(resi_forward(c, p->conv_after_body, w.resi[p->L], c.at<bf16_t>(w.xnf), EP_RES_BF16, g, B, H, W));
  }
  const int s = p->cfg.upscale;
  GemmParams img = {};
  img.outf = y; img.inv_range = 1.0f / p->cfg.img_range; img.Cimg = p->Cimg; img.Hc = H0 * s; img.Wc = W0 * s;
  for (int i = 0; i < 3; ++i) img.mean[i] = p->cfg.mean[i];
  img.mean[3] = 0.f;
  if (p->cfg.upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
    {  // conv_before_upsample + LeakyReLU(0.01)           :816, :742-743
      GemmParams g = {};
      g.outb = c.at<bf16_t>(w.t1); g.scale = 0.01f;
      RUN(run_conv(c, p->conv_before_up, LD_CONV3, EP_LRELU, c.at<bf16_t>(w.fb), B, H, W, g));
    }
    const bf16_t* cur = c.at<bf16_t>(w.t1);
    int h = H, ww = W;
    for (int k = 0; k < p->nstage; ++k) {  // conv + PixelShuffle   :580-588
      GemmParams g = {};
      g.outb = c.at<bf16_t>(w.up[k]); g.r = p->stage_r; g.Cs = 64;
      RUN(run_conv(c, p->up_convs[k], LD_CONV3, EP_PS, cur, B, h, ww, g));
      cur = c.at<bf16_t>(w.up[k]);
      h *= p->stage_r;
      ww *= p->stage_r;
    }
    img.r = 1;
    RUN(run_conv(c, p->conv_last, LD_CONV3, EP_IMG, cur, B, h, ww, img));   // conv_last, /range + mean, crop  :817,:838-840
  } else if (p->cfg.upsampler == SRK_UPSAMPLER_NEAREST_CONV) {
    {  // conv_before_upsample + LeakyReLU(0.01)           :826
      GemmParams g = {};
      g.outb = c.at<bf16_t>(w.t1); g.scale = 0.01f;
      RUN(run_conv(c, p->conv_before_up, LD_CONV3, EP_LRELU, c.at<bf16_t>(w.fb), B, H, W, g));
    }
    const bf16_t* cur = c.at<bf16_t>(w.t1);
    int h = H, ww = W;
    for (int k = 0; k < p->nstage; ++k) {  // lrelu(conv_up_k(nearest x2))   :827-829, LeakyReLU(0.2) :759
      RUN(srk_launch_nn2x_bf16(cur, c.at<bf16_t>(w.nc_u[k]), B, h, ww, 64, st));
      h *= 2;
      ww *= 2;
      GemmParams g = {};
      g.outb = c.at<bf16_t>(w.nc_a[k]); g.scale = 0.2f;
      RUN(run_conv(c, p->up_convs[k], LD_CONV3, EP_LRELU, c.at<bf16_t>(w.nc_u[k]), B, h, ww, g));
      cur = c.at<bf16_t>(w.nc_a[k]);
    }
    {  // lrelu(conv_hr)                                   :830
      GemmParams g = {};
      g.outb = c.at<bf16_t>(w.nc_hr); g.scale = 0.2f;
      RUN(run_conv(c, p->conv_hr, LD_CONV3, EP_LRELU, cur, B, h, ww, g));
    }
    img.r = 1;
    RUN(run_conv(c, p->conv_last, LD_CONV3, EP_IMG, c.at<bf16_t>(w.nc_hr), B, h, ww, img));
  } else if (p->cfg.upsampler == SRK_UPSAMPLER_NONE) {
    img.r = 1;                                 // x + conv_last(res)   :832-836 (x = the normalised, padded input)
    img.res = c.at<float>(w.img4);
    RUN(run_conv(c, p->up_direct, LD_CONV3, EP_PS_IMG, c.at<bf16_t>(w.fb), B, H, W, img));
  } else {
    img.r = s;
    RUN(run_conv(c, p->up_direct, LD_CONV3, EP_PS_IMG, c.at<bf16_t>(w.fb), B, H, W, img));   // UpsampleOneStep :594-615
  }
  return SRK_OK;
}

int srk_swinir_forward_features(srk_swinir_plan* plan, const float* params, const void* packed, const float* f, float* out,
                                void* workspace, int B, int H, int W, srk_stream_t stream_) {
  SRK_REQUIRE(plan && params && packed && f && out && workspace, SRK_E_NULL, "forward_features: null argument");
  PlanOptionScope opts(plan);
  SRK_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, SRK_E_SHAPE,
              "forward_features: H and W must be positive multiples of the window size 8 (got %dx%d)", H, W);
  srk_swinir_plan* p = plan;
  Workspace& w = p->ws;
  if (w.B != B || w.H0 != H || w.W0 != W || w.training != 0) layout_workspace(p, w, B, H, W, 0);
  SRK_REQUIRE(w.T < (1LL << 31) / 4, SRK_E_SHAPE, "forward_features: too many tokens (%lld)", w.T);
  Ctx c = {p, params, reinterpret_cast<const bf16_t*>(packed), side_of(p, packed), reinterpret_cast<unsigned char*>(workspace),
           (hipStream_t)stream_};
  RUN(srk_launch_nchw_tokens(f, c.at<float>(w.f0), B, p->C, p->CP, H * W, 1, c.stream));        // PatchEmbed :524-528
  RUN(forward_body(c, B, nullptr, false));
  // final norm in fp32 (:800), then PatchUnEmbed (:562-565); f0 is free again (the long skip lives outside this function)
  RUN(srk_launch_ln_fwd(c.at<float>(w.layer_out[p->L - 1]), params + p->p_norm_w, params + p->p_norm_b, nullptr, c.at<float>(w.f0),
                        c.at<float>(w.meanf), c.at<float>(w.rstdf), (int)w.T, p->C, p->CP, nullptr, c.stream));
  RUN(srk_launch_nchw_tokens(c.at<float>(w.f0), out, B, p->C, p->CP, H * W, 0, c.stream));
  return SRK_OK;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
static int unpack_group(const Ctx& c, int group, float* grads) {
  const srk_swinir_plan* p = c.p;
  const int nb = p->group_blocks[group];
  const int d0 = p->group_desc_begin[group], d1 = p->group_desc_begin[group + 1];
  if (nb == 0 || d1 == d0) return SRK_OK;
  return srk_launch_unpack_grads(p->d_descs + d0, d1 - d0, nb, c.at<float>(p->ws.gstage_w), c.at<float>(p->ws.gstage_side), grads,
                                 c.stream);
}

static int conv_wgrad(const Ctx& c, const ConvW& cw, const bf16_t* Y, const bf16_t* X, int B, int H, int W, int r) {
  WgradParams q = {};
  q.Y = Y; q.ldy = cw.NP; q.X = X; q.ldx = cw.CinP; q.M = B * H * W; q.N = cw.NP; q.K = cw.CinP;
  q.dW = c.at<float>(c.p->ws.gstage_w) + cw.Wc; q.ldw = 9 * cw.CinP; q.db = c.at<float>(c.p->ws.gstage_side) + cw.bc;
  q.conv = 1; q.B = B; q.H = H; q.W = W; q.r = r; q.Cs = 64;
  q.flops = 2.0 * q.M * cw.Cout * cw.Cin * 9;
  return srk_launch_wgrad(q, c.stream);
}

static WgradParams lin_wgrad(const Ctx& c, const bf16_t* Y, int N, const bf16_t* X, int K, int M, long long Woff, long long boff,
                             double flops) {
  WgradParams q = {};
  q.flops = flops;
  q.Y = Y; q.ldy = N; q.X = X; q.ldx = K; q.M = M; q.N = N; q.K = K;
  q.dW = c.at<float>(c.p->ws.gstage_w) + Woff; q.ldw = K; q.db = c.at<float>(c.p->ws.gstage_side) + boff;
  return q;
}

// weight gradients + input gradient of an RSTB conv / conv_after_body.  gy: bf16 [T][CP] gradient of the conv output, xin: the
// conv's bf16 input; the input gradient leaves through `gout` with epilogue `ep_out` (its A / Wt / shape fields are set here).
static int resi_backward(const Ctx& c, const ResiW& rw, const ResiAct& ra, const bf16_t* gy, const bf16_t* xin, int ep_out,
                         GemmParams gout, int B, int H, int W) {
  const srk_swinir_plan* p = c.p;
  const int T = B * H * W, CP = p->CP;
  hipStream_t st = c.stream;
  const bf16_t* g_in = gy;          // gradient w.r.t. the output of the first conv
  const ConvW* first = &rw.c;
  if (rw.three) {
    const Workspace& w = p->ws;
    RUN(conv_wgrad(c, rw.c2, gy, c.at<bf16_t>(ra.a2), B, H, W, 1));
    {  // through conv .4 and the LeakyReLU in front of it
      GemmParams g = {};
      g.A = gy; g.Wt = c.packed + rw.c2.WcT; g.M = T; g.N = 64; g.K = 9 * CP; g.B = B; g.H = H; g.W = W; g.CinP = CP;
      g.outb = c.at<bf16_t>(w.g3a); g.aux = c.at<bf16_t>(ra.a2); g.scale = 0.2f; g.ldo = 64;
      g.flops = 2.0 * T * rw.c2.Cout * rw.c2.Cin * 9;
      RUN(srk_launch_gemm(LD_CONV3, EP_DLRELU, g, st));
    }
    {  // the 1x1 conv: weight gradient, then through it and the first LeakyReLU
      WgradParams q = lin_wgrad(c, c.at<bf16_t>(w.g3a), 64, c.at<bf16_t>(ra.a1), 64, T, rw.W1, rw.b1s, 2.0 * T * rw.C4 * rw.C4);
      RUN(srk_launch_wgrad(q, st));
      GemmParams g = {};
      g.A = c.at<bf16_t>(w.g3a); g.lda = 64; g.Wt = c.packed + rw.W1T; g.M = T; g.N = 64; g.K = 64;
      g.outb = c.at<bf16_t>(w.g3b); g.aux = c.at<bf16_t>(ra.a1); g.scale = 0.2f; g.ldo = 64; g.flops = 2.0 * T * rw.C4 * rw.C4;
      RUN(srk_launch_gemm(LD_ROWS, EP_DLRELU, g, st));
    }
    g_in = c.at<bf16_t>(w.g3b);
    first = &rw.c0;
  }
  RUN(conv_wgrad(c, *first, g_in, xin, B, H, W, 1));
  GemmParams g = gout;
  g.A = g_in; g.Wt = c.packed + first->WcT; g.M = T; g.N = CP; g.K = 9 * first->NP; g.B = B; g.H = H; g.W = W; g.CinP = first->NP;
  g.ldo = CP; g.flops = 2.0 * T * first->Cout * first->Cin * 9;
  return srk_launch_gemm(LD_CONV3, ep_out, g, st);
}

int srk_swinir_backward(srk_swinir_plan* plan, const float* params, const void* packed, float* grads, const float* d_y,
                        void* workspace, int B, int H0, int W0, const float* drop_scale, int seg_begin, int seg_end,
                        srk_stream_t stream_) {
  SRK_REQUIRE(plan && params && packed && grads && d_y && workspace, SRK_E_NULL, "backward: null argument");
  PlanOptionScope opts(plan);
  srk_swinir_plan* p = plan;
  Workspace& w = p->ws;
  SRK_REQUIRE(w.B == B && w.H0 == H0 && w.W0 == W0 && w.training == 1, SRK_E_STATE,
              "backward: no matching training forward (have B=%d %dx%d training=%d)", w.B, w.H0, w.W0, w.training);
  SRK_REQUIRE(seg_begin >= 0 && seg_begin < seg_end && seg_end <= p->L + 2, SRK_E_SHAPE, "backward: bad segment range %d..%d",
              seg_begin, seg_end);
  Ctx c = {p, params, reinterpret_cast<const bf16_t*>(packed), side_of(p, packed), reinterpret_cast<unsigned char*>(workspace),
           (hipStream_t)stream_};
  const int H = w.H, W = w.W, C = p->C, CP = p->CP, HP = p->HP;
  const int T = (int)w.T, HW = H * W;
  hipStream_t st = c.stream;
  const int s = p->cfg.upscale;
  const double fl_qkv = 2.0 * T * 3 * C * C, fl_proj = 2.0 * T * C * C, fl_mlp = 2.0 * T * (double)p->HID * C;
  const bool fuse_ln = CP == 64 || CP == 128 || CP == 192;   // LayerNorm backward inside the dgrad GEMM epilogue (row = one tile)
  float* gstage_w = c.at<float>(w.gstage_w);
  float* gstage_side = c.at<float>(w.gstage_side);
  // the weight-gradient kernels reduce their row-splits through this region of the caller's workspace; the calling
  // thread's own registration (srk_set_wgrad_workspace) is restored on every way out
  struct WgradWorkspaceScope {
    void* prev_ptr = nullptr;
    size_t prev_bytes = 0;
    WgradWorkspaceScope(void* ptr, size_t bytes) { srk_wgrad_bind_workspace(ptr, bytes, &prev_ptr, &prev_bytes); }
    ~WgradWorkspaceScope() { srk_wgrad_bind_workspace(prev_ptr, prev_bytes, nullptr, nullptr); }
  } wgrad_scope(c.at<float>(w.wgpart), WS_WORKSPACE_BYTES);

  for (int seg = seg_begin; seg < seg_end; ++seg) {
    if (seg == 0) {
      // ---------------- reconstruction tail ----------------
      RUN(srk_launch_zero_f32(gstage_w, (long long)p->packed_elems, st));        // (not hipMemsetAsync: see misc.hip, graph replays)
      RUN(srk_launch_zero_f32(gstage_side, (long long)p->side_floats, st));
      const float inv_range = 1.0f / p->cfg.img_range;
      if (p->cfg.upsampler == SRK_UPSAMPLER_PIXELSHUFFLE) {
        int hs = H, wsz = W;
        for (int k = 0; k < p->nstage; ++k) { hs *= p->stage_r; wsz *= p->stage_r; }
        RUN(srk_launch_img_grad_prep(d_y, c.at<float>(w.gyimg), B, p->Cimg, H0 * s, W0 * s, hs, wsz, 1, 4, inv_range, st));
        const ConvW& cl = p->conv_last;
        RUN(srk_launch_smallconv_wgrad(c.at<bf16_t>(w.up[p->nstage - 1]), c.at<float>(w.gyimg), grads + cl.w, grads + cl.b, B, hs,
                                       wsz, 64, 64, p->Cimg, 4, st));
        RUN(srk_launch_smallconv_dgrad(c.at<float>(w.gyimg), params + cl.w, c.at<bf16_t>(w.gup[p->nstage - 1]), B, hs, wsz, 64, 64,
                                       p->Cimg, 4, st));
        for (int k = p->nstage - 1; k >= 0; --k) {
          hs /= p->stage_r;
          wsz /= p->stage_r;   // resolution of this conv's input / pre-shuffle output
          const bf16_t* prev = k == 0 ? c.at<bf16_t>(w.t1) : c.at<bf16_t>(w.up[k - 1]);
          const ConvW& cw = p->up_convs[k];
          RUN(conv_wgrad(c, cw, c.at<bf16_t>(w.gup[k]), prev, B, hs, wsz, p->stage_r));
          GemmParams g = {};
          g.A = c.at<bf16_t>(w.gup[k]); g.Wt = c.packed + cw.WcT; g.M = B * hs * wsz; g.N = 64; g.K = 9 * cw.NP;
          g.B = B; g.H = hs; g.W = wsz; g.CinP = cw.NP; g.r = p->stage_r; g.Cs = 64; g.ldo = 64;
          g.flops = 2.0 * g.M * cw.Cout * cw.Cin * 9;
          if (k == 0) {
            g.outb = c.at<bf16_t>(w.gt1); g.aux = c.at<bf16_t>(w.t1); g.scale = 0.01f;
            RUN(srk_launch_gemm(LD_CONV3_PS, EP_DLRELU, g, st));
          } else {
            g.outb = c.at<bf16_t>(w.gup[k - 1]);
            RUN(srk_launch_gemm(LD_CONV3_PS, EP_BF16, g, st));
          }
        }
        const ConvW& cb = p->conv_before_up;
        RUN(conv_wgrad(c, cb, c.at<bf16_t>(w.gt1), c.at<bf16_t>(w.fb), B, H, W, 1));
        GemmParams g = {};
        g.A = c.at<bf16_t>(w.gt1); g.Wt = c.packed + cb.WcT; g.M = T; g.N = CP; g.K = 9 * 64; g.B = B; g.H = H; g.W = W; g.CinP = 64;
        g.outb = c.at<bf16_t>(w.gfb); g.ldo = CP; g.flops = 2.0 * T * cb.Cout * cb.Cin * 9;
        RUN(srk_launch_gemm(LD_CONV3, EP_BF16, g, st));
      } else if (p->cfg.upsampler == SRK_UPSAMPLER_NEAREST_CONV) {
        int hs = H, wsz = W;
        for (int k = 0; k < p->nstage; ++k) { hs *= 2; wsz *= 2; }
        const long long out_px = (long long)B * hs * wsz;
        RUN(srk_launch_img_grad_prep(d_y, c.at<float>(w.gyimg), B, p->Cimg, H0 * s, W0 * s, hs, wsz, 1, 4, inv_range, st));
        const ConvW& cl = p->conv_last;
        bf16_t* gA = c.at<bf16_t>(w.ncgA);
        bf16_t* gB = c.at<bf16_t>(w.ncgB);
        bf16_t* gU = c.at<bf16_t>(w.ncgU);
        RUN(srk_launch_smallconv_wgrad(c.at<bf16_t>(w.nc_hr), c.at<float>(w.gyimg), grads + cl.w, grads + cl.b, B, hs, wsz, 64, 64,
                                       p->Cimg, 4, st));
        RUN(srk_launch_smallconv_dgrad(c.at<float>(w.gyimg), params + cl.w, gA, B, hs, wsz, 64, 64, p->Cimg, 4, st));
        RUN(srk_launch_dlrelu_bf16(gA, c.at<bf16_t>(w.nc_hr), 0.2f, out_px * 64, st));        // through lrelu(conv_hr)
        {
          const ConvW& ch = p->conv_hr;
          const bf16_t* x_hr = c.at<bf16_t>(w.nc_a[p->nstage - 1]);
          RUN(conv_wgrad(c, ch, gA, x_hr, B, hs, wsz, 1));
          GemmParams g = {};
          g.A = gA; g.Wt = c.packed + ch.WcT; g.M = (int)out_px; g.N = 64; g.K = 9 * 64; g.B = B; g.H = hs; g.W = wsz; g.CinP = 64;
          g.outb = gB; g.aux = x_hr; g.scale = 0.2f; g.ldo = 64; g.flops = 2.0 * out_px * 64 * 64 * 9;
          RUN(srk_launch_gemm(LD_CONV3, EP_DLRELU, g, st));                                  // ... and lrelu(conv_up_last)
        }
        for (int k = p->nstage - 1; k >= 0; --k) {
          const ConvW& cw = p->up_convs[k];
          const long long px = (long long)B * hs * wsz;
          RUN(conv_wgrad(c, cw, gB, c.at<bf16_t>(w.nc_u[k]), B, hs, wsz, 1));
          GemmParams g = {};
          g.A = gB; g.Wt = c.packed + cw.WcT; g.M = (int)px; g.N = 64; g.K = 9 * 64; g.B = B; g.H = hs; g.W = wsz; g.CinP = 64;
          g.outb = gU; g.ldo = 64; g.flops = 2.0 * px * 64 * 64 * 9;
          RUN(srk_launch_gemm(LD_CONV3, EP_BF16, g, st));                                    // gradient of the upsampled tensor
          hs /= 2;
          wsz /= 2;
          // sum over the 2x2 children + the LeakyReLU of the tensor that was upsampled (t1: slope 0.01, conv_up1 output: 0.2)
          const bf16_t* act = k == 0 ? c.at<bf16_t>(w.t1) : c.at<bf16_t>(w.nc_a[k - 1]);
          bf16_t* dst = k == 0 ? c.at<bf16_t>(w.gt1) : gB;
          RUN(srk_launch_nn2x_sum_dlrelu(gU, act, dst, B, hs, wsz, 64, k == 0 ? 0.01f : 0.2f, st));
        }
        const ConvW& cb = p->conv_before_up;
        RUN(conv_wgrad(c, cb, c.at<bf16_t>(w.gt1), c.at<bf16_t>(w.fb), B, H, W, 1));
        GemmParams g = {};
        g.A = c.at<bf16_t>(w.gt1); g.Wt = c.packed + cb.WcT; g.M = T; g.N = CP; g.K = 9 * 64; g.B = B; g.H = H; g.W = W; g.CinP = 64;
        g.outb = c.at<bf16_t>(w.gfb); g.ldo = CP; g.flops = 2.0 * T * cb.Cout * cb.Cin * 9;
        RUN(srk_launch_gemm(LD_CONV3, EP_BF16, g, st));
      } else {
        // 'pixelshuffledirect' (s*s*in_chans output channels) and the denoising head (s == 1; its "+ x" has no parameters)
        const ConvW& cu = p->up_direct;
        RUN(srk_launch_img_grad_prep(d_y, c.at<float>(w.gyimg), B, p->Cimg, H0 * s, W0 * s, H, W, s, 16, inv_range, st));
        RUN(srk_launch_smallconv_wgrad(c.at<bf16_t>(w.fb), c.at<float>(w.gyimg), grads + cu.w, grads + cu.b, B, H, W, C, CP,
                                       cu.Cout, 16, st));
        RUN(srk_launch_smallconv_dgrad(c.at<float>(w.gyimg), params + cu.w, c.at<bf16_t>(w.gfb), B, H, W, C, CP, cu.Cout, 16, st));
      }
      // conv_after_body (gfb also feeds the long skip into f0, consumed by the head segment)
      {
        GemmParams g = {};
        g.outb = c.at<bf16_t>(w.dxn);
        RUN(resi_backward(c, p->conv_after_body, w.resi[p->L], c.at<bf16_t>(w.gfb), c.at<bf16_t>(w.xnf), EP_BF16, g, B, H, W));
      }
      // final norm
      RUN(srk_launch_ln_bwd(c.at<bf16_t>(w.dxn), c.at<float>(w.layer_out[p->L - 1]), c.at<float>(w.meanf), c.at<float>(w.rstdf),
                            params + p->p_norm_w, c.at<float>(w.gx), c.at<bf16_t>(w.gxb), grads + p->p_norm_w, grads + p->p_norm_b, T,
                            C, CP, nullptr, 0, 0, 0, 0, nullptr, HW, st));
      RUN(unpack_group(c, p->L + 1, grads));
    } else if (seg <= p->L) {
      // ---------------- RSTB l ----------------
      const int l = p->L - seg;
      const int first = p->layer_first_blk[l], depth = p->cfg.depths[l];
      {
        const int last = first + depth - 1;
        GemmParams g = {};
        g.outf = c.at<float>(w.gx2); g.outb = c.at<bf16_t>(w.gxb2);
        g.rowscale = drop_scale ? drop_scale + ((size_t)last * 2 + 1) * B : nullptr; g.rows_per_sample = HW;
        RUN(resi_backward(c, p->layer_conv[l], w.resi[l], c.at<bf16_t>(w.gxb), c.at<bf16_t>(w.layer_xb[l]), EP_F32_BF16, g, B, H, W));
      }
      bool skip_folded = false;
      for (int j = depth - 1; j >= 0; --j) {
        const int bi = first + j;
        const BlockW& bw = p->blocks[bi];
        const BlockAct& ba = w.blk[bi];
        const WinGeom geom = make_wgeom(H, W, bw.shift);
        const float* ds_attn = drop_scale ? drop_scale + ((size_t)bi * 2 + 0) * B : nullptr;
        const float* ds_prev_mlp = (drop_scale && j > 0) ? drop_scale + ((size_t)(bi - 1) * 2 + 1) * B : nullptr;
        if (w.ckpt) {
          // use_checkpoint: refill the shared u / h (fc1 + GELU of the saved norm2 output) and ao (fused qkv + attention forward of
          // the saved norm1 output) buffers of this block -- the same kernels on the same inputs as in the forward pass
          GemmParams g1 = {};
          g1.A = c.at<bf16_t>(ba.xn2); g1.lda = CP; g1.Wt = c.packed + bw.Wfc1; g1.M = T; g1.N = HP; g1.K = CP;
          g1.bias = c.side + bw.bfc1; g1.outb = c.at<bf16_t>(ba.u); g1.outb2 = c.at<bf16_t>(ba.h); g1.ldo = HP; g1.flops = fl_mlp;
          g1.bytes = (double)T * (2.0 * C + 4.0 * p->HID) + 2.0 * C * p->HID;
          RUN(srk_launch_gemm(LD_ROWS, EP_GELU, g1, st));
          const int rc_f = srk_launch_qkv_attn_fwd(c.at<bf16_t>(ba.xn1w), CP, c.packed + bw.Wqkv, c.side + bw.bqkv, bw.scale, nullptr,
                                                   c.side + bw.biasd, c.at<bf16_t>(ba.ao), T / 64, bw.nH, bw.CA, CP, geom, st);
          if (rc_f == SRK_NOT_COVERED) RUN(srk_launch_attn_fwd(c.at<bf16_t>(ba.qkv), c.side + bw.biasd, c.at<bf16_t>(ba.ao), T / 64, bw.nH, geom, st));
          else RUN(rc_f);
        }
        // MLP half as one kernel where it applies: d u = (d x2 . Wfc2) * gelu'(u) stays on the CU, d xn2 = d u . Wfc1 with LN2 backward
        // in its epilogue (gx2 += dx1, gxbw = bf16(gx2 * f_attn) in window order); d u is written once for the fc1 weight gradient
        int rc_mlpb = SRK_NOT_COVERED;
        if (fuse_ln) {
          GemmParams g = {};
          g.A = c.at<bf16_t>(w.gxb2); g.lda = CP; g.Wt = c.packed + bw.Wfc2T; g.K = CP; g.HP = HP; g.aux = c.at<bf16_t>(ba.u);
          g.u_out = c.at<bf16_t>(w.du); g.W2 = c.packed + bw.Wfc1T; g.M = T; g.N = CP; g.ldo = CP;
          g.outf = c.at<float>(w.gx2); g.outb = c.at<bf16_t>(w.gxbw); g.geom = geom; g.rowscale = ds_attn; g.rows_per_sample = HW;
          g.ln_x = c.at<float>(ba.x1); g.ln_mean = c.at<float>(ba.mean2); g.ln_rstd = c.at<float>(ba.rstd2);
          g.ln_gamma = params + bw.n2w; g.ln_dgamma = grads + bw.n2w; g.ln_dbeta = grads + bw.n2b; g.ln_C = C;
          g.ln_rows_window = 0; g.ln_stats_by_m = 0; g.ln_out_window = 1;
          g.flops = 2.0 * fl_mlp;
          g.bytes = (double)T * (2.0 * C + 2.0 * p->HID + 2.0 * p->HID + 4.0 * C + 8.0 * C + 2.0 * C) + 4.0 * C * p->HID;   // d x2, u in; d u out; x1, gx in; gx, gxbw out
          g.u_dgelu = w.u_is_dgelu && !w.ckpt;      // use_checkpoint refills u itself (EP_GELU above)
          rc_mlpb = srk_launch_mlp_fused_bwd(g, st);
          if (rc_mlpb != SRK_NOT_COVERED) RUN(rc_mlpb);
        }
        if (rc_mlpb == SRK_NOT_COVERED && w.u_is_dgelu && !w.ckpt) {
          srk_set_error("train step: the forward stored gelu'(u) for the fused MLP backward, which is not available now "
                        "(mlp_bwd_fused / gemm_stream changed between forward and backward?)");
          return SRK_E_UNSUPPORTED;
        }
        if (rc_mlpb == SRK_NOT_COVERED) {  // d h = d x2 . Wfc2 ; d u = d h * gelu'(u)
          GemmParams g = {};
          g.A = c.at<bf16_t>(w.gxb2); g.lda = CP; g.Wt = c.packed + bw.Wfc2T; g.M = T; g.N = HP; g.K = CP;
          g.outb = c.at<bf16_t>(w.du); g.aux = c.at<bf16_t>(ba.u); g.ldo = HP; g.flops = fl_mlp;
          g.bytes = (double)T * (2.0 * C + 2.0 * p->HID + 2.0 * p->HID) + 2.0 * C * p->HID;   // d x2, u in; d u out
          RUN(srk_launch_gemm(LD_ROWS, EP_DGELU, g, st));
        }
        WgradParams wq[4];   // the four weight gradients of the block go out as one launch (before gxb2 is overwritten)
        wq[0] = lin_wgrad(c, c.at<bf16_t>(w.gxb2), CP, c.at<bf16_t>(ba.h), HP, T, bw.Wfc2, bw.bfc2, fl_mlp);
        if (rc_mlpb == SRK_NOT_COVERED) {  // d xn2 = d u . Wfc1
          GemmParams g = {};
          g.A = c.at<bf16_t>(w.du); g.lda = HP; g.Wt = c.packed + bw.Wfc1T; g.M = T; g.N = CP; g.K = HP;
          g.ldo = CP; g.flops = fl_mlp;
          g.bytes = (double)T * (2.0 * p->HID + 4.0 * C + 8.0 * C + 2.0 * C) + 2.0 * C * p->HID;   // d u, x1, gx in; gx, gxbw out
          if (fuse_ln) {   // ... with LN2 backward fused into the epilogue: gx2 += dx1, gxbw = bf16(gx2 * f_attn) in window order
            g.outf = c.at<float>(w.gx2); g.outb = c.at<bf16_t>(w.gxbw); g.geom = geom; g.rowscale = ds_attn; g.rows_per_sample = HW;
            g.ln_x = c.at<float>(ba.x1); g.ln_mean = c.at<float>(ba.mean2); g.ln_rstd = c.at<float>(ba.rstd2);
            g.ln_gamma = params + bw.n2w; g.ln_dgamma = grads + bw.n2w; g.ln_dbeta = grads + bw.n2b; g.ln_C = C;
            g.ln_rows_window = 0; g.ln_stats_by_m = 0; g.ln_out_window = 1;
            RUN(srk_launch_gemm(LD_ROWS, EP_LNBWD, g, st));
          } else {
            g.outb = c.at<bf16_t>(w.dxn);
            RUN(srk_launch_gemm(LD_ROWS, EP_BF16, g, st));
          }
        }
        wq[1] = lin_wgrad(c, c.at<bf16_t>(w.du), HP, c.at<bf16_t>(ba.xn2), CP, T, bw.Wfc1, bw.bfc1, fl_mlp);
        if (!fuse_ln && rc_mlpb == SRK_NOT_COVERED) {
          // LN2 backward, iterated in window order; emits the (DropPath-scaled) bf16 gradient of x1 in window order
          RUN(srk_launch_ln_bwd(c.at<bf16_t>(w.dxn), c.at<float>(ba.x1), c.at<float>(ba.mean2), c.at<float>(ba.rstd2),
                                params + bw.n2w, c.at<float>(w.gx2), c.at<bf16_t>(w.gxbw), grads + bw.n2w, grads + bw.n2b, T, C, CP,
                                &geom, 0, 0, 1, 1, ds_attn, HW, st));
        }
        wq[2] = lin_wgrad(c, c.at<bf16_t>(w.gxbw), CP, c.at<bf16_t>(ba.ao), bw.CA, T, bw.Wproj, bw.bproj, fl_proj);
        // the slabs of d(bias) are reduced into the table gradient by the reduce launch of the weight gradients below
        int nslab;
        if (w.attn_recompute) {
          // d qkv straight from xn1 (q/k/v re-projected) and d x1 (d attn_out = d x1 . Wproj inside the kernel)
          const int rc_abf = srk_launch_qkv_attn_bwd(c.at<bf16_t>(ba.xn1w), CP, c.packed + bw.Wqkv, c.side + bw.bqkv, bw.scale,
                                                     c.at<bf16_t>(w.gxbw), CP, c.packed + bw.WprojT, c.side + bw.biasd,
                                                     c.at<bf16_t>(w.dqkv), c.at<float>(w.slab), T / 64, bw.nH, bw.CA, CP, geom, st);
          SRK_REQUIRE(rc_abf != SRK_NOT_COVERED, SRK_E_STATE, "backward: the re-projecting attention backward cannot run but the "
                      "forward stored no q/k/v (option attn_bwd_fused changed between forward and backward?)");
          RUN(rc_abf);
          nslab = srk_qkv_attn_bwd_slabs(T / 64, bw.nH, bw.CA, CP);
        } else {
          {  // d attn_out = d x1(window order) . Wproj
            GemmParams g = {};
            g.A = c.at<bf16_t>(w.gxbw); g.lda = CP; g.Wt = c.packed + bw.WprojT; g.M = T; g.N = bw.CA; g.K = CP;
            g.outb = c.at<bf16_t>(w.dao); g.ldo = bw.CA; g.flops = fl_proj; g.bytes = (double)T * 4.0 * C + 2.0 * C * C;
            RUN(srk_launch_gemm(LD_ROWS, EP_BF16, g, st));
          }
          RUN(srk_launch_attn_bwd(c.at<bf16_t>(ba.qkv), c.side + bw.biasd, c.at<bf16_t>(w.dao), c.at<bf16_t>(w.dqkv),
                                  c.at<float>(w.slab), nullptr, T / 64, bw.nH, geom, bw.scale, st));
          nslab = srk_attn_bwd_slabs(T / 64, bw.nH, nullptr);
        }
        const RpbJob rpb = {c.at<float>(w.slab), grads + bw.rpb, nslab, bw.nH};
        wq[3] = lin_wgrad(c, c.at<bf16_t>(w.dqkv), 3 * bw.CA, c.at<bf16_t>(ba.xn1w), CP, T, bw.Wqkv, bw.bqkv, fl_qkv);
        RUN(srk_launch_wgrad_multi_rpb(wq, 4, &rpb, st));   // reads gxb2 (as d x2): must precede the kernel that overwrites it
        {  // d xn1 (window order) = d qkv . Wqkv
          GemmParams g = {};
          g.A = c.at<bf16_t>(w.dqkv); g.lda = 3 * bw.CA; g.Wt = c.packed + bw.WqkvT; g.M = T; g.N = CP; g.K = 3 * bw.CA;
          g.ldo = CP; g.flops = fl_qkv;
          g.bytes = (double)T * (6.0 * C + 4.0 * C + 8.0 * C + 2.0 * C) + 6.0 * C * C;   // d qkv, x, gx in; gx, gxb2 out
          if (fuse_ln) {   // ... with LN1 backward (+ window reverse + un-roll) fused: gx2[tok] += dx, gxb2 = bf16(gx2 * f_mlp(prev))
            // the bf16 copy feeds the previous block's fc2 gradients; behind the first block of the layer nobody reads it (the RSTB skip
            // add below produces the layer's bf16 gradient)
            // -- the RSTB skip add is folded into this epilogue instead: gx = gx + gx2 + dx, gxb its bf16 copy
            g.outf = c.at<float>(w.gx2); g.outb = c.at<bf16_t>(j > 0 ? w.gxb2 : w.gxb); g.geom = geom; g.rowscale = j > 0 ? ds_prev_mlp : nullptr;
            g.rows_per_sample = HW;
            g.ln_skip = j > 0 ? nullptr : c.at<float>(w.gx);
            skip_folded = j == 0;
            g.ln_x = c.at<float>(ba.x_in); g.ln_mean = c.at<float>(ba.mean1); g.ln_rstd = c.at<float>(ba.rstd1);
            g.ln_gamma = params + bw.n1w; g.ln_dgamma = grads + bw.n1w; g.ln_dbeta = grads + bw.n1b; g.ln_C = C;
            g.ln_rows_window = 1; g.ln_stats_by_m = 1; g.ln_out_window = 0;
            RUN(srk_launch_gemm(LD_ROWS, EP_LNBWD, g, st));
          } else {
            g.outb = c.at<bf16_t>(w.dxn);
            RUN(srk_launch_gemm(LD_ROWS, EP_BF16, g, st));
            // LN1 backward (+ window reverse + un-roll); emits the bf16 gradient for the previous block's MLP branch
            RUN(srk_launch_ln_bwd(c.at<bf16_t>(w.dxn), c.at<float>(ba.x_in), c.at<float>(ba.mean1), c.at<float>(ba.rstd1),
                                  params + bw.n1w, c.at<float>(w.gx2), c.at<bf16_t>(w.gxb2), grads + bw.n1w, grads + bw.n1b, T, C, CP,
                                  &geom, 1, 1, 0, 1, ds_prev_mlp, HW, st));
          }
        }
      }
      // RSTB skip: d(layer input) = d(body input) + d(layer output)
      if (!skip_folded) RUN(srk_launch_add_f32_bf16(c.at<float>(w.gx), c.at<float>(w.gx2), c.at<bf16_t>(w.gxb), (long long)T * CP, st));
      RUN(unpack_group(c, l + 1, grads));
    } else {
      // ---------------- head: patch_embed.norm, long skip, conv_first ----------------
      if (p->p_ape >= 0)      // d absolute_pos_embed = batch sum of the gradient of the first layer's input
        RUN(srk_launch_ape_grad(c.at<float>(w.gx), grads + p->p_ape, B, p->cfg.img_size * p->cfg.img_size, C, CP, st));
      RUN(srk_launch_ln_bwd(c.at<bf16_t>(w.gxb), c.at<float>(w.f0), c.at<float>(w.mean_pe), c.at<float>(w.rstd_pe),
                            params + p->p_pe_w, c.at<float>(w.gx2), nullptr, grads + p->p_pe_w, grads + p->p_pe_b, T, C, CP, nullptr, 0,
                            0, 0, 0, nullptr, HW, st));
      RUN(srk_launch_add_bf16_into_f32(c.at<float>(w.gx2), c.at<bf16_t>(w.gfb), (long long)T * CP, st));
      RUN(srk_launch_stem_wgrad(c.at<float>(w.img4), c.at<float>(w.gx2), grads + p->p_conv_first_w, grads + p->p_conv_first_b, B, H, W,
                                p->Cimg, C, CP, st));
      RUN(unpack_group(c, 0, grads));
    }
  }
  return SRK_OK;
}

}  // extern "C"
