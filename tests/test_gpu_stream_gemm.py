"""The persistent LDS-DMA GEMM (csrc/gemm_stream.hip) against torch and against the tile-per-workgroup GEMM.

The streaming kernel takes over the block GEMMs once M >= 64 * #CUs (a full persistent grid); the small
golden-vector models never reach it, so it is pinned here:
  * the plain bf16 epilogue through the C ABI against a torch fp32 product of the same bf16 operands;
  * every fused epilogue (QKV scatter, proj + residual + LayerNorm, GELU, fc2 + residual + next LayerNorm,
    dGELU, fused LayerNorm backward with K = 384 and K = 576, plain dgrad) through a 2-layer SwinIR of the
    cfg3 width (embed 180, 6 heads, bs 4 at 64x64 -> M = 16384 rows), forward + backward, with the kernel
    switched on and off (srk_set_option "gemm_stream"): the two paths compute the same sums in a different
    order, so outputs and gradients agree to fp32-reordering / bf16-flip noise.
"""
import numpy as np
import pytest
import torch

from oracle import swinir_oracle as O
from test_gpu_kernels import bf, close_bf16, dev
from test_gpu_model import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tpu_superresolution_amd import ops as _ops
    return _ops


def set_stream(on):
    from tpu_superresolution_amd._lib import check, lib
    check(lib().srk_set_option(b"gemm_stream", int(on)))


@pytest.mark.parametrize("M,N,K", [(16384, 192, 192), (16384 + 64 * 37, 576, 192), (32768, 384, 192), (16384, 192, 384), (24576, 384, 384)])
def test_stream_linear_bf16_vs_torch(ops, M, N, K):
    torch.manual_seed(M + N + K)
    a, w, b = bf(torch.randn(M, K)), bf(torch.randn(N, K) * 0.1), torch.randn(N)
    ref = a.float() @ w.float().t() + b
    set_stream(1)
    y = ops.linear_bf16(dev(a), dev(w), dev(b))
    close_bf16(y, ref, 2e-3)
    try:
        set_stream(0)
        y0 = ops.linear_bf16(dev(a), dev(w), dev(b))
    finally:
        set_stream(1)
    close_bf16(y0, ref, 2e-3)
    # identical bf16 results except where the fp32 sums straddle a rounding boundary
    assert float((y.float() != y0.float()).float().mean()) < 2e-3


def test_stream_identity_asymmetric(ops):
    """A = tiled identity with an asymmetric W: catches a wrong swizzle / fragment map in the DMA image."""
    K = N = 192
    M = 16384
    a = torch.eye(K).repeat(M // K + 1, 1)[:M]
    w = (torch.arange(N * K).reshape(N, K) % 251).float() / 64.0
    set_stream(1)
    y = ops.linear_bf16(dev(bf(a)), dev(bf(w)), None).cpu().float()
    ref = bf(w).float().t().repeat(M // K + 1, 1)[:M]
    assert torch.equal(y, ref)


def _mid_cfg(upsampler="pixelshuffle"):
    # the one-step head is the light model's (embed 60); at embed 180 its small-Cout dgrad is not covered (raises)
    return O.SwinIRConfig(upscale=2, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=(2, 2),
                          embed_dim=180 if upsampler == "pixelshuffle" else 60,
                          num_heads=(6, 6), mlp_ratio=2.0, upsampler=upsampler, resi_connection="1conv")


@pytest.mark.parametrize("drop", [False, True])
def test_stream_and_tile_paths_agree_on_a_cfg3_width_model(drop):
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=7, scale=1.0)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(4, 3, 64, 64, generator=gen).cuda()
    t = torch.rand(4, 3, 128, 128, generator=gen).cuda()
    ds = None
    if drop:
        keep = 1.0 - torch.linspace(0, 0.3, 4).view(4, 1, 1)
        ds = ((torch.rand(4, 2, 4, generator=gen) < keep).float() / keep).cuda()
    res = {}
    try:
        for on in (1, 0):
            set_stream(on)
            m = build(cfg, sd, train=True, drop_path_rate=0.3 if drop else 0.0)
            out = m(x, drop_scale=ds)
            loss = torch.nn.functional.l1_loss(out, t)
            loss.backward()
            torch.cuda.synchronize()
            res[on] = (out.detach().cpu(), float(loss), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    finally:
        set_stream(1)
    out1, l1, g1 = res[1]
    out0, l0, g0 = res[0]
    assert torch.isfinite(out1).all()
    assert float((out1 - out0).abs().max()) <= 2e-3 * float(out0.abs().max())
    assert abs(l1 - l0) <= 1e-4 * abs(l0)
    rels = {n: float((g1[n] - g0[n]).norm() / (g0[n].norm() + 1e-12)) for n in g0}
    worst = max(rels, key=rels.get)
    print(f"drop={drop}: out diff {float((out1 - out0).abs().max()):.3e}, grad rel-L2 median {np.median(list(rels.values())):.3e}, worst {rels[worst]:.3e} ({worst})")
    assert rels[worst] <= 2e-2, f"{worst}: {rels[worst]:.3e}"
    assert float(np.median(list(rels.values()))) <= 4e-3


@pytest.mark.parametrize("drop", [False, True])
def test_streaming_path_forward_and_gradients_vs_oracle_and_emulation(drop):
    """Direct oracle contact for the persistent kernels at the cfg3 width: embed 180 / 6 heads /
    hidden 360, bs 4 at 64x64 -> M = 16 384 rows, where the streaming GEMMs (incl. the K = 384 / 576 fused
    LayerNorm-backward epilogues), the fused qkv + attention kernel and the ring weight-gradient kernel are the ones that
    run.  Forward, loss and every parameter gradient against the fp32 oracle (bf16-noise tolerances of test_gpu_model.py)
    and against the bf16-rounding emulation (tolerances of test_gpu_emulation.py, ~5x tighter)."""
    from oracle import bf16_emulation as E
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=7, scale=1.0)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(4, 3, 64, 64, generator=gen)
    t = torch.rand(4, 3, 128, 128, generator=gen)
    ds = None
    if drop:
        keep = 1.0 - torch.linspace(0, 0.3, 4).view(4, 1, 1)
        ds = (torch.rand(4, 2, 4, generator=gen) < keep).float() / keep
        ds[1, 0, 0] = 0.0
    set_stream(1)
    m = build(cfg, sd, train=True, drop_path_rate=0.3 if drop else 0.0)
    out = m(x.cuda(), drop_scale=None if ds is None else ds.cuda())
    loss = torch.nn.functional.l1_loss(out, t.cuda())
    loss.backward()
    out = out.detach().cpu()
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    for name, (fn, fwd_tol, loss_tol, g_tol, g_med) in {"oracle": (O.loss_and_grads, 1.2e-2, 2e-3, 0.1, 0.04),
                                                      "emulation": (E.loss_and_grads_emul, 6e-3, 1e-3, 5e-2, 1e-2)}.items():
        loss_r, out_r, grads_r = fn(sd, cfg, x, t, drop_keep=ds)
        err = float((out - out_r).abs().max())
        rels = {n: float((grads[n] - grads_r[n]).norm() / (grads_r[n].norm() + 1e-12)) for n in grads_r}
        worst = max(rels, key=rels.get)
        print(f"[{name}] drop={drop}: fwd err {err:.3e} (max|ref| {float(out_r.abs().max()):.3f}), loss rel "
              f"{abs(float(loss) - float(loss_r)) / float(loss_r):.2e}, grad rel-L2 median {np.median(list(rels.values())):.3e}, "
              f"worst {rels[worst]:.3e} ({worst})")
        assert err <= fwd_tol * float(out_r.abs().max()), name
        assert abs(float(loss) - float(loss_r)) <= loss_tol * float(loss_r), name
        assert rels[worst] <= g_tol, f"{name}: {worst}: {rels[worst]:.3e}"
        assert float(np.median(list(rels.values()))) <= g_med, name


@pytest.mark.parametrize("upsampler", ["pixelshuffle", "pixelshuffledirect"])
def test_all_taps_conv_wgrad_matches_per_tap_tiles_on_the_model(upsampler):
    """csrc/convwgrad.hip (incl. the pixel-shuffled dY of the upsample conv and the MFMA image-head kernels with their
    fp32 -> bf16 hi + lo split of dY) against the per-tap tiles of wgrad.hip / the VALU image-head kernel of misc.hip."""
    from tpu_superresolution_amd._lib import check, lib
    cfg = _mid_cfg(upsampler)
    sd = O.random_state_dict(cfg, seed=9, scale=1.0)
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 64, 64, generator=gen).cuda()
    t = torch.rand(4, 3, 128, 128, generator=gen).cuda()
    res = {}
    try:
        for on in (2, 1, 0):      # 2: all-taps LDS-DMA ring (default), 1: all-taps register-staged, 0: per-tap tiles / VALU head
            check(lib().srk_set_option(b"conv_wgrad_taps", on))
            m = build(cfg, sd, train=True)
            torch.nn.functional.l1_loss(m(x), t).backward()
            torch.cuda.synchronize()
            res[on] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    finally:
        check(lib().srk_set_option(b"conv_wgrad_taps", 2))
    convs = [n for n in res[0] if (".conv." in n or n.startswith(("conv_", "upsample."))) and n != "conv_first.weight"]
    assert len(convs) >= (8 if upsampler == "pixelshuffle" else 5)
    for n in convs:
        rel21 = float((res[2][n] - res[1][n]).norm() / (res[1][n].norm() + 1e-12))
        assert rel21 <= 1e-5, f"{n}: ring vs register-staged {rel21:.3e}"      # same operands, same MFMA order
        rel = float((res[1][n] - res[0][n]).norm() / (res[0][n].norm() + 1e-12))
        # the image-head convs see identical operands in both runs; everything upstream of them also sees the bf16 flips
        # that the MFMA image-head dgrad (hi + lo split) and the fp32 VALU dgrad produce in their bf16 result
        head = n.startswith("conv_last.") or (upsampler == "pixelshuffledirect" and n.startswith("upsample."))
        assert rel <= (1e-4 if head else 2e-3), f"{n}: {rel:.3e}"


@pytest.mark.parametrize("M,N,K", [(8192, 192, 192), (16384, 576, 192), (4096 + 64 * 5, 192, 384)])
def test_streaming_linear_wgrad_vs_torch_and_register_staged_kernel(ops, M, N, K):
    """LDS-DMA ring variant of the 192x192 weight-gradient tile (swizzled transposing reads) against torch and against
    the register-staged kernel; x carries a per-(row, column) pattern so that a wrong swizzle cannot cancel out."""
    from tpu_superresolution_amd._lib import check, lib
    torch.manual_seed(N + K)
    y = bf(torch.randn(M, N) * 0.1)
    x = bf(torch.randn(M, K) + 0.01 * torch.arange(K).float()[None, :] * ((torch.arange(M) % 7).float()[:, None] - 3.0))
    ref_w = y.float().t() @ x.float()
    ref_b = y.float().sum(0)
    out = {}
    # (stream, rows per ring stage, nt loads, split partials via scratch + reduce kernel, eight waves per workgroup)
    variants = {"ring32": (1, 32, 1, 1, 1), "ring32_w4": (1, 32, 1, 1, 0), "ring64": (1, 64, 0, 1, 1), "ring64_w4": (1, 64, 0, 1, 0),
                "ring32_atomics": (1, 32, 1, 0, 1), "ring32_atomics_w4": (1, 32, 1, 0, 0), "staged": (0, 32, 1, 1, 1)}
    try:
        for name, (on, rows, nt, partials, w8) in variants.items():
            check(lib().srk_set_option(b"wgrad_stream_w8", w8))
            check(lib().srk_set_option(b"wgrad_stream", on))
            check(lib().srk_set_option(b"wgrad_stream_rows", rows))
            check(lib().srk_set_option(b"wgrad_stream_nt", nt))
            check(lib().srk_set_option(b"wgrad_partials", partials))
            dw, db = ops.linear_wgrad_bf16(dev(y), dev(x))
            out[name] = (dw.cpu(), db.cpu())
            if name == "ring32":          # fixed summation order: bit-identical when repeated
                dw2, _ = ops.linear_wgrad_bf16(dev(y), dev(x))
                assert torch.equal(dw2.cpu(), out[name][0])
    finally:
        check(lib().srk_set_option(b"wgrad_stream", 1))
        check(lib().srk_set_option(b"wgrad_stream_rows", 32))
        check(lib().srk_set_option(b"wgrad_stream_nt", 1))
        check(lib().srk_set_option(b"wgrad_partials", 1))
        check(lib().srk_set_option(b"wgrad_stream_w8", 1))
    assert torch.equal(out["ring32"][0], out["ring32_w4"][0])        # same per-element summation order in both wave shapes
    scale = max(1.0, float(ref_w.abs().max()))
    for name in variants:
        assert float((out[name][0] - ref_w).abs().max()) < 2e-3 * scale, name
        assert float((out[name][1] - ref_b).abs().max()) < 2e-2, name
        assert float((out[name][0] - out["staged"][0]).abs().max()) < 1e-3 * scale, name


@pytest.mark.parametrize("bs", [8, 7])       # 512 windows (two per workgroup) / 448 (uneven: one or two per workgroup)
def test_fused_qkv_attention_matches_separate_kernels(bs):
    """csrc/attn_fused.hip (projection + attention per window, q/k/v only in LDS on the forward path) against the
    projection GEMM followed by attn_fwd_kernel: same MFMA order and rounding points -> identical outputs; the backward
    pass consumes the q/k/v that the fused kernel wrote."""
    from tpu_superresolution_amd._lib import check, lib
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=11, scale=1.0)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(bs, 3, 64, 64, generator=gen).cuda()          # shifted and unshifted blocks, masked border windows
    t = torch.rand(bs, 3, 128, 128, generator=gen).cuda()
    res = {}
    try:
        for on in (2, 1, 0):       # three workgroups per CU / one 8-wave workgroup per CU / separate kernels
            check(lib().srk_set_option(b"attn_fused", on))
            m = build(cfg, sd, train=True)
            out = m(x)
            torch.nn.functional.l1_loss(out, t).backward()
            torch.cuda.synchronize()
            res[on] = (out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    finally:
        check(lib().srk_set_option(b"attn_fused", 2))
    for on in (2, 1):
        assert torch.isfinite(res[on][0]).all()
        assert torch.equal(res[on][0], res[0][0]), on
        for n in res[0][1]:
            rel = float((res[on][1][n] - res[0][1][n]).norm() / (res[0][1][n].norm() + 1e-12))
            assert rel <= 3e-4, f"{on} {n}: {rel:.3e}"            # bias / LayerNorm / bias-table gradients use fp32 atomics (order-dependent last bits)


def test_wgrad_workspace_is_the_callers_and_optional(ops):
    """The library never allocates: with a registered workspace the row-splits are reduced in a fixed order (bit-identical
    when repeated), without one (null registration) the same entry point falls back to fp32 atomics -- both correct."""
    from tpu_superresolution_amd._lib import check, lib
    torch.manual_seed(5)
    M, N, K = 16384, 192, 192
    y, x = dev(bf(torch.randn(M, N) * 0.1)), dev(bf(torch.randn(M, K)))
    ref = y.float().t() @ x.float()
    st = torch.cuda.current_stream().cuda_stream
    assert int(lib().srk_wgrad_workspace_bytes()) == 256 * 9216 * 16
    assert lib().srk_set_wgrad_workspace(None, 64) < 0                      # null pointer with a size

    def run_raw():
        dw = torch.zeros(N, K, device="cuda")
        check(lib().srk_linear_wgrad_bf16(y.data_ptr(), x.data_ptr(), dw.data_ptr(), None, M, N, K, st))
        return dw

    check(lib().srk_set_wgrad_workspace(None, 0))
    a1 = run_raw()                                                          # atomics
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    check(lib().srk_set_wgrad_workspace(small.data_ptr(), small.numel()))   # too small: atomics again
    a2 = run_raw()
    ws = torch.empty(int(lib().srk_wgrad_workspace_bytes()), dtype=torch.uint8, device="cuda")
    check(lib().srk_set_wgrad_workspace(ws.data_ptr(), ws.numel()))
    b1, b2 = run_raw(), run_raw()
    check(lib().srk_set_wgrad_workspace(None, 0))
    tol = 2e-3 * float(ref.abs().max())
    for got in (a1, a2, b1):
        assert float((got - ref).abs().max()) < tol
    assert torch.equal(b1, b2)


@pytest.mark.parametrize("bs,drop", [(4, False), (5, True)])       # 16384 rows (one 16-row tile list per CU) / uneven lists + DropPath
def test_fused_mlp_matches_separate_fc1_fc2_kernels(bs, drop):
    """csrc/gemm_stream.hip mlp_fused_fwd_kernel (fc1 + GELU + fc2 + residual + next LayerNorm, hidden tile in LDS) against the
    fc1 / fc2 streaming GEMMs: same MFMA order and rounding points -> bit-identical outputs in training and in inference; the
    backward pass consumes the u / h tensors the fused kernel stored."""
    from tpu_superresolution_amd._lib import check, lib
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=13, scale=1.0)
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(bs, 3, 64, 64, generator=gen).cuda()
    t = torch.rand(bs, 3, 128, 128, generator=gen).cuda()
    ds = None
    if drop:
        keep = 1.0 - torch.linspace(0, 0.3, 4).view(4, 1, 1)
        ds = ((torch.rand(4, 2, bs, generator=gen) < keep).float() / keep).cuda()
    res = {}
    check(lib().srk_set_option(b"mlp_dgelu_store", 0))        # the default keeps gelu'(u) in the u buffer: see the test below
    try:
        for on in (1, 0):
            check(lib().srk_set_option(b"mlp_fused", on))
            m = build(cfg, sd, train=True, drop_path_rate=0.3 if drop else 0.0)
            out = m(x, drop_scale=ds)
            torch.nn.functional.l1_loss(out, t).backward()
            torch.cuda.synchronize()
            u = m._engine.activation("blk1.u", torch.bfloat16).clone()
            h = m._engine.activation("blk1.h", torch.bfloat16).clone()
            m.eval()
            with torch.no_grad():
                y_inf = m(x).clone()
            res[on] = (out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}, u.cpu(), h.cpu(), y_inf.cpu())
    finally:
        check(lib().srk_set_option(b"mlp_fused", 1))
        check(lib().srk_set_option(b"mlp_dgelu_store", 1))
    assert torch.isfinite(res[1][0]).all()
    assert torch.equal(res[1][2].view(torch.int16), res[0][2].view(torch.int16)), "u differs"
    assert torch.equal(res[1][3].view(torch.int16), res[0][3].view(torch.int16)), "h differs"
    assert torch.equal(res[1][0], res[0][0]), float((res[1][0] - res[0][0]).abs().max())
    assert torch.equal(res[1][4], res[0][4])
    for n in res[0][1]:
        rel = float((res[1][1][n] - res[0][1][n]).norm() / (res[0][1][n].norm() + 1e-12))
        assert rel <= 1e-4, f"{n}: {rel:.3e}"            # bias / LayerNorm gradients use fp32 atomics (order-dependent last bits)


def test_fused_mlp_pair_keeps_gelu_derivative_between_the_passes():
    """Default of the training plan (option mlp_dgelu_store = 1): the fused MLP forward stores gelu'(u) where the separate kernels
    store u, and the fused MLP backward multiplies by it instead of evaluating erf + exp per element.  The stored tile equals
    bf16(gelu'(u)) up to one bf16 step (u is rounded to bf16 in the other variant, gelu' in this one), the forward outputs are
    bit-identical, and the gradients agree to the bf16 noise of that one rounding."""
    from tpu_superresolution_amd._lib import check, lib
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=13, scale=1.0)
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(4, 3, 64, 64, generator=gen).cuda()
    t = torch.rand(4, 3, 128, 128, generator=gen).cuda()
    res = {}
    try:
        for on in (1, 0):
            check(lib().srk_set_option(b"mlp_dgelu_store", on))
            m = build(cfg, sd, train=True)
            out = m(x)
            torch.nn.functional.l1_loss(out, t).backward()
            torch.cuda.synchronize()
            u = m._engine.activation("blk1.u", torch.bfloat16).float().cpu()
            res[on] = (out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}, u)
    finally:
        check(lib().srk_set_option(b"mlp_dgelu_store", 1))
    assert torch.equal(res[1][0], res[0][0])
    u = res[0][2].double()
    want = 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-u * u / 2) / (2 * torch.pi) ** 0.5
    assert float((res[1][2].double() - want).abs().max()) <= 2.0 ** -7, "stored gelu'(u)"     # |gelu''| <= 0.8, |u| 2^-9 + one bf16 step of a value <= 1.13
    assert float((res[1][2].double() - want).abs().mean()) <= 2e-3
    for n in res[0][1]:
        rel = float((res[1][1][n] - res[0][1][n]).norm() / (res[0][1][n].norm() + 1e-12))
        assert rel <= 5e-3, f"{n}: {rel:.3e}"


def test_two_plans_with_different_options_share_a_process():
    """Per-plan options (srk_swinir_plan_set_option): model A runs the separate kernels (attn_fused 0, mlp fused off), model B the
    defaults, their training forwards and backwards interleaved; each equals its own single run under the thread-level setter, and the
    thread's defaults are untouched afterwards."""
    import ctypes as C
    from tpu_superresolution_amd._lib import check, lib
    cfg = _mid_cfg()
    sd = O.random_state_dict(cfg, seed=13, scale=1.0)
    gen = torch.Generator().manual_seed(8)
    x = torch.rand(8, 3, 64, 64, generator=gen).cuda()
    t = torch.rand(8, 3, 128, 128, generator=gen).cuda()
    plain = {"attn_fused": 0, "mlp_fused": 0, "mlp_bwd_fused": 0, "attn_bwd_fused": 0}

    def run_alone(opts):
        try:
            for k, v in opts.items():
                check(lib().srk_set_option(k.encode(), v))
            m = build(cfg, sd, train=True)
            out = m(x)
            torch.nn.functional.l1_loss(out, t).backward()
            return out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        finally:
            for k, v in (("attn_fused", 2), ("mlp_fused", 1), ("mlp_bwd_fused", 1), ("attn_bwd_fused", 1)):
                check(lib().srk_set_option(k.encode(), v))
    ref_a, ref_b = run_alone(plain), run_alone({})
    ma, mb = build(cfg, sd, train=True), build(cfg, sd, train=True)
    ma.plan_options = dict(plain)
    ya = ma(x)
    yb = mb(x)
    la, lb = torch.nn.functional.l1_loss(ya, t), torch.nn.functional.l1_loss(yb, t)
    lb.backward()
    la.backward()
    torch.cuda.synchronize()
    for (y, m), ref in (((ya, ma), ref_a), ((yb, mb), ref_b)):
        assert torch.equal(y.detach().cpu(), ref[0])
        for n, p in m.named_parameters():
            rel = float((p.grad.cpu() - ref[1][n]).norm() / (ref[1][n].norm() + 1e-12))
            assert rel <= 3e-4, (n, rel)           # fp32 atomics in a few reductions: order-dependent last bits
    assert ma._engine.plan.get_option("attn_fused") == (0, True) and mb._engine.plan.get_option("attn_fused") == (2, False)
    v = C.c_int()
    check(lib().srk_get_option(b"attn_fused", C.byref(v)))
    assert v.value == 2


def test_options_reach_the_autograd_backward_thread():
    """The autograd engine runs CUDA backward nodes on its own thread: an option written with srk_set_option on the main thread must be
    what a backward node reads there (the A/B tests of the backward kernels rely on it), and a different thread id is really involved."""
    import ctypes as C
    import threading
    from tpu_superresolution_amd._lib import check, lib
    seen = {}

    class Spy(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            v = C.c_int()
            check(lib().srk_get_option(b"mlp_bwd_fused", C.byref(v)))
            seen["value"], seen["thread"] = v.value, threading.get_ident()
            return g

    x = torch.ones(4, device="cuda", requires_grad=True)
    try:
        check(lib().srk_set_option(b"mlp_bwd_fused", 0))
        Spy.apply(x).sum().backward()
    finally:
        check(lib().srk_set_option(b"mlp_bwd_fused", 1))
    assert seen["value"] == 0
    assert seen["thread"] != threading.get_ident()
