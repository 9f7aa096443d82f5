"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/srk.h declares, the
plan bookkeeping (pure host code) matches the reference's parameter schema, and the product path
fails loudly without a GPU."""
import ctypes as C

import pytest
import torch

from oracle import swinir_oracle as O


@pytest.fixture(scope="module")
def lib():
    from tpu_superresolution_amd import build
    build.build(verbose=False)
    from tpu_superresolution_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    names = lib.declared_symbols()
    assert len(names) >= 35
    handle = lib.lib()
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    undeclared = [n for n in lib._SIGNATURES if n not in names]
    assert not undeclared, undeclared
    assert b"gfx950" in handle.srk_version()


@pytest.mark.parametrize("cfg", [O.SwinIRConfig.classical_x4(), O.SwinIRConfig.light_x2()])
def test_plan_parameter_table_matches_reference_schema(lib, cfg):
    from tpu_superresolution_amd.engine import SwinIRPlan
    plan = SwinIRPlan(img_size=cfg.img_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depths=cfg.depths,
                      num_heads=cfg.num_heads, window_size=cfg.window_size, mlp_ratio=cfg.mlp_ratio, upscale=cfg.upscale,
                      img_range=cfg.img_range, upsampler=cfg.upsampler)
    schema = [(k, tuple(s)) for k, s, kind in O.state_dict_schema(cfg) if kind == "param"]
    assert [(p.name, p.shape) for p in plan.params] == schema
    offs = [p.offset for p in plan.params]
    assert all(o % 64 == 0 for o in offs) and offs == sorted(offs)
    assert plan.param_floats >= sum(p.numel for p in plan.params)
    # backward segments tile the flat buffer back to front
    rs = plan.segment_ranges
    assert rs[0][1] == plan.param_floats and rs[-1][0] == 0
    assert all(rs[i][0] == rs[i + 1][1] for i in range(len(rs) - 1))
    assert len(rs) == len(cfg.depths) + 2


def test_module_state_dict_schema_and_loud_failure(lib):
    import tpu_superresolution_amd as T
    cfg = O.SwinIRConfig.light_x2()
    m = T.SwinIR(**cfg.kwargs())
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s, _ in O.state_dict_schema(cfg)]
    sd = O.random_state_dict(cfg, 1)
    m.load_state_dict(sd, strict=True)
    assert m.state_dict()["layers.0.residual_group.blocks.1.attn_mask"].shape == (64, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 16, 16))
    with pytest.raises(NotImplementedError):
        m.layers[0].residual_group.blocks[0](torch.rand(1, 64, 60), (8, 8))


def test_unsupported_configs_are_refused_by_the_c_api(lib):
    from tpu_superresolution_amd.engine import SwinIRPlan
    base = dict(img_size=64, in_chans=3, embed_dim=180, depths=[6] * 6, num_heads=[6] * 6, window_size=8, mlp_ratio=2, upscale=4,
                img_range=1.0, upsampler="pixelshuffle")
    with pytest.raises(NotImplementedError, match="window_size == 8"):
        SwinIRPlan(**{**base, "window_size": 7})
    with pytest.raises(NotImplementedError, match="head_dim <= 32"):
        SwinIRPlan(**{**base, "num_heads": [3] * 6})
    with pytest.raises(ValueError, match="scale 5 is not supported"):
        SwinIRPlan(**{**base, "upscale": 5})
    with pytest.raises(NotImplementedError, match="upsamples by 2 or 4"):
        SwinIRPlan(**{**base, "upsampler": "nearest+conv", "upscale": 3})
    with pytest.raises(NotImplementedError, match="upscale must be 1"):
        SwinIRPlan(**{**base, "upsampler": "", "upscale": 2})
    with pytest.raises(ValueError, match="resi_connection"):
        SwinIRPlan(**{**base, "resi_connection": "2conv"})
    # every head / residual connection of the reference constructor has a plan, with the reference's parameter names
    for ups, s in (("nearest+conv", 4), ("nearest+conv", 2), ("", 1), ("pixelshuffledirect", 2)):
        for resi in ("1conv", "3conv"):
            names = [q.name for q in SwinIRPlan(**{**base, "upsampler": ups, "upscale": s, "resi_connection": resi,
                                                   "depths": [2, 2], "num_heads": [6, 6]}).params]
            assert ("conv_up2.weight" in names) == (ups == "nearest+conv" and s == 4)
            assert ("conv_hr.bias" in names) == (ups == "nearest+conv")
            assert ("layers.0.conv.2.weight" in names) == (resi == "3conv") == ("conv_after_body.4.bias" in names)


def test_index_entry_points_validate_arguments_without_a_gpu(lib):
    h = lib.lib()
    assert h.srk_window_partition(None, None, 1, 8, 8, 3, 8, 4, None) == -2
    assert h.srk_shift_mask(C.c_void_p(16), 64, 64, 8, 8, None) == -1
    assert b"shift_size must in 0-window_size" in h.srk_last_error()
    assert h.srk_window_partition(C.c_void_p(16), C.c_void_p(16), 1, 13, 8, 3, 8, 4, None) == -1


def test_options_are_process_wide_with_per_plan_values():
    """SURVEY 8b 're-entrant': srk_set_option writes ONE process-wide value per option (every thread sees it -- the autograd engine runs
    the backward on its own thread); a plan carries its own values (srk_swinir_plan_set_option), applied in a thread-private copy of the
    option set around each of its calls, which leaves the process-wide values as they were.  Host-only: no kernel runs."""
    import ctypes as C
    import threading
    from tpu_superresolution_amd import _lib
    from tpu_superresolution_amd._lib import check, lib
    from tpu_superresolution_amd.engine import SwinIRPlan
    L = lib()

    def get(name):
        v = C.c_int()
        check(L.srk_get_option(name.encode(), C.byref(v)))
        return v.value

    base = get("attn_fused")
    assert base == 2 and get("mlp_fused") == 1 and get("wgrad_stream_rows") == 32
    seen = {}

    def other():
        seen["before"] = get("attn_fused")
        check(L.srk_set_option(b"attn_fused", 0))
        seen["other"] = get("attn_fused")
    check(L.srk_set_option(b"attn_fused", 1))
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["before"] == 1 and seen["other"] == 0 and get("attn_fused") == 0      # one value, whichever thread wrote it
    check(L.srk_set_option(b"attn_fused", base))
    plan = SwinIRPlan(img_size=16, in_chans=3, embed_dim=24, depths=(2,), num_heads=(2,), window_size=8, mlp_ratio=2, upscale=2, img_range=1.0,
                      upsampler="pixelshuffle", options={"attn_fused": 1, "gemm_stream_bm": 32})
    assert plan.get_option("attn_fused") == (1, True) and plan.get_option("gemm_stream_bm") == (32, True)
    assert plan.get_option("mlp_fused") == (1, False)
    assert get("attn_fused") == base and get("gemm_stream_bm") == 0        # setting a plan's option does not touch the process-wide value
    n0 = L.srk_swinir_workspace_bytes(plan.handle, 2, 16, 16, 1)            # a plan call applies and restores
    assert n0 > 0 and get("attn_fused") == base and get("gemm_stream_bm") == 0
    with pytest.raises(Exception, match="gemm_stream_bm"):
        plan.set_option("gemm_stream_bm", 7)
    with pytest.raises(_lib.SrkUnsupported, match="unknown option"):
        plan.set_option("no_such_option", 1)
    assert plan.get_option("gemm_stream_bm") == (32, True)


def test_batched_weight_packing_equals_one_tensor_at_a_time():
    """hat_arch.batched_pack: the packed (bf16 / padded / permuted / transposed) weights of HAT, DAT (inference and training subsets) and
    SwinIR window-16 built by one stack + scatter + cast per weight kind are bit-identical to the tensor-by-tensor pack (pure torch: runs
    on the CPU)."""
    import torch
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd import dat_arch, dat_train, hat_arch as ha, hat_train, swinir_w16

    class Eager:                      # a context whose helpers run eagerly (no active packer)
        def __enter__(self):
            class R:
                def resolve(self, P):
                    pass
            return R()

        def __exit__(self, *a):
            pass

    def both(fn):
        a = fn()
        mods = (ha, dat_arch, swinir_w16)
        real = ha.batched_pack
        for mod in mods:
            mod.batched_pack = Eager
        try:
            b = fn()
        finally:
            for mod in mods:
                mod.batched_pack = real
        assert a.keys() == b.keys()
        for k in a:
            assert isinstance(a[k], torch.Tensor), (k, type(a[k]))
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) and a[k].is_contiguous(), k
        return len(a)

    dev = torch.device("cpu")
    torch.manual_seed(0)
    m = T.HAT(upscale=2, in_chans=3, img_size=32, window_size=16, compress_ratio=3, squeeze_factor=6, conv_scale=0.01, overlap_ratio=0.5,
              img_range=1.0, depths=[2, 2], embed_dim=24, num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv")
    d = T.DAT(img_size=32, in_chans=3, embed_dim=48, split_size=[8, 32], depth=[3, 2], num_heads=[4, 4], expansion_factor=2.0, upscale=2, img_range=1.0)
    s = T.SwinIR(img_size=32, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, mlp_ratio=2, img_range=1.0, upscale=2,
                 upsampler="pixelshuffle")

    def reset(obj, *names):
        for n in names:
            setattr(obj, n, None)

    def hat_p():
        reset(m, "_packed")
        return dict(m._pack(dev))

    def hat_pt():
        reset(m, "_packedT")
        return dict(hat_train.pack_transposed(m, dev))

    def dat_p(train):
        def f():
            reset(d, "_packed")
            return dict(d._pack(dev, train))
        return f

    def dat_pt():
        reset(d, "_packedT")
        return dict(dat_train.pack_train(d, dev))

    def w16_p():
        reset(s, "_w16_packed")
        return dict(swinir_w16.pack(s, dev))

    assert both(hat_p) == 93 and both(hat_pt) == 37 and both(dat_p(False)) == 131 and both(dat_p(True)) == 73 and both(dat_pt) == 58
    assert both(w16_p) == 45


def test_zero_arena_serves_the_second_pass_from_one_buffer():
    """ops.ZeroArena: the first pass learns the size (torch.zeros fallbacks), the following passes carve every request out of one zeroed
    buffer (256-byte aligned pieces, independent of each other), a larger pass than learnt falls back for the excess, and outside an
    arena_scope zeros_f32 is plain torch.zeros."""
    import torch
    from tpu_superresolution_amd import ops
    dev = torch.device("cpu")
    ar = ops.ZeroArena()
    shapes = [(3, 5), (7,), (2, 2, 9), (64,)]
    with ops.arena_scope(ar, dev):
        first = [ops.zeros_f32(s, dev) for s in shapes]
    assert all(t.shape == torch.Size(s) and float(t.abs().sum()) == 0.0 for t, s in zip(first, shapes))
    assert ar.need == 4 * 64 and ar.buf is None
    with ops.arena_scope(ar, dev):
        second = [ops.zeros_f32(s, dev) for s in shapes]
        extra = ops.zeros_f32((100,), dev)                      # beyond what the first pass asked for
    base = ar.buf.data_ptr()
    assert [t.data_ptr() - base for t in second] == [0, 256, 512, 768]
    assert not (base <= extra.data_ptr() < base + ar.buf.numel() * 4)
    second[0].fill_(1.0)
    assert float(second[1].abs().sum()) == 0.0 and float(ar.buf[15:64].abs().sum()) == 0.0
    assert ar.need == 4 * 64 + 128
    kept = second[2]
    with ops.arena_scope(ar, dev):
        third = ops.zeros_f32((3, 5), dev)
    assert third.data_ptr() != second[0].data_ptr() or float(third.abs().sum()) == 0.0     # a fresh buffer per pass: `kept` is untouched
    assert float(kept.abs().sum()) == 0.0 and float(third.abs().sum()) == 0.0
    assert ops.zeros_f32((4,), dev).shape == (4,)
