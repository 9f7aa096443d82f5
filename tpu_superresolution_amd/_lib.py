"""ctypes binding of libsrk.so (include/srk.h).  The library is required: there is no fallback."""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRK_LIB_PATH") or os.path.join(HERE, "libsrk.so")   # SRK_LIB_PATH: developer A/B builds
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "srk.h")

_lib = None


class SrkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsrk error {code}: {message}")
        self.code = code
        self.message = message


class SrkUnsupported(NotImplementedError):
    pass


class WinGeom(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("shift", C.c_int)]


class SwinIRConfig(C.Structure):
    _fields_ = [("img_size", C.c_int), ("in_chans", C.c_int), ("embed_dim", C.c_int), ("num_layers", C.c_int),
                ("depths", C.c_int * 16), ("num_heads", C.c_int * 16), ("window_size", C.c_int),
                ("hidden_dim", C.c_int), ("upscale", C.c_int), ("upsampler", C.c_int), ("img_range", C.c_float),
                ("mean", C.c_float * 3), ("qk_scale", C.c_float), ("resi_connection", C.c_int), ("ape", C.c_int), ("use_checkpoint", C.c_int)]


class GemmArgs(C.Structure):
    """srk_gemm_args (include/srk.h)."""
    _fields_ = [("loader", C.c_int), ("epilogue", C.c_int), ("A", C.c_void_p), ("lda", C.c_int), ("W", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("B", C.c_int), ("H", C.c_int), ("Wd", C.c_int), ("CinP", C.c_int),
                ("r", C.c_int), ("Cs", C.c_int), ("bias", C.c_void_p), ("outf", C.c_void_p), ("outb", C.c_void_p),
                ("outb2", C.c_void_p), ("res", C.c_void_p), ("aux", C.c_void_p), ("ldo", C.c_int), ("scale", C.c_float),
                ("inv_range", C.c_float), ("mean", C.c_float * 4), ("Cimg", C.c_int), ("Hc", C.c_int), ("Wc", C.c_int),
                ("xn_out", C.c_void_p), ("xn_mean", C.c_void_p), ("xn_rstd", C.c_void_p), ("xn_gamma", C.c_void_p),
                ("xn_beta", C.c_void_p), ("xn_C", C.c_int), ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
                ("ln_x", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_dgamma", C.c_void_p),
                ("ln_dbeta", C.c_void_p), ("ln_C", C.c_int)]


class WgradProblem(C.Structure):
    """srk_wgrad_problem (include/srk.h)."""
    _fields_ = [("y", C.c_void_p), ("ldy", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("N", C.c_int), ("K", C.c_int)]


LD_ROWS, LD_CONV3, LD_CONV3_PS = 0, 1, 2
EP_BF16, EP_GELU, EP_RES, EP_LRELU, EP_PS, EP_IMG, EP_PS_IMG, EP_RES_BF16 = 0, 3, 4, 6, 7, 8, 9, 10
EP_DGELU, EP_DLRELU, EP_F32_BF16, EP_LNBWD = 5, 11, 12, 13

UPSAMPLER_PIXELSHUFFLE = 1
UPSAMPLER_PIXELSHUFFLEDIRECT = 2
UPSAMPLER_NEAREST_CONV = 3
UPSAMPLER_NONE = 4
UPSAMPLERS = {"pixelshuffle": UPSAMPLER_PIXELSHUFFLE, "pixelshuffledirect": UPSAMPLER_PIXELSHUFFLEDIRECT,
              "nearest+conv": UPSAMPLER_NEAREST_CONV, "": UPSAMPLER_NONE}
RESI = {"1conv": 0, "3conv": 1}

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_geom_p = C.POINTER(WinGeom)

_SIGNATURES = {
    "srk_version": (C.c_char_p, []),
    "srk_last_error": (C.c_char_p, []),
    "srk_window_partition": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_window_reverse": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_roll2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_pixel_shuffle": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_shift_mask": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "srk_relative_position_index": (_i, [_vp, _i, _vp]),
    "srk_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _geom_p, _vp]),
    "srk_window_attention_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _geom_p, _vp]),
    "srk_window_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _geom_p, _vp]),
    "srk_window_attention_bwd_scratch": (_sz, [_i64, _i]),
    "srk_window_attention_bwd_fused": (_i, [_vp, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _geom_p, _vp]),
    "srk_window_attention_bwd_fused_scratch": (_sz, [_i64, _i]),
    "srk_rel_pos_bias_expand": (_i, [_vp, _vp, _i, _vp]),
    "srk_linear_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_linear_wgrad_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_conv3x3_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_conv3x3_wgrad_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_cast_f32_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "srk_probe_trread": (_i, [_vp, _vp, _vp]),
    "srk_set_option": (_i, [C.c_char_p, _i]),
    "srk_probe_begin": (_i, [_i, _i]),
    "srk_probe_end": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "srk_l1_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _vp]),
    "srk_wgrad_workspace_bytes": (_i64, []),
    "srk_set_wgrad_workspace": (_i, [_vp, _i64]),
    "srk_paired_crop_u8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_batch_psnr_workspace": (_i64, [_i64, _i]),
    "srk_batch_psnr": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp]),
    "srk_eval_psnr_workspace": (_i64, [_i64, _i]),
    "srk_eval_psnr": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "srk_ssim_workspace": (_i64, [_i, _i, _i, _i]),
    "srk_ssim": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "srk_grad_sumsq": (_i, [_vp, _i64, _vp, _vp]),
    "srk_adamw_clip_step": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "srk_gemm_ex": (_i, [C.POINTER(GemmArgs), _vp]),
    "srk_mlp_fused_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "srk_img_prep": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, C.POINTER(C.c_float * 3), _vp]),
    "srk_cab_bwd_workspace": (_sz, [_i, _i, _i]),
    "srk_cab_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_win256_attention_bwd_scratch": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "srk_win256_attention_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "srk_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srk_add_f32_bf16": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "srk_add_bf16_into_f32": (_i, [_vp, _vp, _i64, _vp]),
    "srk_add_f32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "srk_img_grad_prep": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "srk_smallconv_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_smallconv_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_stem_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_conv3x3_wgrad_ps_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_mlp_fused_fwd_train": (_i, [_vp] * 15 + [_i, _vp, _i, _i, _vp]),
    "srk_rowscale_bf16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "srk_stem_conv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_swin_block_fwd": (_i, [_vp] * 16 + [_f] + [_i] * 8 + [_vp]),
    "srk_win256_attention_fwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "srk_win_attention_fwd_padded": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "srk_channel_gate_workspace": (_sz, [_i, _i, _i]),
    "srk_channel_gate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_channel_gate_act": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_dwconv3x3": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "srk_rowln_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "srk_spatial_gate": (_i, [_vp, _i, _vp, _vp, _vp, _f, _i, _vp, _i64, _i, _vp]),
    "srk_spatial_gate_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp]),
    "srk_dual_gate_combine": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "srk_channel_attention_workspace": (_sz, [_i, _i, _i]),
    "srk_channel_attention_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_linear_wgrad_multi_bf16": (_i, [C.POINTER(WgradProblem), _i, _i, _vp]),
    "srk_mlp_fused_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "srk_win_attention_bwd_padded_scratch": (_sz, [_i, _i, _i, _i, _i, _i]),
    "srk_win_attention_bwd_padded": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "srk_bn_train_coeffs": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp]),
    "srk_channel_interaction_covered": (_i, [_i, _i, _i]),
    "srk_channel_interaction_fwd": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srk_channel_interaction_bwd": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "srk_chan_attn_matrix_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_chan_attn_matrix_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_sum_rows_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "srk_bn_train_bwd_coeffs": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "srk_chan_stats_chunks": (_i64, [_i64]),
    "srk_chan_stats": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "srk_affine_act_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "srk_dgelu_affine_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _vp]),
    "srk_lincomb2_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "srk_mul_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "srk_dwconv3x3_wgrad_chunks": (_i, [_i]),
    "srk_dwconv3x3_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "srk_dual_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_spatial_gate_train": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _i, _i, _vp]),
    "srk_rowln_bwd_blocks": (_i64, [_i64]),
    "srk_rowln_bwd_bf16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp]),
    "srk_chan_gram_floats": (_i64, [_i, _i, _i]),
    "srk_chan_gram": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "srk_chan_apply_mat": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "srk_cab_add_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "srk_swinir_plan_create": (_i, [C.POINTER(SwinIRConfig), C.POINTER(_vp)]),
    "srk_swinir_plan_destroy": (None, [_vp]),
    "srk_swinir_plan_set_option": (_i, [_vp, C.c_char_p, _i]),
    "srk_swinir_plan_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "srk_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "srk_swinir_param_floats": (_i64, [_vp]),
    "srk_swinir_param_count": (_i, [_vp]),
    "srk_swinir_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i),
                                   C.POINTER(_i64 * 4)]),
    "srk_swinir_const_bytes": (_sz, [_vp]),
    "srk_swinir_const_init": (_i, [_vp, _vp, _vp]),
    "srk_swinir_packed_bytes": (_sz, [_vp]),
    "srk_swinir_pack": (_i, [_vp, _vp, _vp, _vp]),
    "srk_swinir_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "srk_swinir_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "srk_swinir_forward_features": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "srk_swinir_num_segments": (_i, [_vp]),
    "srk_swinir_segment_range": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "srk_swinir_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "srk_swinir_workspace_lookup": (_i, [_vp, C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
}


def declared_symbols() -> List[str]:
    """Every function name declared in include/srk.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srk_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Load libsrk.so (raises if it has not been built: `python -m tpu_superresolution_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension is required (no fallback path exists); "
                               "build it with `python -m tpu_superresolution_amd.build`")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_claimed_device = None


def claim_device(index: int) -> None:
    """libsrk keeps per-process device state (CU count, per-kernel LDS limits configured once): ONE GPU per process -- the
    data-parallel design is one process per GPU anyway.  The first engine claims its device; another device raises."""
    global _claimed_device
    if _claimed_device is None:
        _claimed_device = int(index)
    elif _claimed_device != int(index):
        raise RuntimeError(f"libsrk is bound to cuda:{_claimed_device} in this process (one GPU per process); "
                           f"cannot also drive cuda:{index} -- start one process per GPU")


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().srk_last_error().decode()
    if rc == -3:
        raise SrkUnsupported(msg)
    if rc == -1 and "is not supported" in msg:
        raise ValueError(msg)
    raise SrkError(rc, msg)
