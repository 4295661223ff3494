"""ctypes binding of librf_flux.so (the C ABI declared in include/rf_flux.h).

The product path has NO fallback: if the shared library is missing or was not built for
gfx950, importing any op raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C reflectionflow_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librf_flux.so")
ABI_VERSION = 15

RF_EPI_STORE, RF_EPI_GELU, RF_EPI_GATE_RES, RF_EPI_QKV, RF_EPI_QKV_GELU = range(5)
# rf_gemm_schedule (rf_gemm_desc.schedule): how ONE launch is cut into workgroups; AUTO everywhere in the product
RF_SCHED_AUTO, RF_SCHED_TILE128, RF_SCHED_TILE256, RF_SCHED_STREAMK, RF_SCHED_PERSISTENT, RF_SCHED_PLAIN256, RF_SCHED_W4, RF_SCHED_W4B = range(8)
# rf_attn_kernel (rf_attn_desc.kernel)
RF_ATTN_AUTO, RF_ATTN_ONLINE128, RF_ATTN_ONLINE256 = 0, 1, 2
RF_ATTN_BOUNDED32, RF_ATTN_BOUNDED16, RF_ATTN_BOUNDED16_SPLIT, RF_ATTN_LAGGED16, RF_ATTN_LAGGED16_SPLIT = 4, 5, 6, 8, 9
RF_ATTN_BOUNDED16_MIX, RF_ATTN_LAGGED16_MIX = 10, 11
# rf_attn_bwd_kernel (rf_attn_bwd_desc.kernel): one dq form | one dK / dV form, 0 = AUTO for that kernel
RF_ATTN_BWD_AUTO, RF_ATTN_BWD_DQ_256, RF_ATTN_BWD_DQ_128, RF_ATTN_BWD_DQ_192 = 0, 1, 2, 3
RF_ATTN_BWD_DKV_128, RF_ATTN_BWD_DKV_192, RF_ATTN_BWD_DKV_128X2 = 1 << 8, 2 << 8, 3 << 8


class RFError(RuntimeError):
    pass


class rf_kseg(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64),
                ("K", C.c_int32), ("_pad", C.c_int32)]


class rf_gemm_group(C.Structure):
    _fields_ = [("seg", rf_kseg * 3), ("bias", C.c_void_p), ("M", C.c_int32), ("tok_offset", C.c_int32),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("gate", C.c_void_p), ("norm_q", C.c_void_p), ("norm_k", C.c_void_p),
                ("a_scale", C.c_void_p), ("w_scale", C.c_void_p)]


class rf_gemm_desc(C.Structure):
    _fields_ = [("N", C.c_int32), ("epilogue", C.c_int32), ("num_groups", C.c_int32), ("n_split", C.c_int32),
                ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("heads", C.c_int32), ("s_pad", C.c_int32),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("norm_eps", C.c_float), ("q_scale", C.c_float),
                ("schedule", C.c_int32), ("clock_probe", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("g", rf_gemm_group * 4)]


class rf_attn_desc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
                ("heads", C.c_int32), ("S", C.c_int32), ("s_pad", C.c_int32), ("n_main", C.c_int32),
                ("ldo", C.c_int64), ("mode", C.c_int32), ("q_prescaled", C.c_int32),
                ("cross_bias", C.c_float), ("scale", C.c_float), ("score_bound", C.c_float), ("lag_thresh", C.c_float),
                ("kernel", C.c_int32), ("mix_small", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("lse", C.c_void_p)]


class rf_attn_bwd_desc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q", "k", "v", "qt", "kt", "o", "dout")] + [("ldo", C.c_int64), ("lddo", C.c_int64)] + [
        (n, C.c_void_p) for n in ("dq", "dk", "dv", "dot", "lse", "dsum")] + [
        ("heads", C.c_int32), ("S", C.c_int32), ("s_pad", C.c_int32), ("mode", C.c_int32), ("lse_given", C.c_int32), ("kernel", C.c_int32)]


class rf_lora_fuse_entry(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("dA", C.c_void_p), ("dB", C.c_void_p), ("n0", C.c_int32), ("n", C.c_int32),
                ("r0", C.c_int32), ("r", C.c_int32), ("scaling", C.c_float), ("_pad", C.c_int32)]


RF_LORA_FUSE_MAX = 8


class rf_w8(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p)]


class rf_lora_seg(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("r_pad", C.c_int32), ("merged", C.c_int32)]


_P = C.c_void_p


class rf_double_block_weights(C.Structure):
    _fields_ = [(n, _P) for n in (
        "w_qkv", "b_qkv", "w_add_qkv", "b_add_qkv", "norm_q", "norm_k", "norm_added_q", "norm_added_k",
        "w_out", "b_out", "w_add_out", "b_add_out", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
        "w_ffc1", "b_ffc1", "w_ffc2", "b_ffc2")] + [
        ("lora_qkv", rf_lora_seg), ("lora_out", rf_lora_seg), ("lora_ff2", rf_lora_seg),
        ("qk_bound", C.c_float), ("_pad", C.c_int32)] + [
        (n, rf_w8) for n in ("q_qkv", "q_add_qkv", "q_out", "q_add_out", "q_ff1", "q_ff2", "q_ffc1", "q_ffc2")]


class rf_single_block_weights(C.Structure):
    _fields_ = [(n, _P) for n in ("w_qkv_mlp", "b_qkv_mlp", "norm_q", "norm_k", "w_out", "b_out")] + [
        ("lora_qkv_mlp", rf_lora_seg), ("lora_out", rf_lora_seg), ("qk_bound", C.c_float), ("_pad", C.c_int32),
        ("q_qkv_mlp", rf_w8), ("q_out", rf_w8)]


class rf_flux_dims(C.Structure):
    _fields_ = [("D", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32), ("S_txt", C.c_int32),
                ("S_img", C.c_int32), ("S_cond", C.c_int32), ("attn_mode", C.c_int32), ("cross_bias", C.c_float),
                ("lora_on_main", C.c_int32), ("add_cond_attn", C.c_int32), ("fp8", C.c_int32), ("_pad", C.c_int32)]


class rf_workspace(C.Structure):
    _fields_ = [("base", C.c_void_p), ("bytes", C.c_int64)]


class rf_flux_model(C.Structure):
    _fields_ = [("num_double", C.c_int32), ("num_single", C.c_int32),
                ("dbl", C.POINTER(rf_double_block_weights)), ("sgl", C.POINTER(rf_single_block_weights)),
                ("w_x_embed", _P), ("b_x_embed", _P), ("lora_x_embed", rf_lora_seg),
                ("w_ctx_embed", _P), ("b_ctx_embed", _P), ("w_proj_out", _P), ("b_proj_out", _P),
                ("in_ch", C.c_int32), ("joint_dim", C.c_int32)]


class rf_vae_conv(C.Structure):
    _fields_ = [("w", _P), ("b", _P), ("cin", C.c_int32), ("cout", C.c_int32),
                ("wf", _P), ("bf", _P), ("fold", C.c_int32), ("_pad", C.c_int32)]


class rf_vae_norm(C.Structure):
    _fields_ = [("gamma", _P), ("beta", _P)]


class rf_vae_resnet(C.Structure):
    _fields_ = [("norm1", rf_vae_norm), ("conv1", rf_vae_conv), ("norm2", rf_vae_norm), ("conv2", rf_vae_conv),
                ("shortcut", rf_vae_conv)]


class rf_vae_attn(C.Structure):
    _fields_ = [("norm", rf_vae_norm), ("w_qk", _P), ("b_qk", _P), ("w_v", _P), ("w_out", _P), ("b_out", _P),
                ("C", C.c_int32), ("_pad", C.c_int32)]


class rf_vae_weights(C.Structure):
    _fields_ = [("levels", C.c_int32), ("res_per_level", C.c_int32), ("groups", C.c_int32), ("has_attn", C.c_int32),
                ("conv_in", rf_vae_conv), ("mid0", rf_vae_resnet), ("mid1", rf_vae_resnet), ("attn", rf_vae_attn),
                ("res", (rf_vae_resnet * 3) * 4), ("resample", rf_vae_conv * 4),
                ("norm_out", rf_vae_norm), ("conv_out", rf_vae_conv)]


class rf_t5_layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln0", "w_qk", "w_v", "w_o", "ln1", "w_wi", "w_wo")]


class rf_t5_weights(C.Structure):
    _fields_ = [("layers", C.c_int32), ("d_model", C.c_int32), ("heads", C.c_int32), ("d_kv", C.c_int32), ("d_ff", C.c_int32), ("vocab", C.c_int32),
                ("eps", C.c_float), ("bias_S", C.c_int32), ("embed", C.c_void_p), ("pos_bias", C.c_void_p), ("final_ln", C.c_void_p),
                ("layer", C.POINTER(rf_t5_layer))]


class rf_clip_layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_scale", "ln1_shift", "w_qk", "b_qk", "w_v", "w_o", "b_o", "ln2_scale", "ln2_shift",
                                          "w_fc1", "b_fc1", "w_fc2", "b_fc2")]


class rf_clip_weights(C.Structure):
    _fields_ = [("layers", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("inter", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32),
                ("eps", C.c_float), ("mask_S", C.c_int32), ("tok_embed", C.c_void_p), ("pos_embed", C.c_void_p), ("mask", C.c_void_p),
                ("final_ln_scale", C.c_void_p), ("final_ln_shift", C.c_void_p), ("layer", C.POINTER(rf_clip_layer))]


# every symbol include/rf_flux.h (the product ABI) declares: (restype, argtypes)
_SIGS = {
    "rf_last_error": (C.c_char_p, []),
    "rf_abi_version": (C.c_int, []),
    "rf_target_arch": (C.c_int, []),
    "rf_gemm_bf16": (C.c_int, [C.POINTER(rf_gemm_desc), _P]),
    "rf_gemm_w8a8": (C.c_int, [C.POINTER(rf_gemm_desc), _P]),
    "rf_layernorm_modulate_fp8": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P, C.c_float, _P]),
    "rf_quant_rows_fp8": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P]),
    "rf_qk_rmsnorm_rope": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P,
                                     C.c_float, _P]),
    "rf_attention_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_float, C.c_float, C.c_int32, C.c_float, _P]),
    "rf_attention_fwd_ws": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_float, C.c_float, C.c_int32, C.c_float, _P, C.c_int64, _P]),
    "rf_attention_ws_bytes": (C.c_int64, []),
    "rf_attention": (C.c_int, [C.POINTER(rf_attn_desc), _P]),
    "rf_layernorm_modulate": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_float, _P]),
    "rf_euler_step": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "rf_silu": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rf_add_inplace": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rf_workspace_bytes": (C.c_int64, [C.POINTER(rf_flux_dims)]),
    "rf_double_block_fwd": (C.c_int, [C.POINTER(rf_flux_dims), C.POINTER(rf_double_block_weights), _P, _P, _P,
                                      C.c_int64, _P, _P, _P, _P, _P, C.POINTER(rf_workspace), _P]),
    "rf_single_block_fwd": (C.c_int, [C.POINTER(rf_flux_dims), C.POINTER(rf_single_block_weights), _P, _P,
                                      C.c_int64, _P, _P, _P, _P, C.POINTER(rf_workspace), _P]),
    "rf_mod_table_cols": (C.c_int64, [C.POINTER(rf_flux_model), C.c_int32]),
    "rf_flux_forward": (C.c_int, [C.POINTER(rf_flux_dims), C.POINTER(rf_flux_model), _P, _P, _P, _P, _P, _P, _P, _P,
                                  C.POINTER(rf_workspace), _P]),
    "rf_flux_denoise": (C.c_int, [C.POINTER(rf_flux_dims), C.POINTER(rf_flux_model), _P, _P, _P, _P, C.c_int64, _P,
                                  _P, _P, C.POINTER(C.c_float), C.c_int32, _P, C.POINTER(rf_workspace), _P]),
    "rf_vae_workspace_bytes": (C.c_int64, [C.POINTER(rf_vae_weights), C.c_int32, C.c_int32, C.c_int32]),
    "rf_vae_decode": (C.c_int, [C.POINTER(rf_vae_weights), _P, C.c_int32, C.c_int32, _P, C.POINTER(rf_workspace), _P]),
    "rf_vae_encode": (C.c_int, [C.POINTER(rf_vae_weights), _P, C.c_int32, C.c_int32, _P, C.POINTER(rf_workspace), _P]),
    "rf_t5_workspace_bytes": (C.c_int64, [C.POINTER(rf_t5_weights), C.c_int32, C.c_int32]),
    "rf_t5_encode": (C.c_int, [C.POINTER(rf_t5_weights), _P, C.c_int32, C.c_int32, _P, C.c_int64, C.POINTER(rf_workspace), _P]),
    "rf_clip_text_workspace_bytes": (C.c_int64, [C.POINTER(rf_clip_weights), C.c_int32, C.c_int32]),
    "rf_clip_text_encode": (C.c_int, [C.POINTER(rf_clip_weights), _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P, _P, C.POINTER(rf_workspace), _P]),
    # training path
    "rf_qkv_train_fwd": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                   _P, _P, _P, _P, _P, _P, _P]),
    "rf_qkv_train_bwd": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                   _P, _P, _P, _P, C.c_int64, _P]),
    "rf_attention_bwd": (C.c_int, [C.POINTER(rf_attn_bwd_desc), _P]),
    "rf_train_partials_bytes": (C.c_int64, [C.c_int32]),
    "rf_layernorm_modulate_bwd": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P, C.c_float,
                                            _P, _P, _P, C.c_int64, _P]),
    "rf_gate_bwd": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    "rf_gate_residual": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "rf_gelu": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "rf_gelu_bwd": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "rf_transpose_bf16": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, C.c_int64, C.c_int32, _P]),
    "rf_lora_fuse": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rf_lora_unfuse_grads": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "rf_gemm_tn_skinny_ws_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "rf_gemm_tn_skinny": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "rf_lora_adamw": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "rf_lora_prodigy_partials_bytes": (C.c_int64, [C.c_int64]),
    "rf_lora_clip_grad_norm": (C.c_int, [_P, C.c_int64, C.c_float, C.c_float, _P, C.c_int64, _P, _P]),
    "rf_lora_prodigy": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P] + [C.c_float] * 6 + [C.c_int32] * 3 + [C.c_float] * 3 +
                        [_P, C.c_int64, _P]),
}
RF_KC_NAMES = ("gemm_main", "gemm_small", "attention", "rowop", "gemm_w8", "quant", "attention_bwd")
# include/rf_flux_debug.h: timing hooks and read-only introspection (tests, bench, tools); not part of the drop-in surface.
# librf_flux.so exports NO kernel-selecting switch: tests pin a kernel per launch through rf_gemm_desc.schedule / rf_attn_desc.kernel.
_EXTRA_SIGS = {
    "rf_time_gemm_w8a8": (C.c_int, [C.POINTER(rf_gemm_desc), C.c_int32, C.POINTER(C.c_float), _P]),
    "rf_time_gemm": (C.c_int, [C.POINTER(rf_gemm_desc), C.c_int32, C.POINTER(C.c_float), _P]),
    "rf_profile_begin": (C.c_int, [C.c_int32]),
    "rf_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "rf_debug_last_attn_path": (C.c_int, []), "rf_debug_last_attn_bwd_path": (C.c_int, []), "rf_debug_last_gemm_path": (C.c_int, []),
               "rf_debug_clock_probe": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
               "rf_debug_sk_plan": (C.c_int, [C.POINTER(rf_gemm_desc), C.c_int32, C.POINTER(C.c_int32)]),
               "rf_debug_attn_mix_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)])}
_lib = None


def load():
    """dlopen librf_flux.so and bind every declared symbol; raises RFError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RFError(f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
                      "run `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm bundles its own libamdhip64 (same soname): import torch FIRST so that librf_flux
    # binds to the HIP runtime torch's streams and allocations live in (two runtimes in one process
    # cannot see each other's devices).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**_SIGS, **_EXTRA_SIGS}.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RFError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.rf_abi_version() != ABI_VERSION:
        raise RFError(f"librf_flux ABI {lib.rf_abi_version()} != binding {ABI_VERSION}: rebuild")
    if lib.rf_target_arch() != 950:
        raise RFError("librf_flux was not built for gfx950")
    _lib = lib
    return lib


def declared_symbols():
    """the product ABI (include/rf_flux.h)"""
    return sorted(_SIGS)


def debug_symbols():
    """measurement / introspection entry points (include/rf_flux_debug.h)"""
    return sorted(_EXTRA_SIGS)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rf_last_error().decode(errors="replace")
        raise RFError(f"{what or 'librf_flux'} failed with status {rc}: {msg}")
