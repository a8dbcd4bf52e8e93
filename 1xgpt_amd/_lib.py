"""ctypes binding of libgenie_hip.so (C ABI: include/genie_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, this raises.  The oracle under
oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GENIE_HIP_LIBRARY: an explicit path to another build of the same ABI (the -DGENIE_STUDY library of 1xgpt_amd/build.py, an A/B
# variant under 1xgpt_amd/build/ab/) for tools/; unset = the shipping library beside this file
LIB_PATH = os.environ.get("GENIE_HIP_LIBRARY") or os.path.join(HERE, "libgenie_hip.so")

PREC_EXACT, PREC_BF16, PREC_F16X3 = 0, 1, 2
LAYOUT_TOKEN_MAJOR, LAYOUT_BCTHW = 0, 1
UNMASK_RANDOM, UNMASK_GREEDY = 0, 1
KC_GEMM, KC_ATTN_SPATIAL, KC_ATTN_TEMPORAL, KC_LAYERNORM, KC_OTHER, KC_FUSED = range(6)
E_ARG, E_SHAPE, E_UNSUPPORTED, E_LAUNCH, E_ASSERT = -1, -2, -3, -4, -5

c_f32p = C.c_void_p  # device pointers travel as plain integers
c_ptr = C.c_void_p


class GenieCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_layers", "num_heads", "head_dim", "d_model", "T", "S", "hidden", "factored_vocab", "num_factored",
        "image_vocab_size", "qk_norm", "use_mup", "qkv_bias", "proj_bias", "mlp_bias")] + [
        ("attn_scale", C.c_float), ("readout_mult", C.c_float), ("precision", C.c_int32)]


class AttnWeights(C.Structure):
    _fields_ = [(n, c_ptr) for n in ("qkv_w", "qkv_b", "proj_w", "proj_b", "norm_w", "norm_b", "qkv_w16", "proj_w16",
                                     "fused_w16")] + [("w16_wide", C.c_int32), ("frame_w16", c_ptr)]


class LayerWeights(C.Structure):
    _fields_ = [("norm1_w", c_ptr), ("norm1_b", c_ptr), ("spatial", AttnWeights), ("temporal", AttnWeights),
                ("norm2_w", c_ptr), ("norm2_b", c_ptr), ("fc1_w", c_ptr), ("fc1_b", c_ptr), ("fc2_w", c_ptr),
                ("fc2_b", c_ptr), ("fc1_w16", c_ptr), ("fc2_w16", c_ptr), ("mlp_fused_w16", c_ptr), ("w16_wide", C.c_int32),
                ("mlp_frame_w16", c_ptr)]


class Weights(C.Structure):
    _fields_ = [("pos_embed", c_ptr), ("mask_embed", c_ptr), ("embed", c_ptr * 4), ("out_w", c_ptr),
                ("out_b", c_ptr), ("out_w16", c_ptr), ("layers_host", C.POINTER(LayerWeights)), ("out_w16_wide", C.c_int32),
                ("out_frame_w16", c_ptr)]

WIDE_QKV, WIDE_PROJ, WIDE_FC1, WIDE_FC2 = 1, 2, 1, 2          # bits of the w16_wide fields (genie_hip.h)
FUSED_QKV_STREAM = 4                                        # spatial attention: fused_w16 = [proj stream | qkv stream]
TEMPORAL_QKV_F16X3_ELEMS = 393216   # f16 values of the f16x3 temporal qkv stream (csrc/kernels_fused_f16x3.hip)
TEMPORAL_FUSED_ELEMS, MLP_FUSED_ELEMS, SPATIAL_PROJ_FUSED_ELEMS, SPATIAL_QKV_FUSED_ELEMS = 262144, 524288, 65536, 196608        # bf16 values of the fused kernels' weight streams
ABI_VERSION = 3


# name -> (restype, argtypes); must list every symbol include/genie_hip.h declares
SIGNATURES = {
    "genie_version": (C.c_int, []),
    "genie_abi_layout": (C.c_int, [C.POINTER(C.c_size_t), C.c_int]),
    "genie_last_error": (C.c_char_p, []),
    "genie_check_config": (C.c_int, [C.POINTER(GenieCfg)]),
    "genie_workspace_bytes": (C.c_size_t, [C.POINTER(GenieCfg), C.c_int]),
    "genie_pack_bf16": (C.c_int, [c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_pack_split_f16": (C.c_int, [c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_embed": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, c_ptr, c_ptr]),
    "genie_layer_norm": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr]),
    "genie_linear": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_linear_lowp": (C.c_int, [C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    c_ptr]),
    "genie_spatial_attention": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, C.c_int, c_ptr]),
    "genie_temporal_attention": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, C.c_int, c_ptr]),
    "genie_attention_core": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_ptr,
                                       c_ptr, c_ptr]),
    "genie_st_block_forward": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(LayerWeights), c_ptr, C.c_int, c_ptr,
                                         C.c_size_t, c_ptr]),
    "genie_decoder_forward": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, c_ptr, C.c_size_t,
                                        c_ptr]),
    "genie_readout_logits": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int,
                                       C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_compute_logits": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int,
                                       C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_prefix_cache_bytes": (C.c_size_t, [C.POINTER(GenieCfg), C.c_int]),
    "genie_clean_pass": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int, c_ptr,
                                   C.c_size_t, c_ptr, C.c_size_t, c_ptr]),
    "genie_masked_frames_logits": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int, c_ptr,
                                             C.c_size_t, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_frame_pass": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, c_ptr, C.c_size_t,
                                   c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_frames_pass": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, C.c_size_t,
                                    c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_generate_workspace_bytes": (C.c_size_t, [C.POINTER(GenieCfg), C.c_int, C.c_int]),
    "genie_generate_cached": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_int, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, C.c_size_t, c_ptr, C.c_size_t, c_ptr]),
    "genie_pack_frame_w16": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr]),
    "genie_frame_linear": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_metric_hits": (C.c_int, [c_ptr, C.c_int64, c_ptr, C.c_int64, C.c_int, C.c_int64, c_ptr, C.c_double, C.c_double,
                                    C.c_double, c_ptr, c_ptr]),
    "genie_factored_ce": (C.c_int, [C.POINTER(GenieCfg), c_ptr, C.c_int, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int,
                                    c_ptr, c_ptr]),
    "genie_readout_ce": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, c_ptr, c_ptr, C.c_int, C.c_int,
                                   C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_train_activation_bytes": (C.c_size_t, [C.POINTER(GenieCfg), C.c_int]),
    "genie_train_workspace_bytes": (C.c_size_t, [C.POINTER(GenieCfg), C.c_int]),
    "genie_train_forward": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, c_ptr, C.c_int, c_ptr, C.c_size_t,
                                      c_ptr, c_ptr]),
    "genie_train_pack_weights": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), C.POINTER(Weights),
                                           C.POINTER(Weights), c_ptr]),
    "genie_train_backward_head": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), C.POINTER(Weights),
                                            C.POINTER(Weights), C.c_int, c_ptr, c_ptr, C.c_size_t, C.c_int, c_ptr]),
    "genie_train_backward_layer": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), C.POINTER(Weights),
                                             C.POINTER(Weights), C.c_int, C.c_int, c_ptr, c_ptr, C.c_size_t, C.c_int,
                                             c_ptr]),
    "genie_train_backward_embed": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, c_ptr, C.c_size_t,
                                             C.c_int, c_ptr]),
    "genie_sumsq": (C.c_int, [c_ptr, C.c_size_t, c_ptr, c_ptr, c_ptr]),
    "genie_adamw_step": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_float, c_ptr, C.c_float, c_ptr]),
    "genie_sample": (C.c_int, [C.POINTER(GenieCfg), c_ptr, C.c_int, C.c_int, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr]),
    "genie_mask_step": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int64, c_ptr, c_ptr, c_ptr, C.c_int64, C.c_int,
                                  C.c_int, c_ptr]),
    "genie_maskgit_generate": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(Weights), c_ptr, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, c_ptr, c_ptr,
                                         C.c_size_t, c_ptr]),
    "genie_profile_enable": (C.c_int, [C.c_int]),
    "genie_profile_reset": (C.c_int, []),
    "genie_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "genie_profile_kernels": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "genie_study_build": (C.c_int, []),
    "genie_pack_temporal_fused_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "genie_pack_temporal_qkv_f16x3": (C.c_int, [c_ptr, c_ptr, c_ptr]),
    "genie_temporal_prefix_fused_bf16": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_temporal_qkv_attn_f16x3": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, C.c_int64, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_pack_mlp_fused_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr]),
    "genie_pack_spatial_proj_fused_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr]),
    "genie_pack_spatial_qkv_fused_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr]),
    "genie_mlp_fused_qkv_bf16": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(LayerWeights), C.POINTER(LayerWeights), c_ptr, c_ptr, C.c_int64, c_ptr]),
    "genie_spatial_attn_proj_fused_bf16": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr]),
    "genie_temporal_fused_bf16": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(AttnWeights), c_ptr, c_ptr, C.c_int, c_ptr]),
    "genie_mlp_fused_bf16": (C.c_int, [C.POINTER(GenieCfg), C.POINTER(LayerWeights), c_ptr, c_ptr, C.c_int64, c_ptr, c_ptr, c_ptr]),
    "genie_bits_from_tokens": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_rescale_u8_bf16": (C.c_int, [c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_rescale_u8_f32": (C.c_int, [c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "genie_pack_conv_weight": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_conv3x3_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, c_ptr]),
    "genie_conv3x3_s2_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        c_ptr]),
    "genie_frames_to_nhwc_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_tokens_from_code_nhwc_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_int, C.c_int, c_ptr]),
    "genie_conv1x1_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_conv_direct_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         c_ptr]),
    "genie_conv_gn_part_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "genie_conv3x3_gn_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, c_ptr, C.c_int, c_ptr]),
    "genie_group_norm_swish_fused_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, c_ptr]),
    "genie_group_norm_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "genie_group_norm_swish_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_float, C.c_int, c_ptr]),
    "genie_bits_from_tokens_nhwc_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_int, C.c_int, c_ptr]),
    "genie_rescale_u8_nhwc_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "genie_tokens_from_bits": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr]),
}

_lib = None


class GenieHipError(RuntimeError):
    def __init__(self, code, where, msg):
        super().__init__(f"{where}: libgenie_hip error {code}: {msg}")
        self.code = code


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python 1xgpt_amd/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback for the HIP path.")
        # torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so, same soname as the /opt/rocm one this library links), and the
        # process must end up with ONE runtime -- torch's, whose streams and device pointers the library is handed.  Loaded the other way
        # round (this library before torch, e.g. __graft_entry__.build() followed by smoke() in one process) the library's launches fail with
        # "no ROCm-capable device is detected" on the GPU box.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.genie_version() != ABI_VERSION:
            raise RuntimeError(f"libgenie_hip ABI version {lib.genie_version()} != {ABI_VERSION}")
        if os.environ.get("GENIE_HIP_LIBRARY") or lib.genie_study_build():
            import sys
            print(f"1xgpt_amd: using {LIB_PATH} (study build: {bool(lib.genie_study_build())}) -- not the shipping library",
                  file=sys.stderr)
        _lib = lib
    return _lib


def check(rc, where):
    if rc != 0:
        msg = load().genie_last_error().decode(errors="replace")
        if rc == E_ASSERT:
            raise AssertionError(msg)
        if rc == E_UNSUPPORTED and "unmask_mode" in msg:
            raise NotImplementedError(msg)
        raise GenieHipError(rc, where, msg)


def make_cfg(config, precision=PREC_EXACT) -> GenieCfg:
    return GenieCfg(
        num_layers=config.num_layers, num_heads=config.num_heads, head_dim=config.d_model // config.num_heads,
        d_model=config.d_model, T=config.T, S=config.S, hidden=int(config.d_model * config.mlp_ratio),
        factored_vocab=config.factored_vocab_size, num_factored=config.num_factored_vocabs,
        image_vocab_size=config.image_vocab_size, qk_norm=int(config.qk_norm), use_mup=int(config.use_mup),
        qkv_bias=int(config.qkv_bias), proj_bias=int(config.proj_bias), mlp_bias=int(config.mlp_bias),
        attn_scale=config.attn_scale, readout_mult=config.readout_mult, precision=precision)
