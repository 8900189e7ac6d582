"""ctypes loader for the C-ABI library ``libctclip_hip.so`` (declared in ``include/ctclip_hip.h``).

The product path FAILS LOUDLY when the HIP extension is missing: there is no eager/CPU fallback.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTCLIP_LIB") or os.path.join(_HERE, "libctclip_hip.so")   # CTCLIP_LIB: profiling builds only

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
_U64, _U32, _D = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_double

# name -> (restype, argtypes).  Every entry point ends with a hipStream_t (void*) unless noted.
SIGNATURES = {
    "ctclip_abi_version": (_I, []),
    "ctclip_last_error": (c_char_p, []),
    "ctclip_target_arch": (c_char_p, []),
    "ctclip_gemm": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P, _L, _P]),
    "ctclip_gemm_workspace": (_L, [_L, _L, _L, _I, _I]),
    "ctclip_gemm_nt2_select": (_I, [_I]),
    "ctclip_gemm_dw_db": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _P]),
    "ctclip_gemm_argmax_workspace": (_L, [_L, _L]),
    "ctclip_gemm_argmax": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _L, _I, _P, _L, _P]),
    "ctclip_gemm_argmax_hilo": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _L, _P, _L, _P]),
    "ctclip_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "ctclip_layernorm_bwd_workspace": (_L, [_L, _I]),
    "ctclip_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
    "ctclip_layernorm_bwd_partials": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
    "ctclip_layernorm_bwd_reduce": (_I, [_P, _P, _P, _L, _I, _P]),
    "ctclip_patch_ln_fwd": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "ctclip_l2norm_rows": (_I, [_P, _P, _P, _L, _I, _L, _F, _I, _I, _P]),
    "ctclip_l2norm_split3": (_I, [_P, _P, _P, _L, _I, _L, _F, _I, _I, _P]),
    "ctclip_segment_sum_workspace": (_L, [_L, _I]),
    "ctclip_segment_sum": (_I, [_P, _I, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P]),
    "ctclip_peg_fwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "ctclip_peg_bwd_workspace": (_L, [_L, _I, _I, _I]),
    "ctclip_peg_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _L, _P]),
    "ctclip_head_transpose": (_I, [_P, _P, _I, _I, _I, _I, _I, _L, _I, _P]),
    "ctclip_qk_norm_fwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _I, _P]),
    "ctclip_qk_norm_bwd_workspace": (_L, [_L, _I, _I]),
    "ctclip_qk_norm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _L, _L, _I, _P, _L, _P]),
    "ctclip_attn_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _F, _F, _U64, _I, _P]),
    "ctclip_attn_bwd_workspace": (_L, [_I, _I, _I]),
    "ctclip_attn_bwd": (_I, [_P] * 10 + [_I, _I] + [_P] * 6 + [_I] * 5 + [_L] * 8 + [_F, _F, _U64, _I, _P, _L, _P]),
    "ctclip_accumulate_f32": (_I, [_P, _P, _L, _P]),
    "ctclip_geglu_weight_interleave": (_I, [_P, _P, _I, _I, _I, _L, _P]),
    "ctclip_gemm_geglu": (_I, [_P, _P, _P, _P, _L, _I, _L, _L, _L, _L, _L, _I, _P]),
    "ctclip_gemm_dgeglu": (_I, [_P, _P, _P, _P, _L, _I, _L, _L, _L, _L, _L, _I, _P]),
    "ctclip_gemm_residual_comp": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _I, _P]),
    "ctclip_attn2_bwd_tok_workspace": (_L, [_I, _I, _I, _I, _I]),
    "ctclip_set_step_state": (_I, [_P]),
    "ctclip_advance_step_state": (_I, [_P, _P]),
    "ctclip_attn2_bwd_fused_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "ctclip_attn2_bwd_fused_workspace": (_L, [_I, _I, _I, _I, _I]),
    "ctclip_attn2_bwd_fused": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _F, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "ctclip_attn2_bwd_tok": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _F, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _P, _L, _P]),
    "ctclip_attn2_unprep_q": (_I, [_P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _P, _L, _P]),
    "ctclip_gemm_headnorm": (_I, [_P, _P, _L, _I, _L, _L, _L, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P, _P, _F, _I, _P]),
    "ctclip_peg_fwd_comp": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "ctclip_shadow_refresh": (_I, [_P, _I, _L, _P]),
    "ctclip_gemm_geglu_bwd": (_I, [_P, _P, _P, _P, _L, _I, _L, _L, _L, _L, _L, _I, _P]),
    "ctclip_preprocess_volume": (_I, [_P, _I, _I, _I, _I, _D, _D, _D, _D, _D, _D, _P, _I, _I, _I, _D, _D, _D, _F, _P]),
    "ctclip_attn_short_supported": (_I, [_I, _I, _I]),
    "ctclip_attn_short_fwd": (_I, [_P, _L, _P, _L, _P, _P, _P, _L, _I, _I, _I, _F, _P]),
    "ctclip_attn_short_bwd_workspace": (_L, [_I, _I]),
    "ctclip_attn_short_bwd": (_I, [_P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _I, _F, _P, _L, _P]),
    "ctclip_attn2_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "ctclip_attn2_prep": (_I, [_P, _P, _P, _L, _L, _L, _P, _P, _F, _P, _P, _P, _P, _P, _L, _I, _P]),
    "ctclip_attn2_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _F, _P, _L, _P, _I, _I, _I, _P]),
    "ctclip_attn2_bwd_workspace": (_L, [_I, _I, _I, _I, _I]),
    "ctclip_attn2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _F, _P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "ctclip_clip_loss_logits": (_I, [_P, _L, _P, _P, _P, _I, _P, _L, _P]),
    "ctclip_l2norm_bwd_rows": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "ctclip_attn2_bwd_dbias": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _F, _P, _P, _I, _I, _I, _P, _L, _P]),
    "ctclip_attn2_unprep_workspace": (_L, []),
    "ctclip_attn2_unprep": (_I, [_P] * 9 + [_F, _P, _P, _P, _L, _L, _L, _P, _P, _L, _I, _P, _L, _P]),
    "ctclip_dropout": (_I, [_P, _P, _P, _L, _F, _U64, _U32, _I, _P]),
    "ctclip_attn_dropout_mask": (_I, [_P, _I, _I, _I, _F, _U64, _P]),
    "ctclip_geglu_fwd": (_I, [_P, _P, _L, _I, _I, _P]),
    "ctclip_geglu_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "ctclip_gelu_fwd": (_I, [_P, _P, _L, _I, _P]),
    "ctclip_gelu_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "ctclip_leaky_relu_fwd": (_I, [_P, _P, _L, _F, _P]),
    "ctclip_leaky_relu_bwd": (_I, [_P, _P, _P, _L, _F, _P]),
    "ctclip_colsum_workspace": (_L, [_L, _I]),
    "ctclip_colsum": (_I, [_P, _P, _L, _I, _L, _I, _P, _L, _P]),
    "ctclip_patch_embed_param_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ctclip_permute0213": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "ctclip_transpose2d": (_I, [_P, _P, _I, _I, _L, _L, _I, _P]),
    "ctclip_pool_fwd": (_I, [_P, _P, _L, _I, _L, _I, _I, _P]),
    "ctclip_pool_bwd": (_I, [_P, _P, _L, _I, _L, _I, _I, _P]),
    "ctclip_convert_pad": (_I, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _P]),
    "ctclip_cpb_expand": (_I, [_P, _P, _I, _I, _I, _P]),
    "ctclip_cpb_reduce": (_I, [_P, _P, _I, _I, _I, _P]),
    "ctclip_bert_embed_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ctclip_vq_gather": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "ctclip_vq_ema_update": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "ctclip_visual_latent_fwd_workspace": (_L, [_I, _I, _L]),
    "ctclip_visual_latent_fwd": (_I, [_P, _P, _P, _I, _I, _L, _I, _P, _L, _P]),
    "ctclip_visual_latent_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _P]),
    "ctclip_clip_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "ctclip_scale_by_scalar": (_I, [_P, _P, _L, _P]),
    "ctclip_relu_dropout": (_I, [_P, _P, _P, _L, _F, _U64, _U32, _P]),
    "ctclip_bce_logits": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "ctclip_pair_softmax_mse": (_I, [_P, _P, _P, _I, _P]),
    "ctclip_latent_similarity": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ctclip_grad_norm_workspace": (_L, []),
    "ctclip_grad_norm_clip": (_I, [_P, _L, _P, _F, _P, _P, _L, _P]),
    "ctclip_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _P, _P]),
    "ctclip_adam_step_zero_grad": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _P, _P]),
    "ctclip_spin": (_I, [_L, _P]),
}

_lib = None


class CtclipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises ImportError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the CT-CLIP HIP kernels are not built. Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc, targets gfx950). "
            "There is deliberately no PyTorch/CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ctclip_abi_version() != 1:
        raise ImportError("libctclip_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ctclip_last_error().decode("utf-8", "replace")
        raise CtclipError(f"{what} failed with code {rc}: {msg}")
