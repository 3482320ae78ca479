"""ctypes binding of ``libvisitron_hip.so`` (the C ABI in ``include/visitron_hip.h``).

There is NO fallback: if the shared library is missing this module raises, and every
op in ``visitron_amd.ops`` refuses tensors that are not on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvisitron_hip.so")
LIB_PATH = os.environ.get("VT_HIP_LIB", LIB_PATH)   # A/B builds of the kernels (tools/): same ABI, other path

# error codes of include/visitron_hip.h
VT_OK, VT_ERR_BAD_SHAPE, VT_ERR_BAD_ALIGN, VT_ERR_NULL, VT_ERR_UNSUPPORTED, VT_ERR_HIP = 0, -1, -2, -3, -4, -5

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
c_uint64, c_uint32 = ctypes.c_uint64, ctypes.c_uint32
DROP = [c_float, c_uint64, c_uint32]  # (p, step seed, site)


class LayerWeights(ctypes.Structure):  # vt_layer_weights
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_in", "b_in", "w_out", "b_out", "ln2_g", "ln2_b")]


class LayerActs(ctypes.Structure):  # vt_layer_acts
    _fields_ = [(n, c_void_p) for n in (
        "qkv", "ctx", "attn_pre", "attn_out", "mid_pre", "mid", "out_pre", "out", "lse",
        "ln1_mean", "ln1_rstd", "ln2_mean", "ln2_rstd", "keep_bits", "ln1_h", "ln2_h")] + [
        ("ln_residual_mode", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class LayerWeightsLn(ctypes.Structure):  # vt_layer_weights_ln
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "g_qkv", "h_qkv", "w_ao", "cb_ao", "gamma_in", "w_in", "g_in", "h_in", "w_out", "cb_out", "ln1_g")]


class LayerWeightsT(ctypes.Structure):  # vt_layer_weights_t
    _fields_ = [(n, c_void_p) for n in ("wt_qkv", "wt_ao", "wt_in", "wt_out")]


class LayerGrads(ctypes.Structure):  # vt_layer_grads
    _fields_ = [(n, c_void_p) for n in (
        "d_w_qkv", "d_b_qkv", "d_w_ao", "d_b_ao", "d_ln1_g", "d_ln1_b", "d_w_in", "d_b_in", "d_w_out", "d_b_out",
        "d_ln2_g", "d_ln2_b")]


class BwdWorkspace(ctypes.Structure):  # vt_bwd_workspace
    _fields_ = [(n, c_void_p) for n in ("g_pre", "g_pre2", "g_mid", "g_ctx", "g_qkv", "delta", "ln_partial", "dq32",
                                        "g_pre_d", "g_pre2_d")]


class WgradProblem(ctypes.Structure):  # vt_wgrad_problem
    _fields_ = [("dY", c_void_p), ("ldy", c_int64), ("X", c_void_p), ("ldx", c_int64), ("dW", c_void_p),
                ("ldw", c_int64), ("db", c_void_p), ("N", c_int), ("K", c_int), ("accumulate", c_int)]


# name -> (restype, argtypes); must list every symbol declared in include/visitron_hip.h
SIGNATURES = {
    "vt_error_string": (ctypes.c_char_p, [c_int]),
    "vt_abi_version": (c_int, []),
    "vt_center_mask": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "vt_batch_row_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vt_batch_row_lists": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vt_action_head_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_float, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vt_linear_splitk_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "vt_embed_table_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p]),
    "vt_debug_set_gemm_variant": (None, [c_int]),
    "vt_debug_set_gemm_trace": (None, [c_void_p]),
    "vt_debug_set_wgrad_kernel": (None, [c_int]),
    "vt_gemm_tune": (None, [c_int, c_int, c_int, c_int, c_int]),
    "vt_debug_set_attn_bwd_waves": (None, [c_int]),
    "vt_set_weight_prefetch": (c_int, [c_int, c_int]),
    "vt_get_weight_prefetch": (c_int, [c_int]),
    "vt_set_attn_dropout_bits": (c_int, [c_int]),
    "vt_get_attn_dropout_bits": (c_int, []),
    "vt_attn_dropout_effective": (ctypes.c_float, [ctypes.c_float]),
    "vt_gemm_reserve_cus": (None, [c_int]),
    "vt_gemm_set_workspace": (c_int, [c_void_p, c_int64]),
    "vt_gemm_workspace_region_bytes": (c_int64, []),
    "vt_gemm_shared_tile_timeouts": (c_int, [ctypes.POINTER(ctypes.c_uint)]),
    "vt_step_counters": (c_int, [c_void_p, c_void_p]),
    "vt_linear_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_linear_bf16_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                  c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int] + DROP + [c_void_p]),
    "vt_linear_lnres_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int] + DROP + [c_void_p]),
    "vt_apply_dropout_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int] + DROP + [c_void_p]),
    "vt_debug_dropout_mask": (c_int, [c_void_p, c_int64] + DROP + [c_int, c_void_p]),
    "vt_attention_probs_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "vt_attention_bwd_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int]
                              + DROP + [c_void_p, c_void_p]),
    "vt_layernorm_bwd_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_int64] + DROP
                              + [c_void_p]),
    "vt_layernorm_bwd_h_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_int64] + DROP
                                + [c_void_p]),
    "vt_embed_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_uint64, c_void_p]),
    "vt_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                              c_float, c_float, c_float, c_float, c_void_p]),
    "vt_adamw_flat_g16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                  c_float, c_float, c_float, c_float, c_void_p]),
    "vt_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "vt_scale_heads_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "vt_mask_tokens": (c_int, [c_void_p] * 10 + [c_int64, c_int64, c_int64, c_float, c_void_p]),
    "vt_assemble_regions": (c_int, [c_void_p] * 13 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_ce_softmax_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                   c_int, c_float, c_void_p]),
    "vt_ce_double_softmax_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                   c_int, c_float, c_void_p]),
    "vt_lstm_step_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_int, c_int, c_int, c_void_p]),
    "vt_lstm_sequence_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_lstm_sequence_persistent_ws_bytes": (c_int64, [c_int, c_int]),
    "vt_lstm_sequence_persistent_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                                                c_void_p]),
    "vt_skinny_linear_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_lstm_sequence_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_lstm_step_train_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vt_lstm_sequence_train_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_void_p]),
    "vt_lstm_step_bwd_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                     c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "vt_lstm_sequence_bwd_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_softdot_attention_bwd_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_softdot_attention_bwd_split_ws_floats": (c_int64, [c_int, c_int, c_int]),
    "vt_softdot_attention_bwd_split_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                                   c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vt_softdot_attention_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_int, c_void_p]),
    "vt_transpose_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vt_transpose_batch_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vt_dgelu_mul_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vt_encoder_backward_bf16": (c_int, [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerWeightsT),
                                         ctypes.POINTER(LayerActs), ctypes.POINTER(LayerGrads), c_int, c_void_p,
                                         c_void_p, c_int, c_void_p, ctypes.POINTER(BwdWorkspace), c_int, c_int, c_int,
                                         c_int, c_int, c_float, c_int, c_float, c_float, c_uint64, c_int, c_void_p]),
    "vt_encoder_backward_overlap_bf16": (c_int, [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerWeightsT),
                                                 ctypes.POINTER(LayerActs), ctypes.POINTER(LayerGrads), c_int, c_void_p,
                                                 c_void_p, c_int, c_void_p, ctypes.POINTER(BwdWorkspace),
                                                 ctypes.POINTER(BwdWorkspace), c_int, c_int, c_int, c_int, c_int, c_float,
                                                 c_int, c_float, c_float, c_uint64, c_int, c_void_p, c_void_p]),
    "vt_attention_fwd_seq_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int]
                                  + DROP + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "vt_attention_bwd_seq_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                          c_int64, c_void_p, c_int, c_int, c_int, c_int] + DROP
                                  + [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vt_encoder_forward_seq_bf16": (c_int, [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerActs), c_int, c_void_p, c_void_p,
                                            c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_uint64, c_int64,
                                            c_void_p, c_void_p, c_void_p]),
    "vt_encoder_backward_seq_bf16": (c_int, [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerWeightsT),
                                             ctypes.POINTER(LayerActs), ctypes.POINTER(LayerGrads), c_int, c_void_p, c_void_p,
                                             ctypes.POINTER(BwdWorkspace), ctypes.POINTER(BwdWorkspace), c_int, c_int, c_int,
                                             c_int, c_int, c_float, c_int, c_float, c_float, c_uint64, c_int, c_int64,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "vt_attention_fwd_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_int, c_int, c_int, c_int] + DROP + [c_void_p, c_void_p]),
    "vt_layernorm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vt_layernorm_h_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_float, c_void_p]),
    "vt_embed_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p, c_float, c_uint64, c_void_p]),
    "vt_pack_concat_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "vt_wgrad_bf16": (c_int, [ctypes.POINTER(WgradProblem), c_int, c_int, c_void_p]),
    "vt_wgrad_turn_timeouts": (c_int, [ctypes.POINTER(ctypes.c_uint)]),
    "vt_linear_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                              c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vt_bmm_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p,
                           c_int64, c_int64, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vt_softmax_rows_f32": (c_int, [c_void_p, c_int64, c_int64, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_int,
                                    c_void_p]),
    "vt_layernorm_rows": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int,
                                  c_float, c_int, c_int, c_void_p]),
    "vt_embed_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p]),
    "vt_linear_ln_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_float,
                                  c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "vt_ln_apply": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64,
                            c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "vt_ln_stream_init": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int64,
                                  c_int, c_float, c_void_p]),
    "vt_encoder_forward_ln_bf16": (c_int, [ctypes.POINTER(LayerWeightsLn), c_int] + [c_void_p] * 10 + [
        c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int64, c_void_p]),
    "vt_encoder_forward_ln_seq_bf16": (c_int, [ctypes.POINTER(LayerWeightsLn), c_int] + [c_void_p] * 9 + [
        c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "vt_encoder_forward_bf16": (c_int, [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerActs), c_int, c_void_p,
                                        c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_float, c_float, c_uint64, c_void_p]),
}

_lib = None


def load():
    """Load the library once; raise with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "visitron_amd: %s not found. Build it first: `make -C visitron_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vt_error_string(int(rc)).decode()
        raise RuntimeError("visitron_hip %s failed: %s (code %d)" % (what, msg, rc))
