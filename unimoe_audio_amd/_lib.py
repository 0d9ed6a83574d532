"""ctypes binding of libumoe_hip.so (the C-ABI declared in include/umoe.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("UMOE_HIP_LIB") or os.path.join(_CSRC, "libumoe_hip.so")   # override: diagnostic builds only
_lib = None


class UmoeError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "umoe.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-j4", "-s"])
    return _SO


vp, i32, i64, f32, f64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint64


class RouterArgs(C.Structure):
    _fields_ = [("x", vp), ("gate_w", vp), ("norm_w", vp), ("h_out", vp), ("logits_in", vp), ("attn_mask", vp),
                ("S", i32), ("D", i32), ("n_dyn", i32), ("n_real", i32), ("n_fix", i32), ("logits_bf16", i32),
                ("top_p", f32), ("fixed_top_k", i32), ("jitter_eps", f64), ("rms_eps", f32),
                ("logits_out", vp), ("top_k", vp), ("sel", vp), ("expert_mask", vp), ("routing_w", vp),
                ("global_w", vp), ("moe_w", vp), ("norm_only", i32), ("gumbel", vp), ("rand_u", vp), ("round_factor", vp), ("x_noise", vp)]


class Group(C.Structure):
    _fields_ = [("w", vp), ("bias", vp), ("rows", vp), ("row_off", vp), ("count", vp), ("static_count", i32),
                ("a_row_base", i32), ("out_row_base", i32), ("n_blocks", i32), ("k", i32), ("a_col_off", i32), ("reserved", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("groups", vp), ("num_groups", i32), ("max_rows", i32), ("max_n_blocks", i32), ("max_k", i32),
                ("a", vp), ("lda", i32), ("norm_w", vp), ("rms_eps", f32), ("resid", vp), ("out", vp), ("ldo", i32),
                ("n_valid", i32), ("prologue", i32), ("epilogue", i32), ("nt", i32), ("waves", i32), ("ksplit", i32), ("part_stride", C.c_long), ("groups_host", vp), ("cache_policy", i32), ("fused_router", vp), ("rider_pub", vp)]


class TGroup(C.Structure):
    _fields_ = [("w", vp), ("w2", vp), ("bias", vp), ("rows", vp), ("row_off", vp), ("count", vp), ("static_count", i32),
                ("a_row_base", i32), ("out_row_base", i32), ("n", i32), ("k", i32), ("ldw", i32), ("a_col_off", i32), ("k_off", vp), ("k_count", vp), ("out_col_off", i32), ("k_compact_a", i32), ("k_compact_w", i32), ("w_kmajor", i32), ("k_w1", i32)]


class TGemmArgs(C.Structure):
    _fields_ = [("groups", vp), ("num_groups", i32), ("max_rows", i32), ("a", vp), ("lda", i32), ("resid", vp), ("out", vp),
                ("ldo", i32), ("epilogue", i32), ("aux_out", vp), ("ld_aux", i32)]


class TnGroup(C.Structure):
    _fields_ = [("p", vp), ("ldp", i32), ("q", vp), ("ldq", i32), ("out", vp), ("p_col_off", i32), ("q_col_off", i32), ("m", i32), ("n", i32),
                ("k_off", i32), ("k", i32), ("k_off_dev", vp), ("k_count_dev", vp), ("out_row_base", i32), ("out_col_off", i32)]


class TGemmTnArgs(C.Structure):
    _fields_ = [("groups", vp), ("num_groups", i32), ("p", vp), ("ldp", i32), ("q", vp), ("ldq", i32), ("out", vp), ("ldo", i32),
                ("k_split", i32), ("ws", vp), ("part_stride", C.c_long)]


class SwigluBwdArgs(C.Structure):
    _fields_ = [("num_groups", i32), ("w_gate", C.POINTER(vp)), ("w_up", C.POINTER(vp)), ("w_down", C.POINTER(vp)), ("D", i32), ("I", i32),
                ("counts", vp), ("offsets", vp), ("slot_token", vp), ("max_rows", i32), ("slot_rows", i32), ("row_base", i32),
                ("x", vp), ("ldx", i32), ("h", vp), ("ldh", i32), ("gu", vp), ("ldgu", i32), ("dy", vp), ("lddy", i32),
                ("dx_slots", vp), ("lddx", i32), ("dw_gate", C.POINTER(vp)), ("dw_up", C.POINTER(vp)), ("dw_down", C.POINTER(vp)),
                ("ws", vp), ("ws_bytes", C.c_size_t), ("w_down_T", vp), ("w_gateup_T", vp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", vp), ("k_cache", vp), ("v_cache", vp), ("kv_start_host", vp), ("d_out", vp), ("rows", i32), ("T", i32), ("H", i32),
                ("KVH", i32), ("hd", i32), ("Lmax", i32), ("scale", f32), ("dq", vp), ("dk_cache", vp), ("dv_cache", vp), ("ws", vp),
                ("ws_bytes", C.c_size_t), ("out", vp), ("lse", vp)]


class CombineArgs(C.Structure):
    _fields_ = [("y_slots", vp), ("slot_of", vp), ("moe_w", vp), ("y_shared", vp), ("global_w", vp), ("resid", vp),
                ("out", vp), ("S", i32), ("D", i32), ("n_real", i32), ("n_dyn", i32), ("n_fix", i32), ("y_parts", vp), ("n_parts", i32), ("part_stride", C.c_long),
                ("shared_row0", i32), ("norm_w", vp),
                ("norm_out", vp), ("rms_eps", f32), ("expert_mask", vp), ("mask_ld", i32), ("dense_rows", i32), ("ep_xfer", vp)]


class RopeArgs(C.Structure):
    _fields_ = [("qkv", vp), ("cos_tab", vp), ("sin_tab", vp), ("pos3", vp), ("kv_pos", vp), ("n_tok", i32), ("T", i32),
                ("H", i32), ("KVH", i32), ("hd", i32), ("sec0", i32), ("sec1", i32), ("sec2", i32), ("Lmax", i32),
                ("q_out", vp), ("k_cache", vp), ("v_cache", vp)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k_cache", vp), ("v_cache", vp), ("kv_start", vp), ("q_pos0", vp), ("rows", i32),
                ("nq", i32), ("H", i32), ("KVH", i32), ("hd", i32), ("Lmax", i32), ("splits", i32), ("scale", f32),
                ("part_o", vp), ("part_ml", vp), ("out", vp), ("qkv_raw", vp), ("cos_tab", vp), ("sin_tab", vp), ("pos3", vp),
                ("sec0", i32), ("sec1", i32), ("sec2", i32), ("lse_out", vp)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", vp), ("B", i32), ("C", i32), ("V", i32), ("cfg_scale", f32), ("temperature", f32),
                ("top_p", f32), ("eos_mul", f32), ("top_k", i32), ("eos", i32), ("enable_eos", i32),
                ("min_tokens", i32), ("step", vp), ("do_sample", i32), ("seed", u64), ("pred", vp), ("probs_out", vp)]


class EngineCfg(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("kv_heads", i32), ("head_dim", i32),
                ("n_dyn", i32), ("n_real", i32), ("n_fix", i32), ("inter_dyn", i32), ("inter_shared", i32),
                ("codec_channels", i32), ("codec_vocab", i32), ("eos", i32), ("pad", i32), ("bos", i32),
                ("mrope0", i32), ("mrope1", i32), ("mrope2", i32), ("rms_eps", f32), ("top_p", f32),
                ("fixed_top_k", i32), ("jitter_eps", f64), ("rows", i32), ("Lmax", i32), ("Tmax", i32),
                ("attn_splits", i32), ("ep_rank", i32), ("ep_size", i32)]


class LayerWeights(C.Structure):
    _fields_ = [("in_norm", vp), ("qkv_w", vp), ("qkv_b", vp), ("o_w", vp), ("post_norm", vp), ("gate_w", vp),
                ("exp_gu", C.POINTER(vp)), ("exp_dn", C.POINTER(vp)), ("sh_gu", C.POINTER(vp)), ("sh_dn", C.POINTER(vp)),
                ("rm_qkv", vp), ("rm_o", vp), ("rm_exp_gate", C.POINTER(vp)), ("rm_exp_up", C.POINTER(vp)),
                ("rm_exp_down", C.POINTER(vp)), ("rm_sh_gate", C.POINTER(vp)), ("rm_sh_up", C.POINTER(vp)),
                ("rm_sh_down", C.POINTER(vp))]


class DecodeIO(C.Structure):
    _fields_ = [("tokens", vp), ("state", vp), ("cfg_scale", f32), ("temperature", f32), ("top_p", f32),
                ("eos_mul", f32), ("top_k", i32), ("do_sample", i32), ("min_tokens", i32), ("seed", u64)]


EXPORTS = [
    "umoe_last_error", "umoe_abi_version", "umoe_packed_elems", "umoe_pack_weight", "umoe_pack_gate_up",
    "umoe_router_fwd", "umoe_router_dispatch_fwd", "umoe_dispatch_build", "umoe_transpose_slots_compact", "umoe_aux_loss_fwd", "umoe_aux_loss_fwd_ws", "umoe_aux_loss_workspace_floats", "umoe_permute_fwd", "umoe_grouped_gemm", "umoe_grouped_swiglu_fwd", "umoe_shared_swiglu_fwd", "umoe_attn_prefill_fwd",
    "umoe_unpermute_combine_fwd", "umoe_rmsnorm_residual_fwd", "umoe_qkv_mrope_kvappend", "umoe_attn_decode",
    "umoe_codec_embed_sum", "umoe_codec_embed_sum_bwd", "umoe_mul_noise", "umoe_codec_head_cfg_sample", "umoe_delay_step", "umoe_rvq_from_codes",
    "umoe_rvq_nearest", "umoe_codec_ce_fwd", "umoe_codec_ce_bwd", "umoe_engine_create", "umoe_engine_destroy", "umoe_engine_set_layer",
    "umoe_engine_set_globals", "umoe_engine_workspace_bytes", "umoe_engine_prefill", "umoe_engine_decode_step",
    "umoe_engine_capture", "umoe_engine_replay", "umoe_engine_buffer", "umoe_engine_profile_step", "umoe_prefetch", "umoe_tiled_gemm", "umoe_tiled_gemm_tn", "umoe_tiled_gemm_tn_workspace_bytes", "umoe_tiled_gemm_tn_split", "umoe_dispatch_build_aligned", "umoe_transpose_slots", "umoe_swiglu_bwd",
    "umoe_unpermute_combine_bwd", "umoe_permute_bwd", "umoe_router_bwd", "umoe_rmsnorm_residual_bwd", "umoe_aux_loss_bwd", "umoe_attn_softmax_fwd", "umoe_attn_softmax_bwd", "umoe_qkv_mrope_bwd",
    "umoe_swiglu_bwd_workspace_bytes", "umoe_grouped_swiglu_bwd", "umoe_shared_swiglu_bwd",
    "umoe_attn_prefill_bwd_workspace_bytes", "umoe_attn_prefill_bwd",
    "umoe_ep_unique_id", "umoe_ep_comm_create", "umoe_ep_comm_destroy", "umoe_ep_all_to_all",
    "umoe_ep_ipc_export", "umoe_ep_ipc_open", "umoe_ep_ipc_close", "umoe_engine_ep_region", "umoe_engine_ep_connect",
    "umoe_engine_ep_error", "umoe_token_drop", "umoe_router_bwd_drop", "umoe_router_bwd_ex", "umoe_dac_conv1d", "umoe_dac_conv_transpose1d", "umoe_dac_resample", "umoe_vision_rope", "umoe_vision_attn", "umoe_swiglu_pair", "umoe_gelu", "umoe_engine_prefill_pos", "umoe_engine_set_probe", "umoe_engine_info", "umoe_engine_prefill_external",
]

EP_PEER, EP_LOOPBACK, EP_RCCL = 0, 1, 2
MAX_EP = 8


def _mirrors():
    return {"umoe_router_args": RouterArgs, "umoe_group_t": Group, "umoe_gemm_args": GemmArgs, "umoe_tgroup_t": TGroup,
            "umoe_tgemm_args": TGemmArgs, "umoe_tn_group_t": TnGroup, "umoe_tgemm_tn_args": TGemmTnArgs, "umoe_swiglu_bwd_args": SwigluBwdArgs, "umoe_attn_bwd_args": AttnBwdArgs,
            "umoe_combine_args": CombineArgs, "umoe_rope_args": RopeArgs, "umoe_attn_args": AttnArgs, "umoe_sample_args": SampleArgs,
            "umoe_engine_cfg": EngineCfg, "umoe_layer_weights": LayerWeights, "umoe_decode_io": DecodeIO}


STRUCT_MIRRORS = _mirrors()          # C struct name -> ctypes mirror (tests/test_abi_cpu.py compares offsets field by field)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise UmoeError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        L = C.CDLL(_SO)
        L.umoe_last_error.restype = C.c_char_p
        L.umoe_packed_elems.restype = C.c_size_t
        L.umoe_packed_elems.argtypes = [i32, i32]
        L.umoe_engine_workspace_bytes.restype = C.c_size_t
        L.umoe_engine_workspace_bytes.argtypes = [vp]
        L.umoe_engine_buffer.restype = vp
        L.umoe_engine_buffer.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t)]
        L.umoe_engine_destroy.restype = None
        L.umoe_engine_destroy.argtypes = [vp]
        L.umoe_pack_weight.argtypes = [vp, i32, i32, vp, vp]
        L.umoe_pack_gate_up.argtypes = [vp, vp, i32, i32, vp, vp]
        L.umoe_router_fwd.argtypes = [C.POINTER(RouterArgs), vp]
        L.umoe_dispatch_build.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp]
        L.umoe_aux_loss_fwd.argtypes = [vp, i32, vp, vp, i32, i32, i32, vp, vp]
        L.umoe_aux_loss_fwd_ws.argtypes = [vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]
        L.umoe_aux_loss_workspace_floats.argtypes = []
        L.umoe_aux_loss_workspace_floats.restype = C.c_size_t
        L.umoe_prefetch.argtypes = [vp, C.c_size_t, i32, vp]
        L.umoe_transpose_slots.argtypes = [vp, i32, i32, vp, vp, vp, i32, i32, vp, i32, vp]
        L.umoe_swiglu_bwd.argtypes = [vp, i32, vp, i32, i32, vp, i32, vp, i32, vp]
        L.umoe_unpermute_combine_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.umoe_permute_bwd.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp, vp]
        L.umoe_router_bwd.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f64, vp, vp]
        L.umoe_router_bwd_drop.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f64, vp, vp]
        L.umoe_router_bwd_ex.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f64, i32, vp, vp, vp]
        L.umoe_dac_conv1d.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, C.POINTER(i32), vp]
        L.umoe_dac_conv_transpose1d.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, C.POINTER(i32), vp]
        L.umoe_dac_resample.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
        L.umoe_vision_rope.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.umoe_vision_attn.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp]
        L.umoe_swiglu_pair.argtypes = [vp, i32, i32, i32, vp, vp]
        L.umoe_gelu.argtypes = [vp, C.c_long, vp]
        L.umoe_engine_prefill_pos.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        L.umoe_token_drop.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.umoe_rmsnorm_residual_bwd.argtypes = [vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, C.c_size_t, vp]
        L.umoe_dispatch_build_aligned.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.umoe_aux_loss_bwd.argtypes = [vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp]
        L.umoe_attn_softmax_fwd.argtypes = [vp, i32, i32, i32, i32, i32, f32, vp, i32, vp]
        L.umoe_attn_softmax_bwd.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp]
        L.umoe_qkv_mrope_bwd.argtypes = [C.POINTER(RopeArgs), vp, vp, vp, vp, vp]
        L.umoe_tiled_gemm_tn.argtypes = [C.POINTER(TGemmTnArgs), vp]
        L.umoe_tiled_gemm_tn_workspace_bytes.argtypes = [C.POINTER(TGemmTnArgs)]
        L.umoe_tiled_gemm_tn_workspace_bytes.restype = C.c_size_t
        L.umoe_tiled_gemm_tn_split.argtypes = [C.POINTER(TGemmTnArgs)]
        L.umoe_swiglu_bwd_workspace_bytes.argtypes = [C.POINTER(SwigluBwdArgs)]
        L.umoe_swiglu_bwd_workspace_bytes.restype = C.c_size_t
        L.umoe_grouped_swiglu_bwd.argtypes = [C.POINTER(SwigluBwdArgs), vp]
        L.umoe_shared_swiglu_bwd.argtypes = [C.POINTER(SwigluBwdArgs), vp]
        L.umoe_attn_prefill_bwd_workspace_bytes.argtypes = [C.POINTER(AttnBwdArgs)]
        L.umoe_attn_prefill_bwd_workspace_bytes.restype = C.c_size_t
        L.umoe_attn_prefill_bwd.argtypes = [C.POINTER(AttnBwdArgs), vp]
        L.umoe_ep_unique_id.argtypes = [vp]
        L.umoe_ep_comm_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
        L.umoe_ep_comm_destroy.argtypes = [vp]
        L.umoe_ep_all_to_all.argtypes = [vp, vp, vp, C.c_size_t, i32, vp]
        L.umoe_ep_ipc_export.argtypes = [vp, vp]
        L.umoe_ep_ipc_open.argtypes = [vp, C.POINTER(vp)]
        L.umoe_ep_ipc_close.argtypes = [vp]
        L.umoe_engine_ep_region.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.umoe_engine_ep_connect.argtypes = [vp, C.POINTER(vp), vp, i32]
        L.umoe_engine_ep_error.argtypes = [vp, vp, C.POINTER(i32)]
        L.umoe_engine_set_probe.argtypes = [vp, vp, vp, vp, vp]
        L.umoe_engine_info.argtypes = [vp, C.c_char_p]
        L.umoe_engine_prefill_external.argtypes = [vp, vp, i32, vp, vp, vp]
        L.umoe_router_dispatch_fwd.argtypes = [C.POINTER(RouterArgs), vp, vp, vp, vp, vp]
        L.umoe_permute_fwd.argtypes = [vp, i32, vp, vp, i32, vp, vp]
        L.umoe_grouped_gemm.argtypes = [C.POINTER(GemmArgs), vp]
        L.umoe_unpermute_combine_fwd.argtypes = [C.POINTER(CombineArgs), vp]
        L.umoe_rmsnorm_residual_fwd.argtypes = [vp, vp, vp, f32, i32, i32, vp, vp, vp]
        L.umoe_qkv_mrope_kvappend.argtypes = [C.POINTER(RopeArgs), vp]
        L.umoe_attn_decode.argtypes = [C.POINTER(AttnArgs), vp]
        L.umoe_attn_prefill_fwd.argtypes = [C.POINTER(AttnArgs), vp]
        L.umoe_codec_embed_sum.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        L.umoe_codec_embed_sum_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        L.umoe_mul_noise.argtypes = [vp, vp, C.c_long, vp, vp]
        L.umoe_codec_head_cfg_sample.argtypes = [C.POINTER(SampleArgs), vp]
        L.umoe_delay_step.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L.umoe_rvq_from_codes.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.umoe_rvq_nearest.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.umoe_codec_ce_fwd.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        L.umoe_codec_ce_bwd.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp]
        L.umoe_engine_create.argtypes = [C.POINTER(EngineCfg), C.POINTER(vp)]
        L.umoe_engine_set_layer.argtypes = [vp, i32, C.POINTER(LayerWeights)]
        L.umoe_engine_set_globals.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
        L.umoe_engine_prefill.argtypes = [vp, vp, vp, i32, vp]
        L.umoe_engine_decode_step.argtypes = [vp, C.POINTER(DecodeIO), vp]
        L.umoe_engine_capture.argtypes = [vp, C.POINTER(DecodeIO), vp]
        L.umoe_engine_replay.argtypes = [vp, vp]
        L.umoe_engine_profile_step.argtypes = [vp, C.POINTER(DecodeIO), vp, C.POINTER(f32), C.POINTER(i32), i32]
        # the mirrors above against the structs the library was BUILT with: a stale libumoe_hip.so (or a stale mirror) must fail here,
        # not read past a shorter struct
        if not hasattr(L, "umoe_struct_size"):
            raise UmoeError(f"{_SO} predates this package (no umoe_struct_size): rebuild it (`make -C unimoe_audio_amd/csrc`)")
        L.umoe_struct_size.restype = C.c_size_t
        L.umoe_struct_size.argtypes = [C.c_char_p]
        for cname, cls in STRUCT_MIRRORS.items():
            built = int(L.umoe_struct_size(cname.encode()))
            if built != C.sizeof(cls):
                raise UmoeError(f"{_SO}: sizeof({cname}) is {built} in the library and {C.sizeof(cls)} in unimoe_audio_amd/_lib.py: "
                                "the library and the package are of different versions -- rebuild (`make -C unimoe_audio_amd/csrc`)")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().umoe_last_error()
        raise UmoeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
