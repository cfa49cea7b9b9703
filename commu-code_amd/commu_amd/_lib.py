"""ctypes binding of libcommu_hip.so (the C ABI declared in include/commu_hip.h).

This is the stub a maintainer of the reference would add to call the MI355X kernels from
Python; it is also the only way the product package reaches the GPU.  There is NO fallback:
if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcommu_hip.so")

c_p = C.c_void_p
c_i = C.c_int
c_f = C.c_float
c_z = C.c_size_t


class AttnDesc(C.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("rd", c_p), ("r_w_bias", c_p), ("r_r_bias", c_p),
                ("reset", c_p), ("ld_qkv", c_i), ("ld_rd", c_i), ("ld_o", c_i), ("T", c_i), ("M", c_i),
                ("B", c_i), ("H", c_i), ("DH", c_i), ("same_length", c_i), ("sshift", c_i), ("scale", c_f),
                ("drop_p", c_f), ("drop_seed", C.c_uint)]


class AttnBwdDesc(C.Structure):
    _fields_ = [("dout", c_p), ("lse", c_p), ("delta", c_p), ("qu2", c_p), ("qv2", c_p), ("dq_ac", c_p),
                ("dk", c_p), ("dv", c_p), ("dsk", c_p), ("du_part", c_p), ("ld_dqkv", c_i), ("ld_dsk", c_i),
                ("du_rows", c_i), ("dsk_wedge", c_i), ("dsk_tiled", c_i), ("p_scratch", c_p), ("o", c_p), ("pf", c_p)]


class ReduceItem(C.Structure):
    _fields_ = [("dst", c_p), ("src_off", C.c_longlong), ("rg", c_i), ("rt", c_i), ("rp", c_i), ("cg", c_i), ("ct", c_i),
                ("cp", c_i)]


class ColsumSource(C.Structure):
    _fields_ = [("X", c_p), ("ldx", c_i), ("rows", c_i), ("alpha", C.c_float), ("pad_", c_i)]


class ColsumTask(C.Structure):
    _fields_ = [("out", c_p), ("cols", c_i), ("src_begin", c_i), ("src_end", c_i), ("pad_", c_i)]


class TransposeItem(C.Structure):
    _fields_ = [("src", c_p), ("dst", c_p), ("ldi", c_i), ("ldo", c_i), ("rows", c_i), ("cols", c_i)]


class TnProblem(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("lda", c_i), ("ldb", c_i), ("N", c_i), ("K", c_i), ("out_off", C.c_longlong),
                ("colsum_off", C.c_longlong)]


# name -> argtypes (all return int unless listed in _RESTYPE); stream is always the last c_void_p
PROTOTYPES = {
    "commu_gemm_nt_bf16": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, C.c_uint, c_f, c_f, c_p],
    "commu_gemm_nt_signbits_words": [c_i, c_i, c_i, c_i, c_i, c_i],
    "commu_gemm_nt_ln_bf16": [c_p, c_i, c_p, c_p, c_i, c_f, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p],
    "commu_gemm_tn_bf16": [c_p, c_i, c_p, c_i, c_p, c_i, c_z, c_i, c_i, c_i, c_i, c_i, c_p],
    "commu_gemm_nt_bf16_batched": [c_p, c_i, C.c_longlong, c_p, c_i, C.c_longlong, c_p, c_i, C.c_longlong, c_i, c_i,
                                   c_i, c_p, c_i, C.c_longlong, c_i, c_i, c_i, c_i, c_p],
    "commu_gemm_tn_slices": [c_i, c_i, c_i],
    "commu_gemm_tn_bf16_batched": [c_p, c_i, C.c_longlong, c_p, c_i, C.c_longlong, c_p, c_i, c_z, c_i, c_i, c_i, c_i,
                                   c_i, c_i, c_i, c_p],
    "commu_gemm_tn_grouped_slices": [C.POINTER(TnProblem), c_i, c_i],
    "commu_gemm_tn_grouped_slices_budget": [C.POINTER(TnProblem), c_i, c_i, c_i],
    "commu_gemm_tn_bf16_grouped": [C.POINTER(TnProblem), c_i, c_i, c_p, C.c_longlong, c_i, c_p],
    "commu_reduce_slabs2d_f32": [c_p, c_i, C.c_longlong, c_p, c_i, c_i, c_i, c_z, c_i, c_i, c_f, c_p],
    "commu_reduce_slabs_f32": [c_p, c_p, c_z, c_i, c_z, c_i, c_f, c_p],
    "commu_quant_mxfp8": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    "commu_gemm_nt_mxfp8": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, C.c_uint, c_f, c_p],
    "commu_reduce_slabs_group_f32": [C.POINTER(ReduceItem), c_i, c_p, c_i, c_z, c_i, c_f, c_p],
    "commu_reduce_slabs_crop_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_z, c_i, c_f, c_p],
    "commu_embed_fwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, C.c_uint, c_f, c_p],
    "commu_embed_bwd": [c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_f, c_i, C.c_uint, c_f, c_p],
    "commu_embed_bwd_ws_rows": [c_i, c_i],
    "commu_embed_bwd_sorted": [c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_f, c_i, C.c_uint, c_f, c_p],
    "commu_posemb_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, C.c_uint, c_f, c_p],
    "commu_layernorm_fwd": [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_i, C.c_uint, c_f, c_p],
    "commu_layernorm_bwd_nblocks": [c_i],
    "commu_layernorm_bwd": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_i, C.c_uint, c_f, c_p],
    "commu_layernorm_bwd_add": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_i, C.c_uint, c_f, c_p],
    "commu_colsum_slabs": [c_i, c_i, c_i],
    "commu_colsum_bf16": [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_f, c_p],
    "commu_colsum_f32": [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_f, c_p],
    "commu_layernorm_bwd_reduce": [c_p, c_i, c_i, c_p, c_p, c_p, c_p],
    "commu_ce_fwd": [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p],
    "commu_ce_bwd": [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "commu_masked_mean": [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p],
    "commu_loss_grad": [c_p, c_i, c_i, c_p, c_f, c_p, c_p],
    "commu_masked_mean_groups": [c_p, c_p, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "commu_loss_grad_groups": [c_p, c_i, c_i, c_p, c_f, c_i, c_i, c_p, c_p],
    "commu_grad_norm": [c_p, c_z, c_p, c_i, c_p, c_p],
    "commu_adam_step": [c_p, c_p, c_p, c_p, c_p, c_z, c_f, c_f, c_f, c_f, c_i, c_p, c_f, c_p],
    "commu_adam_step_dev": [c_p, c_p, c_p, c_p, c_p, c_z, c_p, c_f, c_f, c_f, c_p, c_f, c_p],
    "commu_adam_bias_corrections": [c_f, c_f, c_i, c_p],
    "commu_set_seed_salt": [c_p, c_p],
    "commu_scale_clip_f32": [c_p, c_z, c_p, c_f, c_p],
    "commu_cast_f32_bf16": [c_p, c_p, c_z, c_p],
    "commu_cast_bf16_f32": [c_p, c_p, c_z, c_p],
    "commu_transpose_bf16": [c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    "commu_transpose_f32_bf16": [c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    "commu_copy_bf16": [c_p, c_p, c_z, c_p],
    "commu_transpose_group_bf16": [C.POINTER(TransposeItem), c_i, c_p],
    "commu_mems_update": [c_p, c_z, c_z, c_z, c_p, c_z, c_z, c_z, c_p, c_z, c_i, c_p],
    "commu_relattn_fwd": [C.POINTER(AttnDesc), c_p, c_p, c_p, c_p, c_p],
    "commu_attn_p_scratch_elems": [c_i, c_i, c_i, c_i],
    "commu_relattn_fwd_save": [C.POINTER(AttnDesc), c_p, c_p, c_p, c_p, c_p, c_p],
    "commu_attn_pf_bytes": [c_i, c_i, c_i, c_i],
    "commu_relattn_bwd": [C.POINTER(AttnDesc), C.POINTER(AttnBwdDesc), c_p],
    "commu_relattn_bwd_q": [C.POINTER(AttnDesc), C.POINTER(AttnBwdDesc), c_p],
    "commu_relattn_bwd_kv": [C.POINTER(AttnDesc), C.POINTER(AttnBwdDesc), c_p],
    "commu_colsum_group_f32": [c_p, c_i, c_p, c_i, c_p],
    "commu_colsum_slab_pass": [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "commu_gelu_fwd": [c_p, c_i, c_p, c_i, c_i, c_i, C.c_uint, c_f, c_p],
    "commu_gelu_bwd": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, C.c_uint, c_f, c_p],
    "commu_token_order": [c_p, c_i, c_i, c_p, c_p, c_p, c_p],
    "commu_attn_bwd_qrows": [c_i],
    "commu_attn_fwd_generation": [c_i],
    "commu_attn_bwd_kv_generation": [c_i],
    "commu_attn_band_slabs": [c_i, c_i],
    "commu_attn_band_pairs": [c_i, c_i, c_i],
    "commu_relattn_bwd_band": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "commu_attn_delta": [c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    "commu_transpose_heads": [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "commu_sample_topk": [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_f, c_i, c_p, c_p, c_i, c_p],
    "commu_sample_topk_topp": [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_f, c_i, c_f, c_p, c_p, c_i, c_p],
    "commu_decode_kv_append": [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "commu_decode_attn": [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p],
    "commu_decode_attn_split": [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i,
                                c_i, c_p, c_p, c_p],
    "commu_decode_advance": [c_p, c_p, c_i, c_i, c_p],
    "commu_decode_tail_supported": [c_i, c_i, c_i, c_i],
    "commu_decode_tail_sync_words": [],
    "commu_decode_tail_trace": [c_p],
    "commu_decode_loop_trace": [c_p],
    "commu_decode_layer_tail": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_f, c_i, c_p, c_i, c_p,
                                c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "commu_decode_head": [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p],
    "commu_decode_tail_pack_bytes": [c_i, c_i],
    "commu_decode_tail_pack": [c_p, c_i, c_i, c_i, c_p, c_p],
    "commu_forcing_state_ints": [],
    "commu_forcing_pre": [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "commu_forcing_post": [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "commu_decode_sample_post_pre": [c_p, c_i, c_i, c_p, c_f, c_i, c_f, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i,
                                     c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p],
    "commu_copy_rows_masked_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_p],
    "commu_pack_batch": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, C.c_longlong, c_p, c_p],
    "commu_gemm_nt_f32": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p],
    "commu_embed_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "commu_posemb_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "commu_layernorm_f32": [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "commu_relattn_f32": [c_p, c_i, c_p, c_p, C.c_longlong, C.c_longlong, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i,
                          c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "commu_decode_kv_append_f32": [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "commu_hip_version": [],
}
_RESTYPE = {"commu_decode_tail_pack_bytes": C.c_longlong, "commu_attn_pf_bytes": C.c_longlong, "commu_hip_version": C.c_char_p, "commu_attn_p_scratch_elems": C.c_longlong,
            "commu_gemm_nt_signbits_words": C.c_longlong, "commu_pack_batch": C.c_longlong}
_NOCHECK = {"commu_layernorm_bwd_nblocks", "commu_colsum_slabs", "commu_embed_bwd_ws_rows", "commu_hip_version", "commu_attn_bwd_qrows", "commu_attn_fwd_generation", "commu_attn_bwd_kv_generation", "commu_colsum_slab_pass", "commu_gemm_tn_grouped_slices", "commu_gemm_tn_grouped_slices_budget", "commu_attn_band_slabs", "commu_attn_band_pairs",
            "commu_forcing_state_ints", "commu_decode_tail_supported", "commu_decode_tail_sync_words", "commu_decode_tail_pack_bytes", "commu_attn_p_scratch_elems", "commu_gemm_nt_signbits_words", "commu_pack_batch", "commu_attn_pf_bytes"}

_lib = None


class CommuHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CommuHipError(
            f"{LIB_PATH} not found: build it with `python commu-code_amd/build.py` "
            "(the product path has no CPU fallback)")
    # PyTorch ships its own libamdhip64; import it first so that this library binds to the SAME
    # HIP runtime instance that owns torch's streams and allocations (loading ours first makes
    # launches fail with hipErrorNoDevice).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_i)
    _lib = lib
    return lib


# optional per-entry-point timing with HIP events on the launching stream (bench.py roofline leg)
_prof = None


def profile_start(names, shape_args=None):
    """shape_args: {entry point: tuple of argument positions} -- calls of that entry point are also recorded per shape
    under the key "name:a x b x c" (bench.py: one roofline row per GEMM shape)."""
    global _prof
    _prof = {"names": set(names), "events": {n: [] for n in names}, "on": True, "shape_args": dict(shape_args or {})}


def profile_enable(on: bool):
    """Pause / resume event recording inside a profile_start .. profile_stop window (the events cost a few per cent of a
    training step when every call is bracketed; sampling some of the steps keeps the timed region honest)."""
    if _prof is not None:
        _prof["on"] = bool(on)


def profile_stop():
    """Returns {name: [milliseconds per call]} (synchronises)."""
    global _prof
    import torch
    torch.cuda.synchronize()
    out = {n: [s.elapsed_time(e) for s, e in evs] for n, evs in _prof["events"].items()}
    _prof = None
    return out


def call(name, *args):
    """Call an entry point and raise on a non-zero status."""
    fn = getattr(load(), name)
    if _prof is not None and _prof["on"] and name in _prof["names"]:
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        _prof["events"][name].append((s, e))
        pos = _prof["shape_args"].get(name)
        if pos is not None:
            _prof["events"].setdefault(name + ":" + "x".join(str(int(args[i])) for i in pos), []).append((s, e))
    else:
        rc = fn(*args)
    if name not in _NOCHECK and rc != 0:
        raise CommuHipError(f"{name} failed with status {rc}")
    return rc
