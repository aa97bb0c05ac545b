"""ctypes loader for libquest_hip.so (the C ABI declared in include/quest_hip.h).

There is no fallback: if the library is missing or fails to load, importing raises.  Build it
with ``python -m quest_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUEST_HIP_LIB") or os.path.join(_HERE, "libquest_hip.so")  # override: tuning builds

c_u32, c_i32, c_f32, c_vp = ctypes.c_uint32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p


class PagedKV(ctypes.Structure):
    """quest_paged_kv_t (include/quest_hip.h)."""

    _fields_ = [
        ("data", c_vp),
        ("indices", c_vp),
        ("indptr", c_vp),
        ("num_heads", c_u32),
        ("page_size", c_u32),
        ("head_dim", c_u32),
        ("page_budget", c_u32),
        ("last_page_len", c_u32),
        ("last_page_idx", c_i32),
        ("layout", c_u32),
        ("reserved", c_u32),
    ]


class Batch(ctypes.Structure):
    """quest_batch_t (include/quest_hip.h)."""

    _fields_ = [("n_seqs", c_u32), ("kv_table_stride", c_u32), ("meta_table_stride", c_u32), ("reserved", c_u32),
                ("page_budgets", c_vp)]


# name -> (restype, argtypes); must list every function include/quest_hip.h declares
SIGNATURES = {
    "quest_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "quest_build_info": (ctypes.c_char_p, []),
    "quest_pool_slot": (c_u32, [c_u32, c_u32, c_u32, c_u32, ctypes.c_int]),
    "quest_append_kv_cache_decode": (ctypes.c_int, [c_vp, c_vp, PagedKV, PagedKV, c_vp]),
    "quest_append_kv_cache_prefill": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, PagedKV, PagedKV, c_vp]),
    "quest_estimate_attn_score": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, PagedKV, c_vp]),
    "quest_append_estimate": (ctypes.c_int, [c_vp, c_vp, PagedKV, c_vp, c_vp, c_u32, c_u32, PagedKV, c_vp]),
    "quest_append_estimate_strided": (ctypes.c_int, [c_vp, c_vp, PagedKV, c_vp, c_vp, c_u32, c_u32, c_u32, PagedKV, c_vp]),
    "quest_topk_filtering": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_topk_filtering_strided": (ctypes.c_int, [c_vp, c_u32, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_decode_handler_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_u32]),
    "quest_decode_handler_destroy": (None, [c_vp]),
    "quest_decode_begin_forward": (ctypes.c_int, [c_vp, c_u32, c_u32, c_u32, c_u32, c_u32, c_vp]),
    "quest_decode_end_forward": (ctypes.c_int, [c_vp]),
    "quest_decode_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_vp]),
    "quest_decode_forward_shared": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_vp]),
    "quest_decode_forward_fused_topk": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_vp, c_vp,
                                                        c_vp, c_vp]),
    "quest_decode_forward_fused_topk_strided": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_u32, c_vp,
                                                                c_vp, c_vp, c_vp]),
    "quest_step_state_advance": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_append_estimate_dyn": (ctypes.c_int, [c_vp, c_vp, PagedKV, c_vp, c_vp, c_u32, c_u32, c_u32, PagedKV, c_vp,
                                                  c_vp]),
    "quest_decode_forward_fused_topk_dyn": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_u32, c_vp,
                                                            c_vp, c_vp]),
    "quest_append_estimate_tiles_dyn": (ctypes.c_int, [c_vp, c_vp, PagedKV, c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, PagedKV,
                                                        c_vp, c_vp]),
    "quest_decode_forward_fused_topk_tiles_dyn": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_u32, c_u32,
                                                                  c_vp, c_vp, c_vp]),
    "quest_append_kv_cache_decode_dyn": (ctypes.c_int, [c_vp, c_vp, PagedKV, PagedKV, c_vp, c_vp]),
    "quest_decode_append_forward_shared_dyn": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_vp, c_vp, PagedKV, c_u32, c_vp, c_vp,
                                                              c_vp]),
    "quest_decode_append_forward_shared_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_vp, c_vp, PagedKV, c_u32, c_vp,
                                                                  Batch, c_vp, c_vp]),
    "quest_decode_forward_shared_dyn": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_vp, c_vp]),
    "quest_apply_rope_in_place_dyn": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_f32, c_f32, c_vp, c_vp]),
    "quest_step_state_advance_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, Batch, c_vp]),
    "quest_decode_arm_step_advance": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, Batch]),
    "quest_append_estimate_batched": (ctypes.c_int, [c_vp, c_vp, PagedKV, c_vp, c_vp, c_u32, c_u32, c_u32, PagedKV,
                                                      c_vp, Batch, c_vp]),
    "quest_decode_forward_fused_topk_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_u32,
                                                                c_vp, Batch, c_vp, c_vp]),
    "quest_decode_layer_fused_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_vp, c_vp, PagedKV, c_u32, c_u32, c_vp,
                                                         Batch, c_vp, c_u32, c_vp, c_vp]),
    "quest_append_kv_cache_decode_batched": (ctypes.c_int, [c_vp, c_vp, PagedKV, PagedKV, c_vp, Batch, c_vp]),
    "quest_decode_forward_shared_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, Batch, c_vp, c_vp]),
    "quest_apply_rope_in_place_batched": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_f32, c_f32, c_vp, Batch,
                                                          c_vp]),
    "quest_estimate_attn_score_batched": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, PagedKV, c_vp, Batch, c_vp]),
    "quest_topk_filtering_batched": (ctypes.c_int, [c_vp, c_u32, c_u32, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp, Batch,
                                                     c_vp]),
    "quest_decode_forward_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, PagedKV, c_u32, c_vp, c_u32, c_vp, Batch, c_vp, c_vp]),
    "quest_decode_set_batch": (ctypes.c_int, [c_vp, c_u32]),
    "quest_decode_plan_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_u32), ctypes.POINTER(c_u32)]),
    "quest_decode_debug_workspace": (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(ctypes.c_uint64),
                                                    ctypes.POINTER(c_u32)]),
    "quest_decode_last_launch_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_u32)]),
    "quest_decode_set_pages_per_chunk": (ctypes.c_int, [c_vp, c_u32]),
    "quest_decode_set_skip_merge": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "quest_decode_set_selection_out": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "quest_decode_set_front_end": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "quest_apply_rope_in_place": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, c_u32, c_u32, c_u32, c_f32, c_f32, c_vp]),
    "quest_decode_norm_gemv": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_u32, c_u32, c_vp]),
    "quest_decode_gemv_residual": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_vp]),
    "quest_decode_mlp_gate_up": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_u32, c_u32, c_vp]),
    "quest_decode_qkv_rope": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32,
                                              c_u32, c_f32, c_f32, c_vp, c_vp]),
    "quest_decode_norm_gemv_batched": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_decode_gemv_residual_batched": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_decode_mlp_gate_up_batched": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_u32, c_u32, c_u32, c_vp]),
    "quest_decode_qkv_rope_batched": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_u32, c_u32,
                                                      c_u32, c_u32, c_f32, c_f32, c_vp, c_u32, c_vp]),
    "quest_decode_batched_plan": (ctypes.c_int, [c_u32, c_u32, c_u32, ctypes.c_int, c_vp]),
    "quest_rms_norm_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u32, c_u32, c_f32, c_vp]),
    "quest_prefill_with_paged_kv_cache": (ctypes.c_int, [c_vp, c_vp, c_u32, c_u32, PagedKV, c_u32, ctypes.c_int, c_vp]),
}


def verify_build(lib_path: str, src_root: str = None) -> None:
    """Refuse a library that was not built from the sources beside it: the .so is git-ignored and travels prebuilt, so
    nothing else ties the loaded binary to the code the tests claim to validate.  The library carries build.py's hash of
    csrc/*, include/quest_hip.h and the compiler flags; recompute and compare.  `QUEST_HIP_LIB=<path>` (tuning builds
    with extra -D flags) skips the check -- the override is explicit."""
    from .build import library_hash, source_hash

    try:
        have, want = library_hash(lib_path), source_hash(src_root)
    except OSError as exc:  # a deployment that ships the package + prebuilt library without csrc/ or include/
        raise ImportError(
            f"cannot check {lib_path} against its sources ({exc}): the check reads quest_amd/csrc/* and include/quest_hip.h "
            "under the repository root (or under QUEST_SRC_ROOT).  Ship the sources with the package, point QUEST_SRC_ROOT at "
            "them, or name the library explicitly with QUEST_HIP_LIB=<path> (which skips the check).") from exc
    if have != want:
        raise ImportError(
            f"{lib_path} is stale: it was built from sources with hash {have}, the tree has {want}. "
            "Run `python -m quest_amd.build` (or __graft_entry__.build()).")


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m quest_amd.build` "
            "(there is no CPU fallback for the quest_amd operators).")
    if not os.environ.get("QUEST_HIP_LIB"):
        verify_build(LIB_PATH, os.environ.get("QUEST_SRC_ROOT") or None)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code: int, what: str) -> None:
    """Translate a C-ABI status into the exception type the reference's ops raise
    (TORCH_CHECK -> RuntimeError; std::invalid_argument -> ValueError)."""
    if code == 0:
        return
    msg = lib.quest_error_string(code).decode()
    if code == -1:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed with error code {code}: {msg}")
