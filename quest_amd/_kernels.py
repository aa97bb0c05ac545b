"""Drop-in for the reference's PyBind module ``quest._kernels`` (quest/ops/csrc/bsk_ops.cu:4-20).

Same 7 free functions + 1 class, same argument order and meaning (bsk_ops.h:23-117); each one
validates its tensors the way the reference's CHECK_* macros do (pytorch_extension_utils.h:52-66)
and then calls the matching C-ABI entry point of libquest_hip.so on torch's current stream.
PyTorch is only the owner of device memory here; all arithmetic is in the HIP library.

``prefill_with_paged_kv_cache`` is a hand-written kernel like the others since round 5: the MFMA
flash kernel of ``csrc/prefill.hip`` over the page table (no gathered K/V copy, no torch attention).
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import Batch, PagedKV, check, lib

_NHD, _HND = 0, 1

# The reference compiles its argument validation in or out with -DBSK_TORCH_CHECK
# (quest/ops/CMakeLists.txt:40); here QUEST_TORCH_CHECK=0 turns the same checks off at import time
# (eager decode is host-bound: the checks are ~30 % of an op call's Python time).
import os as _os

_CHECKS_ON = _os.environ.get("QUEST_TORCH_CHECK", "1") != "0"


# ---------------------------------------------------------------- validation (CHECK_* macros)

def _check_input(x: torch.Tensor, name: str) -> None:
    if not _CHECKS_ON:
        return
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _check_rows(x: torch.Tensor, name: str) -> None:
    """A 2-D CUDA tensor whose rows are contiguous (a padded score buffer's ``[:, :n]`` view qualifies)."""
    if not _CHECKS_ON:
        return
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) < x.size(1):
        raise RuntimeError(f"{name} must be 2-D with contiguous rows")


def _check_dim(d: int, x: torch.Tensor, name: str) -> None:
    if _CHECKS_ON and x.dim() != d:
        raise RuntimeError(f"{name} must be a {d}D tensor")


def _check_eq(a, b, what: str) -> None:
    if _CHECKS_ON and a != b:
        raise RuntimeError(f"CHECK_EQ({what}) failed. {a} vs {b}")


def _check_ge(a, b, what: str) -> None:
    if _CHECKS_ON and not a >= b:
        raise RuntimeError(f"CHECK_GE({what}) failed. {a} vs {b}")


def _check_half(x: torch.Tensor, op: str) -> None:
    if x.dtype != torch.float16:  # DISPATCH_PYTORCH_DTYPE_TO_CTYPE has only Half
        raise RuntimeError(f"{op} failed to dispatch with dtype {x.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(x: torch.Tensor) -> int:
    """hipStream_t of torch's current stream on x's device (raw handle: ~0.3 us instead of ~3 us through
    torch.cuda.current_stream(), which matters for the host-bound eager path)."""
    if _raw_stream is not None:
        return _raw_stream(x.device.index)
    return torch.cuda.current_stream(x.device).cuda_stream


def _pool_dims(data: torch.Tensor, layout: int):
    """(page_size, num_heads, head_dim) of a 5-D pool layer view."""
    if layout == _HND:
        return data.size(3), data.size(2), data.size(4)
    return data.size(2), data.size(3), data.size(4)


def _paged(data, indices, indptr, last_page_len, last_page_idx, layout, page_budget=0) -> PagedKV:
    page_size, num_heads, head_dim = _pool_dims(data, layout)
    return PagedKV(
        data.data_ptr(),
        indices.data_ptr() if indices is not None else None,
        indptr.data_ptr() if indptr is not None else None,
        num_heads, page_size, head_dim, page_budget, int(last_page_len), int(last_page_idx), int(layout), 0)


# ---------------------------------------------------------------- free functions

def apply_rope_in_place(q, k, past_kv_len: int, rope_scale: float, rope_theta: float) -> None:
    """page.cu:212-252.  q ``[N, Hq, D]``, k ``[N, Hkv, D]`` rotated in place."""
    _check_input(q, "q")
    _check_input(k, "k")
    _check_dim(3, q, "q")
    _check_dim(3, k, "k")
    _check_eq(q.size(0), k.size(0), "q.size(0), k.size(0)")
    _check_eq(q.size(2), k.size(2), "q.size(2), k.size(2)")
    _check_half(q, "apply_rope_in_place")
    check(lib.quest_apply_rope_in_place(q.data_ptr(), k.data_ptr(), q.size(0), int(past_kv_len), q.size(1),
                                        k.size(1), q.size(2), float(rope_scale), float(rope_theta), _stream(q)),
          "apply_rope_in_place")


def rms_norm_forward(input, weight, output, epsilon: float) -> None:
    """rms_norm.cu:183-212.  input/output ``[1, N, C]``, weight ``[C]``."""
    _check_input(input, "input")
    _check_input(weight, "weight")
    _check_input(output, "output")
    _check_eq(input.dim(), 3, "input.dim(), 3")
    _check_half(input, "rms_norm_forward")
    check(lib.quest_rms_norm_forward(input.data_ptr(), weight.data_ptr(), output.data_ptr(),
                                     input.size(0) * input.size(1), input.size(2), float(epsilon), _stream(input)),
          "rms_norm")


def topk_filtering(estimated_value, estimated_indices, d_out, indices_out, buf, page_budget: int) -> None:
    """topk.cu:7-46.  Rows = heads; selects ``page_budget`` largest of each row.  ``estimated_value`` may be a
    ``[:, :n]`` view of a buffer with padded rows (``decode_append_estimate``'s output): the row stride is passed on."""
    _check_rows(estimated_value, "estimated_value")
    _check_input(estimated_indices, "estimated_indices")
    _check_input(d_out, "d_out")
    _check_input(indices_out, "indices_out")
    _check_dim(2, estimated_value, "estimated_value")
    _check_dim(2, estimated_indices, "estimated_indices")
    num_heads, num_pages = estimated_value.size(0), estimated_value.size(1)
    _check_eq(num_pages, estimated_indices.size(1), "num_pages, estimated_indices.size(1)")
    _check_eq(num_heads, estimated_indices.size(0), "num_heads, estimated_indices.size(0)")
    _check_ge(num_pages, page_budget, "num_pages, page_budget")
    _check_eq(estimated_indices.dtype, torch.int32, "estimated_indices.scalar_type(), torch::kInt32")
    _check_eq(indices_out.dtype, torch.int32, "indices_out.scalar_type(), torch::kInt32")
    _check_eq(page_budget, d_out.size(1), "page_budget, d_out.size(1)")
    _check_eq(page_budget, indices_out.size(1), "page_budget, indices_out.size(1)")
    _check_half(estimated_value, "Top-k filtering")
    check(lib.quest_topk_filtering_strided(estimated_value.data_ptr(), estimated_value.stride(0), estimated_indices.data_ptr(),
                                           d_out.data_ptr(), indices_out.data_ptr(),
                                           buf.data_ptr() if buf is not None else None, num_heads, num_pages,
                                           int(page_budget), _stream(estimated_value)),
          "Top-k filtering")


def estimate_attn_score(q, o, metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len: int,
                        metadata_last_page_idx: int, layout: int) -> None:
    """estimate.cu:6-84.  q ``[1, Hq, D]``; o ``[Hq, n_pages-1]`` written in place."""
    _check_input(q, "q")
    _check_input(o, "o")
    _check_input(metadata_data, "metadata_data")
    _check_input(metadata_indices, "metadata_indices")
    _check_dim(3, q, "q")
    _check_dim(2, o, "o")
    _check_dim(5, metadata_data, "metadata_data")
    _check_dim(1, metadata_indices, "metadata_indices")
    _check_eq(q.size(0), 1, "q.size(0), 1")
    _check_eq(metadata_indices.dtype, torch.int32, "metadata_indices.scalar_type(), torch::kInt32")
    _check_half(q, "Estimate_attn_score")
    page_size, num_kv_heads, head_dim = _pool_dims(metadata_data, layout)
    _check_eq(metadata_data.size(4), q.size(2), "metadata_data.size(4), head_dim")
    _check_eq(o.size(0), q.size(1), "o.size(0), num_heads")
    meta = _paged(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx,
                  layout)
    check(lib.quest_estimate_attn_score(q.data_ptr(), o.data_ptr(), q.size(1), o.size(1), meta, _stream(q)),
          "Estimate_attn_score")


def _check_append(k, v, kv_data, kv_indices, kv_indptr, metadata_data, metadata_indices, metadata_indptr, layout):
    for t, n in ((k, "k"), (v, "v"), (kv_data, "kv_data"), (kv_indices, "kv_indices"),
                 (metadata_data, "metadata_data"), (metadata_indices, "metadata_indices")):
        _check_input(t, n)
    _check_dim(1, kv_indices, "kv_indices")
    _check_dim(1, metadata_indices, "metadata_indices")
    _check_dim(3, k, "k")
    _check_dim(3, v, "v")
    _check_dim(5, kv_data, "kv_data")
    _check_dim(5, metadata_data, "metadata_data")
    for t, n in ((kv_indices, "kv_indices"), (metadata_indices, "metadata_indices"), (kv_indptr, "kv_indptr"),
                 (metadata_indptr, "metadata_indptr")):
        _check_eq(t.dtype, torch.int32, f"{n}.scalar_type(), torch::kInt32")
    page_size, num_heads, head_dim = _pool_dims(kv_data, layout)
    _check_eq(num_heads, k.size(1), "kv_data heads, num_heads")
    _check_eq(head_dim, k.size(2), "kv_data.size(4), head_dim")
    _check_eq(v.size(0), k.size(0), "seq_len, v.size(0)")


def append_kv_cache_prefill(k, v, kv_data, kv_indices, kv_indptr, kv_last_page_len: int, kv_last_page_idx: int,
                            metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len: int,
                            metadata_last_page_idx: int, layout: int) -> None:
    """page.cu:101-210.  k, v ``[N>=2, H, D]`` appended; per-page min/max metadata rebuilt."""
    _check_append(k, v, kv_data, kv_indices, kv_indptr, metadata_data, metadata_indices, metadata_indptr, layout)
    _check_ge(k.size(0), 2, "k.size(0), 2")
    _check_half(k, "Append_kv_cache_prefill")
    kv = _paged(kv_data, kv_indices, kv_indptr, kv_last_page_len, kv_last_page_idx, layout)
    meta = _paged(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx,
                  layout)
    check(lib.quest_append_kv_cache_prefill(k.data_ptr(), v.data_ptr(), k.size(0), kv_indices.size(0), kv, meta,
                                            _stream(k)), "Append_kv_cache_prefill")


def append_kv_cache_decode(k, v, kv_data, kv_indices, kv_indptr, kv_last_page_len: int, kv_last_page_idx: int,
                           metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len: int,
                           metadata_last_page_idx: int, layout: int) -> None:
    """page.cu:6-99.  k, v ``[1, H, D]``."""
    _check_append(k, v, kv_data, kv_indices, kv_indptr, metadata_data, metadata_indices, metadata_indptr, layout)
    _check_eq(k.size(0), 1, "k.size(0), 1")
    _check_half(k, "Append_kv_cache_decode")
    kv = _paged(kv_data, kv_indices, kv_indptr, kv_last_page_len, kv_last_page_idx, layout)
    meta = _paged(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx,
                  layout)
    check(lib.quest_append_kv_cache_decode(k.data_ptr(), v.data_ptr(), kv, meta, _stream(k)),
          "Append_kv_cache_decode")


def append_estimate(k, v, kv_data, kv_indices, kv_indptr, kv_last_page_len: int, kv_last_page_idx: int, q, o,
                    metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len: int,
                    metadata_last_page_idx: int, layout: int) -> None:
    """EXTENSION (not in the reference surface): append_kv_cache_decode + estimate_attn_score in one
    launch; results identical to the two separate ops."""
    _check_append(k, v, kv_data, kv_indices, kv_indptr, metadata_data, metadata_indices, metadata_indptr, layout)
    _check_eq(k.size(0), 1, "k.size(0), 1")
    _check_input(q, "q")
    _check_rows(o, "o")  # [Hq, n_out], rows possibly padded (a view of a wider buffer)
    _check_dim(3, q, "q")
    _check_eq(q.size(0), 1, "q.size(0), 1")
    _check_eq(o.size(0), q.size(1), "o.size(0), num_heads")
    _check_half(k, "Append_kv_cache_decode")
    _check_half(q, "Estimate_attn_score")
    kv = _paged(kv_data, kv_indices, kv_indptr, kv_last_page_len, kv_last_page_idx, layout)
    meta = _paged(metadata_data, metadata_indices, metadata_indptr, metadata_last_page_len, metadata_last_page_idx,
                  layout)
    check(lib.quest_append_estimate_strided(k.data_ptr(), v.data_ptr(), kv, q.data_ptr(), o.data_ptr(), q.size(1),
                                            o.size(1), o.stride(0), meta, _stream(k)), "append_estimate")


# ---- decode-token projections of a decoder layer, fused (EXTENSION: csrc/decode_layer.hip)

def _check_vec(x, n: int, name: str) -> None:
    _check_input(x, name)
    _check_eq(x.numel(), n, f"{name}.numel(), {n}")
    _check_half(x, name)


def _check_weight(w, out_dim: int, in_dim: int, name: str) -> None:
    _check_input(w, name)
    _check_dim(2, w, name)
    _check_eq(tuple(w.shape), (out_dim, in_dim), f"{name}.shape, (out, in)")
    _check_half(w, name)


def decode_norm_gemv(x, gamma, eps: float, w, out) -> None:
    """``out = w @ rmsnorm(x; gamma, eps)`` (``gamma`` None: ``out = w @ x``); x ``[in]``, w ``[out, in]``, out ``[out]``."""
    out_dim, in_dim = w.shape
    _check_vec(x, in_dim, "x")
    _check_vec(out, out_dim, "out")
    _check_weight(w, out_dim, in_dim, "w")
    if gamma is not None:
        _check_vec(gamma, in_dim, "gamma")
    check(lib.quest_decode_norm_gemv(x.data_ptr(), gamma.data_ptr() if gamma is not None else None, float(eps),
                                     w.data_ptr(), out.data_ptr(), in_dim, out_dim, _stream(x)), "decode_norm_gemv")


def decode_gemv_residual(x, w, h) -> None:
    """``h += w @ x`` in place; x ``[in]``, w ``[out, in]``, h ``[out]``."""
    out_dim, in_dim = w.shape
    _check_vec(x, in_dim, "x")
    _check_vec(h, out_dim, "h")
    _check_weight(w, out_dim, in_dim, "w")
    check(lib.quest_decode_gemv_residual(x.data_ptr(), w.data_ptr(), h.data_ptr(), in_dim, out_dim, _stream(x)),
          "decode_gemv_residual")


def decode_mlp_gate_up(h, gamma, eps: float, w_gate, w_up, act) -> None:
    """``act = silu(w_gate @ n) * (w_up @ n)`` with ``n = rmsnorm(h; gamma, eps)``."""
    inter, hidden = w_gate.shape
    _check_vec(h, hidden, "h")
    _check_vec(gamma, hidden, "gamma")
    _check_vec(act, inter, "act")
    _check_weight(w_gate, inter, hidden, "w_gate")
    _check_weight(w_up, inter, hidden, "w_up")
    check(lib.quest_decode_mlp_gate_up(h.data_ptr(), gamma.data_ptr(), float(eps), w_gate.data_ptr(), w_up.data_ptr(),
                                       act.data_ptr(), hidden, inter, _stream(h)), "decode_mlp_gate_up")


def decode_qkv_rope(h, gamma, eps: float, wq, wk, wv, q, k, v, head_dim: int, rope_scale: float, rope_theta: float,
                    state) -> None:
    """q ``[1, Hq, D]``, k / v ``[1, Hkv, D]`` = projections of ``rmsnorm(h; gamma, eps)`` with RoPE on q and k at the
    position ``state.seq_len - 1`` (the device-resident step state, after ``step_state_advance``)."""
    hidden = wq.size(1)
    _check_vec(h, hidden, "h")
    _check_vec(gamma, hidden, "gamma")
    _check_weight(wq, wq.size(0), hidden, "wq")
    _check_weight(wk, wk.size(0), hidden, "wk")
    _check_weight(wv, wk.size(0), hidden, "wv")
    _check_vec(q, wq.size(0), "q")
    _check_vec(k, wk.size(0), "k")
    _check_vec(v, wk.size(0), "v")
    _check_input(state, "state")
    _check_eq(wq.size(0) % head_dim, 0, "wq.size(0) % head_dim, 0")
    check(lib.quest_decode_qkv_rope(h.data_ptr(), gamma.data_ptr(), float(eps), wq.data_ptr(), wk.data_ptr(),
                                    wv.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), hidden,
                                    wq.size(0) // head_dim, wk.size(0) // head_dim, int(head_dim), float(rope_scale),
                                    float(rope_theta), state.data_ptr(), _stream(h)), "decode_qkv_rope")


# ---- the fused decode-layer launches for n <= 16 tokens (one per sequence of a batch)

MAX_BATCHED_TOKENS = 16


def _check_mat(x, n: int, cols: int, name: str) -> None:
    _check_input(x, name)
    _check_eq(x.numel(), n * cols, f"{name}.numel(), {n} * {cols}")
    _check_half(x, name)


def _check_tokens(n: int) -> None:
    if not 1 <= n <= MAX_BATCHED_TOKENS:
        raise RuntimeError(f"batched decode-layer launches take 1..{MAX_BATCHED_TOKENS} tokens, got {n}")


def decode_norm_gemv_batched(x, gamma, eps: float, w, out) -> None:
    """``out[i] = w @ rmsnorm(x[i]; gamma, eps)`` (``gamma`` None: no norm); x ``[n, in]``, out ``[n, out]``."""
    out_dim, in_dim = w.shape
    n = x.numel() // in_dim
    _check_tokens(n)
    _check_mat(x, n, in_dim, "x")
    _check_mat(out, n, out_dim, "out")
    _check_weight(w, out_dim, in_dim, "w")
    if gamma is not None:
        _check_vec(gamma, in_dim, "gamma")
    check(lib.quest_decode_norm_gemv_batched(x.data_ptr(), gamma.data_ptr() if gamma is not None else None, float(eps),
                                             w.data_ptr(), out.data_ptr(), in_dim, out_dim, n, _stream(x)),
          "decode_norm_gemv_batched")


def decode_gemv_residual_batched(x, w, h) -> None:
    """``h[i] += w @ x[i]`` in place; x ``[n, in]``, h ``[n, out]``."""
    out_dim, in_dim = w.shape
    n = x.numel() // in_dim
    _check_tokens(n)
    _check_mat(x, n, in_dim, "x")
    _check_mat(h, n, out_dim, "h")
    _check_weight(w, out_dim, in_dim, "w")
    check(lib.quest_decode_gemv_residual_batched(x.data_ptr(), w.data_ptr(), h.data_ptr(), in_dim, out_dim, n, _stream(x)),
          "decode_gemv_residual_batched")


def decode_mlp_gate_up_batched(h, gamma, eps: float, w_gate, w_up, act) -> None:
    """``act[i] = silu(w_gate @ n_i) * (w_up @ n_i)``, ``n_i = rmsnorm(h[i]; gamma, eps)``; h ``[n, hidden]``."""
    inter, hidden = w_gate.shape
    n = h.numel() // hidden
    _check_tokens(n)
    _check_mat(h, n, hidden, "h")
    _check_vec(gamma, hidden, "gamma")
    _check_mat(act, n, inter, "act")
    _check_weight(w_gate, inter, hidden, "w_gate")
    _check_weight(w_up, inter, hidden, "w_up")
    check(lib.quest_decode_mlp_gate_up_batched(h.data_ptr(), gamma.data_ptr(), float(eps), w_gate.data_ptr(),
                                               w_up.data_ptr(), act.data_ptr(), hidden, inter, n, _stream(h)),
          "decode_mlp_gate_up_batched")


def decode_qkv_rope_batched(h, gamma, eps: float, wq, wk, wv, q, k, v, head_dim: int, rope_scale: float,
                            rope_theta: float, states) -> None:
    """q ``[n, Hq, D]``, k / v ``[n, Hkv, D]`` = projections of ``rmsnorm(h[i]; gamma, eps)``, RoPE on q and k at
    ``states[i].seq_len - 1`` (``states``: the batched step state ``[n, 8]`` int32)."""
    hidden = wq.size(1)
    n = h.numel() // hidden
    _check_tokens(n)
    _check_mat(h, n, hidden, "h")
    _check_vec(gamma, hidden, "gamma")
    _check_weight(wq, wq.size(0), hidden, "wq")
    _check_weight(wk, wk.size(0), hidden, "wk")
    _check_weight(wv, wk.size(0), hidden, "wv")
    _check_mat(q, n, wq.size(0), "q")
    _check_mat(k, n, wk.size(0), "k")
    _check_mat(v, n, wk.size(0), "v")
    _check_input(states, "states")
    _check_eq(states.numel(), n * STEP_STATE_INTS, "states.numel(), n * 8")
    _check_eq(wq.size(0) % head_dim, 0, "wq.size(0) % head_dim, 0")
    check(lib.quest_decode_qkv_rope_batched(h.data_ptr(), gamma.data_ptr(), float(eps), wq.data_ptr(), wk.data_ptr(),
                                            wv.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), hidden,
                                            wq.size(0) // head_dim, wk.size(0) // head_dim, int(head_dim),
                                            float(rope_scale), float(rope_theta), states.data_ptr(), n, _stream(h)),
          "decode_qkv_rope_batched")


# ---- state-driven (graph-replayable) forms: EXTENSIONS, see include/quest_hip.h quest_step_state_t

STEP_STATE_INTS = 8  # int32 fields of quest_step_state_t


def step_state_advance(state, kv_table, meta_table, page_size: int) -> None:
    """Device-side prepare_metadata(1): ``state`` int32[8] (quest_step_state_t), tables int32 up to capacity."""
    for t, n in ((state, "state"), (kv_table, "kv_table"), (meta_table, "meta_table")):
        _check_input(t, n)
        _check_eq(t.dtype, torch.int32, f"{n}.scalar_type(), torch::kInt32")
    _check_eq(state.numel(), STEP_STATE_INTS, "state.numel(), 8")
    check(lib.quest_step_state_advance(state.data_ptr(), kv_table.data_ptr(), meta_table.data_ptr(), int(page_size),
                                       kv_table.numel(), meta_table.numel(), _stream(state)), "step_state_advance")


def tile_max_offset(max_n: int) -> int:
    """Column offset of the tile maxima in a score row of the tiles launches: the scores padded to 8 columns."""
    return (int(max_n) + 7) // 8 * 8


def tiles_row_stride(max_n: int) -> int:
    """Smallest row stride (fp16 columns, a multiple of 8) of a score scratch that also holds the tile maxima."""
    return (tile_max_offset(max_n) + ((int(max_n) + 7) // 8 + 3) // 4 * 4 + 7) // 8 * 8


def append_estimate_dyn(k, v, kv_data, kv_table, q, o, metadata_data, meta_table, state, max_n_out: int,
                        layout: int, tiles: bool = False) -> bool:
    """append_estimate whose lengths / last-page ids / n_out come from ``state``; ``o`` is ``[Hq, stride]``.
    ``tiles``: also store the rows' tile maxima (per 8 pages the largest score, as a 16-bit key) at column
    ``tile_max_offset(max_n_out)`` -- ``o`` must be ``[Hq, >= tiles_row_stride(max_n_out)]``; returns False (nothing
    launched) for QUEST_EUNSUPPORTED: no estimate tile of this shape is a multiple of 8 pages wide (checked before the
    launch, ``csrc/estimate.hip`` ``launch_estimate``), or the group size / head_dim has no instantiation at all -- the
    caller's whole-row launch (``tiles=False``) then raises for the latter."""
    for t, n in ((k, "k"), (v, "v"), (kv_data, "kv_data"), (kv_table, "kv_table"), (q, "q"), (o, "o"),
                 (metadata_data, "metadata_data"), (meta_table, "meta_table"), (state, "state")):
        _check_input(t, n)
    _check_dim(3, k, "k")
    _check_dim(3, q, "q")
    _check_dim(2, o, "o")
    _check_eq(k.size(0), 1, "k.size(0), 1")
    _check_eq(o.size(0), q.size(1), "o.size(0), num_heads")
    _check_ge(o.size(1), max_n_out, "o.size(1), max_n_out")
    _check_half(k, "Append_kv_cache_decode")
    _check_half(q, "Estimate_attn_score")
    kv = _paged(kv_data, kv_table, None, 1, 0, layout)
    meta = _paged(metadata_data, meta_table, None, 1, 0, layout)
    if tiles:
        _check_ge(o.size(1), tiles_row_stride(max_n_out), "o.size(1), tiles_row_stride(max_n_out)")
        code = lib.quest_append_estimate_tiles_dyn(k.data_ptr(), v.data_ptr(), kv, q.data_ptr(), o.data_ptr(), q.size(1),
                                                   o.size(1), int(max_n_out), tile_max_offset(max_n_out), meta,
                                                   state.data_ptr(), _stream(k))
        if code == -2:
            return False
        check(code, "append_estimate_dyn")
        return True
    check(lib.quest_append_estimate_dyn(k.data_ptr(), v.data_ptr(), kv, q.data_ptr(), o.data_ptr(), q.size(1),
                                        o.size(1), int(max_n_out), meta, state.data_ptr(), _stream(k)),
          "append_estimate_dyn")
    return True


def append_kv_cache_decode_dyn(k, v, kv_data, kv_table, metadata_data, meta_table, state, layout: int) -> None:
    """append_kv_cache_decode with lengths / last-page ids from ``state``."""
    for t, n in ((k, "k"), (v, "v"), (kv_data, "kv_data"), (kv_table, "kv_table"), (metadata_data, "metadata_data"),
                 (meta_table, "meta_table"), (state, "state")):
        _check_input(t, n)
    _check_eq(k.size(0), 1, "k.size(0), 1")
    _check_half(k, "Append_kv_cache_decode")
    kv = _paged(kv_data, kv_table, None, 1, 0, layout)
    meta = _paged(metadata_data, meta_table, None, 1, 0, layout)
    check(lib.quest_append_kv_cache_decode_dyn(k.data_ptr(), v.data_ptr(), kv, meta, state.data_ptr(), _stream(k)),
          "Append_kv_cache_decode")


def apply_rope_in_place_dyn(q, k, rope_scale: float, rope_theta: float, state) -> None:
    _check_input(q, "q")
    _check_input(k, "k")
    _check_input(state, "state")
    _check_eq(q.size(0), 1, "q.size(0), 1")
    _check_half(q, "apply_rope_in_place")
    check(lib.quest_apply_rope_in_place_dyn(q.data_ptr(), k.data_ptr(), q.size(1), k.size(1), q.size(2),
                                            float(rope_scale), float(rope_theta), state.data_ptr(), _stream(q)),
          "apply_rope_in_place_dyn")


# ---------------------------------------------------------------------------------------------------------
# Batched state-driven step (EXTENSION; SURVEY 8f-3, BASELINE config 5): n sequences per launch.  The pools
# are shared, page tables are rows of ``[n, stride]`` int32 matrices, ``state`` is ``[n, 8]`` int32,
# q/k/v/o are ``[n, heads, dim]``, scores ``[n, Hq, stride]``.
def _batch(state, kv_tables, meta_tables, budgets=None) -> Batch:
    """quest_batch_t.  ``budgets``: optional int32 ``[n]`` device tensor of per-sequence page budgets (pages a sequence
    attends INCLUDING its current one); None = the budget the handler was planned with, for every sequence."""
    _check_dim(2, state, "state")
    n = state.size(0)
    if budgets is not None:
        _check_input(budgets, "page_budgets")
        _check_dim(1, budgets, "page_budgets")
        _check_eq(budgets.size(0), n, "page_budgets.size(0), n_seqs")
        if budgets.dtype != torch.int32:
            raise RuntimeError("page_budgets must be an int32 tensor")
    for t, name in ((kv_tables, "kv_tables"), (meta_tables, "meta_tables")):
        if t is not None:
            # rows of `capacity` entries at a stride >= capacity: BatchedInferenceController pads the stride to a multiple
            # of 4 entries so that every sequence's table is 16-byte aligned (vector loads of page ids)
            _check_rows(t, name)
            _check_eq(t.size(0), n, f"{name}.size(0), n_seqs")
            if t.dtype != torch.int32:
                raise RuntimeError(f"{name} must be an int32 tensor")
    return Batch(n, 0 if kv_tables is None else kv_tables.stride(0), 0 if meta_tables is None else meta_tables.stride(0), 0,
                 None if budgets is None else budgets.data_ptr())


def step_state_advance_batched(state, kv_tables, meta_tables, page_size: int) -> None:
    _check_input(state, "state")
    b = _batch(state, kv_tables, meta_tables)
    check(lib.quest_step_state_advance_batched(state.data_ptr(), kv_tables.data_ptr(), meta_tables.data_ptr(),
                                               int(page_size), kv_tables.size(1), meta_tables.size(1), b,
                                               _stream(state)), "step_state_advance_batched")


def append_estimate_batched(k, v, kv_data, kv_tables, q, o, metadata_data, meta_tables, state, max_n_out: int,
                            layout: int) -> None:
    for t, n in ((k, "k"), (v, "v"), (kv_data, "kv_data"), (q, "q"), (o, "o"), (metadata_data, "metadata_data"),
                 (state, "state")):
        _check_input(t, n)
    b = _batch(state, kv_tables, meta_tables)
    _check_dim(3, k, "k")
    _check_dim(3, q, "q")
    _check_dim(3, o, "o")
    _check_eq(k.size(0), b.n_seqs, "k.size(0), n_seqs")
    _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
    _check_eq(o.size(0), b.n_seqs, "o.size(0), n_seqs")
    _check_eq(o.size(1), q.size(1), "o.size(1), num_heads")
    _check_ge(o.size(2), max_n_out, "o.size(2), max_n_out")
    _check_half(k, "Append_kv_cache_decode")
    _check_half(q, "Estimate_attn_score")
    kv = _paged(kv_data, kv_tables, None, 1, 0, layout)
    meta = _paged(metadata_data, meta_tables, None, 1, 0, layout)
    check(lib.quest_append_estimate_batched(k.data_ptr(), v.data_ptr(), kv, q.data_ptr(), o.data_ptr(), q.size(1),
                                            o.size(2), int(max_n_out), meta, state.data_ptr(), b, _stream(k)),
          "append_estimate_batched")


def estimate_attn_score_batched(q, o, metadata_data, meta_tables, state, max_n_out: int, layout: int) -> None:
    """estimate_attn_score (estimate.cu:6-84) for ``n`` sequences in one launch: q ``[n, Hq, D]``, o ``[n, Hq, stride]``;
    row i is scored over ``state[i].n_pages - 1`` pages."""
    for t, n in ((q, "q"), (o, "o"), (metadata_data, "metadata_data"), (state, "state")):
        _check_input(t, n)
    b = _batch(state, None, meta_tables)
    _check_dim(3, q, "q")
    _check_dim(3, o, "o")
    _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
    _check_eq(o.size(0), b.n_seqs, "o.size(0), n_seqs")
    _check_eq(o.size(1), q.size(1), "o.size(1), num_heads")
    _check_ge(o.size(2), max_n_out, "o.size(2), max_n_out")
    _check_half(q, "Estimate_attn_score")
    meta = _paged(metadata_data, meta_tables, None, 1, 0, layout)
    check(lib.quest_estimate_attn_score_batched(q.data_ptr(), o.data_ptr(), q.size(1), o.size(2), int(max_n_out), meta,
                                                state.data_ptr(), b, _stream(q)), "Estimate_attn_score")


def topk_filtering_batched(scores, kv_tables, d_out, indices_out, state, max_num_pages: int, page_budget: int,
                           budgets=None) -> None:
    """topk_filtering (topk.cu:7-46) for ``n`` sequences in one launch: scores ``[n, H, stride]``, page ids = each
    sequence's own page table ``kv_tables[i]``, outputs ``[n, H, k_max]`` of which row (i, h) gets its first
    ``min(budget_i - 1, n_pages_i - 1)`` entries written (``budget_i`` from ``budgets`` or ``page_budget``)."""
    for t, n in ((scores, "scores"), (d_out, "d_out"), (indices_out, "indices_out"), (state, "state")):
        _check_input(t, n)
    b = _batch(state, kv_tables, None, budgets)
    _check_dim(3, scores, "scores")
    _check_dim(3, d_out, "d_out")
    _check_dim(3, indices_out, "indices_out")
    _check_eq(scores.size(0), b.n_seqs, "scores.size(0), n_seqs")
    _check_eq(d_out.size(0), b.n_seqs, "d_out.size(0), n_seqs")
    _check_eq(d_out.size(1), scores.size(1), "d_out.size(1), num_heads")
    _check_eq(tuple(indices_out.shape), tuple(d_out.shape), "indices_out.shape, d_out.shape")
    _check_eq(indices_out.dtype, torch.int32, "indices_out.scalar_type(), torch::kInt32")
    _check_ge(scores.size(2), max_num_pages, "scores.size(2), max_num_pages")
    _check_half(scores, "Top-k filtering")
    check(lib.quest_topk_filtering_batched(scores.data_ptr(), scores.size(2), int(max_num_pages), kv_tables.data_ptr(),
                                           d_out.data_ptr(), indices_out.data_ptr(), d_out.size(2), scores.size(1),
                                           int(page_budget), state.data_ptr(), b, _stream(scores)), "Top-k filtering")


def append_kv_cache_decode_batched(k, v, kv_data, kv_tables, metadata_data, meta_tables, state, layout: int) -> None:
    for t, n in ((k, "k"), (v, "v"), (kv_data, "kv_data"), (metadata_data, "metadata_data"), (state, "state")):
        _check_input(t, n)
    b = _batch(state, kv_tables, meta_tables)
    _check_eq(k.size(0), b.n_seqs, "k.size(0), n_seqs")
    _check_half(k, "Append_kv_cache_decode")
    kv = _paged(kv_data, kv_tables, None, 1, 0, layout)
    meta = _paged(metadata_data, meta_tables, None, 1, 0, layout)
    check(lib.quest_append_kv_cache_decode_batched(k.data_ptr(), v.data_ptr(), kv, meta, state.data_ptr(), b,
                                                   _stream(k)), "Append_kv_cache_decode")


def apply_rope_in_place_batched(q, k, rope_scale: float, rope_theta: float, state) -> None:
    _check_input(q, "q")
    _check_input(k, "k")
    _check_input(state, "state")
    b = _batch(state, None, None)
    _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
    _check_eq(k.size(0), b.n_seqs, "k.size(0), n_seqs")
    _check_half(q, "apply_rope_in_place")
    check(lib.quest_apply_rope_in_place_batched(q.data_ptr(), k.data_ptr(), q.size(1), k.size(1), q.size(2),
                                                float(rope_scale), float(rope_theta), state.data_ptr(), b,
                                                _stream(q)), "apply_rope_in_place_batched")


def prefill_with_paged_kv_cache(q, kv_data, kv_indices, kv_last_page_len: int, causal: bool, layout: int,
                                allow_fp16_qk_reduction: bool, rope_scale: float, rope_theta: float):
    """batch_prefill.cu:27-117 -> BatchPrefillWithPagedKVCache (prefill.cuh:1008-1119): attention of the ``n`` new rows
    of ``q`` over the sequence's pages, the new tokens included (row i sees keys 0 .. kv_len - n + i when ``causal``).
    One launch of the MFMA flash kernel of csrc/prefill.hip straight over the page table: no gathered K/V copy, no mask
    tensor, whole prompt and chunked prefill alike, GQA by head index.  ``allow_fp16_qk_reduction`` asks the reference
    for fp16 score accumulation on RTX 4090 (utils/__init__.py:165); scores accumulate in fp32 here either way.  The
    rotary arguments are unused, as in the reference (RotaryMode::kNone, batch_prefill.cu:102).  Returns a new tensor
    (batch_prefill.cu:73)."""
    _check_input(q, "q")
    _check_input(kv_data, "kv_data")
    _check_input(kv_indices, "kv_indices")
    _check_dim(3, q, "q")
    _check_dim(5, kv_data, "kv_data")
    _check_dim(1, kv_indices, "kv_indices")
    _check_eq(kv_indices.dtype, torch.int32, "kv_indices.scalar_type(), torch::kInt32")
    _check_eq(q.size(2), kv_data.size(4), "q.size(2), kv_data.size(4)")
    _check_half(q, "BatchPrefillWithPagedKVCache")
    o = torch.empty_like(q)
    if q.size(0) == 0:  # nothing to attend from (the reference launches an empty grid)
        return o
    kv = _paged(kv_data, kv_indices, None, kv_last_page_len, 0, layout)
    check(lib.quest_prefill_with_paged_kv_cache(q.data_ptr(), o.data_ptr(), q.size(0), q.size(1), kv, kv_indices.size(0),
                                                1 if causal else 0, _stream(q)), "BatchPrefillWithPagedKVCache")
    return o


# ---------------------------------------------------------------- handler class

class BatchDecodeWithPagedKVCachePyTorchWrapper:
    """bsk_ops.h:88-117 / approx_attn.cu:27-150: begin_forward plans, forward launches."""

    def __init__(self, layout: int):
        self._layout = int(layout)
        h = ctypes.c_void_p()
        check(lib.quest_decode_handler_create(ctypes.byref(h), self._layout), "BatchDecodeWithPagedKVCache")
        self._h = h
        self._destroy = lib.quest_decode_handler_destroy  # keep alive for __del__ at interpreter exit

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._destroy(h)

    def begin_forward(self, indptr, num_qo_heads: int, num_kv_heads: int, head_dim: int, page_size: int,
                      empty_data) -> None:
        _check_dim(1, indptr, "indptr")
        _check_eq(indptr.dtype, torch.int32, "indptr.scalar_type(), torch::kInt32")
        if empty_data.dtype != torch.float16:
            raise RuntimeError(f"BatchDecodeWithPagedKVCache failed to dispatch with dtype {empty_data.dtype}")
        # the planner needs the selected-page count on the host; the reference copies indptr back
        # too (decode_attn.cuh:866-873).  A CPU indptr skips the device sync.
        host = indptr if not indptr.is_cuda else indptr.cpu()
        n_sel = int(host[-1]) - int(host[0])
        stream = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
        check(lib.quest_decode_begin_forward(self._h, n_sel, int(num_qo_heads), int(num_kv_heads), int(head_dim),
                                             int(page_size), stream), "BatchDecodeWithPagedKVCache")

    def end_forward(self) -> None:
        check(lib.quest_decode_end_forward(self._h), "BatchDecodeWithPagedKVCache")

    def forward(self, q, o, paged_kv_data, paged_kv_indices, paged_kv_indptr, paged_kv_last_page_len: int,
                paged_kv_last_page_idx: int, rope_scale: float, rope_theta: float) -> None:
        _check_input(q, "q")
        _check_input(o, "o")
        _check_input(paged_kv_data, "paged_kv_data")
        _check_input(paged_kv_indices, "paged_kv_indices")
        _check_dim(3, q, "q")
        _check_dim(2, paged_kv_indices, "paged_kv_indices")
        _check_dim(5, paged_kv_data, "paged_kv_data")
        _check_eq(paged_kv_indices.size(0), q.size(1), "paged_kv_indices.size(0), num_qo_heads")
        _check_eq(paged_kv_data.size(1), 2, "paged_kv_data.size(1), 2")
        _check_eq(paged_kv_data.size(4), q.size(2), "paged_kv_data.size(4), head_dim")
        _check_eq(paged_kv_indices.dtype, torch.int32, "paged_kv_indices.scalar_type(), torch::kInt32")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, paged_kv_indices, paged_kv_indptr, paged_kv_last_page_len,
                    paged_kv_last_page_idx, self._layout, page_budget=paged_kv_indices.size(1))
        check(lib.quest_decode_forward(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), None, _stream(q)),
              "BatchDecodeWithPagedKVCache")

    def forward_shared(self, q, o, paged_kv_data, page_list, paged_kv_last_page_len: int,
                       paged_kv_last_page_idx: int) -> bool:
        """EXTENSION: forward when all query heads share ONE page list ``[n_selected]`` (full-KV decode).  With
        GQA each K/V tile is fetched once per kv head.  False (nothing launched) if the shape is outside the
        shared kernel's set (page_size 16, head_dim 64/128)."""
        _check_input(q, "q")
        _check_input(o, "o")
        _check_input(paged_kv_data, "paged_kv_data")
        _check_input(page_list, "page_list")
        _check_dim(1, page_list, "page_list")
        _check_eq(page_list.dtype, torch.int32, "page_list.scalar_type(), torch::kInt32")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, page_list, None, paged_kv_last_page_len, paged_kv_last_page_idx, self._layout)
        code = lib.quest_decode_forward_shared(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), None, _stream(q))
        if code == -2:
            return False
        check(code, "BatchDecodeWithPagedKVCache")
        return True

    def forward_fused_topk(self, q, o, paged_kv_data, page_table, scores, topk_val_out, topk_idx_out,
                           paged_kv_last_page_len: int, paged_kv_last_page_idx: int) -> bool:
        """EXTENSION: topk_filtering + forward in one launch.  ``page_table`` is the sequence's page table
        ``[n_pages]`` (kv_indices_with_last), ``scores`` the estimate output ``[Hq, n_pages-1]``.  Returns
        False (nothing launched) when the current plan's chunks are too large for the fused front end."""
        _check_input(q, "q")
        _check_input(o, "o")
        _check_input(paged_kv_data, "paged_kv_data")
        _check_input(page_table, "page_table")
        _check_rows(scores, "scores")  # [Hq, n_scores], rows possibly padded (decode_append_estimate's output)
        _check_dim(3, q, "q")
        _check_dim(1, page_table, "page_table")
        _check_dim(5, paged_kv_data, "paged_kv_data")
        _check_eq(scores.size(0), q.size(1), "scores.size(0), num_qo_heads")
        _check_eq(page_table.size(0), scores.size(1) + 1, "page_table.size(0), n_scores + 1")
        _check_eq(page_table.dtype, torch.int32, "page_table.scalar_type(), torch::kInt32")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        _check_half(scores, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, page_table, None, paged_kv_last_page_len, paged_kv_last_page_idx, self._layout)
        code = lib.quest_decode_forward_fused_topk_strided(
            self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), scores.data_ptr(), scores.size(1), scores.stride(0),
            topk_val_out.data_ptr() if topk_val_out is not None else None,
            topk_idx_out.data_ptr() if topk_idx_out is not None else None, None, _stream(q))
        if code == -2:  # QUEST_EUNSUPPORTED: chunk larger than the fused front end stages
            return False
        check(code, "BatchDecodeWithPagedKVCache")
        return True

    def forward_shared_dyn(self, q, o, paged_kv_data, page_table, state) -> None:
        """forward_shared over all pages of the sequence, live length from ``state`` (graph replay)."""
        for t, n in ((q, "q"), (o, "o"), (paged_kv_data, "paged_kv_data"), (page_table, "page_table"), (state, "state")):
            _check_input(t, n)
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, page_table, None, 1, 0, self._layout)
        check(lib.quest_decode_forward_shared_dyn(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), state.data_ptr(),
                                                  None, _stream(q)), "BatchDecodeWithPagedKVCache")

    def append_forward_shared_dyn(self, k, v, metadata_data, meta_table, q, o, paged_kv_data, page_table, state) -> bool:
        """``append_kv_cache_decode_dyn`` + ``forward_shared_dyn`` in ONE launch (the new token's row is taken from k / v
        by the workgroup that attends the current page, written to the pool and folded into the metadata there).  False
        when the shape is outside the group-shared kernel: the caller issues the two launches."""
        for t, n in ((k, "k"), (v, "v"), (metadata_data, "metadata_data"), (meta_table, "meta_table"), (q, "q"), (o, "o"),
                     (paged_kv_data, "paged_kv_data"), (page_table, "page_table"), (state, "state")):
            _check_input(t, n)
        _check_half(q, "BatchDecodeWithPagedKVCache")
        _check_half(k, "BatchDecodeWithPagedKVCache")
        _check_eq(tuple(k.shape), tuple(v.shape), "k.shape, v.shape")
        kv = _paged(paged_kv_data, page_table, None, 1, 0, self._layout)
        _check_eq(tuple(k.shape[-2:]), (kv.num_heads, kv.head_dim), "k.shape[-2:], (kv heads, head_dim)")
        meta = _paged(metadata_data, meta_table, None, 1, 0, self._layout)
        code = lib.quest_decode_append_forward_shared_dyn(self._h, k.data_ptr(), v.data_ptr(), meta, q.data_ptr(), o.data_ptr(),
                                                          kv, q.size(1), state.data_ptr(), None, _stream(q))
        if code == -2:
            return False
        check(code, "BatchDecodeWithPagedKVCache")
        return True

    def append_forward_shared_batched(self, k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables, state) -> bool:
        """``append_forward_shared_dyn`` for every sequence of a batch (k, v ``[n, Hkv, D]``)."""
        for t, n in ((k, "k"), (v, "v"), (metadata_data, "metadata_data"), (q, "q"), (o, "o"),
                     (paged_kv_data, "paged_kv_data"), (state, "state")):
            _check_input(t, n)
        b = _batch(state, kv_tables, meta_tables)
        _check_dim(3, q, "q")
        _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
        _check_eq(k.size(0), b.n_seqs, "k.size(0), n_seqs")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        _check_half(k, "BatchDecodeWithPagedKVCache")
        _check_eq(tuple(k.shape), tuple(v.shape), "k.shape, v.shape")
        kv = _paged(paged_kv_data, kv_tables, None, 1, 0, self._layout)
        _check_eq(tuple(k.shape[-2:]), (kv.num_heads, kv.head_dim), "k.shape[-2:], (kv heads, head_dim)")
        meta = _paged(metadata_data, meta_tables, None, 1, 0, self._layout)
        code = lib.quest_decode_append_forward_shared_batched(self._h, k.data_ptr(), v.data_ptr(), meta, q.data_ptr(),
                                                              o.data_ptr(), kv, q.size(1), state.data_ptr(), b, None, _stream(q))
        if code == -2:
            return False
        check(code, "BatchDecodeWithPagedKVCache")
        return True

    def forward_fused_topk_dyn(self, q, o, paged_kv_data, page_table, scores, state, max_n_scores: int,
                               tiles: bool = False) -> bool:
        """forward_fused_topk whose row length and current page come from ``state`` (graph replay).
        ``tiles``: the rows carry their tile maxima (``append_estimate_dyn(..., tiles=True)`` wrote them): two short
        selection passes instead of passes over the whole row; returns False (nothing launched) where the plan is
        outside what the tiles launch serves (more than 256 selected pages, page size != 16)."""
        for t, n in ((q, "q"), (o, "o"), (paged_kv_data, "paged_kv_data"), (page_table, "page_table"),
                     (scores, "scores"), (state, "state")):
            _check_input(t, n)
        _check_dim(2, scores, "scores")
        _check_eq(scores.size(0), q.size(1), "scores.size(0), num_qo_heads")
        _check_ge(scores.size(1), max_n_scores, "scores.size(1), max_n_scores")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, page_table, None, 1, 0, self._layout)
        if tiles:
            _check_ge(scores.size(1), tiles_row_stride(max_n_scores), "scores.size(1), tiles_row_stride(max_n_scores)")
            code = lib.quest_decode_forward_fused_topk_tiles_dyn(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1),
                                                                 scores.data_ptr(), scores.size(1), int(max_n_scores),
                                                                 tile_max_offset(max_n_scores), state.data_ptr(), None,
                                                                 _stream(q))
            if code == -2:
                return False
            check(code, "BatchDecodeWithPagedKVCache")
            return True
        check(lib.quest_decode_forward_fused_topk_dyn(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1),
                                                      scores.data_ptr(), scores.size(1), int(max_n_scores),
                                                      state.data_ptr(), None, _stream(q)),
              "BatchDecodeWithPagedKVCache")
        return True

    def set_batch(self, n_seqs: int) -> None:
        """Sequences per launch the NEXT begin_forward plans for (workspace, work split)."""
        check(lib.quest_decode_set_batch(self._h, int(n_seqs)), "set_batch")

    def forward_batched(self, q, o, paged_kv_data, indices, state, budgets=None) -> None:
        """forward for ``n`` sequences in one launch: q/o ``[n, Hq, D]``, indices ``[n, Hq, k_max]`` (int32) of which
        row (i, h) is read up to ``min(budget_i - 1, n_pages_i - 1)`` entries; the current page comes from ``state``."""
        for t, n in ((q, "q"), (o, "o"), (paged_kv_data, "paged_kv_data"), (indices, "indices"), (state, "state")):
            _check_input(t, n)
        b = _batch(state, None, None, budgets)
        _check_dim(3, q, "q")
        _check_dim(3, indices, "indices")
        _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
        _check_eq(indices.size(0), b.n_seqs, "indices.size(0), n_seqs")
        _check_eq(indices.size(1), q.size(1), "indices.size(1), num_qo_heads")
        _check_eq(indices.dtype, torch.int32, "indices.scalar_type(), torch::kInt32")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, None, None, 1, 0, self._layout)
        check(lib.quest_decode_forward_batched(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), indices.data_ptr(),
                                               indices.size(2), state.data_ptr(), b, None, _stream(q)),
              "BatchDecodeWithPagedKVCache")

    def forward_fused_topk_batched(self, q, o, paged_kv_data, kv_tables, scores, state, max_n_scores: int,
                                   budgets=None) -> None:
        """forward_fused_topk_dyn for ``n`` sequences in one launch: q/o ``[n, Hq, D]``, scores ``[n, Hq, stride]``."""
        for t, n in ((q, "q"), (o, "o"), (paged_kv_data, "paged_kv_data"), (scores, "scores"), (state, "state")):
            _check_input(t, n)
        b = _batch(state, kv_tables, None, budgets)
        _check_dim(3, q, "q")
        _check_dim(3, scores, "scores")
        _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
        _check_eq(scores.size(0), b.n_seqs, "scores.size(0), n_seqs")
        _check_eq(scores.size(1), q.size(1), "scores.size(1), num_qo_heads")
        _check_ge(scores.size(2), max_n_scores, "scores.size(2), max_n_scores")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, kv_tables, None, 1, 0, self._layout)
        check(lib.quest_decode_forward_fused_topk_batched(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1),
                                                          scores.data_ptr(), scores.size(2), int(max_n_scores),
                                                          state.data_ptr(), b, None, _stream(q)),
              "BatchDecodeWithPagedKVCache")

    def layer_fused_batched(self, k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables, state,
                            max_n_scores: int, budgets=None, scores_out=None) -> bool:
        """One launch per layer of a batched step: ``append_estimate_batched`` + ``forward_fused_topk_batched`` as one
        grid of (sequence, head) workgroups that keep the page scores in LDS (csrc/layer_device.cuh).  Same pool bytes,
        selections and outputs.  ``scores_out``: optional ``[n, Hq, >= max_n_scores]`` fp16 inspection copy of the
        scores.  Returns False (nothing launched) when the plan / shape is outside what the launch serves."""
        for t, n in ((k, "k"), (v, "v"), (metadata_data, "metadata_data"), (q, "q"), (o, "o"),
                     (paged_kv_data, "paged_kv_data"), (state, "state")):
            _check_input(t, n)
        b = _batch(state, kv_tables, meta_tables, budgets)
        _check_dim(3, q, "q")
        _check_dim(3, k, "k")
        _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
        _check_eq(k.size(0), b.n_seqs, "k.size(0), n_seqs")
        _check_eq(tuple(o.shape), tuple(q.shape), "o.shape, q.shape")
        _check_eq(tuple(k.shape), tuple(v.shape), "k.shape, v.shape")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        _check_half(k, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, kv_tables, None, 1, 0, self._layout)
        _check_eq(tuple(k.shape[-2:]), (kv.num_heads, kv.head_dim), "k.shape[-2:], (kv heads, head_dim)")
        meta = _paged(metadata_data, meta_tables, None, 1, 0, self._layout)
        sc_ptr, sc_stride = None, 0
        if scores_out is not None:
            _check_input(scores_out, "scores_out")
            _check_dim(3, scores_out, "scores_out")
            _check_eq(tuple(scores_out.shape[:2]), tuple(q.shape[:2]), "scores_out.shape[:2], (n_seqs, num_qo_heads)")
            _check_ge(scores_out.size(2), max_n_scores, "scores_out.size(2), max_n_scores")
            _check_half(scores_out, "BatchDecodeWithPagedKVCache")
            sc_ptr, sc_stride = scores_out.data_ptr(), scores_out.size(2)
        code = lib.quest_decode_layer_fused_batched(self._h, k.data_ptr(), v.data_ptr(), meta, q.data_ptr(), o.data_ptr(), kv,
                                                    q.size(1), int(max_n_scores), state.data_ptr(), b, sc_ptr, sc_stride,
                                                    None, _stream(q))
        if code == -2:
            return False
        check(code, "BatchDecodeWithPagedKVCache")
        return True

    def forward_shared_batched(self, q, o, paged_kv_data, kv_tables, state) -> None:
        for t, n in ((q, "q"), (o, "o"), (paged_kv_data, "paged_kv_data"), (state, "state")):
            _check_input(t, n)
        b = _batch(state, kv_tables, None)
        _check_dim(3, q, "q")
        _check_eq(q.size(0), b.n_seqs, "q.size(0), n_seqs")
        _check_half(q, "BatchDecodeWithPagedKVCache")
        kv = _paged(paged_kv_data, kv_tables, None, 1, 0, self._layout)
        check(lib.quest_decode_forward_shared_batched(self._h, q.data_ptr(), o.data_ptr(), kv, q.size(1),
                                                      state.data_ptr(), b, None, _stream(q)),
              "BatchDecodeWithPagedKVCache")

    # introspection used by the bench / tuning sweeps (not part of the reference surface)
    def plan_info(self):
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        check(lib.quest_decode_plan_info(self._h, ctypes.byref(a), ctypes.byref(b)), "plan_info")
        return a.value, b.value

    def last_launch_info(self) -> dict:
        """Which kernel instantiation the handler's most recent per-head-list launch took: keys per thread of the fused
        front end (0 = index-tensor launch), waves per workgroup, front-end variant (DecodeParams.vec_front: 0 / 1 / 3
        first generation staged / vector-staged / direct, 2 second generation, 8 tiles, 7 the one-launch layer), whether
        the one-variant instantiation was launched, workgroups per head, sequences.  So that tests and benches can assert
        they run the same kernel."""
        info = (ctypes.c_uint32 * 6)()
        check(lib.quest_decode_last_launch_info(self._h, info), "last_launch_info")
        return {"keys_per_thread": info[0], "waves": info[1], "front_end_variant": info[2], "specialised": bool(info[3]),
                "workgroups_per_head": info[4], "n_seqs": info[5]}

    def set_pages_per_chunk(self, ppc: int) -> None:
        check(lib.quest_decode_set_pages_per_chunk(self._h, int(ppc)), "set_pages_per_chunk")

    def set_front_end(self, generation: int) -> None:
        """Tuning / test aid: force the fused launches' top-k front end (0 = automatic; 1 = first generation, 2 / 3 =
        second generation without / with its histogram pre-filter)."""
        check(lib.quest_decode_set_front_end(self._h, int(generation)), "set_front_end")

    def set_selection_out(self, val_out, idx_out) -> None:
        """Inspection aid: the state-driven / batched fused launches also write their selection into
        ``val_out`` fp16 / ``idx_out`` int32, both ``[n_seqs, Hq, n_selected]`` (None, None: off).  The caller keeps
        the tensors alive while set."""
        if idx_out is not None:
            _check_input(idx_out, "idx_out")
            _check_eq(idx_out.dtype, torch.int32, "idx_out.scalar_type(), torch::kInt32")
        if val_out is not None:
            _check_input(val_out, "val_out")
            _check_half(val_out, "set_selection_out")
        self._sel_keepalive = (val_out, idx_out)
        check(lib.quest_decode_set_selection_out(self._h, val_out.data_ptr() if val_out is not None else None,
                                                 idx_out.data_ptr() if idx_out is not None else None),
              "set_selection_out")

    def arm_step_advance(self, state, kv_tables, meta_tables, page_size: int) -> None:
        """The NEXT token's reservation (``step_state_advance``) rides in this handler's next forward call: in its merge
        launch where the plan has one, as its own launch behind the attention kernel otherwise.  ``state`` is the step state
        ``[8]`` (one sequence, tables ``[capacity]``) or ``[n, 8]`` (tables ``[n, capacity]``).  Call right before the LAST
        layer's forward of a step whose first token was reserved by ``step_state_advance`` (quest_hip.h)."""
        for t, n in ((state, "state"), (kv_tables, "kv_tables"), (meta_tables, "meta_tables")):
            _check_input(t, n)
            _check_eq(t.dtype, torch.int32, f"{n}.scalar_type(), torch::kInt32")
        if state.dim() == 1:
            b = Batch(1, 0, 0, 0, None)
            max_kv, max_meta = kv_tables.numel(), meta_tables.numel()
        else:
            b = _batch(state, kv_tables, meta_tables)
            max_kv, max_meta = kv_tables.size(1), meta_tables.size(1)
        check(lib.quest_decode_arm_step_advance(self._h, state.data_ptr(), kv_tables.data_ptr(), meta_tables.data_ptr(),
                                                int(page_size), max_kv, max_meta, b), "arm_step_advance")

    def set_skip_merge(self, skip: bool) -> None:
        """Measurement aid: launch only the attention kernel (partial states stay unmerged, ``o`` unwritten)."""
        check(lib.quest_decode_set_skip_merge(self._h, int(bool(skip))), "set_skip_merge")
