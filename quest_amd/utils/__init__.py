"""Operator wrappers with the reference's names and argument meaning (quest/utils/__init__.py:11-276).

Each wrapper turns ``InferenceController`` state into the argument list of one ``_kernels`` op.
All tensors are NHD/HND per ``iController.layout``; q/k/v projections are ``[tokens, heads, dim]``.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .. import _kernels
from .controller import BatchedInferenceController, InferenceController
from .decode_wrapper import BatchDecodeWithPagedKVCacheWrapper
from .kv_cache import KvCache
from .utils import TensorLayout

__all__ = [
    "TensorLayout",
    "KvCache",
    "InferenceController",
    "BatchDecodeWithPagedKVCacheWrapper",
    "append_kv",
    "prefill_forward",
    "decode_estimate",
    "decode_topk",
    "decode_sparse_attn",
    "rms_norm_forward",
    "apply_rope_in_place",
    # extensions (fused launches; same results as the op pairs they replace)
    "decode_append_estimate",
    "decode_topk_sparse_attn",
    # state-driven forms for hipGraph replay across tokens
    "step_advance_dyn",
    "score_scratch",
    "decode_layer_dyn",
    "decode_layer_dense_dyn",
    # batched state-driven forms: n sequences per launch
    "BatchedInferenceController",
    "step_advance_batched",
    "decode_layer_batched",
    "decode_layer_dense_batched",
    # the four operators one by one for a whole batch (eager), per-sequence page budgets
    "append_kv_batched",
    "decode_estimate_batched",
    "decode_topk_batched",
    "decode_sparse_attn_batched",
]


def _rope_defaults(rope_scale, rope_theta):
    return (1.0 if rope_scale is None else rope_scale), (1e4 if rope_theta is None else rope_theta)


def apply_rope_in_place(q: torch.Tensor, k: torch.Tensor, past_kv_len: int, rope_scale: Optional[float] = None,
                        rope_theta: Optional[float] = None) -> None:
    """Rotate q ``[N, Hq, D]`` and k ``[N, Hkv, D]`` in place; row i sits at position ``past_kv_len + i``."""
    scale, theta = _rope_defaults(rope_scale, rope_theta)
    _kernels.apply_rope_in_place(q, k, past_kv_len, scale, theta)


def rms_norm_forward(input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> torch.Tensor:
    out = torch.empty_like(input)
    _kernels.rms_norm_forward(input, weight, out, epsilon)
    return out


def _append_args(ctl: InferenceController, layer_idx: int):
    kv, meta = ctl.kv_cache, ctl.metadata_cache
    return (kv.buf_layer(layer_idx), ctl.kv_indices_with_last, ctl.kv_indptr_for_append, kv.last_page_len,
            ctl.kv_last_page_idx, meta.buf_layer(layer_idx), ctl.metadata_indices, ctl.metadata_indptr_for_append,
            meta.last_page_len, ctl.metadata_last_page_idx, ctl.layout)


def append_kv(k: torch.Tensor, v: torch.Tensor, iController: InferenceController, layer_idx: int) -> None:
    """Write new keys/values ``[N, Hkv, D]`` into the paged cache of ``layer_idx`` and fold the keys into
    the per-page (max, min) metadata.  N > 1 takes the prefill kernel, N == 1 the decode kernel."""
    op = _kernels.append_kv_cache_prefill if k.size(0) > 1 else _kernels.append_kv_cache_decode
    op(k, v, *_append_args(iController, layer_idx))


def prefill_forward(q: torch.Tensor, iController: InferenceController, layer_idx: int,
                    rope_scale: Optional[float] = None, rope_theta: Optional[float] = None) -> torch.Tensor:
    """Causal attention of q ``[N, Hq, D]`` over the (already appended) cache.  Not on the sparse path."""
    scale, theta = _rope_defaults(rope_scale, rope_theta)
    return _kernels.prefill_with_paged_kv_cache(
        q, iController.kv_cache.buf_layer(layer_idx), iController.kv_indices_with_last,
        iController.kv_cache.last_page_len, True, iController.layout, False, scale, theta)


def decode_estimate(q: torch.Tensor, iController: InferenceController, layer_idx: int) -> torch.Tensor:
    """Upper-bound criticality score of every page but the current one: ``[Hq, n_pages - 1]`` fp16."""
    meta = iController.metadata_cache
    # metadata seqlen == number of KV pages; the last entry belongs to the current page
    o = torch.empty((iController.num_heads, meta.seqlen - 1), dtype=q.dtype, device=q.device)
    _kernels.estimate_attn_score(q, o, meta.buf_layer(layer_idx), iController.metadata_indices,
                                 iController.metadata_indptr_for_append, meta.last_page_len,
                                 iController.metadata_last_page_idx, iController.layout)
    return o


def decode_topk(estimated_attn_score: torch.Tensor, iController: InferenceController) -> None:
    """Pick the ``budget - 1`` best pages per head into ``iController.topk_dindices_buffer``."""
    # (decode_append_estimate's padded rows are served in place: the op takes the row stride)
    _kernels.topk_filtering(estimated_attn_score, iController.kv_indices_without_last, iController.topk_dout_buffer,
                            iController.topk_dindices_buffer, iController.topk_buf,
                            iController.inference_page_budget - 1)


def decode_sparse_attn(q: torch.Tensor, iController: InferenceController, layer_idx: int, topk_indices: torch.Tensor,
                       rope_scale: Optional[float] = None, rope_theta: Optional[float] = None) -> torch.Tensor:
    """Attention of q ``[1, Hq, D]`` over the pages in ``topk_indices`` ``[Hq, budget - 1]`` plus the
    current page."""
    o = torch.empty_like(q)
    if topk_indices is iController.kv_indices_without_last and iController.kv_indices_with_last is not None:
        # full-KV decode: every head attends the same pages (the tensor is the page table repeated per head,
        # controller.py:106) -> one shared list, K/V fetched once per kv head (matters with GQA)
        n_sel = topk_indices.size(1)
        if iController._decode_handler.forward_shared(q, o, iController.kv_cache.buf_layer(layer_idx),
                                                      iController.kv_indices_with_last[:n_sel],
                                                      iController.kv_cache.last_page_len,
                                                      iController.kv_last_page_idx):
            return o
    iController._decode_handler.forward(q, o, iController.kv_cache.buf_layer(layer_idx), topk_indices,
                                        iController.kv_indptr_for_approx_decode, iController.kv_cache.last_page_len,
                                        iController.kv_last_page_idx, rope_scale, rope_theta)
    return o


# ---------------------------------------------------------------------------- fused extensions
# The reference issues five launches per layer for a decode token (append, estimate, top-k, attention,
# merge).  On MI355X each dependent launch costs ~1.5-2 us of boundary + a memory round trip, comparable
# to the 5 us the data movement itself takes, so the two pairs below are also offered as single launches.
# They are bit-identical to the pairs (tests/test_gpu_parity.py::test_fused_equals_unfused).

def decode_append_estimate(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, iController: InferenceController,
                           layer_idx: int) -> torch.Tensor:
    """``append_kv(k, v)`` + ``decode_estimate(q)`` in one launch; returns the scores ``[Hq, n_pages-1]`` -- a VIEW of a
    buffer whose rows are padded to a multiple of 8 columns (16 bytes), which is what lets ``decode_topk_sparse_attn`` run
    its vector-load front ends and serve rows beyond 4096 pages in one launch (the reference's contiguous layout,
    utils/__init__.py:171-205, has 2-byte aligned rows).  ``.contiguous()`` gives the reference layout back."""
    meta = iController.metadata_cache
    n_out = meta.seqlen - 1
    o = torch.empty((iController.num_heads, (n_out + 7) // 8 * 8), dtype=q.dtype, device=q.device)[:, :n_out]
    a = _append_args(iController, layer_idx)
    _kernels.append_estimate(k, v, a[0], a[1], a[2], a[3], a[4], q, o, a[5], a[6], a[7], a[8], a[9], a[10])
    return o


def decode_topk_sparse_attn(q: torch.Tensor, estimated_attn_score: torch.Tensor, iController: InferenceController,
                            layer_idx: int, write_topk: bool = True) -> torch.Tensor:
    """``decode_topk(scores)`` + ``decode_sparse_attn(q, topk_dindices_buffer)`` in one launch.  With
    ``write_topk`` the selection also lands in ``topk_dout_buffer`` / ``topk_dindices_buffer``."""
    o = torch.empty_like(q)
    ok = iController._decode_handler.forward_fused_topk(
        q, o, iController.kv_cache.buf_layer(layer_idx), iController.kv_indices_with_last, estimated_attn_score,
        iController.topk_dout_buffer if write_topk else None,
        iController.topk_dindices_buffer if write_topk else None,
        iController.kv_cache.last_page_len, iController.kv_last_page_idx)
    if not ok:  # plan with very large chunks: take the two-launch path
        decode_topk(estimated_attn_score, iController)
        return decode_sparse_attn(q, iController, layer_idx, iController.topk_dindices_buffer)
    return o


# ---------------------------------------------------------------------------- graph-replayable step
# With ``iController.enable_device_state()`` the per-token quantities live in device memory; a step built
# from the two calls below can be captured once (torch.cuda.graph) and replayed for every new token:
#
#     ctl.prepare_metadata(1); ctl.begin_forward(1)            # plan for the budget (host, once)
#     ctl.end_forward(); <undo the host-side reservation: see tests/test_gpu_graph_decode.py>
#     with torch.cuda.graph(g):
#         step_advance_dyn(ctl)                                # device-side prepare_metadata(1)
#         for layer: o[layer] = decode_layer_dyn(q[layer], k[layer], v[layer], ctl, layer, scores)
#     per token:  fill q/k/v buffers; g.replay(); ctl.prepare_metadata(1)   # host mirror only

def score_scratch(controller) -> torch.Tensor:
    """The fp16 page-score scratch of the state-driven launches: ``[Hq, stride]`` for an ``InferenceController``,
    ``[n_seqs, Hq, stride]`` for a ``BatchedInferenceController``, with ``stride`` = the pool capacity in pages rounded
    up to 8 columns -- 16-byte aligned rows let the attention launch read a thread's 4 scores with one 8-byte
    load (second-generation top-k front end, csrc/topk_bitmap.cuh).  Any ``[.., >= max_pages]`` fp16 tensor works;
    other strides take the first-generation front end (rows up to 4096 pages)."""
    stride = (controller.max_pages + 7) // 8 * 8
    if not isinstance(controller, BatchedInferenceController):
        # room for the rows' tile maxima behind the scores (long-row launches, ``decode_layer_dyn``)
        stride = max(stride, _kernels.tiles_row_stride(controller.max_pages - 1))
    shape = (controller.num_heads, stride)
    if isinstance(controller, BatchedInferenceController):
        shape = (controller.n_seqs,) + shape
    return torch.empty(shape, dtype=torch.float16, device=controller.device)


def _need_state(controller) -> None:
    """The state-driven ops read lengths and page tables from device memory: `enable_device_state()` must have been
    called after the prefill (and again after `clean_states()` / `quest_clear()`, which drop it)."""
    state = getattr(controller, "step_state", None) if not isinstance(controller, BatchedInferenceController) \
        else getattr(controller, "step_states", None)
    if state is None:
        raise RuntimeError("the controller has no device-resident step state: call enable_device_state() after the "
                           "prefill (clean_states() / quest_clear() drop it)")


def step_advance_dyn(iController: InferenceController) -> None:
    _need_state(iController)
    _kernels.step_state_advance(iController.step_state, iController.kv_table_full, iController.meta_table_full,
                                iController.page_size)


def decode_layer_dyn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, iController: InferenceController,
                     layer_idx: int, scores: torch.Tensor, rope_scale: Optional[float] = None,
                     rope_theta: Optional[float] = None, apply_rope: bool = False,
                     tiles: Optional[bool] = None, advance_after: bool = False) -> torch.Tensor:
    """One layer of a decode token in the sparse regime (pages > budget), every length read from the
    device-resident state: [RoPE] -> append+estimate -> top-k+attention (+merge).  ``scores`` is a
    caller-owned ``[Hq, >= max_pages]`` fp16 scratch (``score_scratch``).  ``tiles``: None = the tiles launches for pools
    of at least ``iController.tiles_min_pages`` pages (where the scratch has room and the plan allows), True / False force.
    ``advance_after``: the LAST layer of a step -- the next token's reservation (``step_advance_dyn``) rides in this layer's
    merge launch instead of heading the next step as a launch of its own (the first token is reserved by one
    ``step_advance_dyn`` before the first step; the host mirror ``prepare_metadata(1)`` follows every step as before)."""
    ctl = iController
    _need_state(ctl)
    if advance_after:
        ctl._decode_handler.arm_step_advance(ctl.step_state, ctl.kv_table_full, ctl.meta_table_full, ctl.page_size)
    if apply_rope:
        scale, theta = _rope_defaults(rope_scale, rope_theta)
        _kernels.apply_rope_in_place_dyn(q, k, scale, theta, ctl.step_state)
    max_n = ctl.max_pages - 1
    # long rows: the estimate hands the rows' tile maxima (per 8 pages the largest score) to the attention launch, which
    # then selects in two short passes (csrc/decode_device.cuh sparse_decode_tiles_body); same selection, same outputs
    tiles = (tiles if tiles is not None else max_n >= ctl.tiles_min_pages) and scores.dim() == 2 \
        and scores.size(1) >= _kernels.tiles_row_stride(max_n) and ctl.inference_page_budget - 1 <= 256
    tiles = _kernels.append_estimate_dyn(k, v, ctl.kv_cache.buf_layer(layer_idx), ctl.kv_table_full, q, scores,
                                         ctl.metadata_cache.buf_layer(layer_idx), ctl.meta_table_full, ctl.step_state,
                                         max_n, ctl.layout, tiles=True) if tiles else False
    if not tiles:
        _kernels.append_estimate_dyn(k, v, ctl.kv_cache.buf_layer(layer_idx), ctl.kv_table_full, q, scores,
                                     ctl.metadata_cache.buf_layer(layer_idx), ctl.meta_table_full, ctl.step_state, max_n,
                                     ctl.layout)
    o = torch.empty_like(q)
    if not (tiles and ctl._decode_handler.forward_fused_topk_dyn(q, o, ctl.kv_cache.buf_layer(layer_idx), ctl.kv_table_full,
                                                                 scores, ctl.step_state, max_n, tiles=True)):
        ctl._decode_handler.forward_fused_topk_dyn(q, o, ctl.kv_cache.buf_layer(layer_idx), ctl.kv_table_full, scores,
                                                   ctl.step_state, max_n)
    return o


def decode_layer_dense_dyn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, iController: InferenceController,
                           layer_idx: int, rope_scale: Optional[float] = None, rope_theta: Optional[float] = None,
                           apply_rope: bool = False, fuse_append: bool = True, advance_after: bool = False) -> torch.Tensor:
    """One FULL-KV layer of a decode token (``advance_after``: see ``decode_layer_dyn``), every length read from the device-resident state:
    [RoPE] -> append + attention over all pages in ONE launch (group-shared kernel; the workgroup that attends the
    current page takes the new token from k / v, writes it to the pool and folds it into the metadata) (+merge).  Shapes
    outside the group-shared kernel take the separate append launch.  Needs ``begin_graph_decode(dense_layers=True)``."""
    ctl = iController
    _need_state(ctl)
    if apply_rope:
        scale, theta = _rope_defaults(rope_scale, rope_theta)
        _kernels.apply_rope_in_place_dyn(q, k, scale, theta, ctl.step_state)
    o = torch.empty_like(q)
    kvb, mb = ctl.kv_cache.buf_layer(layer_idx), ctl.metadata_cache.buf_layer(layer_idx)
    if advance_after:
        ctl._dense_handler.arm_step_advance(ctl.step_state, ctl.kv_table_full, ctl.meta_table_full, ctl.page_size)
    if not (fuse_append and ctl._dense_handler.append_forward_shared_dyn(k, v, mb, ctl.meta_table_full, q, o, kvb,
                                                                         ctl.kv_table_full, ctl.step_state)):
        _kernels.append_kv_cache_decode_dyn(k, v, kvb, ctl.kv_table_full, mb, ctl.meta_table_full, ctl.step_state, ctl.layout)
        ctl._dense_handler.forward_shared_dyn(q, o, kvb, ctl.kv_table_full, ctl.step_state)
    return o



# --------------------------------------------------------------------------------------------------
# Batched state-driven step: the same three launches per layer serve every sequence of a
# ``BatchedInferenceController`` (grid.z = sequence).  q/k/v are ``[n_seqs, heads, dim]``.

def step_advance_batched(bController: BatchedInferenceController) -> None:
    _kernels.step_state_advance_batched(bController.step_states, bController.kv_tables, bController.meta_tables,
                                        bController.page_size)


def decode_layer_batched(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bController: BatchedInferenceController,
                         layer_idx: int, scores: torch.Tensor, rope_scale: Optional[float] = None,
                         rope_theta: Optional[float] = None, apply_rope: bool = False,
                         out: Optional[torch.Tensor] = None, one_launch: Optional[bool] = None,
                         write_scores: bool = False) -> torch.Tensor:
    """``decode_layer_dyn`` for every sequence at once; ``scores`` is ``[n_seqs, Hq, >= max_pages]`` fp16.

    ``one_launch``: None (default) = the controller's choice (``BatchedInferenceController.one_launch_layers``): ONE
    launch per layer where the plan allows it -- one workgroup per (sequence, head) that appends, scores its head's pages
    into LDS, selects from LDS and gathers (csrc/layer_device.cuh; the batch must fill the chip) -- else the two launches
    append+estimate | top-k+attention; True / False force one of them (True raises where the one-launch form does not
    apply).  Both produce the same pool bytes, selections and outputs.
    ``write_scores``: the one-launch form also writes its page scores to ``scores`` (inspection; they otherwise never
    leave the CU).

    NOTE -- ``scores`` after the call: only the TWO-launch form fills it.  Where the one-launch form runs (the default for
    MHA batches that fill the chip) ``scores`` keeps whatever it held before unless ``write_scores=True``; code that reads
    the scratch after this call must pass ``write_scores=True`` or ``one_launch=False``.  Which form ran is left in
    ``bController.last_layer_launches`` (1 or 2)."""
    b = bController
    _need_state(b)
    if apply_rope:
        scale, theta = _rope_defaults(rope_scale, rope_theta)
        _kernels.apply_rope_in_place_batched(q, k, scale, theta, b.step_states)
    max_n = b.max_pages - 1
    if (b.one_launch_layers if one_launch is None else one_launch):
        o = torch.empty_like(q) if out is None else out
        if b._decode_handler.layer_fused_batched(k, v, b.metadata_layer(layer_idx), b.meta_tables, q, o, b.kv_layer(layer_idx),
                                                 b.kv_tables, b.step_states, max_n, b.page_budgets,
                                                 scores if write_scores else None):
            b.last_layer_launches = 1
            return o
        if one_launch:
            raise RuntimeError("the one-launch layer does not serve this plan / shape (needs one workgroup per head, "
                               "page_size 16, head_dim 64 / 128, <= 255 selected pages, rows <= 4096 pages)")
    _kernels.append_estimate_batched(k, v, b.kv_layer(layer_idx), b.kv_tables, q, scores, b.metadata_layer(layer_idx),
                                     b.meta_tables, b.step_states, max_n, b.layout)
    o = torch.empty_like(q) if out is None else out
    b._decode_handler.forward_fused_topk_batched(q, o, b.kv_layer(layer_idx), b.kv_tables, scores, b.step_states, max_n,
                                                 b.page_budgets)
    b.last_layer_launches = 2
    return o


# ---- the reference's four operators of a decode step, one by one, for a whole batch (eager: no captured graph needed;
# lengths come from the device state uploaded by ``BatchedInferenceController.begin_forward``).  Per sequence the
# results are the single-sequence operators' bits (tests/test_gpu_batched.py).

def append_kv_batched(k: torch.Tensor, v: torch.Tensor, bController: BatchedInferenceController, layer_idx: int) -> None:
    """``append_kv`` of one decode token per sequence: k, v ``[n_seqs, Hkv, D]``."""
    b = bController
    _need_state(b)
    _kernels.append_kv_cache_decode_batched(k, v, b.kv_layer(layer_idx), b.kv_tables, b.metadata_layer(layer_idx),
                                            b.meta_tables, b.step_states, b.layout)


def decode_estimate_batched(q: torch.Tensor, bController: BatchedInferenceController, layer_idx: int,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``decode_estimate`` for every sequence: returns ``[n_seqs, Hq, stride]`` page scores (row i valid up to its
    page count - 1)."""
    b = bController
    _need_state(b)
    o = score_scratch(b) if out is None else out
    _kernels.estimate_attn_score_batched(q, o, b.metadata_layer(layer_idx), b.meta_tables, b.step_states, b.max_pages - 1,
                                         b.layout)
    return o


def decode_topk_batched(estimated_attn_score: torch.Tensor, bController: BatchedInferenceController) -> None:
    """``decode_topk`` for every sequence into ``topk_dout_buffer`` / ``topk_dindices_buffer`` ``[n_seqs, Hq, k_max]``:
    row (i, h) holds ``min(budget_i - 1, pages_i - 1)`` (score, physical page) pairs in ascending column order."""
    b = bController
    _need_state(b)
    _kernels.topk_filtering_batched(estimated_attn_score, b.kv_tables, b.topk_dout_buffer, b.topk_dindices_buffer,
                                    b.step_states, b.max_pages - 1, b.inference_page_budget, b.page_budgets)


def decode_sparse_attn_batched(q: torch.Tensor, bController: BatchedInferenceController, layer_idx: int,
                               topk_indices: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``decode_sparse_attn`` for every sequence over ``topk_indices`` ``[n_seqs, Hq, k_max]`` + the current pages."""
    b = bController
    _need_state(b)
    o = torch.empty_like(q) if out is None else out
    b._decode_handler.forward_batched(q, o, b.kv_layer(layer_idx), topk_indices, b.step_states, b.page_budgets)
    return o


def decode_layer_dense_batched(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                               bController: BatchedInferenceController, layer_idx: int,
                               rope_scale: Optional[float] = None, rope_theta: Optional[float] = None,
                               apply_rope: bool = False, out: Optional[torch.Tensor] = None,
                               fuse_append: bool = True) -> torch.Tensor:
    """``decode_layer_dense_dyn`` for every sequence at once (needs ``begin_graph_decode(dense_layers=True)``)."""
    b = bController
    _need_state(b)
    if apply_rope:
        scale, theta = _rope_defaults(rope_scale, rope_theta)
        _kernels.apply_rope_in_place_batched(q, k, scale, theta, b.step_states)
    o = torch.empty_like(q) if out is None else out
    kvb, mb = b.kv_layer(layer_idx), b.metadata_layer(layer_idx)
    if not (fuse_append and b._dense_handler.append_forward_shared_batched(k, v, mb, b.meta_tables, q, o, kvb, b.kv_tables,
                                                                           b.step_states)):
        _kernels.append_kv_cache_decode_batched(k, v, kvb, b.kv_tables, mb, b.meta_tables, b.step_states, b.layout)
        b._dense_handler.forward_shared_batched(q, o, kvb, b.kv_tables, b.step_states)
    return o
