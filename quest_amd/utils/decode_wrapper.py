"""Python face of the decode handler (reference: quest/utils/decode_wrapper.py:11-81)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _kernels
from .utils import TensorLayout


class BatchDecodeWithPagedKVCacheWrapper:
    """Owns the plan (pages per workgroup, partial-state workspace) that ``begin_forward`` makes
    once per decode step and every layer's ``forward`` reuses."""

    def __init__(self, kv_layout: str = "NHD"):
        self.kv_layout = kv_layout
        self._wrapper = _kernels.BatchDecodeWithPagedKVCachePyTorchWrapper(TensorLayout.parse(kv_layout))

    def begin_forward(self, indptr: torch.Tensor, num_qo_heads: int, num_kv_heads: int, head_dim: int,
                      page_size: int, data_type) -> None:
        # dtype travels as an empty tensor, as in the reference's PyBind signature
        self._wrapper.begin_forward(indptr, num_qo_heads, num_kv_heads, head_dim, page_size,
                                    torch.empty(0, dtype=data_type))

    def end_forward(self) -> None:
        self._wrapper.end_forward()

    def forward(self, q: torch.Tensor, o: torch.Tensor, paged_kv_data: torch.Tensor, paged_kv_indices: torch.Tensor,
                paged_kv_indptr: torch.Tensor, paged_kv_last_page_len: int, paged_kv_last_page_idx: int,
                rope_scale: Optional[float] = None, rope_theta: Optional[float] = None) -> None:
        self._wrapper.forward(q, o, paged_kv_data, paged_kv_indices, paged_kv_indptr, paged_kv_last_page_len,
                              paged_kv_last_page_idx, 1.0 if rope_scale is None else rope_scale,
                              1e4 if rope_theta is None else rope_theta)

    def forward_shared(self, q, o, paged_kv_data, page_list, paged_kv_last_page_len: int,
                       paged_kv_last_page_idx: int) -> bool:
        return self._wrapper.forward_shared(q, o, paged_kv_data, page_list, paged_kv_last_page_len,
                                            paged_kv_last_page_idx)

    def forward_fused_topk(self, q, o, paged_kv_data, page_table, scores, topk_val_out, topk_idx_out,
                           paged_kv_last_page_len: int, paged_kv_last_page_idx: int) -> bool:
        return self._wrapper.forward_fused_topk(q, o, paged_kv_data, page_table, scores, topk_val_out, topk_idx_out,
                                                paged_kv_last_page_len, paged_kv_last_page_idx)

    def forward_shared_dyn(self, q, o, paged_kv_data, page_table, state) -> None:
        self._wrapper.forward_shared_dyn(q, o, paged_kv_data, page_table, state)

    def append_forward_shared_dyn(self, k, v, metadata_data, meta_table, q, o, paged_kv_data, page_table, state) -> bool:
        return self._wrapper.append_forward_shared_dyn(k, v, metadata_data, meta_table, q, o, paged_kv_data, page_table, state)

    def append_forward_shared_batched(self, k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables, state) -> bool:
        return self._wrapper.append_forward_shared_batched(k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables,
                                                           state)

    def forward_fused_topk_dyn(self, q, o, paged_kv_data, page_table, scores, state, max_n_scores: int,
                               tiles: bool = False) -> bool:
        return self._wrapper.forward_fused_topk_dyn(q, o, paged_kv_data, page_table, scores, state, max_n_scores, tiles)

    def set_batch(self, n_seqs: int) -> None:
        self._wrapper.set_batch(n_seqs)

    def forward_fused_topk_batched(self, q, o, paged_kv_data, kv_tables, scores, state, max_n_scores: int,
                                   budgets=None) -> None:
        self._wrapper.forward_fused_topk_batched(q, o, paged_kv_data, kv_tables, scores, state, max_n_scores, budgets)

    def layer_fused_batched(self, k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables, state, max_n_scores: int,
                            budgets=None, scores_out=None) -> bool:
        return self._wrapper.layer_fused_batched(k, v, metadata_data, meta_tables, q, o, paged_kv_data, kv_tables, state,
                                                 max_n_scores, budgets, scores_out)

    def forward_batched(self, q, o, paged_kv_data, indices, state, budgets=None) -> None:
        self._wrapper.forward_batched(q, o, paged_kv_data, indices, state, budgets)

    def forward_shared_batched(self, q, o, paged_kv_data, kv_tables, state) -> None:
        self._wrapper.forward_shared_batched(q, o, paged_kv_data, kv_tables, state)

    def plan_info(self):
        return self._wrapper.plan_info()

    def last_launch_info(self) -> dict:
        return self._wrapper.last_launch_info()

    def set_pages_per_chunk(self, ppc: int) -> None:
        self._wrapper.set_pages_per_chunk(ppc)

    def set_skip_merge(self, skip: bool) -> None:
        self._wrapper.set_skip_merge(skip)

    def arm_step_advance(self, state, kv_tables, meta_tables, page_size: int) -> None:
        self._wrapper.arm_step_advance(state, kv_tables, meta_tables, page_size)

    def set_front_end(self, generation: int) -> None:
        self._wrapper.set_front_end(generation)

    def set_selection_out(self, val_out, idx_out) -> None:
        self._wrapper.set_selection_out(val_out, idx_out)
