"""Per-sequence decode state: page tables, budgets, top-k buffers, handler lifecycle.

Mirrors the reference's ``InferenceController`` (quest/utils/controller.py:7-146) attribute for
attribute, because ``quest.utils``' wrappers and ``QuestAttention`` read these fields directly.
Behavioural contract kept: ``prepare_metadata`` -> ``begin_forward`` -> ops -> ``end_forward``;
``page_budget`` is in PAGES and includes the current page (controller.py:14); the first call of a
decode step may pass ``updateTensor=False`` to re-plan with another budget without touching the
index tensors (llama.py:434-439 layer-skip).

What differs is cost, not meaning: index tensors are views/expansions of a device-resident page
table (no Python-list -> tensor rebuild per token), small indptr tensors are cached by value, and
top-k output buffers are reused while the budget is unchanged.  Extension: ``num_kv_heads`` for
GQA pools (per-query-head selection).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .decode_wrapper import BatchDecodeWithPagedKVCacheWrapper
from .kv_cache import KvCache
from .utils import TensorLayout


class InferenceController:
    def __init__(self, num_layers, num_heads, head_dim, page_size,
                 page_budget,  # pages, including the last (current) page
                 max_seq_len,  # capacity of the KV / metadata pools, in tokens
                 dtype, device, num_kv_heads: Optional[int] = None, layout: int = TensorLayout.NHD,
                 shuffle_seed: Optional[int] = None, kv_pool=None, metadata_pool=None):
        self.num_heads = num_heads
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        if self.num_heads % self.num_kv_heads != 0:
            raise ValueError(f"num_heads {num_heads} is not a multiple of num_kv_heads {num_kv_heads}")
        self.head_dim = head_dim
        self.page_size = page_size
        self.layout = TensorLayout.parse(layout)
        self.device = device
        self.dtype = dtype

        max_kv_pages = (max_seq_len + page_size - 1) // page_size
        self.kv_cache = KvCache(num_layers, self.num_kv_heads, head_dim, max_seq_len, page_size, dtype, device,
                                self.layout, shuffle_seed, pool=kv_pool)
        # one metadata entry (max in the K slot, min in the V slot) per KV page: controller.py:29-37
        self.metadata_cache = KvCache(num_layers, self.num_kv_heads, head_dim, max_kv_pages, page_size, dtype, device,
                                      self.layout, None if shuffle_seed is None else shuffle_seed + 1,
                                      pool=metadata_pool)

        self._page_budget = page_budget
        self._decode_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout=TensorLayout.FORMAT2STR[self.layout])

        self.kv_indices_with_last = None
        self.kv_indices_without_last = None
        self.metadata_indices = None
        self.kv_last_page_idx = None
        self.metadata_last_page_idx = None
        self.kv_indptr_for_append = None
        self.metadata_indptr_for_append = None
        self.kv_indptr_for_approx_decode = None
        self.inference_page_budget = None
        self.topk_dout_buffer = None
        self.topk_dindices_buffer = None
        self.topk_buf = None

        self._indptr_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._without_last_pages = -1
        # top-k output buffers, one pair per distinct width (the model alternates between the full-KV plan of its
        # first layers and the real budget on every token, llama.py:428-439: keyed by width they are allocated once)
        self._topk_buffers: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        # device-resident step state (enable_device_state); state_epoch changes whenever graphs captured over it go stale
        self.kv_table_full = self.meta_table_full = self.step_state = None
        self.state_epoch = 0

    # ------------------------------------------------------------------ budgets
    @property
    def page_budget_pages(self) -> int:
        return self._page_budget

    @property
    def token_budget(self) -> int:
        return self._page_budget * self.page_size

    def set_page_budget(self, page_budget: int) -> None:
        self._page_budget = page_budget

    # ------------------------------------------------------------------ step lifecycle
    def prepare_metadata(self, seq_len: int) -> None:
        """Reserve pages for ``seq_len`` new tokens and metadata entries for the pages they open."""
        new_pages = self.kv_cache.append_seq(seq_len)
        self.metadata_cache.append_seq(new_pages)

    def _indptr(self, n: int) -> torch.Tensor:
        t = self._indptr_cache.get((0, n))
        if t is None:
            if len(self._indptr_cache) > 64:
                self._indptr_cache.clear()
            t = torch.tensor([0, n], dtype=torch.int32, device=self.device)
            self._indptr_cache[(0, n)] = t
        return t

    def begin_forward(self, seq_len: int, updateTensor: bool = True) -> None:
        n_pages = len(self.kv_cache.indicies)
        n_meta = len(self.metadata_cache.indicies)
        if updateTensor:
            self.kv_indptr_for_append = self._indptr(n_pages)
            self.metadata_indptr_for_append = self._indptr(n_meta)
            self.kv_last_page_idx = self.kv_cache.indicies[-1]
            self.metadata_last_page_idx = self.metadata_cache.indicies[-1]
            self.kv_indices_with_last = self.kv_cache.device_table()
            self.metadata_indices = self.metadata_cache.device_table()

        if seq_len > 1:
            return  # prefill: append_kv_cache_prefill + prefill_with_paged_kv_cache need nothing else

        assert n_pages >= 1, "decode needs a non-empty cache"  # one page: attention over the current page only
        if updateTensor and self._without_last_pages != n_pages:
            # input ids of the top-k ([H, n_pages-1], controller.py:106), rebuilt only when a page was added
            self.kv_indices_without_last = self.kv_indices_with_last[:-1].unsqueeze(0).expand(
                self.num_heads, n_pages - 1).contiguous()
            self._without_last_pages = n_pages

        budget = min(self._page_budget, n_pages)
        # keyed on the buffer WIDTH, not on inference_page_budget (begin_graph_decode sets the latter without
        # allocating): a buffer narrower than budget - 1 would be written out of bounds by topk_filtering
        if self.topk_dout_buffer is None or self.topk_dout_buffer.size(1) != budget - 1:
            pair = self._topk_buffers.get(budget - 1)
            if pair is None:
                if len(self._topk_buffers) >= 8:  # a growing dense-regime sequence asks for a new width per page
                    self._topk_buffers.clear()
                pair = (torch.zeros((self.num_heads, budget - 1), dtype=self.dtype, device=self.device),
                        torch.zeros((self.num_heads, budget - 1), dtype=torch.int32, device=self.device))
                self._topk_buffers[budget - 1] = pair
            self.topk_dout_buffer, self.topk_dindices_buffer = pair
            if self.topk_buf is None:  # scratch argument of the reference's RAFT call; unused here
                self.topk_buf = torch.zeros((self.num_heads, 8), dtype=self.dtype, device=self.device)
        self.inference_page_budget = budget
        self.kv_indptr_for_approx_decode = self._indptr(budget - 1)
        # the planner wants the count on the host: hand it a CPU indptr (no device round trip)
        self._decode_handler.begin_forward(torch.tensor([0, budget - 1], dtype=torch.int32), self.num_heads,
                                           self.num_kv_heads, self.head_dim, self.page_size, self.dtype)

    def end_forward(self) -> None:
        self._decode_handler.end_forward()

    # ------------------------------------------------------------------ device-resident step state
    def enable_device_state(self) -> None:
        """Put the sequence's step state (lengths, last-page ids) in device memory so that a decode step
        captured in a hipGraph can be replayed while the sequence grows (EXTENSION).  Call
        after prefill; the state describes the cache BEFORE the next token -- the graph's first node,
        ``_kernels.step_state_advance``, moves it forward by one token on the device."""
        self.kv_table_full = self.kv_cache.full_device_table()
        self.meta_table_full = self.metadata_cache.full_device_table()
        self.max_pages = self.kv_table_full.numel()
        # decode_layer_dyn: pools from this many pages up take the tiles launches (tile maxima handed from the estimate
        # to the attention launch); below, every workgroup of a head selects over the whole row
        self.tiles_min_pages = getattr(self, "tiles_min_pages", 4097)
        self.step_state = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.state_epoch += 1  # graphs captured before this call are stale
        self.sync_device_state()

    def begin_graph_decode(self, dense_layers: bool = False) -> None:
        """Plan the sparse decode for the configured budget once, for a graph that will be replayed over
        many tokens (the plan depends only on the budget).  With
        ``dense_layers`` a second handler is planned for full-KV layers (the model's first layers,
        llama.py:428-430) over the pool's capacity."""
        # The sequence may still be shorter than the budget: the state-driven attention launch then selects all
        # of its pages (k = n), which is the reference's full-attention branch (QuestAttention.py:123-132), and
        # moves into the sparse regime by itself as the sequence grows -- same graph.
        assert len(self.kv_cache.indicies) >= 1, "prefill first"
        budget = min(self._page_budget, self.max_pages)
        self.inference_page_budget = budget
        self._decode_handler.begin_forward(torch.tensor([0, budget - 1], dtype=torch.int32), self.num_heads,
                                           self.num_kv_heads, self.head_dim, self.page_size, self.dtype)
        if dense_layers:
            if getattr(self, "_dense_handler", None) is None:
                self._dense_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout=TensorLayout.FORMAT2STR[self.layout])
            self._dense_handler.begin_forward(torch.tensor([0, self.max_pages - 1], dtype=torch.int32), self.num_heads,
                                              self.num_kv_heads, self.head_dim, self.page_size, self.dtype)

    def sync_device_state(self) -> None:
        kv, meta = self.kv_cache, self.metadata_cache
        host = torch.tensor([kv.seqlen, len(kv.indicies), kv.last_page_len, kv.indicies[-1], len(meta.indicies),
                             meta.last_page_len, meta.indicies[-1], 0], dtype=torch.int32)
        self.step_state.copy_(host)

    def need_estimate(self) -> bool:
        if self.inference_page_budget is None:
            return False
        return len(self.kv_cache.indicies) > self.inference_page_budget

    def clean_states(self) -> None:
        self.kv_cache.release()
        self.metadata_cache.release()
        self._without_last_pages = -1
        self.inference_page_budget = None
        # a captured decode graph holds the page tables / step state of the request that just ended: drop them so
        # that a stale graph cannot be replayed over the next request's pages (callers re-capture after prefill)
        self.invalidate_device_state()

    def invalidate_device_state(self) -> None:
        self.kv_table_full = self.meta_table_full = self.step_state = None
        self.state_epoch += 1


class BatchedInferenceController:
    """``n_seqs`` independent sequences decoded with ONE launch per op (EXTENSION; SURVEY 8f-3: the reference
    fixes ``batch_size = 1``, approx_attn.cu:113; BASELINE config 5 decodes 8 sequences per GPU).

    All sequences share one KV pool and one metadata pool (each reserves its ``max_seq_len`` worth of pages up
    front); ``self.seqs[i]`` is an ordinary ``InferenceController`` over those pools, so prefill and the eager
    single-sequence ops work on each sequence exactly as before.  After prefill, ``enable_device_state()``
    stacks the page tables / step states and the ``*_batched`` wrappers of ``quest_amd.utils`` run every
    sequence of a decode step in one grid (grid.z = sequence); results per sequence are bit-identical to the
    single-sequence path."""

    def __init__(self, n_seqs, num_layers, num_heads, head_dim, page_size, page_budget, max_seq_len, dtype, device,
                 num_kv_heads: Optional[int] = None, layout: int = TensorLayout.NHD,
                 shuffle_seed: Optional[int] = None):
        from .kv_cache import KvPool

        if n_seqs <= 0:
            raise ValueError("n_seqs must be positive")
        self.n_seqs = n_seqs
        self.num_heads = num_heads
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.head_dim, self.page_size, self.device, self.dtype = head_dim, page_size, device, dtype
        self.layout = TensorLayout.parse(layout)
        self._page_budget = page_budget
        pages = (max_seq_len + page_size - 1) // page_size
        meta_pages = (pages + page_size - 1) // page_size
        self.kv_pool = KvPool(num_layers, self.num_kv_heads, head_dim, n_seqs * pages, page_size, dtype, device,
                              self.layout, shuffle_seed)
        self.metadata_pool = KvPool(num_layers, self.num_kv_heads, head_dim, n_seqs * meta_pages, page_size, dtype,
                                    device, self.layout, None if shuffle_seed is None else shuffle_seed + 1)
        self.seqs = [InferenceController(num_layers, num_heads, head_dim, page_size, page_budget, max_seq_len, dtype,
                                         device, num_kv_heads=num_kv_heads, layout=layout, kv_pool=self.kv_pool,
                                         metadata_pool=self.metadata_pool) for _ in range(n_seqs)]
        self._decode_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout=TensorLayout.FORMAT2STR[self.layout])
        self._dense_handler = None
        self.kv_tables = self.meta_tables = self.step_states = None
        self.state_epoch = 0
        # per-sequence page budgets: ONE persistent int32 [n_seqs] device buffer (a captured graph holds its address),
        # rewritten in place by set_page_budgets; kNoBudget = "the planned budget" (the kernels clamp to the plan)
        self._budget_buf = None
        self.page_budgets = None      # the buffer once set_page_budgets has been called with a list, else None
        self._host_budgets = None
        self._graph_budget_ptr = 0    # what begin_graph_decode planned with: 0 = no per-sequence budgets
        self.topk_dout_buffer = self.topk_dindices_buffer = None
        self._planned = None
        self._graph_planned = 0       # budget begin_graph_decode planned for (0: no graph plan)
        # decode_layer_batched: one launch per layer (csrc/layer_device.cuh) where the plan allows it AND it pays: its grid
        # is one workgroup per (sequence, head), each bound by what ONE CU moves, so it needs the chip filled in whole
        # rounds -- measured at cfg-3 shapes, us per sequence-layer one / two launches: 5 sequences (160 workgroups on 256
        # CUs) 13.27 / 13.01, 6: 12.03 / 12.47, 7: 11.64 / 11.96, 8: 11.15 / 11.60, 12 (1.5 per CU) 12.10 / 11.86, 16: 11.02 /
        # 11.16 (profiles/r05_sweep_sequences_per_gpu_one_vs_two_launches.txt).  With GQA the query heads of a group would
        # each stream their kv head's metadata (cfg 5: 68.9 vs 58.8 us per layer): two launches.
        self.one_launch_layers = self.num_heads == self.num_kv_heads and self._fills_the_chip(n_seqs * num_heads, device)
        self.last_layer_launches = None  # 1 / 2: which form utils.decode_layer_batched ran last (its docstring)

    @staticmethod
    def _fills_the_chip(workgroups: int, device) -> bool:
        """Does a grid of `workgroups` one-per-CU-sized workgroups use the chip's CUs in (nearly) whole rounds?"""
        cus = 256  # MI355X
        try:
            dev = torch.device(device)
            if dev.type == "cuda" and torch.cuda.is_available():
                cus = torch.cuda.get_device_properties(dev).multi_processor_count
        except Exception:  # noqa: BLE001  (CPU-only host logic tests)
            pass
        rounds = -(-workgroups // cus)
        return workgroups / (rounds * cus) >= (0.75 if rounds == 1 else 0.8)

    # ---- per-sequence page budgets + eager (host-planned) batched steps.  The reference keeps one controller and one
    # page budget per request and loops over requests in Python (controller.py:39-41, :80-129: five list -> tensor
    # copies per token per request); here a step of the whole batch is ONE small host -> device copy of the sequences'
    # lengths, and every operator runs once for all sequences (quest_amd.utils.*_batched).
    def set_page_budgets(self, budgets) -> None:
        """Pages each sequence attends per step INCLUDING its current page (None: the constructor's budget for all).
        A sequence with fewer pages than its budget attends all of them."""
        if budgets is None:
            if self._graph_budget_ptr:
                # a captured step reads the buffer: "no per-sequence budget" = the sentinel, in place.  The kernels clamp
                # to the CAPTURED plan, so the constructor's budget can only be honoured if the plan covers it
                if min(self._page_budget, self.max_pages) > self._graph_planned:
                    raise RuntimeError(f"the constructor's budget ({self._page_budget} pages) exceeds the {self._graph_planned} "
                                       "pages the captured plan was made for; call begin_graph_decode() (and re-capture) "
                                       "after dropping the per-sequence budgets")
                self._budget_buf.fill_(self.kNoBudget)
                self._host_budgets = None
                return
            self.page_budgets = self._host_budgets = None
            return
        budgets = [int(x) for x in budgets]
        if len(budgets) != self.n_seqs or min(budgets) < 1:
            raise ValueError("one page budget >= 1 per sequence")
        if self._graph_planned and not self._graph_budget_ptr:
            raise RuntimeError("per-sequence budgets set after begin_graph_decode() planned without them: a captured "
                               "graph would ignore them; call set_page_budgets() before begin_graph_decode()")
        if self._graph_planned and max(budgets) > self._graph_planned:
            raise RuntimeError(f"budget {max(budgets)} exceeds the {self._graph_planned} pages the captured plan was "
                               "made for; call begin_graph_decode() (and re-capture) after raising budgets")
        self._host_budgets = budgets
        if self._budget_buf is None:
            self._budget_buf = torch.empty(self.n_seqs, dtype=torch.int32, device=self.device)
        self._budget_buf.copy_(torch.tensor(budgets, dtype=torch.int32))
        self.page_budgets = self._budget_buf

    kNoBudget = 2 ** 31 - 1

    def max_page_budget(self) -> int:
        return max(self._host_budgets) if self._host_budgets is not None else self._page_budget

    def begin_forward(self) -> None:
        """Plan one eager decode step of the batch after ``prepare_metadata(1)``: upload the sequences' lengths
        (``[n_seqs, 8]`` int32, one copy) and (re)plan the decode handler for the largest budget."""
        if self.step_states is None:
            raise RuntimeError("call enable_device_state() after the prefill")
        self.sync_device_state()
        budget = min(self.max_page_budget(), self.max_pages)
        if self._planned != budget:
            self._decode_handler.set_batch(self.n_seqs)
            self._decode_handler.begin_forward(torch.tensor([0, budget - 1], dtype=torch.int32), self.num_heads,
                                               self.num_kv_heads, self.head_dim, self.page_size, self.dtype)
            self._planned = budget
        # keyed on the buffers' WIDTH, not on _planned: begin_graph_decode plans without allocating them
        k = max(budget - 1, 1)
        if self.topk_dout_buffer is None or self.topk_dout_buffer.size(2) != k:
            self.topk_dout_buffer = torch.zeros(self.n_seqs, self.num_heads, k, dtype=self.dtype, device=self.device)
            self.topk_dindices_buffer = torch.zeros(self.n_seqs, self.num_heads, k, dtype=torch.int32, device=self.device)
        self.inference_page_budget = budget

    def end_forward(self) -> None:
        pass  # nothing to release: tables, states and the handler's workspace persist across steps

    def kv_layer(self, layer_idx: int) -> torch.Tensor:
        return self.kv_pool.layer(layer_idx)

    def metadata_layer(self, layer_idx: int) -> torch.Tensor:
        return self.metadata_pool.layer(layer_idx)

    def prepare_metadata(self, seq_len: int = 1) -> None:
        """Host mirror of the device-side reservation (``step_advance_batched``) for every sequence."""
        for c in self.seqs:
            c.prepare_metadata(seq_len)

    def enable_device_state(self) -> None:
        self.kv_tables = self._stack_tables([c.kv_cache.full_device_table() for c in self.seqs])
        self.meta_tables = self._stack_tables([c.metadata_cache.full_device_table() for c in self.seqs])
        self.max_pages = self.kv_tables.size(1)
        self.step_states = torch.zeros(self.n_seqs, 8, dtype=torch.int32, device=self.device)
        self.state_epoch += 1
        self.sync_device_state()

    @staticmethod
    def _stack_tables(tables) -> torch.Tensor:
        """``[n_seqs, capacity]`` view of a buffer whose rows are padded to a multiple of 4 entries: every sequence's
        table then starts 16-byte aligned, which the fused attention launch needs to fetch 4 page ids with one load
        (a stride of exactly `capacity` entries -- 2055, 2065, 2179 in the bench -- made every batched launch take the
        scalar-load front end).  The pad entries repeat the row's last page and are never indexed."""
        cap = tables[0].numel()
        stride = (cap + 3) // 4 * 4
        buf = torch.empty(len(tables), stride, dtype=torch.int32, device=tables[0].device)
        for row, t in zip(buf, tables):
            row[:cap] = t
            row[cap:] = t[-1]
        return buf[:, :cap]

    def sync_device_state(self) -> None:
        rows = []
        for c in self.seqs:
            kv, meta = c.kv_cache, c.metadata_cache
            rows.append([kv.seqlen, len(kv.indicies), kv.last_page_len, kv.indicies[-1], len(meta.indicies),
                         meta.last_page_len, meta.indicies[-1], 0])
        self.step_states.copy_(torch.tensor(rows, dtype=torch.int32))

    def begin_graph_decode(self, dense_layers: bool = False) -> None:
        """Plan the batched sparse decode once for the configured budget (every sequence must already be in
        the sparse regime), and optionally the batched full-KV decode over the per-sequence capacity."""
        # sequences shorter than the budget attend all of their pages (see InferenceController.begin_graph_decode)
        assert all(len(c.kv_cache.indicies) >= 1 for c in self.seqs), "prefill every sequence first"
        budget = min(self.max_page_budget(), self.max_pages)
        self.inference_page_budget = budget
        self._planned = budget
        # what a graph captured after this call bakes in: the plan's page count and whether the launches read budgets
        self._graph_planned = budget
        self._graph_budget_ptr = self.page_budgets.data_ptr() if self.page_budgets is not None else 0
        self._decode_handler.set_batch(self.n_seqs)
        self._decode_handler.begin_forward(torch.tensor([0, budget - 1], dtype=torch.int32), self.num_heads,
                                           self.num_kv_heads, self.head_dim, self.page_size, self.dtype)
        if dense_layers:
            if self._dense_handler is None:
                self._dense_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout=TensorLayout.FORMAT2STR[self.layout])
            self._dense_handler.set_batch(self.n_seqs)
            self._dense_handler.begin_forward(torch.tensor([0, self.max_pages - 1], dtype=torch.int32), self.num_heads,
                                              self.num_kv_heads, self.head_dim, self.page_size, self.dtype)

    def clean_states(self) -> None:
        for c in self.seqs:
            c.clean_states()
        self.kv_tables = self.meta_tables = self.step_states = None
        self._graph_planned, self._graph_budget_ptr = 0, 0  # graphs over the dropped state are stale anyway
        self.state_epoch += 1
