"""Paged KV storage for one sequence (reference: quest/utils/kv_cache.py:7-133).

Same public surface -- ``KvPool(buf, alloc_block, free_block, ...)`` and
``KvCache(seqlen, last_page_len, indicies, buf_layer, append_seq, release)`` -- with two
differences that matter at 32K-128K tokens on a 288 GB part:

* the pool is one ``[layers, capacity, 2, ...]`` allocation per sequence exactly as in the
  reference (kv_cache.py:20-23), in either layout;
* the page table is also mirrored in a device int32 tensor that is extended in place when
  pages are added, so the per-token controller step does not rebuild it from a Python list.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .utils import TensorLayout


class KvPool:
    """Fixed-capacity pool of pages shared by all layers of one sequence."""

    def __init__(self, num_layers: int, num_heads: int, head_dim: int, capacity: int, block_len: int,
                 dtype: torch.dtype, device, layout: int = TensorLayout.NHD, shuffle_seed: Optional[int] = None):
        self._layout = TensorLayout.parse(layout)
        inner = (num_heads, block_len, head_dim) if self._layout == TensorLayout.HND else (block_len, num_heads, head_dim)
        self._buf = torch.empty((num_layers, capacity, 2) + inner, dtype=dtype, device=device)
        # LIFO free list; ascending by default (the reference's set.pop() on small ints is too),
        # optionally shuffled so tests exercise arbitrary physical placement.
        order = list(range(capacity))
        if shuffle_seed is not None:
            g = torch.Generator().manual_seed(int(shuffle_seed))
            order = torch.randperm(capacity, generator=g).tolist()
        self._free: List[int] = order[::-1]
        self._in_use = [False] * capacity
        self._layers = [self._buf[i] for i in range(num_layers)]  # per-layer views, built once (hot in eager decode)

    @property
    def layout(self) -> int:
        return self._layout

    @property
    def buf(self) -> torch.Tensor:
        return self._buf

    @property
    def num_layers(self) -> int:
        return self._buf.shape[0]

    def layer(self, layer_idx: int) -> torch.Tensor:
        return self._layers[layer_idx]

    @property
    def capacity(self) -> int:
        return self._buf.shape[1]

    @property
    def block_len(self) -> int:
        return self._buf.shape[4] if self._layout == TensorLayout.HND else self._buf.shape[3]

    @property
    def num_free_blocks(self) -> int:
        return len(self._free)

    def alloc_block(self) -> int:
        if not self._free:
            raise RuntimeError("KvPool exhausted: max_seq_len too small for this sequence")
        idx = self._free.pop()
        self._in_use[idx] = True
        return idx

    def free_block(self, idx: int) -> None:
        assert 0 <= idx < self.capacity
        assert self._in_use[idx], "double free of a KV page"
        self._in_use[idx] = False
        self._free.append(idx)


class KvCache:
    """Key-value cache (or its min/max metadata) of one sequence."""

    def __init__(self, num_layers, num_heads, head_dim, max_seq_len: int, page_size, dtype: torch.dtype, device,
                 layout: int = TensorLayout.NHD, shuffle_seed: Optional[int] = None, pool: Optional[KvPool] = None):
        if max_seq_len <= 0:
            raise ValueError("init_len must be non-negative")
        capacity = (max_seq_len + page_size - 1) // page_size
        if pool is None:
            self._pool = KvPool(num_layers, num_heads, head_dim, capacity, page_size, dtype, device, layout, shuffle_seed)
            self._reserve: Optional[List[int]] = None
        else:
            # EXTENSION (batched decode): several sequences share one pool; each takes its `capacity` pages up
            # front so that the order it will use them in is known (full_device_table) whatever the others do
            self._pool = pool
            self._reserve = [pool.alloc_block() for _ in range(capacity)][::-1]
        self._capacity = capacity
        self._indicies: List[int] = []
        self._seqlen = 0
        self._table = torch.empty(capacity, dtype=torch.int32, device=device)  # device mirror of _indicies
        self._table_len = 0

    @property
    def pool(self) -> KvPool:
        return self._pool

    @property
    def seqlen(self) -> int:
        return self._seqlen

    @property
    def last_page_len(self) -> int:
        return (self._seqlen - 1) % self._pool.block_len + 1

    @property
    def indicies(self) -> List[int]:  # spelling kept: reference attribute name (kv_cache.py:107-109)
        return self._indicies

    indices = indicies

    def buf_layer(self, layer_idx: int) -> torch.Tensor:
        return self._pool.layer(layer_idx)  # IndexError past the last layer

    def append_seq(self, seq_len: int) -> int:
        """Reserve room for ``seq_len`` more tokens; returns how many pages were added."""
        if seq_len <= 0:
            return 0
        S = self._pool.block_len
        need = (self._seqlen + seq_len + S - 1) // S - len(self._indicies)
        for _ in range(need):
            self._indicies.append(self._alloc())
        self._seqlen += seq_len
        return need

    @property
    def capacity_pages(self) -> int:
        return self._capacity

    def _alloc(self) -> int:
        if self._reserve is None:
            return self._pool.alloc_block()
        if not self._reserve:
            raise RuntimeError("KvPool exhausted: max_seq_len too small for this sequence")
        return self._reserve.pop()

    def device_table(self) -> torch.Tensor:
        """int32 device view of the page table, synchronised lazily (only the new tail is copied)."""
        n = len(self._indicies)
        if self._table_len < n:
            tail = torch.tensor(self._indicies[self._table_len:n], dtype=torch.int32)
            self._table[self._table_len:n].copy_(tail, non_blocking=True)
            self._table_len = n
        return self._table[:n]

    def full_device_table(self) -> torch.Tensor:
        """int32 device table of EVERY page this sequence will ever use, in allocation order: the pages in
        use followed by the free list in the order ``alloc_block`` will hand them out.  Valid because the
        pool belongs to this sequence alone (kv_cache.py:86-94 builds one pool per KvCache).  Used by the
        device-resident step state: page i of the sequence is ``table[i]`` before it is allocated."""
        future = list(reversed(self._pool._free if self._reserve is None else self._reserve))
        return torch.tensor(self._indicies + future, dtype=torch.int32, device=self._table.device)

    def release(self) -> None:
        self._seqlen = 0
        if self._reserve is None:
            # newest page first: the LIFO free list then hands the pages out again in the order they were used,
            # so the next request sees the same physical order (as the reserved branch below guarantees)
            for idx in reversed(self._indicies):
                self._pool.free_block(idx)
        else:  # pages stay reserved for this sequence's next request, to be used in the same order
            self._reserve.extend(reversed(self._indicies))
        self._indicies.clear()
        self._table_len = 0
