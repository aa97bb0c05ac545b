"""Deterministic, library-independent synthetic inputs (TEST INFRASTRUCTURE ONLY).

``normal_f16(seed, shape)`` is a counter-based generator built from integer
hashing and exact float arithmetic only (no libm, no numpy RNG), so the golden
maker in the build container and the tests on the GPU box regenerate identical
bits.  Values are an Irwin-Hall(4) approximation of N(0,1) -- the reference's
tests and benches use N(0,1) inputs (test_approx_attention.py:134-137,
kernels/src/include/cpu_utils.h:64-71).
"""
from __future__ import annotations

import numpy as np

from . import NHD, HND, Paged, append_prefill

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def normal_f16(seed: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
        r = _splitmix64(ctr)
    s = np.zeros(n, dtype=np.int64)
    for i in range(4):
        s += ((r >> np.uint64(16 * i)) & np.uint64(0xFFFF)).astype(np.int64)
    # sum of 4 U{0..65535}: mean 131070, variance 4*(65536^2-1)/12
    x = (s - 131070).astype(np.float64) / 37837.22  # exact int -> double, one IEEE division
    return x.astype(np.float32).astype(np.float16).reshape(shape)


def page_permutation(seed: int, capacity: int) -> np.ndarray:
    """A seeded physical-page order (reference pools hand out arbitrary pages,
    quest/utils/kv_cache.py:55-57)."""
    with np.errstate(over="ignore"):
        keys = _splitmix64(np.arange(capacity, dtype=np.uint64) + (np.uint64(seed + 7919) << np.uint64(32)))
    return np.argsort(keys, kind="stable").astype(np.int32)


def build_sequence(k: np.ndarray, v: np.ndarray, page_size: int = 16, layout: int = NHD,
                   perm_seed: int | None = None, slack_pages: int = 3):
    """Lay out k, v ``[L, H, D]`` into a KV pool + metadata pool through the oracle's
    append-prefill, the way InferenceController does (controller.py:19-37).

    Returns ``(kv: Paged, meta: Paged)`` for a single layer.
    """
    L, H, D = k.shape
    n_pages = (L + page_size - 1) // page_size
    n_meta = (n_pages + page_size - 1) // page_size
    cap, mcap = n_pages + slack_pages, n_meta + slack_pages
    shape = (lambda c: (c, 2, H, page_size, D)) if layout == 1 else (lambda c: (c, 2, page_size, H, D))  # 1 = HND
    kv_data = np.zeros(shape(cap), dtype=np.float16)
    meta_data = np.zeros(shape(mcap), dtype=np.float16)
    if perm_seed is None:
        kv_idx = np.arange(n_pages, dtype=np.int32)
        meta_idx = np.arange(n_meta, dtype=np.int32)
    else:
        kv_idx = page_permutation(perm_seed, cap)[:n_pages]
        meta_idx = page_permutation(perm_seed + 1, mcap)[:n_meta]
    kv = Paged(kv_data, kv_idx, (L - 1) % page_size + 1, layout)
    meta = Paged(meta_data, meta_idx, (n_pages - 1) % page_size + 1, layout)
    append_prefill(kv, meta, k, v)
    return kv, meta
