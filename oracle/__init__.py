"""CPU oracle for the Quest sparse-decode hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``quest_amd`` never does.  The arithmetic lives in
``quest_oracle.c`` (plain C, each function citing the reference file:line it
restates); this module is a numpy/ctypes face for it plus ``torch_ref`` (an
eager-PyTorch restatement used as the timed CPU baseline).

Parity status: pinned against fixtures generated from the reference's own
pure-torch oracles (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libquest_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile quest_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "quest_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libquest_oracle.so"])
    return _LIB_PATH


class _PagedC(ctypes.Structure):
    _fields_ = [
        ("num_heads", ctypes.c_uint32),
        ("page_size", ctypes.c_uint32),
        ("head_dim", ctypes.c_uint32),
        ("layout", ctypes.c_uint32),
        ("data", ctypes.c_void_p),
        ("indices", ctypes.c_void_p),
        ("n_pages", ctypes.c_int32),
        ("last_page_len", ctypes.c_uint32),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.qo_h2f.restype = ctypes.c_float
        _lib.qo_h2f.argtypes = [ctypes.c_uint16]
        _lib.qo_f2h.restype = ctypes.c_uint16
        _lib.qo_f2h.argtypes = [ctypes.c_float]
    return _lib


NHD, HND, NHD_ROT = 0, 1, 2  # 2: this build's row-rotated NHD (include/quest_hip.h QUEST_LAYOUT_NHD_ROT), same shape as NHD


@dataclass
class Paged:
    """One layer of a paged pool + the page table of one sequence.

    ``data``: float16 ``[max_pages, 2, S, H, D]`` (NHD) or ``[max_pages, 2, H, S, D]`` (HND),
    as quest/utils/kv_cache.py:20-23 allocates it per layer.
    """

    data: np.ndarray
    indices: np.ndarray  # int32 [n_pages]
    last_page_len: int
    layout: int = NHD

    def __post_init__(self):
        assert self.data.dtype == np.float16 and self.data.ndim == 5 and self.data.flags.c_contiguous
        self.indices = np.ascontiguousarray(self.indices, dtype=np.int32)

    @property
    def page_size(self) -> int:
        return self.data.shape[3] if self.layout == HND else self.data.shape[2]

    @property
    def num_heads(self) -> int:
        return self.data.shape[2] if self.layout == HND else self.data.shape[3]

    @property
    def head_dim(self) -> int:
        return self.data.shape[4]

    def c(self) -> _PagedC:
        return _PagedC(
            self.num_heads,
            self.page_size,
            self.head_dim,
            self.layout,
            self.data.ctypes.data,
            self.indices.ctypes.data,
            len(self.indices),
            self.last_page_len,
        )


def _f16(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float16)


def _p(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def append_decode(kv: Paged, meta: Paged, k, v) -> None:
    k, v = _f16(k), _f16(v)
    assert k.shape == (1, kv.num_heads, kv.head_dim) == v.shape
    ck, cm = kv.c(), meta.c()
    lib().qo_append_decode(ctypes.byref(ck), ctypes.byref(cm), _p(k), _p(v))


def append_prefill(kv: Paged, meta: Paged, k, v) -> None:
    k, v = _f16(k), _f16(v)
    assert k.shape == v.shape and k.shape[1:] == (kv.num_heads, kv.head_dim)
    ck, cm = kv.c(), meta.c()
    lib().qo_append_prefill(ctypes.byref(ck), ctypes.byref(cm), _p(k), _p(v), ctypes.c_int32(k.shape[0]))


def estimate(q, meta: Paged) -> np.ndarray:
    q = _f16(q)
    assert q.ndim == 3 and q.shape[0] == 1
    hq = q.shape[1]
    n = (len(meta.indices) - 1) * meta.page_size + meta.last_page_len - 1
    out = np.zeros((hq, n), dtype=np.float16)
    cm = meta.c()
    lib().qo_estimate(_p(q), ctypes.byref(cm), ctypes.c_uint32(hq), _p(out))
    return out


def topk(values, in_idx, k: int):
    values = _f16(values)
    in_idx = np.ascontiguousarray(in_idx, dtype=np.int32)
    rows, n = values.shape
    assert in_idx.shape == (rows, n)
    k = min(k, n)
    out_v = np.zeros((rows, k), dtype=np.float16)
    out_i = np.zeros((rows, k), dtype=np.int32)
    lib().qo_topk(_p(values), _p(in_idx), ctypes.c_uint32(rows), ctypes.c_uint32(n), ctypes.c_uint32(k),
                  _p(out_v), _p(out_i))
    return out_v, out_i


def sparse_attn(q, kv: Paged, idx, n_sel: int, last_page_idx: int, last_page_len: int):
    """Returns (o [1,Hq,D] f16, lse [Hq] f32 natural log)."""
    q = _f16(q)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    hq = q.shape[1]
    assert idx.ndim == 2 and idx.shape[0] == hq and idx.shape[1] >= n_sel
    out = np.zeros_like(q)
    lse = np.zeros((hq,), dtype=np.float32)
    ck = kv.c()
    lib().qo_sparse_attn(_p(q), ctypes.byref(ck), _p(idx), ctypes.c_uint32(idx.shape[1]),
                         ctypes.c_uint32(n_sel), ctypes.c_int32(last_page_idx),
                         ctypes.c_uint32(last_page_len), ctypes.c_uint32(hq), _p(out), _p(lse))
    return out, lse


def rope_in_place(x: np.ndarray, past_len: int, rope_scale: float = 1.0, rope_theta: float = 1e4) -> None:
    assert x.dtype == np.float16 and x.ndim == 3 and x.flags.c_contiguous
    n, h, d = x.shape
    lib().qo_rope(_p(x), ctypes.c_uint32(n), ctypes.c_uint32(h), ctypes.c_uint32(d),
                  ctypes.c_uint32(past_len), ctypes.c_float(rope_scale), ctypes.c_float(rope_theta))


def rms_norm(x, w, eps: float) -> np.ndarray:
    x, w = _f16(x), _f16(w)
    rows, cols = int(np.prod(x.shape[:-1])), x.shape[-1]
    out = np.zeros_like(x)
    lib().qo_rms_norm(_p(x), _p(w), ctypes.c_uint32(rows), ctypes.c_uint32(cols), ctypes.c_float(eps), _p(out))
    return out


def half_key(values: np.ndarray) -> np.ndarray:
    """Order-preserving uint16 key of fp16 bit patterns (same as qo_key)."""
    b = np.ascontiguousarray(values, dtype=np.float16).view(np.uint16)
    return np.where(b & 0x8000, ~b, b | 0x8000).astype(np.uint16)
