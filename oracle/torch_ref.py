"""Eager-PyTorch restatement of the reference's CPU-runnable oracles (TEST INFRASTRUCTURE ONLY).

This is the "port" that bench.py times as ``cpu_baseline`` and that tests pin
against fixtures produced by the reference's own functions:

* ``page_scores``      <- quest/tests/test_estimate.py:17-75 (``_ref_cpu_estimate``)
* ``sparse_decode``    <- quest/tests/test_approx_attention.py:17-110 (``_ref_self_approx_attention``)
* ``dense_decode``     <- quest/tests/test_decode_attention.py:17-44 (``_ref_self_attention``)
* ``prefill_attention`` <- quest/tests/test_prefill_attention.py:17-44 (``_ref_self_attention``: bottom-right causal mask
  ``tril(diagonal=kv_len - qo_len)``), in fp32 or fp64 so that it can serve as the checker at a tighter bar than the
  reference's own fp16 arithmetic; pinned against fixtures of the reference function (tests/golden/prefill_ref_golden.npz)

Like the reference's eager path it materialises the full ``[H, 1, L]`` logits and
masks them, so its cost does not shrink with the page budget.  Layout of all
inputs is NHD: q ``[1, H, D]``, k/v ``[L, H, D]``.
"""
from __future__ import annotations

import math

import torch


def _paged_extrema(k_hld: torch.Tensor, page_size: int):
    """Per-page element-wise (max, min) of keys; k_hld is ``[H, L, D]``."""
    H, L, D = k_hld.shape
    n_pages = -(-L // page_size)
    pad = n_pages * page_size - L
    lo = torch.finfo(k_hld.dtype).min
    hi = torch.finfo(k_hld.dtype).max
    k_for_max = torch.nn.functional.pad(k_hld, (0, 0, 0, pad), value=lo)
    k_for_min = torch.nn.functional.pad(k_hld, (0, 0, 0, pad), value=hi)
    kmax = k_for_max.view(H, n_pages, page_size, D).amax(dim=2)
    kmin = k_for_min.view(H, n_pages, page_size, D).amin(dim=2)
    return kmax, kmin


def page_scores_f32(q: torch.Tensor, k: torch.Tensor, page_size: int) -> torch.Tensor:
    """Upper-bound criticality score of every page except the current (last) one, fp32 ``[H, N-1]``.

    score[h, p] = sum_d max(q[h,d] * Kmax[h,p,d], q[h,d] * Kmin[h,p,d])
    (kernels/include/decode/decode_attn.cuh:152-156; the reference oracle writes the same
    quantity with the sign trick, test_estimate.py:39-71).
    """
    qh = q.transpose(0, 1).float()  # [H, 1, D]
    kmax, kmin = _paged_extrema(k.transpose(0, 1), page_size)
    s = torch.maximum(qh * kmax.float(), qh * kmin.float()).sum(dim=-1)  # [H, N]
    return s[:, :-1].contiguous()


def page_scores(q: torch.Tensor, k: torch.Tensor, page_size: int) -> torch.Tensor:
    return page_scores_f32(q, k, page_size).to(q.dtype)


def dense_decode(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Full-KV single-token attention, ``[1, H, D]``."""
    D = q.shape[-1]
    qh, kh, vh = q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1)
    logits = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(D)  # [H, 1, L] in q.dtype
    probs = torch.softmax(logits, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(probs, vh).transpose(0, 1)


def sparse_decode(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, page_size: int, page_budget: int):
    """Quest decode attention: top-(budget-1) pages by ``page_scores`` + the last page.

    Returns ``(o [1,H,D], topk_pages [H, budget-1] int32 or None)``; page ids are logical
    (position in the sequence).  ``page_budget`` is in pages and includes the last page
    (quest/utils/controller.py:14).
    """
    L, H, D = k.shape
    n_pages = -(-L // page_size)
    if n_pages <= page_budget:
        return dense_decode(q, k, v), None
    scores = page_scores_f32(q, k, page_size)  # fp32, like the reference oracle's internal top-k
    sel = scores.topk(page_budget - 1, dim=-1).indices  # [H, B-1]
    pages = torch.cat([sel, torch.full((H, 1), n_pages - 1, dtype=sel.dtype)], dim=1)
    tok = (pages.unsqueeze(-1) * page_size + torch.arange(page_size)).reshape(H, -1)
    keep = torch.zeros(H, n_pages * page_size, dtype=torch.bool)
    keep.scatter_(1, tok, True)
    keep = keep[:, :L]

    qh, kh, vh = q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1)
    logits = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(D)  # full [H, 1, L]
    logits = logits.masked_fill(~keep.unsqueeze(1), torch.finfo(logits.dtype).min)
    probs = torch.softmax(logits, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(probs, vh).transpose(0, 1)
    return o, sel.to(torch.int32)


def prefill_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = True,
                      dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Attention of ``qo_len`` query rows ``[qo_len, Hq, D]`` over ``kv_len >= qo_len`` cached tokens ``[kv_len, Hkv, D]``;
    row i sees keys ``0 .. kv_len - qo_len + i`` when ``causal`` (test_prefill_attention.py:38-39).  GQA: query head h
    reads kv head ``h // (Hq // Hkv)`` (evaluation/quest_attention.py:139-184's repeat_kv).  Returns ``[qo_len, Hq, D]``
    in ``dtype``."""
    qo_len, Hq, D = q.shape
    kv_len, Hkv, _ = k.shape
    assert (kv_len >= qo_len or not causal) and Hq % Hkv == 0
    qh = q.to(dtype).transpose(0, 1)
    kh = k.to(dtype).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
    vh = v.to(dtype).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
    logits = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(D)  # [Hq, qo_len, kv_len]
    if causal:
        keep = torch.ones(qo_len, kv_len, dtype=torch.bool, device=q.device).tril(diagonal=kv_len - qo_len)
        logits = logits.masked_fill(~keep, float("-inf"))
    return torch.matmul(torch.softmax(logits, dim=-1), vh).transpose(0, 1)
