#!/usr/bin/env python3
"""Time of the prefill attention op (quest.utils.prefill_forward -> csrc/prefill.hip, the MFMA flash kernel over the paged
cache) for a whole prompt and for a 2048-token chunk at the end of it, Llama-2-7B head shapes (and Llama-3.1-8B GQA with
--gqa; --head-dim 64 / 256 for the other two built head sizes); beside it torch's fused attention on CONTIGUOUS K/V (is_causal flash backend for the whole prompt; the masked
memory-efficient backend for the chunk -- what the op was built on until round 5), which does not pay for gathering the
pages.      python scripts/prefill_bench.py [--gqa] [--head-dim D] [--lens 4096,16384,32768]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quest_amd.utils as qu  # noqa: E402

dev = torch.device("cuda", 0)
Hq, D = 32, 128
if "--head-dim" in sys.argv:
    D = int(sys.argv[sys.argv.index("--head-dim") + 1])
Hkv = 8 if "--gqa" in sys.argv else 32
lens = [4096, 16384, 32768]
if "--lens" in sys.argv:
    lens = [int(x) for x in sys.argv[sys.argv.index("--lens") + 1].split(",")]


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        f()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def flops(n, kv, causal=True):
    """4 D flops per (query, visible key) pair and head."""
    pairs = n * kv - (n * (n - 1)) // 2 if causal else n * kv
    return 4.0 * D * Hq * pairs


sdpa = torch.nn.functional.scaled_dot_product_attention
for L in lens:
    ctl = qu.InferenceController(1, Hq, D, 16, 128, L + 64, torch.float16, dev, num_kv_heads=Hkv)
    g = torch.Generator(device=dev).manual_seed(L)
    k = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn_like(k)
    q = torch.randn(L, Hq, D, generator=g, device=dev, dtype=torch.float16)
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    qu.append_kv(k, v, ctl, 0)
    kh, vh = k.transpose(0, 1).unsqueeze(0), v.transpose(0, 1).unsqueeze(0)
    for name, qq in (("whole prompt", q), ("last 2048 rows (chunked)", q[-2048:])):
        n = qq.size(0)
        ms = timed(lambda: qu.prefill_forward(qq, ctl, 0))
        qh = qq.transpose(0, 1).unsqueeze(0)
        if n == L:
            ms_t = timed(lambda: sdpa(qh, kh, vh, is_causal=True, scale=1 / math.sqrt(D), enable_gqa=Hkv != Hq))
        else:
            mask = torch.arange(L, device=dev).unsqueeze(0) <= (L - n + torch.arange(n, device=dev)).unsqueeze(1)
            ms_t = timed(lambda: sdpa(qh, kh, vh, attn_mask=mask, scale=1 / math.sqrt(D), enable_gqa=Hkv != Hq))
        fl = flops(n, L)
        print(f"L={L} Hkv={Hkv} D={D} {name}: {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s   (torch SDPA on contiguous K/V: "
              f"{ms_t:.3f} ms = {fl / ms_t / 1e9:.0f} TFLOP/s)", flush=True)
    ctl.end_forward()
    del ctl, k, v, q, kh, vh
    torch.cuda.empty_cache()
