#!/usr/bin/env python3
"""Time of the prefill attention op (quest.utils.prefill_forward: torch SDPA over the paged cache) for a whole prompt and
for a 2048-token chunk at the end of it; Llama-2-7B head shapes.   python scripts/prefill_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quest_amd.utils as qu
dev = torch.device("cuda", 0)
H, D = 32, 128
for L in (4096, 32768):
    ctl = qu.InferenceController(1, H, D, 16, 128, L + 64, torch.float16, dev)
    g = torch.Generator(device=dev).manual_seed(L)
    k = torch.randn(L, H, D, generator=g, device=dev, dtype=torch.float16); v = torch.randn_like(k); q = torch.randn_like(k)
    ctl.prepare_metadata(L); ctl.begin_forward(L)
    qu.append_kv(k, v, ctl, 0)
    for name, qq in (("whole prompt", q), ("last 2048 rows (chunked)", q[-2048:])):
        o = qu.prefill_forward(qq, ctl, 0); torch.cuda.synchronize()
        t0 = time.perf_counter(); o = qu.prefill_forward(qq, ctl, 0); torch.cuda.synchronize()
        print(f"L={L} {name}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    ctl.end_forward()
