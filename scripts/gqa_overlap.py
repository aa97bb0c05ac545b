#!/usr/bin/env python3
"""How much do the query heads of a GQA group overlap in the pages they select?  (SURVEY 7 hard part 5 / VERDICT r1
item 5: "measure overlap first".)  cfg-5 shapes, synthetic N(0,1) data, the real kernels' selection."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quest_amd.utils as qu

dev = torch.device("cuda:0")
L, Hq, Hkv, D, B = 32768, 32, 8, 128, 128
g = torch.Generator(device=dev).manual_seed(0)
ctl = qu.InferenceController(1, Hq, D, 16, B, L + 64, torch.float16, dev, num_kv_heads=Hkv, shuffle_seed=1)
k = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
v = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
ctl.prepare_metadata(L)
ctl.begin_forward(L)
qu.append_kv(k, v, ctl, 0)
ctl.end_forward()
ctl.prepare_metadata(1)
ctl.begin_forward(1)
unions = []
for trial in range(8):
    q = torch.randn(1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    est = qu.decode_estimate(q, ctl, 0)
    qu.decode_topk(est, ctl)
    sel = ctl.topk_dindices_buffer.view(Hkv, Hq // Hkv, B - 1)
    for h in range(Hkv):
        unions.append(torch.unique(sel[h]).numel())
ctl.end_forward()
G = Hq // Hkv
u = torch.tensor(unions, dtype=torch.float32)
n = L // 16 - 1
p = (B - 1) / n
print(f"group of {G} heads x {B - 1} selected pages = {G * (B - 1)} gathers; distinct pages per group: "
      f"mean {u.mean():.1f} (min {u.min():.0f}, max {u.max():.0f}) = {100 * u.mean() / (G * (B - 1)):.1f} % "
      f"(independent selections would give {n * (1 - (1 - p) ** G):.1f})")
