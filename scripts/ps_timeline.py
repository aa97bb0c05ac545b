#!/usr/bin/env python3
"""In-kernel timeline of the persistent n-token projection kernel (tuning): builds libquest_hip_pstl.so with
-DQUEST_PS_TIMELINE, runs o_proj / down_proj shaped residual launches at 8 tokens and prints the stamps (us from entry) of
(first | last workgroup) x (wave 0 | 15).    python scripts/ps_timeline.py [--build]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "quest_amd", "libquest_hip_pstl.so")
if "--build" in sys.argv:
    from quest_amd.build import build_variant
    print(build_variant(LIB, ["-DQUEST_PS_TIMELINE"]))
    sys.exit(0)
os.environ["QUEST_HIP_LIB"] = LIB
import torch
from quest_amd import _kernels
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
dbg = torch.zeros(4 * 16, dtype=torch.int64, device=dev)
os.environ["QUEST_PS_DEBUG_PTR"] = hex(dbg.data_ptr())
n = 8
for in_dim, out_dim in ((4096, 4096), (11008, 4096), (4096, 22016)):
    ws = [(torch.randn(out_dim, in_dim, generator=g, device=dev, dtype=torch.float16) * 0.02) for _ in range(6)]
    x = torch.randn(n, in_dim, generator=g, device=dev, dtype=torch.float16)
    h = torch.zeros(n, out_dim, device=dev, dtype=torch.float16)
    for w in ws:  # the last launch's stamps stay (weights cold in every launch: 6 sets rotate past the caches)
        _kernels.decode_gemv_residual_batched(x, w, h)
    torch.cuda.synchronize()
    t = dbg.cpu().view(4, 16).tolist()
    print(f"== residual launch in={in_dim} out={out_dim}, {n} tokens (us at 2.1 GHz)")
    for name, row in zip(("wg 0 wave 0", "wg 0 wave 15", "last wg wave 0", "last wg wave 15"), t):
        base = row[0]
        print(f"  {name:16s}", " ".join(f"{(v - base) / 2100.0:5.2f}" for v in row if v))
