#!/usr/bin/env python3
"""Per-launch time of the fused decoder-layer GEMV launches at Llama-2-7B shapes (rotating weight sets past the Infinity
Cache, hipGraph + HIP events).  QUEST_GEMV_CFG="RW,U" forces a workgroup shape.   python scripts/gemv_bench.py
With --tokens N (2..16): the N-token launches (csrc/decode_layer.hip skinny_kernel; QUEST_SKINNY_CFG="NW,U") next to
torch.nn.functional.linear on the same [N, in] inputs (hipBLASLt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quest_amd import _kernels
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from kbench_reference_rows import graph_time

dev = torch.device("cuda", 0)
H, I, NSET = 4096, 11008, 8
g = torch.Generator(device=dev).manual_seed(0)
def w(o, i): return [(torch.randn(o, i, generator=g, device=dev, dtype=torch.float16) * 0.02) for _ in range(NSET)]
x = torch.randn(H, generator=g, device=dev, dtype=torch.float16)
xi = torch.randn(I, generator=g, device=dev, dtype=torch.float16)
gamma = torch.ones(H, device=dev, dtype=torch.float16)
h = torch.zeros(H, device=dev, dtype=torch.float16)
act = torch.empty(I, device=dev, dtype=torch.float16)
q = torch.empty(1, 32, 128, device=dev, dtype=torch.float16); k = torch.empty_like(q); v = torch.empty_like(q)
state = torch.tensor([1000, 1, 1, 0, 1, 1, 0, 0], dtype=torch.int32, device=dev)
wo, wd, wg, wu, wq, wk, wv = w(H, H), w(H, I), w(I, H), w(I, H), w(H, H), w(H, H), w(H, H)
rows = []
rows.append(("o_proj+res    32 MiB", graph_time(lambda i: _kernels.decode_gemv_residual(x, wo[i], h), NSET, 20), 32))
rows.append(("down+res      86 MiB", graph_time(lambda i: _kernels.decode_gemv_residual(xi, wd[i], h), NSET, 20), 86))
rows.append(("gate/up+silu 172 MiB", graph_time(lambda i: _kernels.decode_mlp_gate_up(x, gamma, 1e-5, wg[i], wu[i], act), NSET, 20), 172))
rows.append(("qkv+rope      96 MiB", graph_time(lambda i: _kernels.decode_qkv_rope(x, gamma, 1e-5, wq[i], wk[i], wv[i], q, k, v, 128, 1.0, 1e4, state), NSET, 20), 96))
print(os.environ.get("QUEST_GEMV_CFG", "default"), " | ".join(f"{n}: {t:6.2f} us {m * 1.048576 / t:5.2f} TB/s" for n, t, m in rows))

if "--tokens" in sys.argv:
    n = int(sys.argv[sys.argv.index("--tokens") + 1])
    X = torch.randn(n, H, generator=g, device=dev, dtype=torch.float16)
    XI = torch.randn(n, I, generator=g, device=dev, dtype=torch.float16)
    HB = torch.zeros(n, H, device=dev, dtype=torch.float16)
    ACT = torch.empty(n, I, device=dev, dtype=torch.float16)
    Q = torch.empty(n, 32, 128, device=dev, dtype=torch.float16); K = torch.empty_like(Q); V = torch.empty_like(Q)
    ST = torch.tensor([[1000 + 17 * i, 1, 1, 0, 1, 1, 0, 0] for i in range(n)], dtype=torch.int32, device=dev)
    rows = []
    rows.append(("o_proj+res    32 MiB", graph_time(lambda i: _kernels.decode_gemv_residual_batched(X, wo[i], HB), NSET, 20), 32))
    rows.append(("down+res      86 MiB", graph_time(lambda i: _kernels.decode_gemv_residual_batched(XI, wd[i], HB), NSET, 20), 86))
    rows.append(("gate/up+silu 172 MiB", graph_time(lambda i: _kernels.decode_mlp_gate_up_batched(X, gamma, 1e-5, wg[i], wu[i], ACT), NSET, 20), 172))
    rows.append(("qkv+rope      96 MiB", graph_time(lambda i: _kernels.decode_qkv_rope_batched(X, gamma, 1e-5, wq[i], wk[i], wv[i], Q, K, V, 128, 1.0, 1e4, ST), NSET, 20), 96))
    print(f"{n} tokens", os.environ.get("QUEST_SKINNY_CFG", "default"), " | ".join(f"{nm}: {t:6.2f} us {m * 1.048576 / t:5.2f} TB/s" for nm, t, m in rows))
    lin = torch.nn.functional.linear
    rows = []
    rows.append(("linear o_proj  32 MiB", graph_time(lambda i: lin(X, wo[i]), NSET, 20), 32))
    rows.append(("linear down    86 MiB", graph_time(lambda i: lin(XI, wd[i]), NSET, 20), 86))
    rows.append(("linear gate    86 MiB", graph_time(lambda i: lin(X, wg[i]), NSET, 20), 86))
    print(f"{n} tokens torch ", " | ".join(f"{nm}: {t:6.2f} us {m * 1.048576 / t:5.2f} TB/s" for nm, t, m in rows))
