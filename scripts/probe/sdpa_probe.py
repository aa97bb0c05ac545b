"""Which SDPA backends this PyTorch-ROCm build runs for the prefill shapes (is_causal, no mask), and how fast."""
import time, torch
from torch.nn.attention import SDPBackend, sdpa_kernel
dev = "cuda:0"
H, D = 32, 128
for L in (4096, 16384):
    q = torch.randn(1, H, L, D, device=dev, dtype=torch.float16); k = torch.randn_like(q); v = torch.randn_like(q)
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        if name == "math" and L > 4096:
            continue
        try:
            with sdpa_kernel(be):
                o = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
            flops = 4 * L * L * D * H / 2
            print(f"L={L} {name}: {dt*1e3:.2f} ms, {flops/dt/1e12:.1f} TFLOP/s")
        except Exception as e:
            print(f"L={L} {name}: FAILED {type(e).__name__}: {str(e)[:120]}")

# chunked prefill: n query rows at the end of kv_len keys (bottom-right causal) through a boolean / additive mask
for n, kv in ((2048, 16384), (2048, 16383)):
    q = torch.randn(1, H, n, D, device=dev, dtype=torch.float16)
    k = torch.randn(1, H, kv, D, device=dev, dtype=torch.float16); v = torch.randn_like(k)
    cols = torch.arange(kv, device=dev); limit = (kv - n + torch.arange(n, device=dev)).unsqueeze(1)
    mb = cols.unsqueeze(0) <= limit
    ref = None
    for name, be, mask in (("efficient bool", SDPBackend.EFFICIENT_ATTENTION, mb),
                           ("efficient additive", SDPBackend.EFFICIENT_ATTENTION, torch.zeros(n, kv, device=dev, dtype=torch.float16).masked_fill(~mb, float("-inf"))),
                           ("flash bool", SDPBackend.FLASH_ATTENTION, mb), ("math bool", SDPBackend.MATH, mb)):
        try:
            with sdpa_kernel(be):
                o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            if ref is None:
                ref = o
            print(f"n={n} kv={kv} {name}: {dt*1e3:.2f} ms  max diff vs first {float((o.float()-ref.float()).abs().max()):.2e}")
        except Exception as e:
            print(f"n={n} kv={kv} {name}: FAILED {type(e).__name__}: {str(e)[:100]}")
# GQA without materialising the repeated K/V
try:
    q = torch.randn(1, 32, 4096, D, device=dev, dtype=torch.float16); k = torch.randn(1, 8, 4096, D, device=dev, dtype=torch.float16); v = torch.randn_like(k)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
    o2 = torch.nn.functional.scaled_dot_product_attention(q, k.repeat_interleave(4, 1), v.repeat_interleave(4, 1), is_causal=True)
    print("enable_gqa ok, max diff", float((o.float() - o2.float()).abs().max()))
except Exception as e:
    print("enable_gqa FAILED", type(e).__name__, str(e)[:100])
