#!/usr/bin/env python3
"""Developer micro-bench: per-operator time at cfg 3 shapes from hipGraph replays of back-to-back
launches that rotate over `--layers` distinct KV/metadata pools (so nothing is cache-resident).
Times are per launch INCLUDING the dependent-kernel boundary, i.e. what the op costs inside a step.

    python scripts/kbench.py [--layers 8 --ppc 0 4 8 16 --layout NHD]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def graph_time(fn, layers, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for l in range(layers):
            fn(l)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in range(layers):
            fn(l)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--ppc", type=int, nargs="*", default=[0])
    ap.add_argument("--layout", default="NHD")
    ap.add_argument("--seqlen", type=int, default=32768)
    ap.add_argument("--token-budget", type=int, default=2048)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--dense", action="store_true")
    a0 = ap.parse_args()
    sys.argv = [sys.argv[0]]
    a = bench.parse()
    a.mode = "graph-static"  # per-op timing uses the by-value (reference-signature) entry points
    a.layers, a.layout, a.seqlen, a.token_budget, a.heads, a.kv_heads = (a0.layers, a0.layout, a0.seqlen,
                                                                        a0.token_budget, a0.heads, a0.kv_heads)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.Workload(a, dev)
    qu, ctl = w.qu, w.ctl
    bpl = bench.bytes_per_layer(a)
    L = a.layers
    print(f"# layout={a.layout} layers={L} seqlen={a.seqlen} budget_pages={w.page_budget} Hq={a.heads} Hkv={a.kv_heads}")
    for ppc in a0.ppc:
        ctl._decode_handler.set_pages_per_chunk(ppc)
        ctl.set_page_budget(w.page_budget)
        ctl.begin_forward(1)
        plan = ctl._decode_handler.plan_info()
        est = [qu.decode_estimate(w.q[l], ctl, l) for l in range(L)]
        qu.decode_topk(est[0], ctl)
        idx = ctl.topk_dindices_buffer
        t_s = graph_time(lambda l: qu.decode_sparse_attn(w.q[l], ctl, l, idx), L)
        print(f"sparse_attn(+merge) ppc={plan[0]:4d} chunks={plan[1]:4d}: {t_s:7.2f} us  {bpl['attn'] / t_s / 1e3:7.1f} GB/s")
        t_ts = graph_time(lambda l: qu.decode_topk_sparse_attn(w.q[l], est[l], ctl, l, write_topk=False), L)
        print(f"topk+sparse_attn(+merge) ppc={plan[0]:4d} chunks={plan[1]:4d}: {t_ts:7.2f} us")
        if ppc == a0.ppc[0]:
            t_e = graph_time(lambda l: qu.decode_estimate(w.q[l], ctl, l), L)
            print(f"estimate: {t_e:7.2f} us  {bpl['estimate'] / t_e / 1e3:7.1f} GB/s")
            t_t = graph_time(lambda l: qu.decode_topk(est[l], ctl), L)
            print(f"topk:     {t_t:7.2f} us")
            t_a = graph_time(lambda l: qu.append_kv(w.k1[l], w.v1[l], ctl, l), L)
            print(f"append:   {t_a:7.2f} us")

            def chain(l):
                qu.append_kv(w.k1[l], w.v1[l], ctl, l)
                e = qu.decode_estimate(w.q[l], ctl, l)
                qu.decode_topk(e, ctl)
                qu.decode_sparse_attn(w.q[l], ctl, l, ctl.topk_dindices_buffer)

            t_c = graph_time(chain, L)
            print(f"chain:    {t_c:7.2f} us  {bpl['chain'] / t_c / 1e3:7.1f} GB/s")
            t_ae = graph_time(lambda l: qu.decode_append_estimate(w.q[l], w.k1[l], w.v1[l], ctl, l), L)
            print(f"append+estimate (1 launch): {t_ae:7.2f} us")
            t_ts = graph_time(lambda l: qu.decode_topk_sparse_attn(w.q[l], est[l], ctl, l, write_topk=False), L)
            print(f"topk+sparse_attn(+merge):   {t_ts:7.2f} us")

            def fchain(l):
                e = qu.decode_append_estimate(w.q[l], w.k1[l], w.v1[l], ctl, l)
                qu.decode_topk_sparse_attn(w.q[l], e, ctl, l, write_topk=False)

            t_fc = graph_time(fchain, L)
            print(f"fused chain: {t_fc:7.2f} us  {bpl['chain'] / t_fc / 1e3:7.1f} GB/s")
        ctl.end_forward()
    if a0.dense:
        ctl._decode_handler.set_pages_per_chunk(0)
        ctl.set_page_budget(1 << 20)
        ctl.begin_forward(1, updateTensor=False)
        t_d = graph_time(lambda l: qu.decode_sparse_attn(w.q[l], ctl, l, ctl.kv_indices_without_last), L, reps=5)
        print(f"dense:    {t_d:7.2f} us  {bpl['dense'] / t_d / 1e3:7.1f} GB/s plan={ctl._decode_handler.plan_info()}")
        ctl.end_forward()


if __name__ == "__main__":
    main()
