"""Per-op timing of the batched state-driven launches (8 sequences, cfg-3 shapes): append+estimate,
top-k+attention(+merge), and the whole layer, for a sweep of pages-per-chunk."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=8)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--ppc", type=int, nargs="*", default=[0, 16, 32, 64, 128])
    x = ap.parse_args()
    sys.argv = [sys.argv[0]]
    a = bench.parse()
    a.layers, a.steps, a.warmup = x.layers, 8, 2
    dev = torch.device("cuda:0")
    from quest_amd import _kernels
    import quest_amd.utils as qu

    w = bench.BatchedWorkload(a, dev, x.seqs)
    b = w.ctl
    bpl = bench.bytes_per_layer(a)
    max_n = b.max_pages - 1

    def ae(l):
        _kernels.append_estimate_batched(w.k1[l], w.v1[l], b.kv_layer(l), b.kv_tables, w.q[l], w.scores,
                                         b.metadata_layer(l), b.meta_tables, b.step_states, max_n, b.layout)

    def ts(l):
        b._decode_handler.forward_fused_topk_batched(w.q[l], w.o[l], b.kv_layer(l), b.kv_tables, w.scores,
                                                     b.step_states, max_n)

    def both(l):
        ae(l)
        ts(l)

    qu.step_advance_batched(b)
    t_ae = bench.time_kernel_loop(ae, a.layers, 10)
    print(f"append+estimate batched: {t_ae:.1f} us per launch = {t_ae / x.seqs:.2f} us/seq, "
          f"{x.seqs * (bpl['append'] + bpl['estimate']) / t_ae / 1e3:.0f} GB/s")
    for ppc in x.ppc:
        b._decode_handler.set_pages_per_chunk(ppc)
        try:
            b.begin_graph_decode()
            t_ts = bench.time_kernel_loop(ts, a.layers, 10)
            t_l = bench.time_kernel_loop(both, a.layers, 10)
        except Exception as e:  # unsupported split
            print(f"ppc {ppc}: {e}")
            continue
        print(f"ppc {ppc:3d} plan {b._decode_handler.plan_info()}: topk+attn(+merge) {t_ts:.1f} us = {t_ts / x.seqs:.2f} us/seq, "
              f"{x.seqs * bpl['attn'] / t_ts / 1e3:.0f} GB/s; layer {t_l:.1f} us = {t_l / x.seqs:.2f} us/seq")


if __name__ == "__main__":
    main()
