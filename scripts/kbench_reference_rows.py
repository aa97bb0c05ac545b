#!/usr/bin/env python3
"""The reference's own kernel benches, row for row, on the MI355X -- same shapes, same inputs, same byte accounting.

    python scripts/kbench_reference_rows.py [--out gpurun_out/kernel_sweep.json] [--reps 20]      (or: bench.py --kernel-sweep)

Rows (all fp16, page size 16, head_dim 128, 32 query = 32 kv heads, batch 1, NHD -- the reference benches' axes):

* attention kernel ALONE, random DISTINCT page ids per head (kernels/src/bench/bench_batch_decode.cu:36-63: per head a
  shuffled iota of the sequence's pages, the first budget-1 kept; the last page is the current one), through the
  handler (kernel + merge, like the reference's cooperative path :88-111):
    - seqlen 4096 x page budgets 64 / 128 / 256 / 512 (clamped to 256) = the four published `fig-kernel-bench` rows
      (RTX 6000 Ada);
    - seqlen 32768 x page budgets 256 / 640 / 896 = scripts/bench_kernels.sh:12-14;
  bytes as bench_batch_decode.cu:82-86: q + budget * 2 * Hkv * page * D * 2 + indptr + indices read, o written.
* top-k at the six LongBench (average length / 16, token budget / 16) pairs of scripts/bench_kernels.sh:7-24
  (bench_decode_select_k.cu: 32 rows of random fp16 scores, shuffled page ids); bytes = scores + ids read, selected
  values + ids written (the reference registers none).
* estimate at those six lengths (bench_max_possible.cu:40-73: shuffled metadata pages, N(0,1)-like pool bytes);
  bytes as :70-73: q + the whole metadata pool + indptr + indices read, scores written.

Timing: every row is a hipGraph of `pools` back-to-back launches, each on its OWN pool / score / index tensors (the set is
sized well past the 256 MiB Infinity Cache, the counterpart of nvbench's cold, L2-flushed runs), replayed `reps` times
between two HIP events on the launch stream.  The figure is the average per launch INCLUDING the dependent-launch
boundary (~1.5 us), i.e. an upper bound of the kernel's own duration; rocprofv3 --kernel-trace --stats of this command
gives the kernel-only averages (profiles/r03_kernel_sweep_*).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
H, D, PAGE = 32, 128, 16
LONGBENCH_LEN = (5819, 15370, 11984, 14101, 24723, 8154)      # scripts/bench_kernels.sh:7
LONGBENCH_BUDGET = (256, 512, 1024, 512, 4096, 512)           # :8 (tokens)
# published on RTX 6000 Ada: page budget -> (us, GB/s, % of its peak)
ADA_ROWS = {64: (35.370, 475.0, 49.48), 128: (63.551, 528.5, 55.05), 256: (116.135, 578.3, 60.23), 512: (117.629, 570.9, 59.47)}


def graph_time(fn, n, reps):
    """Average us per launch of fn(0..n-1) captured once and replayed `reps` times (HIP events on the launch stream)."""
    import torch

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(n):
            fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def n_pools_for(bytes_per_pool, floor_bytes=1 << 30, lo=4, hi=32):
    return max(lo, min(hi, -(-floor_bytes // max(1, bytes_per_pool))))


def row(name, us, rd, wr, **extra):
    gbs = (rd + wr) / (us * 1e-6) / 1e9
    r = {"row": name, "us": round(us, 3), "read_MiB": round(rd / 2**20, 3), "write_MiB": round(wr / 2**20, 4),
         "GBps": round(gbs, 1), "pct_of_8TBps": round(100 * gbs / HBM_PEAK_GBS, 2)}
    r.update(extra)
    return r


def attention_rows(reps, dev):
    import torch
    from quest_amd import _kernels

    out = []
    for seqlen, budgets in ((4096, (64, 128, 256, 512)), (32768, (256, 640, 896))):
        n_pages = -(-seqlen // PAGE)
        last_len = (seqlen - 1) % PAGE + 1
        pool_bytes = n_pages * 2 * PAGE * H * D * 2
        n_pools = n_pools_for(pool_bytes)
        g = torch.Generator(device=dev).manual_seed(seqlen)
        pools = [torch.empty(n_pages, 2, PAGE, H, D, dtype=torch.float16, device=dev).normal_(generator=g)
                 for _ in range(n_pools)]
        q = torch.randn(n_pools, 1, H, D, generator=g, device=dev, dtype=torch.float16)
        o = torch.empty_like(q)
        for budget in budgets:
            b = min(budget, n_pages)  # bench_batch_decode.cu:49 "adjust page_budget"
            # per head: the first b-1 of a shuffled iota over the pages but the last (bench_batch_decode.cu:53-63)
            idx = [torch.stack([torch.randperm(n_pages - 1, generator=g, device=dev)[:b - 1] for _ in range(H)]).int().contiguous()
                   for _ in range(n_pools)]
            indptr = torch.tensor([0, b - 1], dtype=torch.int32, device=dev)
            wrapper = _kernels.BatchDecodeWithPagedKVCachePyTorchWrapper(0)
            wrapper.begin_forward(indptr.cpu(), H, H, D, PAGE, torch.empty(0, dtype=torch.float16))

            def launch(i):
                wrapper.forward(q[i], o[i], pools[i], idx[i], indptr, last_len, n_pages - 1, 1.0, 1e4)

            us = graph_time(launch, n_pools, reps)
            wrapper.set_skip_merge(True)
            us_kernel = graph_time(launch, n_pools, reps)
            wrapper.set_skip_merge(False)
            ppc, chunks = wrapper.plan_info()
            wrapper.end_forward()
            rd = H * D * 2 + b * 2 * H * PAGE * D * 2 + 2 * 4 + H * (b - 1) * 4
            extra = {"op": "sparse_decode_attention", "seqlen": seqlen, "page_budget": budget, "pages_read": b,
                     "us_kernel_without_merge_launch": round(us_kernel, 3), "workgroups_per_head": chunks, "pools_rotated": n_pools}
            if seqlen == 4096:
                a = ADA_ROWS[budget]
                extra["published_rtx6000ada"] = {"us": a[0], "GBps": a[1], "pct_of_its_peak": a[2]}
            out.append(row(f"attention seqlen={seqlen} page_budget={budget}", us, rd, H * D * 2, **extra))
        del pools, q, o
        torch.cuda.empty_cache()
    return out


def topk_rows(reps, dev):
    import torch
    from quest_amd import _kernels

    out = []
    g = torch.Generator(device=dev).manual_seed(7)
    n_sets = 32
    for length, budget in zip(LONGBENCH_LEN, LONGBENCH_BUDGET):
        n, k = length // PAGE, budget // PAGE  # bench_kernels.sh:19-22
        vals = torch.randn(n_sets, H, n, generator=g, device=dev, dtype=torch.float16)
        ids = torch.stack([torch.stack([torch.randperm(n, generator=g, device=dev) for _ in range(H)]) for _ in range(n_sets)]).int()
        d_out = torch.empty(n_sets, H, k, dtype=torch.float16, device=dev)
        i_out = torch.empty(n_sets, H, k, dtype=torch.int32, device=dev)
        us = graph_time(lambda i: _kernels.topk_filtering(vals[i], ids[i], d_out[i], i_out[i], None, k), n_sets, reps)
        out.append(row(f"top-k seq_len={n} k={k}", us, H * n * 6, H * k * 6, op="topk_filtering",
                       longbench_avg_tokens=length, token_budget=budget, published="5-10 us for seqlen < 128K (RTX 4090, paper 4.3.1)"))
    return out


def estimate_rows(reps, dev):
    import torch
    from quest_amd import _kernels

    out = []
    g = torch.Generator(device=dev).manual_seed(9)
    for length in LONGBENCH_LEN:
        n_pages = -(-length // PAGE)
        n_chunks = -(-n_pages // PAGE)                      # bench_max_possible.cu:44-47
        last_chunk_len = (n_pages - 1) % PAGE + 1
        pool_bytes = n_chunks * 2 * PAGE * H * D * 2
        n_pools = n_pools_for(pool_bytes, hi=64)
        pools = [torch.empty(n_chunks, 2, PAGE, H, D, dtype=torch.float16, device=dev).normal_(generator=g) for _ in range(n_pools)]
        table = torch.randperm(n_chunks, generator=g, device=dev).int()  # shuffled chunk pages (:50)
        indptr = torch.tensor([0, n_chunks], dtype=torch.int32, device=dev)
        q = torch.randn(n_pools, 1, H, D, generator=g, device=dev, dtype=torch.float16)
        o = torch.empty(n_pools, H, n_pages - 1, dtype=torch.float16, device=dev)
        last_idx = int(table[-1])
        us = graph_time(lambda i: _kernels.estimate_attn_score(q[i], o[i], pools[i], table, indptr, last_chunk_len, last_idx, 0),
                        n_pools, reps)
        rd = H * D * 2 + pool_bytes + 2 * 4 + n_chunks * 4
        out.append(row(f"estimate seqlen={length}", us, rd, H * (n_pages - 1) * 2, op="estimate_attn_score", pages=n_pages,
                       pools_rotated=n_pools))
        del pools, q, o
        torch.cuda.empty_cache()
    return out


def run(reps=20, out_path=None):
    import torch

    assert torch.cuda.is_available(), "the kernel sweep needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rows = attention_rows(reps, dev) + topk_rows(reps, dev) + estimate_rows(reps, dev)
    res = {"what": "reference kernel benches row for row (bench_batch_decode.cu, scripts/bench_kernels.sh) on MI355X",
           "timing": "hipGraph of back-to-back launches over rotating pools (past the Infinity Cache), HIP events; us per "
                     "launch incl. the dependent-launch boundary", "peak_GBps": HBM_PEAK_GBS, "rows": rows}
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        json.dump(res, open(out_path, "w"), indent=1)
    return res


def markdown(res):
    lines = ["| row | us / launch | MiB read | GB/s | % of 8 TB/s | reference (published) |", "|---|---|---|---|---|---|"]
    for r in res["rows"]:
        pub = r.get("published_rtx6000ada")
        ref = f"{pub['us']} us, {pub['GBps']} GB/s, {pub['pct_of_its_peak']} % (RTX 6000 Ada)" if pub else r.get("published", "--")
        us = f"{r['us']}" + (f" (kernel without the merge launch: {r['us_kernel_without_merge_launch']})" if "us_kernel_without_merge_launch" in r else "")
        lines.append(f"| {r['row']} | {us} | {r['read_MiB']} | {r['GBps']} | {r['pct_of_8TBps']} | {ref} |")
    return "\n".join(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernel_sweep.json"))
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    res = run(a.reps, a.out)
    print(markdown(res))
    print(json.dumps({"kernel_sweep": a.out, "rows": len(res["rows"])}))
