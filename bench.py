#!/usr/bin/env python3
"""Headline bench: Quest self-attention decode (append -> estimate -> top-k -> sparse attention)
at BASELINE.json's metric point -- Yarn-Llama-2-7B shapes (32 layers, 32 heads, D=128, fp16),
seqlen 32768, token budget 2048 = 128 pages of 16 -- on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

A "step" is one decode token of one sequence per GPU: the four-operator chain run for every
layer of the model against that layer's own KV / metadata pools (so consecutive kernels touch
different memory and the 256 MiB Infinity Cache cannot hold the working set).  Everything goes
through the C ABI of libquest_hip.so; tokens/s is attention-only (no weights offline).
Sequences are independent, so GPUs never exchange data inside a step; with N > 1 each step ends
with one RCCL all_gather of the sampled token ids (weak scaling: one sequence per GPU).

One JSON line on stdout (rank 0).  Extra objects: "roofline" (dominant kernel: sparse paged decode
attention, algorithmic bytes / HIP-event launch time) and "cpu_baseline" (oracle/torch_ref eager
port timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=32768)
    ap.add_argument("--token-budget", type=int, default=2048)
    ap.add_argument("--page-size", type=int, default=16)
    ap.add_argument("--layout", choices=["NHD", "HND"], default="NHD")
    ap.add_argument("--mode", choices=["graph", "graph-static", "eager"], default="graph",
                    help="graph: ONE hipGraph of the whole step, replayed while the sequence grows by a token per "
                         "step (device-resident step state); graph-static: the same graph without the state "
                         "(re-decodes the same position); eager: Python op by op")
    ap.add_argument("--skip-layers", type=int, default=0,
                    help="run the first n layers with full KV like quest/models/llama.py:428-439 (default 0)")
    ap.add_argument("--unfused", action="store_true",
                    help="issue the reference's five launches per layer instead of the fused append+estimate and "
                         "top-k+attention launches (same results)")
    ap.add_argument("--seqs-per-gpu", type=int, default=1,
                    help="independent sequences per GPU (BASELINE configs[4] uses 8); each runs its chain on its "
                         "own HIP stream inside the step graph so their latency phases overlap")
    ap.add_argument("--multi-seq-mode", choices=["batched", "streams"], default="batched",
                    help="with --seqs-per-gpu > 1: batched = ONE launch per op for all sequences over a shared pool "
                         "(grid.z = sequence, quest_*_batched); streams = one chain per sequence on its own stream")
    ap.add_argument("--pages-per-chunk", type=int, default=0, help="override the decode planner (tuning)")
    ap.add_argument("--no-multi-seq", action="store_true",
                    help="skip the side measurement with 8 sequences per GPU (BASELINE configs[4]'s per-GPU load)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--cpu-sample-s", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic K/V/q (SURVEY 8d: seeds 0, 1, 2)")
    return ap.parse_args()


class Workload:
    """One sequence: controller + filled pools + per-layer decode inputs."""

    def __init__(self, a, dev, seq_id=0):
        import quest_amd.utils as qu

        self.qu = qu
        self.a = a
        self.dev = dev
        self.page_budget = a.token_budget // a.page_size
        L = a.seqlen
        self.ctl = qu.InferenceController(a.layers, a.heads, a.head_dim, a.page_size, self.page_budget,
                                          L + 2 * a.steps + a.warmup + 4 * a.page_size, torch.float16, dev,  # 2x: the N > 1 no-gather rerun
                                          num_kv_heads=a.kv_heads,
                                          layout=a.layout, shuffle_seed=1234)
        g = torch.Generator(device=dev).manual_seed(1000 + dev.index + 97 * seq_id + 7919 * a.seed)
        ctl = self.ctl
        # prefill L-1 tokens (device-side append with fused min/max metadata), then one decode token
        ctl.prepare_metadata(L - 1)
        ctl.begin_forward(L - 1)
        kbuf = torch.empty(L - 1, a.kv_heads, a.head_dim, dtype=torch.float16, device=dev)
        vbuf = torch.empty_like(kbuf)
        for layer in range(a.layers):
            kbuf.normal_(generator=g)
            vbuf.normal_(generator=g)
            qu.append_kv(kbuf, vbuf, ctl, layer)
        ctl.end_forward()
        del kbuf, vbuf
        self.q = torch.randn(a.layers, 1, a.heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.k1 = torch.randn(a.layers, 1, a.kv_heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.v1 = torch.randn(a.layers, 1, a.kv_heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.outs = [None] * a.layers
        self.dyn = a.mode == "graph" and a.skip_layers == 0 and not a.unfused
        if self.dyn:
            # state-driven stepping: the graph's first node reserves the token on the device
            ctl.enable_device_state()
            ctl.begin_graph_decode()
            self.scores = torch.empty(a.heads, ctl.max_pages, dtype=torch.float16, device=dev)
        else:
            ctl.prepare_metadata(1)

    def step_dyn(self):
        """One decode token, every length read from device memory (replayable as the sequence grows)."""
        qu, ctl, a = self.qu, self.ctl, self.a
        qu.step_advance_dyn(ctl)
        for layer in range(a.layers):
            self.outs[layer] = qu.decode_layer_dyn(self.q[layer], self.k1[layer], self.v1[layer], ctl, layer,
                                                   self.scores)

    def after_replay(self):
        if self.dyn:
            self.ctl.prepare_metadata(1)  # host mirror of the device-side reservation (Python ints only)

    def step(self):
        """One decode token: llama.py:424-439 controller sequence + QuestAttention.py:99-157 per layer."""
        if self.dyn:
            return self.step_dyn()
        qu, ctl, a = self.qu, self.ctl, self.a
        skip = a.skip_layers
        if skip > 0:
            ctl.set_page_budget(1 << 20)
            ctl.begin_forward(1)
        else:
            ctl.set_page_budget(self.page_budget)
            ctl.begin_forward(1)
        for layer in range(a.layers):
            if skip > 0 and layer == skip:
                ctl.end_forward()
                ctl.set_page_budget(self.page_budget)
                ctl.begin_forward(1, updateTensor=False)
            q = self.q[layer]
            if not ctl.need_estimate():
                qu.append_kv(self.k1[layer], self.v1[layer], ctl, layer)
                o = qu.decode_sparse_attn(q, ctl, layer, ctl.kv_indices_without_last)
            elif a.unfused:
                qu.append_kv(self.k1[layer], self.v1[layer], ctl, layer)
                est = qu.decode_estimate(q, ctl, layer)
                qu.decode_topk(est, ctl)
                o = qu.decode_sparse_attn(q, ctl, layer, ctl.topk_dindices_buffer)
            else:
                est = qu.decode_append_estimate(q, self.k1[layer], self.v1[layer], ctl, layer)
                o = qu.decode_topk_sparse_attn(q, est, ctl, layer, write_topk=False)
            self.outs[layer] = o
        ctl.end_forward()
        return o


class BatchedWorkload:
    """n sequences sharing one pool, decoded with one launch per op (quest_amd.utils.*_batched)."""

    dyn = True

    def __init__(self, a, dev, n_seqs, seq_id0=0):
        import quest_amd.utils as qu

        self.qu, self.a, self.dev, self.n = qu, a, dev, n_seqs
        self.page_budget = a.token_budget // a.page_size
        L = a.seqlen
        self.ctl = qu.BatchedInferenceController(n_seqs, a.layers, a.heads, a.head_dim, a.page_size, self.page_budget,
                                                 L + 2 * a.steps + a.warmup + 4 * a.page_size, torch.float16, dev,  # 2x: the N > 1 no-gather rerun
                                                 num_kv_heads=a.kv_heads, layout=a.layout, shuffle_seed=1234)
        kbuf = torch.empty(L - 1, a.kv_heads, a.head_dim, dtype=torch.float16, device=dev)
        vbuf = torch.empty_like(kbuf)
        g = torch.Generator(device=dev).manual_seed(1000 + dev.index + 97 * seq_id0 + 7919 * a.seed)
        for c in self.ctl.seqs:
            c.prepare_metadata(L - 1)
            c.begin_forward(L - 1)
            for layer in range(a.layers):
                kbuf.normal_(generator=g)
                vbuf.normal_(generator=g)
                qu.append_kv(kbuf, vbuf, c, layer)
            c.end_forward()
        del kbuf, vbuf
        self.q = torch.randn(a.layers, n_seqs, a.heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.k1 = torch.randn(a.layers, n_seqs, a.kv_heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.v1 = torch.randn(a.layers, n_seqs, a.kv_heads, a.head_dim, generator=g, device=dev, dtype=torch.float16)
        self.o = torch.empty_like(self.q)
        self.ctl.enable_device_state()
        if a.pages_per_chunk:
            self.ctl._decode_handler.set_pages_per_chunk(a.pages_per_chunk)
        self.ctl.begin_graph_decode()
        self.scores = torch.empty(n_seqs, a.heads, self.ctl.max_pages, dtype=torch.float16, device=dev)

    def step(self):
        qu, b, a = self.qu, self.ctl, self.a
        qu.step_advance_batched(b)
        for layer in range(a.layers):
            qu.decode_layer_batched(self.q[layer], self.k1[layer], self.v1[layer], b, layer, self.scores,
                                    out=self.o[layer])

    def after_replay(self):
        self.ctl.prepare_metadata(1)

    def sync(self):
        self.ctl.sync_device_state()


def bytes_per_layer(a):
    """Algorithmic bytes of one layer-step, SURVEY.md 8(d) accounting (= the reference benches')."""
    S, D, Hq, Hkv = a.page_size, a.head_dim, a.heads, a.kv_heads
    N = (a.seqlen + S - 1) // S
    B = min(a.token_budget // S, N)
    app = 4 * Hkv * D * 2 * 2 + 2 * 2 * Hkv * D * 2
    est = N * Hkv * 2 * D * 2 + Hq * D * 2 + 4 * ((N + S - 1) // S) + Hq * (N - 1) * 2
    topk = Hq * (N - 1) * (2 + 4) + Hq * (B - 1) * (2 + 4)
    att = B * S * 2 * Hq * D * 2 + Hq * D * 2 + Hq * (B - 1) * 4 + Hq * D * 2
    dense = N * S * 2 * Hkv * D * 2 + Hq * D * 2 + Hq * (N - 1) * 4 + Hq * D * 2  # every kv head read once
    return {"append": app, "estimate": est, "topk": topk, "attn": att, "chain": app + est + topk + att, "dense": dense}


def time_kernel_loop(fn, layers, reps):
    """Average duration of one `fn(layer)` launch: `layers` back-to-back launches (one per layer, so
    every launch reads a different pool) are captured into a hipGraph on torch's current stream -- the
    stream the kernels are launched on -- and `reps` replays are bracketed by two HIP events.  The
    figure includes the dependent-launch boundary (~1.5 us), i.e. what the op costs inside a step."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for layer in range(layers):
            fn(layer)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for layer in range(layers):
            fn(layer)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * layers)  # us


def batched_op_times(bw, bpl):
    """HIP-event time of the two batched launches of a layer (graph of one launch per layer, as in
    time_kernel_loop) and the HBM figures they amount to."""
    from quest_amd import _kernels

    b, n = bw.ctl, bw.n
    max_n = b.max_pages - 1

    def ae(l):
        _kernels.append_estimate_batched(bw.k1[l], bw.v1[l], b.kv_layer(l), b.kv_tables, bw.q[l], bw.scores,
                                         b.metadata_layer(l), b.meta_tables, b.step_states, max_n, b.layout)

    def ts(l):
        b._decode_handler.forward_fused_topk_batched(bw.q[l], bw.o[l], b.kv_layer(l), b.kv_tables, bw.scores,
                                                     b.step_states, max_n)

    t_ae = time_kernel_loop(ae, bw.a.layers, 10)
    t_ts = time_kernel_loop(ts, bw.a.layers, 10)
    ppc, chunks = b._decode_handler.plan_info()
    # full-KV decode of the same batch in one launch (group-shared kernel over every page of every sequence)
    b.begin_graph_decode(dense_layers=True)

    def dense(l):
        b._dense_handler.forward_shared_batched(bw.q[l], bw.o[l], b.kv_layer(l), b.kv_tables, b.step_states)

    t_dense = time_kernel_loop(dense, bw.a.layers, 3)
    return {"batched_append_estimate_us": t_ae, "batched_topk_sparse_attn_us": t_ts,
            "batched_dense_full_kv_us": t_dense, "batched_dense_full_kv_us_per_sequence": t_dense / n,
            "batched_dense_gbs": n * bpl["dense"] / (t_dense * 1e-6) / 1e9,
            "speedup_vs_batched_dense": t_dense / (t_ae + t_ts),
            "batched_append_estimate_gbs": n * (bpl["append"] + bpl["estimate"]) / (t_ae * 1e-6) / 1e9,
            "batched_topk_sparse_attn_gbs": n * (bpl["topk"] + bpl["attn"]) / (t_ts * 1e-6) / 1e9,
            "batched_sparse_attn_frac_of_hbm_peak": n * bpl["attn"] / (t_ts * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "plan": {"pages_per_workgroup": ppc, "workgroups_per_head": chunks}}


def multi_seq_side_measurement(a, dev, bpl, dense_us, n_seqs=8, layers=8, mode="batched"):
    """Side figure (not `value`): the same chain with 8 independent sequences on one GPU -- the per-GPU load
    of BASELINE configs[4].  batched: one launch per op serves all sequences (shared pool, grid.z = sequence);
    streams: each sequence's chain on its own stream inside one step graph.  Either way the fixed latencies
    of one sequence's kernels overlap the data movement of another's.  8 layers per sequence keep it short."""
    import copy

    b = copy.copy(a)
    b.layers, b.steps, b.warmup = layers, 30, 5
    if mode == "batched":
        b.mode = "graph"
        bw = BatchedWorkload(b, dev, n_seqs, 100)
        ws = [bw]

        def step_all():
            bw.step()
    else:
        b.mode = "graph-static"
        ws = [Workload(b, dev, 100 + i) for i in range(n_seqs)]
        streams = [torch.cuda.Stream() for _ in ws]

        def step_all():
            cur = torch.cuda.current_stream()
            for st, wl in zip(streams, ws):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    wl.step()
            for st in streams:
                cur.wait_stream(st)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step_all()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if mode == "batched":
        bw.sync()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step_all()

    def replay():
        g.replay()
        if mode == "batched":
            bw.after_replay()

    for _ in range(b.warmup):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(b.steps):
        replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    us = el * 1e6 / b.steps / layers / n_seqs
    res = {"sequences": n_seqs, "layers_per_sequence": layers, "mode": mode, "us_per_sequence_layer": us,
           "chain_frac_of_hbm_peak": bpl["chain"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           # a step of the full model yields n_seqs tokens in us * model_layers * n_seqs microseconds
           "tokens_per_s_scaled_to_model_layers": 1.0 / (us * 1e-6 * a.layers)}
    if dense_us is not None:
        res["speedup_vs_dense"] = dense_us / us
    if mode == "batched":
        res.update(batched_op_times(bw, bpl))
    del ws, g
    torch.cuda.empty_cache()
    return res


def cpu_baseline(a, budget_s):
    """The eager-PyTorch port of the reference's CPU-runnable oracle, on the host cores."""
    from oracle import torch_ref

    torch.manual_seed(0)
    L, H, D = a.seqlen, a.heads, a.head_dim
    q = torch.randn(1, H, D).half()
    k = torch.randn(L, H, D).half()
    v = torch.randn(L, H, D).half()
    B = a.token_budget // a.page_size
    with torch.inference_mode():
        torch_ref.sparse_decode(q, k, v, a.page_size, B)  # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            torch_ref.sparse_decode(q, k, v, a.page_size, B)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 64:
                break
    per_layer = el / n
    return {"value": 1.0 / (per_layer * a.layers), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n} layer-steps of oracle.torch_ref.sparse_decode (eager fp16 CPU, L={L}, budget {B} pages, "
                      f"H={H}); {per_layer * 1e3:.0f} ms per layer-step, scaled to {a.layers} layers"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("QUEST_BENCH_FORCE_DIST") == "1":  # the latter: rehearse the RCCL path on 1 GPU
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("QUEST_BENCH_REHEARSE") == "1":
            # rehearsal of the N > 1 control flow on a ONE-GPU box: all ranks share cuda:0, gloo collectives
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the quest_amd operators have no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from quest_amd.parallel import gather_tokens

    batched = (a.seqs_per_gpu > 1 and a.multi_seq_mode == "batched" and a.mode == "graph" and a.skip_layers == 0
               and not a.unfused)
    n_local = a.seqs_per_gpu
    if batched:
        ws = [BatchedWorkload(a, dev, n_local)]
    else:
        ws = [Workload(a, dev, i) for i in range(n_local)]
    w = ws[0]
    streams = [torch.cuda.Stream() for _ in ws] if len(ws) > 1 else None
    torch.cuda.synchronize()

    def step_all():
        """One decode token for every local sequence; sequences are independent, so with more than one
        each chain goes to its own stream (fork/join around the step)."""
        if streams is None:
            return w.step()
        cur = torch.cuda.current_stream()
        for st, wl in zip(streams, ws):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                wl.step()
        for st in streams:
            cur.wait_stream(st)

    # ---- the step, eager or captured
    if a.mode in ("graph", "graph-static"):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_all()  # warm the allocator / plan before capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for wl in ws:
            if wl.dyn:
                wl.ctl.sync_device_state()  # the warm-up advanced the device state; start from the prefilled cache
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step_all()

        def run():
            graph.replay()
            for wl in ws:
                wl.after_replay()
    else:
        run = step_all

    tok = torch.zeros(n_local, dtype=torch.int64, device=dev)  # stand-in for the sampled token ids of the local sequences

    def one_step():
        run()
        if dist is not None:
            gather_tokens(tok, dist)

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / a.steps
    value = world * n_local * a.steps / elapsed  # one token per local sequence per step

    # side figure for N > 1 ("report both with and without the gather"): the same K steps again
    # without the all_gather of token ids; not part of `value`
    ms_no_gather = None
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            run()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_no_gather = float(t.item()) * 1e3 / a.steps

    out = None
    if rank == 0:
        qu = w.qu
        # per-op timings run on ONE sequence (the first of a batch) through the reference's op sequence
        ctl = w.ctl.seqs[0] if batched else w.ctl
        wq = [w.q[l][:1] for l in range(a.layers)] if batched else [w.q[l] for l in range(a.layers)]
        wk = [w.k1[l][:1] for l in range(a.layers)] if batched else [w.k1[l] for l in range(a.layers)]
        wv = [w.v1[l][:1] for l in range(a.layers)] if batched else [w.v1[l] for l in range(a.layers)]
        bpl = bytes_per_layer(a)
        # ---- per-operator launch time with HIP events on the launch stream (steady state of this very step)
        if w.dyn:
            ctl.end_forward()
        ctl.set_page_budget(w.page_budget)
        ctl.begin_forward(1)
        ppc, chunks = ctl._decode_handler.plan_info()
        est0 = [qu.decode_estimate(wq[l], ctl, l) for l in range(a.layers)]
        for l in range(a.layers):
            qu.decode_topk(est0[l], ctl)
        idx = ctl.topk_dindices_buffer
        reps = 10
        t_att = time_kernel_loop(lambda l: qu.decode_sparse_attn(wq[l], ctl, l, idx), a.layers, reps)
        # the dominant kernel by itself (merge launch skipped; partial states stay in the workspace)
        ctl._decode_handler.set_skip_merge(True)
        t_att_kernel = time_kernel_loop(lambda l: qu.decode_sparse_attn(wq[l], ctl, l, idx), a.layers, reps)
        ctl._decode_handler.set_skip_merge(False)
        t_est = time_kernel_loop(lambda l: qu.decode_estimate(wq[l], ctl, l), a.layers, reps)
        t_topk = time_kernel_loop(lambda l: qu.decode_topk(est0[l], ctl), a.layers, reps)
        t_app = time_kernel_loop(lambda l: qu.append_kv(wk[l], wv[l], ctl, l), a.layers, reps)
        t_ae = time_kernel_loop(lambda l: qu.decode_append_estimate(wq[l], wk[l], wv[l], ctl, l), a.layers, reps)
        t_ts = time_kernel_loop(lambda l: qu.decode_topk_sparse_attn(wq[l], est0[l], ctl, l, write_topk=False),
                                a.layers, reps)
        ctl.end_forward()
        ops = {"append_us": t_app, "estimate_us": t_est, "topk_us": t_topk, "sparse_attn_plus_merge_us": t_att,
               "sparse_attn_kernel_only_us": t_att_kernel,
               "fused_append_estimate_us": t_ae, "fused_topk_sparse_attn_plus_merge_us": t_ts,
               "chain_us_in_step": ms_per_step * 1e3 / a.layers / n_local,
               "note": "per launch inside a hipGraph of 32 back-to-back launches (one per layer), "
                       "dependent-launch boundary included"}
        dense_us = None
        if not a.no_dense:
            ctl.set_page_budget(1 << 20)
            ctl.begin_forward(1, updateTensor=False)
            dense_us = time_kernel_loop(
                lambda l: qu.decode_sparse_attn(wq[l], ctl, l, ctl.kv_indices_without_last), a.layers, 3)
            ctl.end_forward()
        achieved = bpl["attn"] / (t_att_kernel * 1e-6) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")  # PMC bytes/launch, filled from rocprofv3 --pmc runs
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("sparse_decode_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "self-attn decode tokens/s (attention-only), seqlen=32768 token_budget=2048, 1/2/4/8 GPU",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: Yarn-Llama-2-7B-128K shapes, {n_local} sequence(s) per GPU, "
                                   "self-attention chain (append+estimate+top-k+sparse attn) x all layers per token",
                       "layers": a.layers, "num_qo_heads": a.heads, "num_kv_heads": a.kv_heads,
                       "head_dim": a.head_dim, "seqlen": a.seqlen, "seqlen_after_run": ctl.kv_cache.seqlen, "page_size": a.page_size,
                       "token_budget": a.token_budget, "page_budget_pages": a.token_budget // a.page_size,
                       "kv_layout": a.layout, "mode": a.mode, "seed": a.seed, "skip_layers": a.skip_layers,
                       "launches_per_layer": "5 (reference op sequence)" if a.unfused else
                       "3 (append+estimate | top-k+sparse attn | merge)",
                       "sequences_per_gpu": n_local,
                       "multi_sequence": ("batched launches, shared pool" if batched else "one stream per sequence")
                       if n_local > 1 else None, "parallelism": f"sequence-sharded x{world}, all_gather(token ids)"},
            "roofline": {"bound": "hbm", "kernel": "sparse_decode_kernel (the decode_sparse_attn op without its merge launch; "
                                                   "op incl. merge: op_us)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bpl["attn"], "launch_us": t_att_kernel, "op_us": t_att,
                         "op_frac": bpl["attn"] / (t_att * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "plan": {"pages_per_workgroup": ppc, "workgroups_per_head": chunks}},
            "ops_us": ops,
            "ms_per_step_without_token_gather": ms_no_gather,
            "selfattn_us_per_layer": ms_per_step * 1e3 / a.layers / n_local,
            "chain_bytes_per_layer": bpl["chain"],
            "chain_frac_of_hbm_peak": bpl["chain"] * n_local / (ms_per_step * 1e-3 / a.layers) / 1e9 / HBM_PEAK_GBS,
        }
        if dense_us is not None:
            out["dense_full_kv_us"] = dense_us
            out["dense_gbs"] = bpl["dense"] / (dense_us * 1e-6) / 1e9
            out["speedup_vs_dense"] = dense_us / (ms_per_step * 1e3 / a.layers / n_local)
        if batched:
            out["batched_ops"] = batched_op_times(w, bpl)
        if world == 1 and n_local == 1 and not a.no_multi_seq:
            out["eight_sequences_per_gpu"] = multi_seq_side_measurement(a, dev, bpl, dense_us, mode=a.multi_seq_mode)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, a.cpu_sample_s)
        else:
            out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
