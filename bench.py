#!/usr/bin/env python3
"""Headline bench: Quest self-attention decode (append -> estimate -> top-k -> sparse attention)
at BASELINE.json's metric point -- Yarn-Llama-2-7B shapes (32 layers, 32 heads, D=128, fp16),
seqlen 32768, token budget 2048 = 128 pages of 16 -- on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W] [--config {2,3,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: it starts N
fresh rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), never touches
the GPU itself, and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is a rank.

A "step" is one decode token of every sequence of every GPU: the four-operator chain run for every
layer of the model against that layer's own KV / metadata pools (so consecutive kernels touch
different memory and the 256 MiB Infinity Cache cannot hold the working set).  Everything goes
through the C ABI of libquest_hip.so; tokens/s is attention-only (no weights offline).
Sequences are independent, so GPUs never exchange data inside a step; with N > 1 each step ends
with one RCCL all_gather of the sampled token ids (weak scaling: the same sequences per GPU).  (Measured
over RCCL, world size 1: +1 % per step; issuing the gather asynchronously behind the next step was
tried and is slower, +16 % -- the RCCL kernel then runs beside the step's kernels.)

One JSON line on stdout (rank 0).  Extra objects: "roofline" (the dominant kernel OF THE TIMED STEP:
sparse_decode_kernel with its top-k front end; algorithmic bytes / HIP-event launch time) and
"cpu_baseline" (oracle/torch_ref eager port timed on the host cores, bounded sample).  Side objects of the default N = 1 run,
measured after the headline and outside `value`: "batched_8seq", "cfg5_8seq_gqa" (8 sequences per GPU), the same and the
headline on the reference's NHD pool layout ("reference_layout_nhd", "batched_8seq_nhd", "cfg5_8seq_gqa_nhd": the default
layout is this build's row-rotated NHD since round 6, config.kv_layout) and
"prefill_attention" (the prefill operator on the headline shapes: the MFMA kernel's TFLOP/s against the dense fp16 peak).
"""
import argparse
import datetime
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# SURVEY.md 8 "BASELINE configs -> concrete sizes" (cfg n = BASELINE.json configs[n-1]); budgets in TOKENS here
CONFIGS = {
    2: dict(label="BASELINE configs[1]: LongChat-7B-v1.5-32K shapes (page budget 512 >= 256 pages: full KV)",
            layers=32, heads=32, kv_heads=32, seqlen=4096, token_budget=512 * 16, seqs_per_gpu=1),
    3: dict(label="BASELINE configs[2]: Yarn-Llama-2-7B-128K shapes", layers=32, heads=32, kv_heads=32, seqlen=32768,
            token_budget=2048, seqs_per_gpu=1),
    4: dict(label="BASELINE configs[3]: Llama-3.1-8B-Instruct (GQA) shapes", layers=32, heads=32, kv_heads=8,
            seqlen=131072, token_budget=4096, seqs_per_gpu=1),
    5: dict(label="BASELINE configs[4]: Llama-3.1-8B (GQA) shapes, 64 x 32K sequences sharded 8 per GPU", layers=32,
            heads=32, kv_heads=8, seqlen=32768, token_budget=2048, seqs_per_gpu=8),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed decode tokens per sequence (default: ~1 s of timed region at the headline config)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="SURVEY cfg number: 3 = BASELINE configs[2] (headline), 4 = 128K GQA, 5 = 8 x 32K GQA per GPU, "
                         "2 = 4K full-KV; the shape flags below override single fields")
    ap.add_argument("--layers", type=int)
    ap.add_argument("--heads", type=int)
    ap.add_argument("--kv-heads", type=int)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--seqlen", type=int)
    ap.add_argument("--token-budget", type=int)
    ap.add_argument("--page-size", type=int, default=16)
    ap.add_argument("--layout", choices=["NHD", "HND", "NHD_ROT"], default="NHD_ROT",
                    help="pool layout: this build's row-rotated NHD (default since round 6: QUEST_LAYOUT_NHD_ROT, "
                         "include/quest_hip.h -- the NHD shape with the heads of an entry rotated by the entry, so that every "
                         "head's 256-byte pieces cycle through the address classes MI355X serves unevenly; same bits as NHD), "
                         "or the reference's NHD / HND.  config.kv_layout names it; the default run reports the NHD figures "
                         "beside it (side objects `reference_layout_nhd`, `batched_8seq_nhd`, `cfg5_8seq_gqa_nhd`)")
    ap.add_argument("--mode", choices=["graph", "graph-static", "eager"], default="graph",
                    help="graph: ONE hipGraph of the whole step, replayed while the sequence grows by a token per "
                         "step (device-resident step state); graph-static: the same graph without the state "
                         "(re-decodes the same position); eager: Python op by op")
    ap.add_argument("--skip-layers", type=int, default=0,
                    help="run the first n layers with full KV like quest/models/llama.py:428-439 (default 0)")
    ap.add_argument("--unfused", action="store_true",
                    help="issue the reference's five launches per layer instead of the fused append+estimate and "
                         "top-k+attention launches (same results)")
    ap.add_argument("--seqs-per-gpu", type=int,
                    help="independent sequences per GPU (config 5 uses 8), decoded with ONE launch per op over a "
                         "shared pool (grid.z = sequence) unless --multi-seq-mode streams")
    ap.add_argument("--multi-seq-mode", choices=["batched", "streams"], default="batched")
    ap.add_argument("--seq-groups", type=int, default=1,
                    help="batched mode: split the local sequences into this many groups, each a batched workload over its "
                         "own pool on its own stream (one hipGraph for all): a group's selection prologue and dispatch "
                         "gaps run under another group's streaming (experiment, DESIGN.md 3.5)")
    ap.add_argument("--layer-launches", choices=["auto", "one", "two"], default="auto",
                    help="batched sparse layers: one launch per layer (a workgroup per (sequence, head) appends, scores into "
                         "LDS, selects and gathers: csrc/layer_device.cuh) or the two launches append+estimate | "
                         "top-k+attention; auto = one launch where the plan allows it (MHA batches that fill the chip)")
    ap.add_argument("--tiles", choices=["auto", "on", "off"], default="auto",
                    help="single-sequence sparse layers: the tiles launches (the estimate hands per-8-page score maxima to the "
                         "attention launch, which selects in two short passes); auto = pools of 4097 pages and more (cfg 4)")
    ap.add_argument("--pages-per-chunk", type=int, default=0, help="override the decode planner (tuning)")
    ap.add_argument("--separate-dense-append", action="store_true",
                    help="full-KV layers: issue the decode append as its own launch (three launches per layer, as before "
                         "round 4) instead of folded into the attention launch (A/B)")
    ap.add_argument("--separate-advance", action="store_true",
                    help="single-sequence graph steps: head every step with the step_state_advance launch (as before round 6) "
                         "instead of letting the next token's reservation ride in the last layer's merge launch (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--kernel-sweep", action="store_true",
                    help="instead of the step bench: the reference's own kernel benches row for row (attention kernel "
                         "alone at seqlen 4096 x budgets 64-512 and 32768 x 256/640/896, top-k and estimate at the six "
                         "LongBench pairs of scripts/bench_kernels.sh) -> gpurun_out/kernel_sweep.json + a markdown table")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side configurations (8 sequences per GPU) the default headline run measures after its "
                         "timed region")
    ap.add_argument("--side-steps", type=int, default=100, help="timed steps of each side configuration")
    ap.add_argument("--side-warmup", type=int, default=None,
                    help="untimed steps before each side configuration's timed steps (default: max(--warmup, 20))")
    ap.add_argument("--cpu-sample-s", type=float, default=15.0)
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic K/V/q (SURVEY 8d: seeds 0, 1, 2)")
    a = ap.parse_args(argv)
    preset = CONFIGS[a.config]
    custom = []
    for key in ("layers", "heads", "kv_heads", "seqlen", "token_budget", "seqs_per_gpu"):
        if getattr(a, key) is None:
            setattr(a, key, preset[key])
        elif getattr(a, key) != preset[key]:
            custom.append(f"{key}={getattr(a, key)}")
    a.workload_label = preset["label"] if not custom else f"custom ({preset['label']} with {', '.join(custom)})"
    return a


# ------------------------------------------------------------------------------------------------ launcher

def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


RENDEZVOUS_TIMEOUT_S = 300  # init_process_group / collectives: a peer that never shows up must not stall the node (the
# other ranks also wait this long at the closing barrier while rank 0 takes its per-operator side measurements: ~10 s)


def launch_ranks(n, poll_s=0.1):
    """Start n rank processes of this script and relay rank 0's stdout.  The launcher never initialises the
    GPU (no torch.cuda / HIP call) and no process that has done so is ever re-exec'd: every rank is a fresh
    child.  ALL children are polled: on the first non-zero exit the others are terminated (a rank that dies
    before the rendezvous would otherwise leave its peers waiting for the process-group timeout) and the
    launcher returns that exit code promptly, after printing one JSON error line if rank 0 produced none."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # drains rank 0's pipe while every child is being watched
    failed = None  # (rank, exit code) of the first rank seen to fail
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                failed = (r, code)
                break
        if live and failed is None:
            time.sleep(poll_s)
    if failed is not None:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + 10
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    reader.join(timeout=10)
    out = (chunks[0] if chunks else b"").decode()
    if failed is not None and not any(l.startswith("{") for l in out.splitlines()):
        out += json.dumps({"error": f"rank {failed[0]} of {n} exited with code {failed[1]}; the other ranks were "
                                    "terminated", "n_gpus": n, "failed_rank": failed[0], "exit_code": failed[1]}) + "\n"
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0 if failed is None else (failed[1] if failed[1] > 0 else 1)


# ------------------------------------------------------------------------------------------------ workloads

class Workload:
    """One sequence: controller + filled pools + per-layer decode inputs."""

    def __init__(self, a, dev, seq_id=0):
        import torch
        import quest_amd.utils as qu

        self.qu = qu
        self.a = a
        self.dev = dev
        self.n = 1
        self.page_budget = a.token_budget // a.page_size
        L = a.seqlen
        cap_tokens = L + 2 * a.steps + a.warmup + 4 * a.page_size  # 2x: the N > 1 no-gather rerun
        self.ctl = qu.InferenceController(a.layers, a.heads, a.head_dim, a.page_size, self.page_budget, cap_tokens,
                                          torch.float16, dev, num_kv_heads=a.kv_heads, layout=a.layout,
                                          shuffle_seed=1234)
        self.gen_seed = 1000 + dev.index + 97 * seq_id + 7919 * a.seed
        g = torch.Generator(device=dev).manual_seed(self.gen_seed)
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
        n_pages_end = (cap_tokens + a.page_size - 1) // a.page_size
        self.dense = self.page_budget >= n_pages_end  # budget covers the cache for the whole run: full-KV decode
        self.dyn = a.mode == "graph" and a.skip_layers == 0 and not a.unfused
        self.fold = self.dyn and not a.separate_advance  # see prime()
        if self.dyn:
            # state-driven stepping: the graph's first node reserves the token on the device
            ctl.enable_device_state()
            if a.tiles != "auto":
                ctl.tiles_min_pages = 0 if a.tiles == "on" else 1 << 30
            if a.pages_per_chunk:
                ctl._decode_handler.set_pages_per_chunk(a.pages_per_chunk)
            ctl.begin_graph_decode(dense_layers=self.dense)
            self.scores = qu.score_scratch(ctl)
            self.tiles = ctl.max_pages - 1 >= ctl.tiles_min_pages and ctl.inference_page_budget - 1 <= 256
        else:
            ctl.prepare_metadata(1)

    def prime(self):
        """Folded stepping (round 6): the NEXT token's reservation rides in the last layer's merge launch instead of heading
        every step as a 1-thread launch of its own (4.7 us of dependent latency per token) -- so the first token is reserved
        here, once, before the first step; device state and host mirror then run one reserved token ahead."""
        if self.dyn and self.fold:
            self.qu.step_advance_dyn(self.ctl)
            self.ctl.prepare_metadata(1)

    def step_dyn(self):
        """One decode token, every length read from device memory (replayable as the sequence grows)."""
        qu, ctl, a = self.qu, self.ctl, self.a
        if not self.fold:
            qu.step_advance_dyn(ctl)
        for layer in range(a.layers):
            last = self.fold and layer == a.layers - 1
            if self.dense:
                self.outs[layer] = qu.decode_layer_dense_dyn(self.q[layer], self.k1[layer], self.v1[layer], ctl, layer,
                                                             fuse_append=not a.separate_dense_append, advance_after=last)
            else:
                self.outs[layer] = qu.decode_layer_dyn(self.q[layer], self.k1[layer], self.v1[layer], ctl, layer,
                                                       self.scores, advance_after=last)

    def after_replay(self):
        if self.dyn:
            self.ctl.prepare_metadata(1)  # host mirror of the device-side reservation (Python ints only)

    def sync(self):
        if self.dyn:
            self.ctl.sync_device_state()

    def step(self):
        """One decode token: llama.py:424-439 controller sequence + QuestAttention.py:99-157 per layer."""
        if self.dyn:
            return self.step_dyn()
        qu, ctl, a = self.qu, self.ctl, self.a
        skip = a.skip_layers
        ctl.set_page_budget(1 << 20 if skip > 0 else self.page_budget)
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
    dense = False

    def __init__(self, a, dev, n_seqs, seq_id0=0):
        import torch
        import quest_amd.utils as qu

        self.qu, self.a, self.dev, self.n = qu, a, dev, n_seqs
        self.page_budget = a.token_budget // a.page_size
        L = a.seqlen
        self.ctl = qu.BatchedInferenceController(n_seqs, a.layers, a.heads, a.head_dim, a.page_size, self.page_budget,
                                                 L + 2 * a.steps + a.warmup + 4 * a.page_size, torch.float16, dev,
                                                 num_kv_heads=a.kv_heads, layout=a.layout, shuffle_seed=1234)
        kbuf = torch.empty(L - 1, a.kv_heads, a.head_dim, dtype=torch.float16, device=dev)
        vbuf = torch.empty_like(kbuf)
        self.gen_seed = 1000 + dev.index + 97 * seq_id0 + 7919 * a.seed
        g = torch.Generator(device=dev).manual_seed(self.gen_seed)
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
        self.scores = qu.score_scratch(self.ctl)
        if a.layer_launches != "auto":
            self.ctl.one_launch_layers = a.layer_launches == "one"
        self.one_launch = False  # set by the first step: did the layers take the one-launch kernel?

    def step(self):
        qu, b, a = self.qu, self.ctl, self.a
        qu.step_advance_batched(b)
        for layer in range(a.layers):
            qu.decode_layer_batched(self.q[layer], self.k1[layer], self.v1[layer], b, layer, self.scores,
                                    out=self.o[layer])
        self.one_launch = b._decode_handler.last_launch_info()["front_end_variant"] == 7

    def after_replay(self):
        self.ctl.prepare_metadata(1)

    def sync(self):
        self.ctl.sync_device_state()


class StubWorkload:
    """QUEST_BENCH_STUB=1: no GPU, no kernels -- a trivial CPU step, so that the launcher, the rendezvous, the
    per-step token gather, the max-over-ranks timing and the JSON contract can be exercised on a CPU-only
    box (tests/test_parallel_gloo.py).  The line it produces is marked "data": "stub" and measures nothing."""

    dyn = False
    dense = False

    def __init__(self, a):
        import torch

        self.n = a.seqs_per_gpu
        self.x = torch.zeros(16)

    def step(self):
        self.x += 1

    def after_replay(self):
        pass

    def sync(self):
        pass


def bytes_per_layer(a):
    """Algorithmic bytes of one layer-step of ONE sequence, SURVEY.md 8(d) accounting (= the reference benches')."""
    S, D, Hq, Hkv = a.page_size, a.head_dim, a.heads, a.kv_heads
    N = (a.seqlen + S - 1) // S
    B = min(a.token_budget // S, N)
    app = 4 * Hkv * D * 2 * 2 + 2 * 2 * Hkv * D * 2
    est = N * Hkv * 2 * D * 2 + Hq * D * 2 + 4 * ((N + S - 1) // S) + Hq * (N - 1) * 2
    topk = Hq * (N - 1) * (2 + 4) + Hq * (B - 1) * (2 + 4)
    att = B * S * 2 * Hq * D * 2 + Hq * D * 2 + Hq * (B - 1) * 4 + Hq * D * 2
    dense = N * S * 2 * Hkv * D * 2 + Hq * D * 2 + Hq * (N - 1) * 4 + Hq * D * 2  # every kv head read once
    sparse = B < N
    chain = app + est + topk + att if sparse else app + dense
    return {"append": app, "estimate": est, "topk": topk, "attn": att, "chain": chain, "dense": dense,
            "pages": N, "budget_pages": B, "sparse": sparse}


def time_kernel_loop(fn, layers, reps):
    """Average duration of one `fn(layer)` launch: `layers` back-to-back launches (one per layer, so
    every launch reads a different pool) are captured into a hipGraph on torch's current stream -- the
    stream the kernels are launched on -- and `reps` replays are bracketed by two HIP events.  The
    figure includes the dependent-launch boundary (~1.5 us), i.e. what the op costs inside a step."""
    import torch

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


def step_op_times(w, a, bpl, reps=10):
    """HIP-event time of the launches the timed step is made of, each as a graph of one launch per layer
    on the stream the kernels are launched on, reading the same device-resident state as the step.
    Returns (ops dict, roofline dict).  The roofline describes the DOMINANT kernel of the step."""
    from quest_amd import _kernels

    n = w.n
    if isinstance(w, BatchedWorkload):
        b = w.ctl
        max_n = b.max_pages - 1
        handler = b._decode_handler

        def ae(l):
            _kernels.append_estimate_batched(w.k1[l], w.v1[l], b.kv_layer(l), b.kv_tables, w.q[l], w.scores,
                                             b.metadata_layer(l), b.meta_tables, b.step_states, max_n, b.layout)

        def ts(l):
            handler.forward_fused_topk_batched(w.q[l], w.o[l], b.kv_layer(l), b.kv_tables, w.scores, b.step_states, max_n)
    else:
        ctl = w.ctl
        max_n = ctl.max_pages - 1
        handler = ctl._decode_handler
        o = [w.q[l].clone() for l in range(a.layers)]

        def ae(l):
            assert _kernels.append_estimate_dyn(w.k1[l], w.v1[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.q[l],
                                                w.scores, ctl.metadata_cache.buf_layer(l), ctl.meta_table_full,
                                                ctl.step_state, max_n, ctl.layout, tiles=w.tiles)

        def ts(l):
            assert handler.forward_fused_topk_dyn(w.q[l], o[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.scores,
                                                  ctl.step_state, max_n, tiles=w.tiles)

    ppc, chunks = handler.plan_info()
    if isinstance(w, BatchedWorkload) and w.one_launch:
        # the timed step's layer is ONE launch (layer_decode_kernel): it is the dominant -- the only -- kernel
        def layer(l):
            assert handler.layer_fused_batched(w.k1[l], w.v1[l], b.metadata_layer(l), b.meta_tables, w.q[l], w.o[l],
                                               b.kv_layer(l), b.kv_tables, b.step_states, max_n, b.page_budgets)

        t_layer = time_kernel_loop(layer, a.layers, reps)
        li = handler.last_launch_info()
        assert li["front_end_variant"] == 7, li
        kname = f"layer_decode_kernel<{a.head_dim},{li['keys_per_thread']},{li['waves']}>"
        alg = n * bpl["chain"]
        achieved = alg / (t_layer * 1e-6) / 1e9
        t_ae = time_kernel_loop(ae, a.layers, reps)   # the two launches it replaces, for comparison
        t_ts = time_kernel_loop(ts, a.layers, reps)
        ops = {"layer_us": t_layer, "two_launch_form_append_estimate_us": t_ae, "two_launch_form_topk_sparse_attn_us": t_ts,
               "note": "per launch inside a hipGraph of one launch per layer (each on its own pool), dependent-launch "
                       "boundary included; layer_us = the launch of the timed step"}
        roof = {"bound": "hbm",
                "kernel": f"{kname} (the whole layer in one launch: a workgroup per (sequence, head) appends, scores its "
                          "head's pages into LDS, selects from LDS, gathers the selected K/V pages; algorithmic bytes = the "
                          "chain's, SURVEY 8d: A + E + T + S)",
                "kernel_name": kname, "launch": li,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg, "launch_us": t_layer, "sequences_per_launch": n,
                "plan": {"pages_per_workgroup": ppc, "workgroups_per_head": chunks}}
        t_ae_ts = t_layer
    else:
        t_ae_ts = None
    if t_ae_ts is None:
        ops, roof, t_ae_ts = _two_launch_op_times(w, a, bpl, reps, ae, ts, handler, ppc, chunks, n)
    if isinstance(w, BatchedWorkload) and not a.no_dense:
        # full-KV decode of the same batch in one launch (group-shared kernel over every page of every sequence)
        b.begin_graph_decode(dense_layers=True)
        t_dense = time_kernel_loop(lambda l: b._dense_handler.forward_shared_batched(w.q[l], w.o[l], b.kv_layer(l),
                                                                                      b.kv_tables, b.step_states),
                                   a.layers, 3)
        ops.update({"batched_dense_full_kv_us": t_dense, "batched_dense_full_kv_us_per_sequence": t_dense / n,
                    "batched_dense_gbs": n * bpl["dense"] / (t_dense * 1e-6) / 1e9,
                    "speedup_vs_batched_dense_ops": t_dense / t_ae_ts})
    return ops, roof


def _two_launch_op_times(w, a, bpl, reps, ae, ts, handler, ppc, chunks, n):
    t_ae = time_kernel_loop(ae, a.layers, reps)
    t_ts = time_kernel_loop(ts, a.layers, reps)
    handler.set_skip_merge(True)  # the attention kernel by itself (partial states stay in the workspace)
    t_ts_kernel = time_kernel_loop(ts, a.layers, reps)
    handler.set_skip_merge(False)
    alg = n * (bpl["attn"] + bpl["topk"])
    achieved = alg / (t_ts_kernel * 1e-6) / 1e9
    li = handler.last_launch_info()  # what the timed launches ARE (keys per thread, waves, front-end variant)
    fc, waves = li["keys_per_thread"], li["waves"]
    kname = (f"sparse_decode_kernel<{a.head_dim},{a.page_size if a.page_size == 16 else 0},{fc},{waves},"
             f"{li['front_end_variant'] if li['specialised'] else -1}>")
    ops = {"append_estimate_us": t_ae, "topk_sparse_attn_plus_merge_us": t_ts, "topk_sparse_attn_kernel_only_us": t_ts_kernel,
           "append_estimate_gbs": n * (bpl["append"] + bpl["estimate"]) / (t_ae * 1e-6) / 1e9,
           "append_estimate_frac_of_hbm_peak": n * (bpl["append"] + bpl["estimate"]) / (t_ae * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "note": "per launch inside a hipGraph of one launch per layer (each on its own pool), dependent-launch "
                   "boundary included; state-driven entry points = the launches of the timed step"}
    roof = {"bound": "hbm",
            "kernel": f"{kname} (top-k front end + gather of the selected K/V pages; the dominant kernel of the timed "
                      "step, merge launch excluded)",
            "kernel_name": kname, "launch": li,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg, "launch_us": t_ts_kernel, "sequences_per_launch": n,
            "op_us_with_merge": t_ts, "plan": {"pages_per_workgroup": ppc, "workgroups_per_head": chunks}}
    return ops, roof, t_ae + t_ts


def reference_op_times(w, a, bpl, reps=10):
    """Side figures: the reference's five-op sequence (and the two fused launches) on ONE sequence through the
    host-planned entry points, plus the full-KV (dense) decode the speed-up is quoted against."""
    import torch

    qu = w.qu
    batched = isinstance(w, BatchedWorkload)
    ctl = w.ctl.seqs[0] if batched else w.ctl
    wq = [w.q[l][:1] for l in range(a.layers)] if batched else [w.q[l] for l in range(a.layers)]
    wk = [w.k1[l][:1] for l in range(a.layers)] if batched else [w.k1[l] for l in range(a.layers)]
    wv = [w.v1[l][:1] for l in range(a.layers)] if batched else [w.v1[l] for l in range(a.layers)]
    ops = {}
    if w.dyn:
        ctl.end_forward()
    if bpl["sparse"]:
        ctl.set_page_budget(w.page_budget)
        ctl.begin_forward(1)
        est0 = [qu.decode_estimate(wq[l], ctl, l) for l in range(a.layers)]
        for l in range(a.layers):
            qu.decode_topk(est0[l], ctl)
        idx = ctl.topk_dindices_buffer
        h = ctl._decode_handler
        ops["sparse_attn_plus_merge_us"] = time_kernel_loop(lambda l: qu.decode_sparse_attn(wq[l], ctl, l, idx), a.layers, reps)
        h.set_skip_merge(True)
        t_k = time_kernel_loop(lambda l: qu.decode_sparse_attn(wq[l], ctl, l, idx), a.layers, reps)
        h.set_skip_merge(False)
        ops["sparse_attn_kernel_only_us"] = t_k  # the index-tensor instantiation (no front end): NOT on the timed path
        ops["sparse_attn_kernel_only_frac_of_hbm_peak"] = bpl["attn"] / (t_k * 1e-6) / 1e9 / HBM_PEAK_GBS
        ops["estimate_us"] = time_kernel_loop(lambda l: qu.decode_estimate(wq[l], ctl, l), a.layers, reps)
        ops["topk_us"] = time_kernel_loop(lambda l: qu.decode_topk(est0[l], ctl), a.layers, reps)
        ops["append_us"] = time_kernel_loop(lambda l: qu.append_kv(wk[l], wv[l], ctl, l), a.layers, reps)
        ctl.end_forward()
    dense_us = None
    if not a.no_dense:
        ctl.set_page_budget(1 << 20)
        ctl.begin_forward(1, updateTensor=not bpl["sparse"])
        dense_us = time_kernel_loop(lambda l: qu.decode_sparse_attn(wq[l], ctl, l, ctl.kv_indices_without_last),
                                    a.layers, 3)
        ctl.end_forward()
    ops["note"] = ("reference op sequence on one sequence, host-planned entry points; per launch inside a hipGraph of "
                   "one launch per layer, dependent-launch boundary included")
    return ops, dense_us


def cpu_baseline(a, w, budget_s):
    """The eager-PyTorch port of the reference's CPU-runnable oracle on the host cores, on the GPU run's own
    layer-0 inputs of the first sequence (regenerated from the workload's seed, copied to the host): one
    warm-up, then the median of >= 5 runs within the time budget."""
    import torch
    from oracle import torch_ref

    L = a.seqlen
    g = torch.Generator(device=w.dev).manual_seed(w.gen_seed)
    kd = torch.empty(L - 1, a.kv_heads, a.head_dim, dtype=torch.float16, device=w.dev).normal_(generator=g)
    vd = torch.empty_like(kd).normal_(generator=g)  # same draw order as the workload's layer 0
    batched = isinstance(w, BatchedWorkload)
    k = torch.cat([kd, (w.k1[0][:1] if batched else w.k1[0]).reshape(1, a.kv_heads, a.head_dim)]).cpu()
    v = torch.cat([vd, (w.v1[0][:1] if batched else w.v1[0]).reshape(1, a.kv_heads, a.head_dim)]).cpu()
    q = (w.q[0][:1] if batched else w.q[0]).reshape(1, a.heads, a.head_dim).cpu()
    del kd, vd
    G = a.heads // a.kv_heads
    if G > 1:  # the eager oracle is written for expanded K/V (evaluation/quest_attention.py:139-184 repeat_kv)
        k, v = k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)
    B = a.token_budget // a.page_size
    times = []
    with torch.inference_mode():
        torch_ref.sparse_decode(q, k, v, a.page_size, B)  # warm-up
        t_begin = time.perf_counter()
        while len(times) < 5 or (time.perf_counter() - t_begin < budget_s and len(times) < 64):
            t0 = time.perf_counter()
            torch_ref.sparse_decode(q, k, v, a.page_size, B)
            times.append(time.perf_counter() - t0)
    per_layer = statistics.median(times)
    return {"value": 1.0 / (per_layer * a.layers), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"median of {len(times)} layer-steps (after 1 warm-up) of oracle.torch_ref.sparse_decode on the GPU "
                      f"run's layer-0 inputs (eager fp16 CPU, L={L}, budget {B} pages, Hq={a.heads}, Hkv={a.kv_heads}); "
                      f"{per_layer * 1e3:.0f} ms per layer-step, scaled to {a.layers} layers, one sequence"}


def pmc_traffic(a, n_local, kernel_name=None):
    """HBM bytes (read + write) per launch of the dominant kernel of this configuration, from the COMMITTED rocprofv3
    --pmc passes of this very command (profiles/traffic_latest.json, gfx950 FETCH_SIZE correction applied) -- a lookup,
    not a measurement of the run that prints it.  Keyed by configuration AND by the kernel instantiation that was
    launched (`roofline.kernel_name`): a profile of another instantiation is not reported as this run's traffic.
    Returns (bytes or None, source description)."""
    tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if "custom" in a.workload_label and not (a.config == 3 and n_local == 8):
        return None, None
    try:
        t = json.load(open(tp))
        rec = t["bytes_per_launch"].get(f"cfg{a.config}_seqs{n_local}")
        if rec is None or (kernel_name is not None and rec.get("kernel") not in (None, kernel_name)):
            return None, (f"profiles/traffic_latest.json holds no entry for {kernel_name} at cfg{a.config}_seqs{n_local}"
                          if rec is not None else None)
        return rec["bytes"], ("committed profile profiles/traffic_latest.json (" + t.get("source", "rocprofv3 --pmc") +
                              f", kernel {rec.get('kernel')}); not re-measured by this run")
    except Exception:
        return None, None


def add_traffic(roof, a, n_local):
    """roofline.traffic + its source + the fraction of the HBM peak the PMC bytes amount to (next to `frac`, which is
    algorithmic bytes / time)."""
    roof["traffic"], roof["traffic_source"] = (pmc_traffic(a, n_local, roof["kernel_name"]) if roof.get("kernel_name")
                                               else (None, None))  # no named instantiation: nothing to look up
    roof["frac_hbm_pmc"] = (roof["traffic"] / (roof["launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                            if roof["traffic"] and roof.get("launch_us") else None)
    if roof["traffic"] and roof["traffic"] < 0.95 * roof.get("algorithmic_bytes_per_launch", 0):
        roof["traffic_note"] = ("HBM bytes below the algorithmic bytes: with GQA the query heads of a kv-head group "
                                "run on one XCD (XCD-aware grid order) and re-read each other's pages from its L2")


def measure(a, dev, dist, world_seen, rank, stub, side=False):
    """Build the workload `a` describes, capture the step, time EXACTLY a.steps steps after a.warmup (barrier +
    synchronize on both sides, max over ranks) and, on rank 0, return the result line as a dict (None elsewhere).
    side=True: a side configuration measured after the headline -- no CPU baseline, no reference-op side figures."""
    import torch

    sync = (lambda: None) if stub else torch.cuda.synchronize
    from quest_amd.parallel import gather_tokens

    n_local = a.seqs_per_gpu
    batched = (n_local > 1 and a.multi_seq_mode == "batched" and a.mode == "graph" and a.skip_layers == 0
               and not a.unfused)
    build_error = None
    try:
        if stub:
            if os.environ.get("QUEST_BENCH_STUB_BUILD_FAIL") == f"{rank}:{int(side)}":
                raise MemoryError("stub: this rank cannot build its workload")  # tests/test_parallel_gloo.py
            ws = [StubWorkload(a)]
        elif batched and a.seq_groups > 1:
            assert n_local % a.seq_groups == 0, "--seq-groups must divide the sequences per GPU"
            per = n_local // a.seq_groups
            ws = [BatchedWorkload(a, dev, per, seq_id0=i * per) for i in range(a.seq_groups)]
        elif batched:
            ws = [BatchedWorkload(a, dev, n_local)]
        else:
            ws = [Workload(a, dev, i) for i in range(n_local)]
    except Exception as exc:  # most likely memory: the pools of a side configuration
        if dist is None:
            raise
        build_error = exc
    if dist is not None:
        # every rank says whether it built its workload BEFORE any other collective of this measurement: a rank that failed
        # would otherwise leave through its exception while its peers wait in the barrier / token gather below until the
        # process-group timeout.  All ranks leave together, with the failing rank's message where it is known.
        ok = torch.tensor([0 if build_error is not None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError(f"rank {rank}: " + (f"{type(build_error).__name__}: {build_error}" if build_error is not None
                                                    else "a peer rank failed to build its workload"))
    w = ws[0]
    streams = [torch.cuda.Stream() for _ in ws] if len(ws) > 1 else None
    sync()

    def step_all():
        """One decode token for every local sequence; unbatched sequences each go to their own stream."""
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
    if a.mode in ("graph", "graph-static") and not stub:
        for wl in ws:
            if hasattr(wl, "prime"):
                wl.prime()  # folded stepping: reserve the first token once (the host mirror follows)
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            step_all()  # warm the allocator / plan before capture
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        for wl in ws:
            wl.sync()  # the warm-up advanced the device state; start from the host mirror's (prefilled [+ primed]) cache
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
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / a.steps
    value = world_seen * n_local * a.steps / elapsed  # one token per local sequence per step

    # side figure for N > 1 ("report both with and without the gather"): the same K steps again
    # without the all_gather of token ids; not part of `value`
    ms_no_gather = None
    if dist is not None:
        sync()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            run()
        sync()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_no_gather = float(t.item()) * 1e3 / a.steps

    out = None
    if rank == 0:
        bpl = bytes_per_layer(a)
        us_per_seq_layer = ms_per_step * 1e3 / a.layers / n_local
        out = {
            "metric": f"self-attn decode tokens/s (attention-only), seqlen={a.seqlen} token_budget={a.token_budget}, "
                      "1/2/4/8 GPU",
            "value": value, "unit": "tokens/s", "n_gpus": world_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "stub" if stub else "synthetic",
            "config": {"workload": f"{a.workload_label}; {n_local} sequence(s) per GPU; self-attention chain "
                                   + ("(append + estimate + top-k + sparse attention)" if bpl["sparse"] else
                                      "(append + full-KV attention: the page budget covers the cache)")
                                   + " x all layers per token",
                       "survey_cfg": a.config, "layers": a.layers, "num_qo_heads": a.heads, "num_kv_heads": a.kv_heads,
                       "head_dim": a.head_dim, "seqlen": a.seqlen, "page_size": a.page_size,
                       "token_budget": a.token_budget, "page_budget_pages": a.token_budget // a.page_size,
                       "kv_layout": a.layout, "mode": a.mode, "seed": a.seed, "skip_layers": a.skip_layers,
                       "step_state_advance": ("rides in the last layer's merge launch (quest_decode_arm_step_advance)"
                                              if getattr(w, "fold", False) else "its own launch at the head of the step"),
                       "launches_per_layer": ("3 (append | full-KV attention | merge)" if a.separate_dense_append else
                                              "2 (full-KV attention with the decode append folded in | merge)")
                       if not bpl["sparse"] else "5 (reference op sequence)" if a.unfused else
                       "1 (a workgroup per (sequence, head): append + estimate into LDS + top-k from LDS + sparse attn)"
                       if getattr(w, "one_launch", False) else "3 (append+estimate | top-k+sparse attn | merge)",
                       "sequences_per_gpu": n_local,
                       "multi_sequence": ("batched launches, shared pool" if batched else "one stream per sequence")
                       if n_local > 1 else None,
                       "world_size_seen": world_seen, "gpus_requested": a.gpus,
                       "parallelism": f"sequence-sharded x{world_seen}, all_gather(token ids)"},
            "ms_per_step_without_token_gather": ms_no_gather,
            "selfattn_us_per_layer": us_per_seq_layer,
            "chain_bytes_per_layer": bpl["chain"],
            "chain_frac_of_hbm_peak": bpl["chain"] / (us_per_seq_layer * 1e-6) / 1e9 / HBM_PEAK_GBS,
        }
        if not stub:
            ctl0 = w.ctl.seqs[0] if batched else w.ctl
            out["config"]["seqlen_after_run"] = ctl0.kv_cache.seqlen
            if w.dyn and not w.dense:
                ops, roof = step_op_times(w, a, bpl)
                add_traffic(roof, a, n_local)
                out["roofline"], out["ops_us"] = roof, ops
            dense_us = None
            if not side:
                ref_ops, dense_us = reference_op_times(w, a, bpl)
                out["reference_op_sequence_us"] = ref_ops
            if w.dense and w.dyn and not batched:  # full-KV config: the dense kernel IS the dominant kernel of the timed step
                ctl0.begin_graph_decode(dense_layers=True)  # (reference_op_times re-planned the handlers)
                hd = ctl0._dense_handler
                od = [w.q[l].clone() for l in range(a.layers)]

                fused_append = [True]

                def dense_launch(l):  # the timed step's launch: decode append folded into the group-shared attention
                    if not hd.append_forward_shared_dyn(w.k1[l], w.v1[l], ctl0.metadata_cache.buf_layer(l), ctl0.meta_table_full,
                                                        w.q[l], od[l], ctl0.kv_cache.buf_layer(l), ctl0.kv_table_full,
                                                        ctl0.step_state):
                        fused_append[0] = False  # shapes outside the group-shared kernel: attention alone (the step appends separately)
                        hd.forward_shared_dyn(w.q[l], od[l], ctl0.kv_cache.buf_layer(l), ctl0.kv_table_full, ctl0.step_state)

                t_op = time_kernel_loop(dense_launch, a.layers, 10)
                hd.set_skip_merge(True)
                t_k = time_kernel_loop(dense_launch, a.layers, 10)
                hd.set_skip_merge(False)
                bpl_now = bytes_per_layer(argparse.Namespace(**{**vars(a), "seqlen": ctl0.kv_cache.seqlen}))
                # what was launched decides the name and the bytes: with the append folded in, or (fallback) attention alone --
                # then no instantiation of the group-shared kernel is claimed and no committed traffic figure is looked up
                alg = bpl_now["dense"] + (bpl_now["append"] if fused_append[0] else 0)
                ach = alg / (t_k * 1e-6) / 1e9
                kname = f"shared_decode_kernel<{a.head_dim},{a.heads // a.kv_heads},4,true>" if fused_append[0] else None
                out["roofline"] = {"bound": "hbm", "kernel": (f"{kname} (full-KV decode with the decode append folded in, K/V "
                                   "read once per kv head; the dominant kernel of the timed step, merge launch excluded)"
                                   if kname else "full-KV attention launch of the state-driven step (per-head-list kernel over "
                                   "the one page table; the append is its own launch at this shape)"),
                                   "kernel_name": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
                                   "launch_us": t_k, "op_us_with_merge": t_op,
                                   "sequence_length_at_measurement": ctl0.kv_cache.seqlen}
                add_traffic(out["roofline"], a, n_local)
            if dense_us is not None:
                out["dense_full_kv_us"] = dense_us
                out["dense_gbs"] = bpl["dense"] / (dense_us * 1e-6) / 1e9
                out["speedup_vs_dense"] = dense_us / us_per_seq_layer
            if "ops_us" in out and "batched_dense_full_kv_us_per_sequence" in out["ops_us"]:
                out["speedup_vs_batched_dense"] = out["ops_us"]["batched_dense_full_kv_us_per_sequence"] / us_per_seq_layer
            if world_seen == 1 and not a.no_cpu_baseline and not side:
                out["cpu_baseline"] = cpu_baseline(a, w, a.cpu_sample_s)
            else:
                out["cpu_baseline"] = None
        else:
            out["roofline"], out["cpu_baseline"] = None, None
    return out


def prefill_side(a, dev):
    """Side object of the default N = 1 run: the prefill attention operator (prefill_with_paged_kv_cache -> the MFMA flash
    kernel of csrc/prefill.hip) on the headline's shapes -- the whole seqlen-token prompt of one layer, causal, straight
    over the paged cache -- timed with events on the launch stream.  MFMA-bound: flops = 4 D per visible (query, key) pair
    and head; peak = the dense fp16 matrix peak of MI355X_MICROARCH.md.  Not part of `value`."""
    import torch

    import quest_amd.utils as qu

    L, Hq, Hkv, D = a.seqlen, a.heads, a.kv_heads, a.head_dim
    ctl = qu.InferenceController(1, Hq, D, a.page_size, 128, L + 4 * a.page_size, torch.float16, dev, num_kv_heads=Hkv)
    g = torch.Generator(device=dev).manual_seed(a.seed)
    k = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    q = torch.randn(L, Hq, D, generator=g, device=dev, dtype=torch.float16)
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    qu.append_kv(k, v, ctl, 0)
    del k, v
    o = qu.prefill_forward(q, ctl, 0)
    torch.cuda.synchronize()
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = qu.prefill_forward(q, ctl, 0)
    e1.record()
    torch.cuda.synchronize()
    ctl.end_forward()
    ms = e0.elapsed_time(e1) / reps
    assert torch.isfinite(o[-1].float()).all()
    flops = 4.0 * D * Hq * (L * L - (L * (L - 1)) // 2)
    tf = flops / (ms * 1e-3) / 1e12
    return {"workload": f"prefill attention of one layer, causal, {L} tokens x {Hq} heads x {D} (kv heads {Hkv}), paged cache",
            "kernel": f"prefill_kernel<{D},{'true' if a.page_size == 16 else 'false'}>", "launches": reps, "ms": ms,
            "flops": flops, "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0},
            "note": "measured after the headline timing in the same process; not part of `value`"}


_LINE_OUT = None  # the process's real stdout once main() has parked it (see there)


def main():
    a = parse()
    if a.kernel_sweep:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import kbench_reference_rows as sweep

        path = os.path.join(ROOT, "gpurun_out", "kernel_sweep.json")
        res = sweep.run(out_path=path)
        print(sweep.markdown(res), file=sys.stderr)
        print(json.dumps({"kernel_sweep": os.path.relpath(path, ROOT), "rows": res["rows"]}), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner from C
    # when the first communicator is created): park the real stdout for that line and point fd 1 at stderr for the rest
    # of the run.
    global _LINE_OUT
    sys.stdout.flush()
    line_out = _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("QUEST_BENCH_STUB") == "1"
    if stub and os.environ.get("QUEST_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(3)  # tests/test_parallel_gloo.py: a peer that exits before the rendezvous
    rehearse = os.environ.get("QUEST_BENCH_REHEARSE") == "1" or stub
    if world > 1 or os.environ.get("QUEST_BENCH_FORCE_DIST") == "1":  # the latter: rehearse the RCCL path on 1 GPU
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            # rehearsal of the N > 1 control flow on a ONE-GPU (or, with the stub, CPU-only) box: gloo collectives
            local = 0
            if not stub:
                torch.cuda.set_device(0)
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S))
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                    timeout=datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S))
        world_seen = dist.get_world_size()
    else:
        dist = None
        world_seen = 1
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the quest_amd operators have no CPU fallback")
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    out = measure(a, dev, dist, world_seen, rank, stub)

    # ---- side configurations, measured AFTER the headline and outside `value` (the headline stays cfg 3 x 1 sequence per
    # GPU so that N = 1 equals BENCH): where the fixed costs of a launch are shared by 8 sequences per GPU -- the per-GPU
    # load of BASELINE configs[4].  N = 1: the headline shapes x 8 sequences (MHA) and configs[4]'s per-GPU load (GQA).
    # N > 1: EVERY rank runs configs[4]'s per-GPU load with the per-step all_gather of its 8 token
    # ids, so that a SCALE record shows the configuration BASELINE names -- 8 GQA sequences per GPU x N GPUs -- as
    # aggregate tokens/s and per-GPU chain fraction.  A failure here must not cost the headline line: it is printed in
    # any case.
    try:
        if (a.config == 3 and not a.no_side and "custom" not in a.workload_label and a.mode == "graph"):
            import gc

            sides = [("batched_8seq", dict(config=3, seqs_per_gpu=8)), ("cfg5_8seq_gqa", dict(config=5))]
            if dist is not None:
                sides = sides[1:]
            elif a.layout != "NHD":
                # the same three measurements on the REFERENCE's pool layout (a headline on another layout
                # carries the NHD figure beside it)
                sides += [("reference_layout_nhd", dict(config=3, layout="NHD")),
                          ("batched_8seq_nhd", dict(config=3, seqs_per_gpu=8, layout="NHD")),
                          ("cfg5_8seq_gqa_nhd", dict(config=5, layout="NHD"))]
            for name, overrides in sides:
                gc.collect()
                if not stub:
                    torch.cuda.empty_cache()
                side_warmup = a.side_warmup if a.side_warmup is not None else max(a.warmup, 20)
                a2 = parse(["--config", str(overrides["config"]), "--steps", str(a.side_steps), "--warmup", str(side_warmup),
                            "--seed", str(a.seed), "--no-cpu-baseline", "--gpus", str(a.gpus),
                            "--layer-launches", a.layer_launches, "--layout", overrides.get("layout", a.layout)]
                           + (["--seqs-per-gpu", str(overrides["seqs_per_gpu"])] if "seqs_per_gpu" in overrides else []))
                try:
                    full = measure(a2, dev, dist, world_seen, rank, stub, side=True)
                except RuntimeError as exc:  # the headline line must survive a side measurement that fails (e.g. memory)
                    # (under N > 1 a workload that cannot be built makes EVERY rank raise here, at the same point -- see
                    # measure(): nobody is left inside a collective; any other exception leaves through the outer handler)
                    if dist is not None and "failed to build its workload" not in str(exc) and not str(exc).startswith("rank "):
                        raise
                    if out is not None:
                        out[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    continue
                except Exception as exc:
                    if dist is not None:
                        raise  # the peers may be inside collectives: leave through the outer handler
                    out[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    continue
                if out is None:
                    continue  # ranks > 0 took part in the measurement; rank 0 reports it
                roof, ops = full.get("roofline") or {}, full.get("ops_us") or {}
                out[name] = {
                    "workload": full["config"]["workload"], "sequences_per_gpu": full["config"]["sequences_per_gpu"],
                    "kv_layout": full["config"]["kv_layout"],
                    "n_gpus": full["n_gpus"], "steps": full["steps"], "tokens_per_s": full["value"],
                    "ms_per_step": full["ms_per_step"],
                    "ms_per_step_without_token_gather": full.get("ms_per_step_without_token_gather"),
                    "us_per_sequence_layer": full["selfattn_us_per_layer"],
                    "chain_frac_of_hbm_peak": full["chain_frac_of_hbm_peak"],
                    "dominant_kernel": roof.get("kernel"), "dominant_kernel_launch_us": roof.get("launch_us"),
                    "dominant_kernel_frac_algorithmic": roof.get("frac"), "dominant_kernel_frac_hbm_pmc": roof.get("frac_hbm_pmc"),
                    "dominant_kernel_traffic_bytes": roof.get("traffic"), "traffic_source": roof.get("traffic_source"),
                    "launches_per_layer": full["config"]["launches_per_layer"],
                    "two_launch_form_us": ([ops.get("two_launch_form_append_estimate_us"),
                                            ops.get("two_launch_form_topk_sparse_attn_us")]
                                           if "layer_us" in ops else None),
                    "append_estimate_us": ops.get("append_estimate_us"),
                    "append_estimate_frac_of_hbm_peak": ops.get("append_estimate_frac_of_hbm_peak"),
                    "batched_dense_full_kv_us_per_sequence": ops.get("batched_dense_full_kv_us_per_sequence"),
                    "speedup_vs_batched_dense": full.get("speedup_vs_batched_dense"),
                    "speedup_vs_dense": full.get("speedup_vs_dense"),
                    "speedup_vs_single_sequence_dense": (out["dense_full_kv_us"] / full["selfattn_us_per_layer"]
                                                         if overrides["config"] == 3 and out.get("dense_full_kv_us") else None),
                    "note": ("measured after the headline timing in the same process(es); not part of `value`"
                             + ("; every rank decodes 8 sequences, each step ends with the all_gather of the token ids; "
                                "tokens_per_s is the aggregate over all ranks, the fractions are per GPU (rank 0's)"
                                if dist is not None else "")),
                }
        if (dist is None and out is not None and not stub and not a.no_side and a.config == 3
                and "custom" not in a.workload_label):
            try:
                out["prefill_attention"] = prefill_side(a, dev)
            except Exception as exc:  # never at the cost of the headline line
                out["prefill_attention"] = {"error": f"{type(exc).__name__}: {exc}"}
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as exc:
        if out is not None:
            out["side_error"] = f"{type(exc).__name__}: {exc}"
        else:
            raise
    finally:
        if out is not None:
            print(json.dumps(out), file=line_out, flush=True)


if __name__ == "__main__":
    try:
        main()
    except Exception as exc:  # one machine-readable line for whoever parses stdout, then the traceback as usual
        if os.environ.get("RANK", "0") == "0":
            print(json.dumps({"error": f"{type(exc).__name__}: {exc}", "rank": 0}), file=_LINE_OUT or sys.stdout, flush=True)
        raise
