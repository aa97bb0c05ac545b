"""Per-op timing of the single-sequence state-driven launches (cfg-3 shapes) next to the by-value ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]]
a = bench.parse()
a.layers = 32
dev = torch.device("cuda:0")
w = bench.Workload(a, dev)  # mode graph -> state-driven
from quest_amd import _kernels  # noqa: E402
import quest_amd.utils as qu  # noqa: E402

ctl = w.ctl
qu.step_advance_dyn(ctl)
max_n = ctl.max_pages - 1


def ae(l):
    _kernels.append_estimate_dyn(w.k1[l], w.v1[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.q[l], w.scores,
                                 ctl.metadata_cache.buf_layer(l), ctl.meta_table_full, ctl.step_state, max_n, ctl.layout)


outs = [torch.empty_like(w.q[0]) for _ in range(a.layers)]


def ts(l):
    ctl._decode_handler.forward_fused_topk_dyn(w.q[l], outs[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.scores,
                                               ctl.step_state, max_n)


def both(l):
    ae(l)
    ts(l)


print(f"capacity pages {ctl.max_pages}, live n {len(ctl.kv_cache.indicies)}")
print(f"append+estimate dyn: {bench.time_kernel_loop(ae, a.layers, 10):.2f} us")
print(f"topk+attn(+merge) dyn: {bench.time_kernel_loop(ts, a.layers, 10):.2f} us")
print(f"layer dyn: {bench.time_kernel_loop(both, a.layers, 10):.2f} us")

# the bench's whole step (advance + 32 layers) as one graph, timed by HIP events over back-to-back replays
ctl.sync_device_state()
w.prime()  # folded stepping (round 6): the first token's reservation, once; every step then ends with the next one's
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    w.step_dyn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
ctl.sync_device_state()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    w.step_dyn()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"whole step graph: {e0.elapsed_time(e1) * 1e3 / 20 / a.layers:.2f} us per layer (HIP events, 20 back-to-back replays)")
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"whole step graph: {(time.perf_counter() - t0) * 1e6 / 20 / a.layers:.2f} us per layer (host clock)")
