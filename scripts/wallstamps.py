#!/usr/bin/env python3
"""Per-workgroup wall-clock stamps of the fused top-k + attention launch in the regime the bench measures (a hipGraph of
one launch per layer, each on its own pool: cold caches).  Builds a -DQUEST_WALLSTAMPS variant of the library, loads it
via QUEST_HIP_LIB, replays the graph and reads the stamps every workgroup of the LAST layer's launch left in its
partial-state record.

    python scripts/wallstamps.py --build            (here, cross-compiles)
    python scripts/wallstamps.py [--front-end N] [--config N] [--ppc N] [--seqs N]
                                                    (on the GPU box; N = quest_decode_set_front_end value, default 0;
                                                     --seqs N: the batched launches of N sequences per GPU)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "quest_amd", "libquest_hip_wallstamps.so")

if "--build" in sys.argv:
    from quest_amd.build import build_variant
    print(build_variant(VARIANT, ["-DQUEST_WALLSTAMPS"]))
    sys.exit(0)

os.environ["QUEST_HIP_LIB"] = VARIANT
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from quest_amd._lib import check, lib  # noqa: E402

fe = int(sys.argv[sys.argv.index("--front-end") + 1]) if "--front-end" in sys.argv else 0
cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "3"
ppc = int(sys.argv[sys.argv.index("--ppc") + 1]) if "--ppc" in sys.argv else 0
seqs = int(sys.argv[sys.argv.index("--seqs") + 1]) if "--seqs" in sys.argv else 1
a = bench.parse(["--config", cfg, "--steps", "100" if seqs > 1 else "300"] + (["--pages-per-chunk", str(ppc)] if ppc else [])
                + (["--seqs-per-gpu", str(seqs)] if seqs > 1 else [])
                + (sys.argv[sys.argv.index("--bench-args") + 1].split() if "--bench-args" in sys.argv else []))
dev = torch.device("cuda", 0)
from quest_amd import _kernels  # noqa: E402

if a.seqs_per_gpu > 1:
    seqs = a.seqs_per_gpu
    w = bench.BatchedWorkload(a, dev, seqs)
    ctl = w.ctl
    h = ctl._decode_handler
    h.set_front_end(fe)
    max_n = ctl.max_pages - 1
    w.qu.step_advance_batched(ctl)

    one_launch = "--one-launch" in sys.argv  # the one-launch layer (csrc/layer_device.cuh) instead of the pair

    def layer(l):  # the step's pair of launches (the scores must be this layer's)
        if one_launch:
            assert h.layer_fused_batched(w.k1[l], w.v1[l], ctl.metadata_layer(l), ctl.meta_tables, w.q[l], w.o[l],
                                         ctl.kv_layer(l), ctl.kv_tables, ctl.step_states, max_n)
            return
        _kernels.append_estimate_batched(w.k1[l], w.v1[l], ctl.kv_layer(l), ctl.kv_tables, w.q[l], w.scores,
                                         ctl.metadata_layer(l), ctl.meta_tables, ctl.step_states, max_n, ctl.layout)
        h.forward_fused_topk_batched(w.q[l], w.o[l], ctl.kv_layer(l), ctl.kv_tables, w.scores, ctl.step_states, max_n)
else:
    w = bench.Workload(a, dev)
    ctl = w.ctl
    h = ctl._decode_handler
    h.set_front_end(fe)
    max_n = ctl.max_pages - 1
    o = [w.q[l].clone() for l in range(a.layers)]
    w.qu.step_advance_dyn(ctl)

    tiles = "--tiles" in sys.argv  # the tiles launches (tile maxima from the estimate to the attention launch)

    def layer(l):  # the step's pair of launches (the scores must be this layer's)
        assert _kernels.append_estimate_dyn(w.k1[l], w.v1[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.q[l], w.scores,
                                            ctl.metadata_cache.buf_layer(l), ctl.meta_table_full, ctl.step_state, max_n,
                                            ctl.layout, tiles=tiles)
        assert h.forward_fused_topk_dyn(w.q[l], o[l], ctl.kv_cache.buf_layer(l), ctl.kv_table_full, w.scores, ctl.step_state,
                                        max_n, tiles=tiles)


h.set_skip_merge(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for l in range(a.layers):
        layer(l)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for l in range(a.layers):
        layer(l)
info = h.last_launch_info()
ptr, nbytes, rec = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint32()
check(lib.quest_decode_debug_workspace(h._wrapper._h, ctypes.byref(ptr), ctypes.byref(nbytes), ctypes.byref(rec)), "ws")
hip = ctypes.CDLL("libamdhip64.so")
H, C, R, D = a.heads * seqs, info["workgroups_per_head"], rec.value, a.head_dim
host = np.empty(H * C * R, np.float32)
rows = []
for rep in range(12):
    g.replay()
    torch.cuda.synchronize()
    assert hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ptr, ctypes.c_size_t(host.nbytes), 2) == 0
    u = host.view(np.uint32).reshape(H, C, R)[:, :, D + 2:D + 8].astype(np.int64)
    if rep >= 2:
        rows.append(u.copy())
print(f"== cfg {cfg}, {seqs} sequence(s) per launch, front end {fe}: variant {info['front_end_variant']}, {info['waves']} waves, {C} workgroups per head "
      f"(last layer's launch, {len(rows)} replays; 10 ns ticks -> us)")
for name, fn in ((("kernel span (last end - first entry)", lambda u, t0: (u[..., 3].max() - t0)),
                 ("entry of the last workgroup to start", lambda u, t0: (u[..., 0].max() - t0)),
                 ("median page list known", lambda u, t0: np.median(u[..., 1] - t0)),
                 ("  latest page list known", lambda u, t0: (u[..., 1].max() - t0)),
                 ("median wave-0 pages folded", lambda u, t0: np.median(u[..., 2] - t0)),
                 ("  latest wave-0 pages folded", lambda u, t0: (u[..., 2].max() - t0)),
                 ("median partial written", lambda u, t0: np.median(u[..., 3] - t0)))
                + ((("median wave-0 scores done (one-launch layer)", lambda u, t0: np.median(u[..., 5] - t0)),
                    ("  latest wave-0 scores done", lambda u, t0: (u[..., 5].max() - t0)),
                    ("median end", lambda u, t0: np.median(u[..., 3] - t0)),
                    ("  p10 / p90 end", lambda u, t0: np.percentile(u[..., 3] - t0, 10) + 1j * np.percentile(u[..., 3] - t0, 90)))
                   if info["front_end_variant"] == 7 else ())):
    vals = [fn(u, u[..., 0].min()) * 0.01 for u in rows]
    if np.iscomplexobj(np.asarray(vals)):
        print(f"  {name:42s} {np.median([v.real for v in vals]):6.2f} / {np.median([v.imag for v in vals]):6.2f} us")
        continue
    print(f"  {name:42s} {np.median(vals):6.2f} us  (min {min(vals):.2f}, max {max(vals):.2f})")
u = rows[-1]
t0 = u[..., 0].min()
end = (u[..., 3] - t0) * 0.01
pages = u[..., 4]
fe_t = (u[..., 1] - u[..., 0]) * 0.01
ga_t = (u[..., 2] - u[..., 1]) * 0.01
print("  by pages of the workgroup: count, mean front end, mean gather (wave 0), mean end")
for p in sorted(set(pages.ravel().tolist())):
    m = pages == p
    print(f"    {p:3d} pages: {int(m.sum()):4d} workgroups  fe {fe_t[m].mean():5.2f}  gather {ga_t[m].mean():5.2f}  end {end[m].mean():5.2f}  max end {end[m].max():5.2f}")
order = np.argsort(end.ravel())[::-1][:8]
print("  last workgroups to finish (head, chunk, pages, entry, page list, folded, end):")
for i in order:
    hq, c = divmod(int(i), C)
    r = u[hq, c]
    print(f"    ({hq:2d},{c:2d}) {int(r[4]):3d}  {(r[0]-t0)*0.01:5.2f} {(r[1]-t0)*0.01:5.2f} {(r[2]-t0)*0.01:5.2f} {(r[3]-t0)*0.01:5.2f}")
if "--per-head" in sys.argv:  # front end per head, mean over the head's workgroups and the replays: data- or placement-dependent?
    fe_all = np.stack([(x[..., 1] - x[..., 0]) * 0.01 for x in rows])  # [replay][head][chunk]
    per_head = fe_all.mean(axis=(0, 2))
    spread = fe_all.mean(axis=2).std(axis=0)
    print("  front end per head (us, mean over workgroups and replays; std over replays):")
    for h0 in range(0, H, 8):
        print("    " + "  ".join(f"{h0 + i:3d}: {per_head[h0 + i]:5.2f}±{spread[h0 + i]:.2f}" for i in range(min(8, H - h0))))
    per_chunk = fe_all.mean(axis=(0, 1))
    print("  front end per chunk index (us): " + " ".join(f"{x:.2f}" for x in per_chunk))
if "--matrix" in sys.argv and seqs > 1:  # medians over the replays, one row per sequence, one column per head
    allr = np.stack(rows)  # [replay][seq*head][chunk][field]
    t0s = allr[..., 0].min(axis=(1, 2))[:, None, None]
    for name, col in (("scores done (wave 0)", 5), ("page list known", 1), ("end", 3)):
        if col == 5 and info["front_end_variant"] != 7:
            continue
        m = np.median((allr[..., col] - t0s) * 0.01, axis=0)[:, 0].reshape(seqs, a.heads)
        print(f"  {name}, us after the first entry (rows = sequences, columns = heads):")
        for r in m:
            print("    " + " ".join(f"{x:5.1f}" for x in r))
        print("    column means: " + " ".join(f"{x:5.1f}" for x in m.mean(axis=0)))
if "--per-head" in sys.argv:  # end of the gather / of the workgroup per head: is a head's address class (NHD: bits 8-9 = head mod 4) visible?
    allr = np.stack(rows)
    t0s = allr[..., 0].min(axis=(1, 2))[:, None, None]
    for name, col in (("wave-0 pages folded", 2), ("end", 3)):
        m = np.median((allr[..., col] - t0s) * 0.01, axis=0)  # [head][chunk]
        print(f"  {name} per head (us after the first entry; mean / max over the head's workgroups):")
        for h0 in range(0, H, 8):
            print("    " + "  ".join(f"{h0 + i:3d}: {m[h0 + i].mean():5.2f}/{m[h0 + i].max():5.2f}" for i in range(min(8, H - h0))))
