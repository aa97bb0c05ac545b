#!/usr/bin/env python3
"""Phase timeline of one workgroup of the fused top-k + sparse attention kernel (cfg-3 shapes).
Builds a -DQUEST_TIMELINE variant of the library next to the normal one and loads it via QUEST_HIP_LIB.

    python scripts/timeline.py --build      (here, cross-compiles)
    python scripts/timeline.py              (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "quest_amd", "libquest_hip_timeline.so")

if "--build" in sys.argv:
    from quest_amd.build import build_variant
    # TL_FIRST_HEAD=1: stamp a workgroup of head 0 (first workgroup of its CU) instead of the middle head
    print(build_variant(VARIANT, ["-DQUEST_TIMELINE"] + (["-DQUEST_TL_FIRST_HEAD"] if os.environ.get("TL_FIRST_HEAD") else [])))
    sys.exit(0)

os.environ["QUEST_HIP_LIB"] = VARIANT
import torch  # noqa: E402

import bench  # noqa: E402

cfg_args = ["--config", os.environ.get("TL_CONFIG", "3")]
sys.argv = [sys.argv[0]]
a = bench.parse(cfg_args)
a.mode, a.layers = "graph-static", 4
dev = torch.device("cuda", 0)
w = bench.Workload(a, dev)
from quest_amd import _kernels  # noqa: E402
from quest_amd._lib import lib, check  # noqa: E402

qu, ctl = w.qu, w.ctl
ctl.set_page_budget(w.page_budget)
ctl.begin_forward(1)
est = [qu.decode_estimate(w.q[l], ctl, l) for l in range(a.layers)]
names = ["entry", "loads issued+hist cleared", "scores arrived, keys in LDS", "barrier", "topk_select done",
         "page list in LDS (barrier)", "all K/V folded", "row butterfly + LDS write", "barrier", "partial written"]
h = ctl._decode_handler
# state-driven twin of the same sequence (8-wave workgroups, lengths from the device state)
a2 = bench.parse(cfg_args)
a2.mode, a2.layers = "graph", a.layers
w2 = bench.Workload(a2, dev)
c2 = w2.ctl
qu.step_advance_dyn(c2)
sc2 = [qu.score_scratch(c2) for _ in range(a.layers)]
for l in range(a.layers):
    _kernels.append_estimate_dyn(w2.k1[l], w2.v1[l], c2.kv_cache.buf_layer(l), c2.kv_table_full, w2.q[l], sc2[l],
                                 c2.metadata_cache.buf_layer(l), c2.meta_table_full, c2.step_state, c2.max_pages - 1,
                                 c2.layout)
h2 = c2._decode_handler
variants = ("dyn",) if a.seqlen > 65536 else ("dyn", True, False)
for fused in variants:
    acc = torch.zeros(a.layers * 4, 32, device=dev)
    for rep in range(4):
        for l in range(a.layers):
            lse = acc[rep * a.layers + l]
            q = w.q[l]
            o = torch.empty_like(q)
            if fused == "dyn":
                kv = _kernels._paged(c2.kv_cache.buf_layer(l), c2.kv_table_full, None, 1, 0, c2.layout)
                check(lib.quest_decode_forward_fused_topk_dyn(h2._wrapper._h, w2.q[l].data_ptr(), o.data_ptr(), kv, q.size(1),
                                                              sc2[l].data_ptr(), sc2[l].size(1), c2.max_pages - 1,
                                                              c2.step_state.data_ptr(), lse.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "dyn")
            elif fused:
                kv = _kernels._paged(ctl.kv_cache.buf_layer(l), ctl.kv_indices_with_last, None, ctl.kv_cache.last_page_len,
                                     ctl.kv_last_page_idx, ctl.layout)
                check(lib.quest_decode_forward_fused_topk(h._wrapper._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), est[l].data_ptr(),
                                                          est[l].size(1), None, None, lse.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "fused")
            else:
                qu.decode_topk(est[l], ctl)
                idx = ctl.topk_dindices_buffer
                kv = _kernels._paged(ctl.kv_cache.buf_layer(l), idx, None, ctl.kv_cache.last_page_len,
                                     ctl.kv_last_page_idx, ctl.layout, page_budget=idx.size(1))
                check(lib.quest_decode_forward(h._wrapper._h, q.data_ptr(), o.data_ptr(), kv, q.size(1), lse.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "plain")
    torch.cuda.synchronize()
    t = acc[a.layers:].cpu().median(dim=0).values  # skip the first (cold) round
    cyc_per_us = float(t[9] / (t[10] / 100.0)) if t[10] > 0 else float("nan")
    label = {"dyn": "state-driven, default front end", True: "host-planned fused launch",
             False: "index list from memory (FC = 0)"}[fused]
    info = (h2 if fused == "dyn" else h).last_launch_info()
    print(f"== {label}: variant {info['front_end_variant']}, {info['workgroups_per_head']} workgroups per head; "
          f"{cyc_per_us:.0f} cycles/us, workgroup lifetime {float(t[9]) / cyc_per_us:.2f} us")
    for i, nme in enumerate(names):
        print(f"  {float(t[i]) / cyc_per_us:6.2f} us  {nme}")
    if info["front_end_variant"] == 2:
        subn = ["keys + range published", "barrier A", "hist atomics issued", "barrier B", "threshold (per wave)",
                "bitmaps written", "barrier D", "ranks scanned"]
        for i, nme in enumerate(subn):
            print(f"      {float(t[16 + i]) / cyc_per_us:6.2f} us  fe2_select: {nme}")
    elif fused:
        subn = ["hist1 atomics issued", "barrier", "bins read + summed", "block scan", "threshold bin published (barrier)",
                "hist2 + barrier", "exact T (wave 0) + barrier", "gt/eq counted", "block scan 2"]
        for i, nme in enumerate(subn):
            print(f"      {float(t[16 + i]) / cyc_per_us:6.2f} us  topk_select: {nme}")
