#!/usr/bin/env python3
"""Per-workgroup wall-clock trace of the chained launch (append + estimate + top-k + attention in one grid).
Builds a -DQUEST_CHAIN_TRACE variant of the library next to the normal one and loads it via QUEST_HIP_LIB.

    python scripts/chain_trace.py --build [extra -D flags]    (here, cross-compiles)
    CT_CONFIG=3 QUEST_CHAIN_LEAD=2 python scripts/chain_trace.py   (on the GPU box)

Prints, per role and head group, when its workgroups started, got past the wait (attention) / finished their work
(estimate), and ended -- in us from the first start of the launch (100 MHz clock: 0.01 us resolution).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "quest_amd", "libquest_hip_chaintrace.so")

if "--build" in sys.argv:
    from quest_amd.build import build_variant
    print(build_variant(VARIANT, ["-DQUEST_CHAIN_TRACE"] + [f for f in sys.argv[1:] if f.startswith("-D")]))
    sys.exit(0)

os.environ["QUEST_HIP_LIB"] = os.environ.get("CT_LIB", VARIANT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

a = bench.parse(["--config", os.environ.get("CT_CONFIG", "3")])
a.mode, a.layers = "graph", 8
dev = torch.device("cuda", 0)
w = bench.Workload(a, dev)
from quest_amd._lib import lib, check  # noqa: E402

qu, ctl = w.qu, w.ctl
h = ctl._decode_handler
scores = qu.score_scratch(ctl)
N = 4096
trace = np.zeros((N, 4), np.int64)
rows = []
for rep in range(6):
    qu.step_advance_dyn(ctl)
    for l in range(a.layers):
        o = torch.empty_like(w.q[l])
        assert h.chain_decode_dyn(w.k1[l], w.v1[l], w.q[l], o, ctl.kv_cache.buf_layer(l), ctl.kv_table_full,
                                  ctl.metadata_cache.buf_layer(l), ctl.meta_table_full, scores, ctl.step_state,
                                  ctl.max_pages - 1)
    torch.cuda.synchronize()
    ctl.prepare_metadata(1)
    check(lib.quest_chain_trace(h._wrapper._h, trace.ctypes.data, N), "trace")
    rows.append(trace.copy())
assert h.chain_error() == 0
t = rows[-1]  # last layer of the last step
used = t[:, 1] > 0
t = t[used]
role, group = t[:, 0] >> 32, t[:, 0] & 0xffffffff
t0 = t[:, 1].min()
us = (t[:, 1:] - t0) / 100.0
print(f"workgroups {used.sum()}, launch span {us[:, 2].max():.2f} us")
names = {0: "attention", 1: "estimate", 2: "append"}
second = {0: "past wait", 1: "work done", 2: "work done"}
for r in (2, 1, 0):
    for g in sorted(set(group[role == r])):
        m = (role == r) & (group == g)
        x = us[m]
        print(f"{names[r]:9s} group {g}: n={m.sum():4d}  start {x[:, 0].min():6.2f}..{x[:, 0].max():6.2f}   "
              f"{second[r]} {x[:, 1].min():6.2f}..{np.median(x[:, 1]):6.2f}..{x[:, 1].max():6.2f}   "
              f"end {x[:, 2].min():6.2f}..{np.median(x[:, 2]):6.2f}..{x[:, 2].max():6.2f}")

# durations
for r in (1, 0):
    m = role == r
    x = us[m]
    first, second_, = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1]
    order = np.argsort(x[:, 0])
    print(f"{names[r]}: start->{second[r]} p10/p50/p90/max {np.percentile(first, 10):.2f} {np.percentile(first, 50):.2f} "
          f"{np.percentile(first, 90):.2f} {first.max():.2f};  ->end {np.percentile(second_, 10):.2f} "
          f"{np.percentile(second_, 50):.2f} {np.percentile(second_, 90):.2f} {second_.max():.2f}")
    # by start-time decile
    for q in range(0, 10, 1):
        sel = order[len(order) * q // 10: len(order) * (q + 1) // 10]
        print(f"   start {x[sel, 0].min():6.2f}..{x[sel, 0].max():6.2f}: {second[r]} after {np.median(first[sel]):5.2f} (max {first[sel].max():5.2f}), "
              f"end after {np.median(second_[sel]):5.2f} more (max {second_[sel].max():5.2f})")
