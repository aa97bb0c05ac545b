#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per counter) into per-kernel medians.
usage: pmc_summary.py OUT.json DIR_FETCH DIR_WRITE   (gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2)"""
import csv
import glob
import json
import statistics
import sys
from collections import defaultdict


def load(d, counter):
    vals = defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals


def short(name):
    for key in ("layer_decode_kernel", "estimate_kernel", "sparse_decode_kernel", "shared_decode_kernel", "merge_states", "topk_kernel",
                "append_decode", "append_prefill", "step_state_advance", "rope_kernel"):
        if key in name:
            i = name.find(key)
            j = name.find("(", i)
            return name[i:j if j > 0 else i + 60]
    return None


out, fetch, write = sys.argv[1], load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
rows = []
for k in sorted(fetch):
    s = short(k)
    if s is None:
        continue
    f = statistics.median(fetch[k])
    w = statistics.median(write.get(k, [0.0]))
    rows.append({"kernel": s, "dispatches": len(fetch[k]), "FETCH_SIZE_KB_median_raw": f, "WRITE_SIZE_KB_median": w,
                 "hbm_read_bytes_corrected": int(2 * f * 1024), "hbm_write_bytes": int(w * 1024)})
json.dump({"correction": "gfx950: read bytes = 2 x FETCH_SIZE KB x 1024 (MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE as reported", "kernels": rows}, open(out, "w"), indent=1)
for r in rows:
    print(r)
