#!/bin/bash
# SQ counters of the chain's kernels (one pass, <= 8 SQ counters): where the waves' cycles go.  Run on the GPU box.
set -e
tag=${SQ_TAG:-sq}; out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out  # SQ_TAG / QUEST_HIP_LIB select a build variant
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/a -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense "$@" > /dev/null 2> $out/a.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/b -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense "$@" > /dev/null 2> $out/b.err
python3 - <<PY
import csv, glob, statistics, json
from collections import defaultdict
res = defaultdict(lambda: defaultdict(list))
for d in ("a","b"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("estimate_kernel", "sparse_decode_kernelILi128ELi16ELi8ELi8ELi3", "sparse_decode_kernelILi128ELi16ELi0ELi4", "merge_states", "topk_kernel"):
                if key in k:
                    res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(out, open("$out/sq_summary.json", "w"), indent=1)
for k, cs in out.items():
    w = cs.get("SQ_WAVES", 1)
    print(k, {c: round(v / w, 1) for c, v in cs.items()}, "waves", w)
PY
rm -rf $out/a $out/b
