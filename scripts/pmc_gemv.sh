#!/bin/bash
# HBM bytes (FETCH_SIZE, gfx950 x2 correction) and texture-addresser / L2 counters of the decoder-layer projection kernels
# at 8 tokens next to the batch-1 kernels (scripts/gemv_bench.py --tokens 8), one rocprofv3 --pmc pass per counter group.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_gemv; rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d $O/$tag -o r -- python3 scripts/gemv_bench.py --tokens 8 > $O/$tag.log 2> $O/$tag.err || { tail -3 $O/$tag.err; }
done
python3 - <<'PY'
import csv, glob, statistics
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/prof_gemv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemv_kernel" in k or "persist_kernel" in k or "skinny_kernel" in k:
            vals[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print(k, {c: round(statistics.median(v), 1) for c, v in sorted(vals[k].items())}, "n", len(next(iter(vals[k].values()))))
PY
