#!/bin/bash
# Matrix-pipe, vector-issue and LDS counters of the prefill kernel (scripts/prefill_bench.py --lens $1, default 16384), one
# rocprofv3 --pmc pass per counter group, plus a --kernel-trace --stats pass.   bash scripts/pmc_prefill.sh [LEN] [TAG]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
LEN=${1:-16384}; TAG=${2:-prefill}
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 scripts/prefill_bench.py --lens $LEN > $O/stats.log 2> $O/stats.err || tail -3 $O/stats.err
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d $O/$tag -o r -- python3 scripts/prefill_bench.py --lens $LEN > $O/$tag.log 2> $O/$tag.err || { tail -3 $O/$tag.err; }
done
python3 - $O <<'PY'
import csv, glob, statistics, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "prefill_kernel" in r["Kernel_Name"]:
            vals[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals, key=int):
    print("grid", k, {c: round(statistics.median(v), 1) for c, v in sorted(vals[k].items())})
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "prefill" in r["Name"] or "attn" in r["Name"].lower() or "fmha" in r["Name"].lower():
            print(r["Name"][:90], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
