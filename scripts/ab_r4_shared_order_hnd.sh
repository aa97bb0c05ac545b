set -o pipefail
O=$PWD/gpurun_out; R=$O/r04_ab_shared_grid_order_hnd.txt; : > $R
export QUEST_TUNING=1
for rep in 1 2; do
for hf in 0 1; do
  for spec in "2 1" "3 1" "3 8"; do
    set -- $spec
    QUEST_SHARED_HEADS_FIRST=$hf python bench.py --config $1 --seqs-per-gpu $2 --layout HND --no-side --no-cpu-baseline > $O/ab_order.json 2> $O/ab_order.err || { tail -5 $O/ab_order.err; exit 1; }
    python - "$hf" "$1" "$2" >> $R <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_order.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}; o=d.get("ops_us") or {}
print("HND heads_first", sys.argv[1], "cfg", sys.argv[2], "seqs", sys.argv[3], "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "dense_full_kv_us", d.get("dense_full_kv_us"), "batched_dense/seq", o.get("batched_dense_full_kv_us_per_sequence"), "roof_launch_us", r.get("launch_us"))
PY
  done
done
done
cat $R
