set -e
for ppc in 0 8 16 32 64; do
  echo "== ppc $ppc"
  timeout -k 10 240 python bench.py --seqs-per-gpu 8 --layers 8 --steps 30 --warmup 5 --no-cpu-baseline --no-dense --pages-per-chunk $ppc 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['selfattn_us_per_layer'], d['chain_frac_of_hbm_peak'], d['roofline']['plan'])
"
done
echo "== streams"
timeout -k 10 240 python bench.py --seqs-per-gpu 8 --layers 8 --steps 30 --warmup 5 --no-cpu-baseline --no-dense --multi-seq-mode streams 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['selfattn_us_per_layer'], d['chain_frac_of_hbm_peak'])
"
