import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    r=d.get("roofline") or {}
    o=d.get("ops_us") or {}
    print(f.split('/')[-1], "us/layer", round(d["selfattn_us_per_layer"],2), "chainfrac", round(d["chain_frac_of_hbm_peak"],3), "spd", round(d.get("speedup_vs_dense") or 0,2), "spdB", round(d.get("speedup_vs_batched_dense") or 0,2),
          "| A+E", round(o.get("append_estimate_us",0),2), "T+S+M", round(o.get("topk_sparse_attn_plus_merge_us",0),2), "T+S", round(o.get("topk_sparse_attn_kernel_only_us",0),2), "frac", round(r.get("frac") or 0,3))
