#!/usr/bin/env python3
"""profiles/traffic_latest.json from the round's rocprofv3 --pmc passes (gpurun_out/prof_<round>_<tag>/pmc_traffic.json +
line.json): HBM bytes per launch of the DOMINANT kernel of each configuration, keyed by configuration and carrying the
kernel instantiation the bytes belong to (bench.py's `roofline.kernel_name`), so that bench.py never reports one
instantiation's traffic for another.

    python scripts/make_traffic_latest.py r04
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
TAGS = {"cfg3": "cfg3_seqs1", "cfg3x8": "cfg3_seqs8", "cfg5": "cfg5_seqs8", "cfg4": "cfg4_seqs1", "cfg2": "cfg2_seqs1"}


def pretty(mangled):
    """sparse_decode_kernelILi128ELi16ELi8ELi8ELi3EEEv... -> sparse_decode_kernel<128,16,8,8,3>"""
    m = re.match(r"(\w+?_kernel)I((?:L[ib]n?\d+E)+)E", mangled)
    if not m:
        return mangled
    args = []
    for t, neg, v in re.findall(r"L([ib])(n?)(\d+)E", m.group(2)):
        args.append(("true" if v == "1" else "false") if t == "b" else ("-" if neg else "") + v)
    return f"{m.group(1)}<{','.join(args)}>"


out = {"source": f"profiles/{rnd}_*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of `python3 bench.py "
                 f"--steps 20 --warmup 5 --no-side [--config N | --seqs-per-gpu 8]`, round {rnd[1:].lstrip('0')} closing tree)",
       "correction": "gfx950 FETCH_SIZE counts 128-B requests at 64 B: read bytes = 2 x FETCH_SIZE KB x 1024 (MI355X_MICROARCH.md, "
                     "HBM section); WRITE_SIZE as reported",
       "bytes_per_launch": {}}
for tag, key in TAGS.items():
    d = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{tag}")
    try:
        line = json.loads(open(os.path.join(d, "line.json")).read().strip().splitlines()[-1])
        pmc = json.load(open(os.path.join(d, "pmc_traffic.json")))
    except Exception as exc:  # noqa: BLE001
        print(f"{tag}: skipped ({exc})")
        continue
    want = (line.get("roofline") or {}).get("kernel_name")
    rows = {pretty(r["kernel"]): r for r in pmc["kernels"]}
    r = rows.get(want)
    if r is None:
        print(f"{tag}: no PMC row for {want}; have {sorted(rows)}")
        continue
    out["bytes_per_launch"][key] = {"kernel": want, "read_bytes": r["hbm_read_bytes_corrected"], "write_bytes": r["hbm_write_bytes"],
                                    "bytes": r["hbm_read_bytes_corrected"] + r["hbm_write_bytes"], "dispatches": r["dispatches"]}
    print(tag, want, out["bytes_per_launch"][key])
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
