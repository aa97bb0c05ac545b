#!/usr/bin/env python3
"""Copy a rocprofv3 --kernel-trace --stats summary into profiles/ (kernel names shortened)."""
import csv
import glob
import sys

src_dir, dst = sys.argv[1], sys.argv[2]
f = glob.glob(f"{src_dir}/*/*kernel_stats.csv")[0]
rows = list(csv.reader(open(f)))
with open(dst, "w", newline="") as out:
    w = csv.writer(out)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)
print(dst)
