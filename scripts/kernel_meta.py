#!/usr/bin/env python3
"""Code-object metadata of the built library's kernels (VGPRs, spills, scratch, LDS, code bytes), filtered by a substring.

    python scripts/kernel_meta.py [substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.environ.get("QUEST_HIP_LIB") or os.path.join(ROOT, "quest_amd", "libquest_hip.so")
pats = sys.argv[1:]
notes, syms = "", ""
with tempfile.TemporaryDirectory() as d:
    # the .so carries one offload bundle per source file, concatenated in .hip_fatbin
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for i, st in enumerate(starts):
        part = os.path.join(d, f"b{i}.bin")
        open(part, "wb").write(blob[st:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
        co = os.path.join(d, f"k{i}.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        syms += subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", co], text=True)
size = {}
for line in syms.splitlines():
    f = line.split()
    if len(f) >= 8 and f[3] == "FUNC":
        size[f[7]] = int(f[2])
cur = {}
rows = []
for line in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and "agpr_count" not in cur and cur.get("_in_args"):
        continue
    if k == "agpr_count":
        cur = {"agpr": v}
    elif k in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
               "vgpr_spill_count", "symbol"):
        cur[k] = v
        if k == "symbol":
            pass
    if k == "wavefront_size":
        rows.append(cur)
        cur = {}
for r in rows:
    sym = r.get("symbol", "?").replace("'", "").replace(".kd", "")
    try:
        dem = subprocess.check_output([f"{LLVM}/llvm-cxxfilt", sym], text=True).strip()
    except Exception:
        dem = sym
    if pats and not any(p in dem for p in pats):
        continue
    print(f"{dem[:110]:110s} vgpr {r.get('vgpr_count'):>4s} spill {r.get('vgpr_spill_count'):>3s} scratch "
          f"{r.get('private_segment_fixed_size'):>4s} lds {r.get('group_segment_fixed_size'):>6s} sgpr {r.get('sgpr_count'):>4s} "
          f"code {size.get(sym, 0):>6d}")
