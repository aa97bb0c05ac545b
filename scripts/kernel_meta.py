#!/usr/bin/env python3
"""Code-object metadata of the built library's kernels (VGPRs, spills, scratch, LDS, code bytes), filtered by substrings
of the mangled name (template arguments read ILi128ELi16E...).

    python scripts/kernel_meta.py [substring ...]          (QUEST_HIP_LIB=<path> inspects another build)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quest_amd.build import LIB, kernel_metadata  # noqa: E402

meta = kernel_metadata(os.environ.get("QUEST_HIP_LIB") or LIB)
for name in sorted(meta):
    if sys.argv[1:] and not any(p in name for p in sys.argv[1:]):
        continue
    r = meta[name]
    short = name[len("_ZN5quest"):] if name.startswith("_ZN5quest") else name
    print(f"{short[:90]:90s} vgpr {r['vgpr']:4d} spill {r['vgpr_spill']:3d} scratch {r['scratch']:4d} lds {r['lds']:6d} "
          f"sgpr {r['sgpr']:4d} code {r['code_bytes']:6d}")
