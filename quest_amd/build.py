"""Build libquest_hip.so for gfx950 with hipcc (in-tree, next to the sources).

    python -m quest_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libquest_hip.so")
SOURCES = ["append.hip", "estimate.hip", "topk.hip", "sparse_attn.hip", "rope_norm.hip", "decode_layer.hip", "prefill.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         # first 16 kernarg dwords in SGPRs at wave launch (the kernels keep their pointers first): the first
         # loads of a workgroup do not wait for a scalar load of the argument block (measured -0.3 us / launch)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


HASH_TAG = b"quest_hip gfx950 src="  # quest_build_info() = this tag + 16 hex digits (+ " r4")


def source_hash(src_root: str = None, extra_flags=()) -> str:
    """16 hex digits over everything the library is built from: every file under csrc/, include/quest_hip.h, and the
    compiler flags.  Compiled into the library (-DQUEST_SRC_HASH, returned by quest_build_info()); quest_amd._lib
    recomputes it at import and refuses a library built from other sources.  `src_root` = a directory holding
    quest_amd/csrc and include/ (default: this tree)."""
    import hashlib

    root = src_root or os.path.dirname(HERE)
    csrc = os.path.join(root, "quest_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cuh", ".h")))
    files.append(os.path.join(root, "include", "quest_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    h.update(" ".join(list(FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def library_hash(lib_path: str = LIB):
    """The source hash a built library carries (read from the file, nothing is loaded); None if it has none."""
    try:
        blob = open(lib_path, "rb").read()
    except OSError:
        return None
    i = blob.find(HASH_TAG)
    if i < 0:
        return None
    h = blob[i + len(HASH_TAG): i + len(HASH_TAG) + 16]
    return h.decode() if len(h) == 16 and all(c in b"0123456789abcdef" for c in h) else None


def needs_build() -> bool:
    """True when there is no library or it was built from other sources / flags (content hash, not mtimes: the .so is
    git-ignored and travels prebuilt, so file times say nothing about what it was built from)."""
    return library_hash(LIB) != source_hash()


def build_variant(out_path: str, extra_flags) -> str:
    """Tuning aid: build a second .so with extra -D flags (loaded with QUEST_HIP_LIB=<path>)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = f'-DQUEST_SRC_HASH="{source_hash(extra_flags=extra_flags)}"'
    subprocess.check_call([hipcc] + FLAGS + list(extra_flags) + [tag] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out_path])
    return out_path


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [f'-DQUEST_SRC_HASH="{source_hash()}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def kernel_metadata(lib_path: str = LIB):
    """Code-object metadata of every kernel in the built library: {mangled name: {vgpr, vgpr_spill, scratch, lds, sgpr,
    code_bytes}} (clang-offload-bundler + llvm-readelf from the ROCm LLVM; no GPU needed).  Used by
    scripts/kernel_meta.py and by the CPU test that guards the headline kernels' resources."""
    import re
    import tempfile

    llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    notes, syms = "", ""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")  # one offload bundle per source file, concatenated in .hip_fatbin
        subprocess.check_call([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for i, st in enumerate(starts):
            part, co = os.path.join(d, f"b{i}.bin"), os.path.join(d, f"k{i}.co")
            open(part, "wb").write(blob[st:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes += subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co], text=True)
            syms += subprocess.check_output([f"{llvm}/llvm-readelf", "-sW", co], text=True)
    size = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    keys = {"group_segment_fixed_size": "lds", "private_segment_fixed_size": "scratch", "sgpr_count": "sgpr",
            "vgpr_count": "vgpr", "vgpr_spill_count": "vgpr_spill"}
    rows, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":  # first key of a kernel's record
            cur = {}
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
        elif cur is not None and k == "symbol":
            cur["symbol"] = v.strip("'").replace(".kd", "")
        elif cur is not None and k == "wavefront_size":  # last key
            rows.append(cur)
            cur = None
    out = {}
    for r in rows:  # keyed by the MANGLED name (GNU c++filt does not know _Float16; template arguments read ILi<n>E...)
        r["code_bytes"] = size.get(r["symbol"], 0)
        out[r["symbol"]] = r
    return out


if __name__ == "__main__":
    # python -m quest_amd.build [--force]                       the product library
    # python -m quest_amd.build --variant OUT.so -DFOO [-DBAR]  a tuning build (loaded with QUEST_HIP_LIB=OUT.so)
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(os.path.abspath(sys.argv[i + 1]), [a for a in sys.argv[i + 2:] if a.startswith("-")]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
