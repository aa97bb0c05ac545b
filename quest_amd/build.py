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
SOURCES = ["append.hip", "estimate.hip", "topk.hip", "sparse_attn.hip", "rope_norm.hip", "decode_layer.hip"]
HEADERS = ["quest_common.cuh", "topk_select.cuh", "topk_bitmap.cuh", "append_device.cuh", "estimate_device.cuh", "decode_device.cuh", os.path.join("..", "..", "include", "quest_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         # first 16 kernarg dwords in SGPRs at wave launch (the kernels keep their pointers first): the first
         # loads of a workgroup do not wait for a scalar load of the argument block (measured -0.3 us / launch)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(out_path: str, extra_flags) -> str:
    """Tuning aid: build a second .so with extra -D flags (loaded with QUEST_HIP_LIB=<path>)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out_path])
    return out_path


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
