#!/usr/bin/env python3
"""Build examples/pybind_binding.cpp (the reference-side PyBind binding of INTEGRATION.md section B) against
libquest_hip.so with torch.utils.cpp_extension and check that it produces the same bits as quest_amd._kernels.

    python scripts/check_cpp_binding.py --build-only     (no GPU needed)
    python scripts/check_cpp_binding.py                  (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.utils.cpp_extension import load

import quest_amd.build as qb

qb.build()
lib_dir = os.path.join(ROOT, "quest_amd")
build_dir = os.path.join(ROOT, "build", "binding_example")  # in-tree (git-ignored), nothing under ~/.cache
os.makedirs(build_dir, exist_ok=True)
ext = load(name="quest_binding_example", sources=[os.path.join(ROOT, "examples", "pybind_binding.cpp")],
           extra_include_paths=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
           extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
           extra_ldflags=[f"-L{lib_dir}", "-lquest_hip", f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-lamdhip64"],
           build_directory=build_dir, verbose=False)
print("built:", ext.__file__)
if "--build-only" in sys.argv:
    sys.exit(0)

import numpy as np  # noqa: E402
from _harness import cuda, fill, inputs, make_controller  # noqa: E402

import quest_amd.utils as qu  # noqa: E402
from quest_amd import _kernels  # noqa: E402

L, H, D, B = 613, 8, 128, 9
q, k, v = inputs(5, L, H, H, D)
ctl = make_controller(L, H, H, D, 16, B, shuffle_seed=3)
fill(ctl, k, v)
qd = cuda(q)
meta = ctl.metadata_cache
n_out = meta.seqlen - 1
o_py = torch.empty(H, n_out, dtype=torch.float16, device="cuda:0")
o_cc = torch.empty_like(o_py)
args = (meta.buf_layer(0), ctl.metadata_indices, ctl.metadata_indptr_for_append, meta.last_page_len,
        ctl.metadata_last_page_idx, ctl.layout)
_kernels.estimate_attn_score(qd, o_py, *args)
ext.estimate_attn_score(qd, o_cc, *args)
assert torch.equal(o_py, o_cc), "estimate differs"
idx = ctl.kv_indices_without_last
outs = []
for fn in (_kernels.topk_filtering, ext.topk_filtering):
    dv = torch.zeros(H, B - 1, dtype=torch.float16, device="cuda:0")
    di = torch.zeros(H, B - 1, dtype=torch.int32, device="cuda:0")
    fn(o_py, idx, dv, di, ctl.topk_buf, B - 1)
    outs.append((dv, di))
assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "top-k differs"
try:
    ext.topk_filtering(o_py, idx, outs[0][0], outs[0][1], ctl.topk_buf, n_out + 5)
    raise SystemExit("expected an error for page_budget > num_pages")
except RuntimeError as e:
    assert "topk_filtering failed" in str(e)
print("C++ binding == quest_amd._kernels on estimate and top-k; error path raises RuntimeError: OK")
