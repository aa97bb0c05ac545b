#!/usr/bin/env python3
"""Generate tests/golden/prefill_ref_golden.npz from the REFERENCE's own pure-torch prefill oracle
(`_ref_self_attention`, quest/tests/test_prefill_attention.py:17-44), on (qo_len, kv_len) pairs of the reference's own
sweep (:46) at reduced head counts.  Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_prefill_golden.py

The reference function is imported by file path, exactly as make_golden.py does it (empty stand-in for the unbuilt
`quest._kernels`).  Inputs are regenerated from seeds by oracle.synth, so the fixture holds seeds, shapes and outputs."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from oracle.synth import normal_f16  # noqa: E402

CASES = [(13, 33, 2), (24, 66, 2), (51, 129, 2), (77, 400, 2), (244, 700, 1), (311, 1110, 1), (502, 700, 1)]


def prefill_inputs(seed, qo_len, kv_len, H, D=128):
    return (normal_f16(seed * 3 + 0, (qo_len, H, D)), normal_f16(seed * 3 + 1, (kv_len, H, D)),
            normal_f16(seed * 3 + 2, (kv_len, H, D)))


def main():
    stub = types.ModuleType("quest._kernels")
    stub.BatchDecodeWithPagedKVCachePyTorchWrapper = object
    sys.modules["quest._kernels"] = stub
    spec = importlib.util.spec_from_file_location("ref_test_prefill_attention",
                                                  os.path.join(REF, "quest", "tests", "test_prefill_attention.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    torch.set_num_threads(8)
    out = {"cases": np.array([(700 + i, qo, kv, H) for i, (qo, kv, H) in enumerate(CASES)], np.int32)}
    for i, (qo, kv, H) in enumerate(CASES):
        q, k, v = prefill_inputs(700 + i, qo, kv, H)
        with torch.inference_mode():
            o = m._ref_self_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v))
        out[f"o_{qo}_{kv}"] = o.contiguous().numpy()
        print(qo, kv, H, out[f"o_{qo}_{kv}"].shape, out[f"o_{qo}_{kv}"].dtype)
    np.savez_compressed(os.path.join(HERE, "prefill_ref_golden.npz"), **out)


if __name__ == "__main__":
    main()
