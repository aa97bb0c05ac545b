#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own pure-torch oracles.

Runs only in the build container (needs /root/reference; the GPU box never sees it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference functions are imported, not copied: the test modules under
/root/reference/quest/tests are loaded by file path.  They ``import quest.utils``
which needs the unbuilt CUDA extension ``quest._kernels``; an empty module object
is registered under that name so the import succeeds (nothing in it is ever
called -- only the ``_ref_*`` pure-torch functions are used).

Inputs are regenerated from seeds by oracle.synth (bit-identical everywhere), so
the fixtures hold only seeds, shapes and the reference's outputs.
"""
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


def _load_ref_tests():
    stub = types.ModuleType("quest._kernels")
    stub.BatchDecodeWithPagedKVCachePyTorchWrapper = object
    sys.modules["quest._kernels"] = stub
    mods = {}
    for name in ("test_estimate", "test_approx_attention", "test_decode_attention", "test_rope"):
        path = os.path.join(REF, "quest", "tests", name + ".py")
        spec = importlib.util.spec_from_file_location("ref_" + name, path)
        m = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(m)
            mods[name] = m
        except Exception as e:  # ordinary python error (e.g. transformers API drift)
            print(f"[make_golden] {name}: not importable here: {type(e).__name__}: {e}")
    return mods


def _t(a):
    return torch.from_numpy(a)


def inputs(seed, L, H, D=128):
    q = normal_f16(seed * 3 + 0, (1, H, D))
    k = normal_f16(seed * 3 + 1, (L, H, D))
    v = normal_f16(seed * 3 + 2, (L, H, D))
    return q, k, v


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mods = _load_ref_tests()
    page_size = 16
    out = {}

    # --- estimate: the reference's own kv_len grid (test_estimate.py:77), H=32 as there
    est = mods["test_estimate"]._ref_cpu_estimate
    cases = []
    for i, L in enumerate([27, 61, 113, 482, 577, 1110, 1541, 2047, 3330]):
        H = 32 if L <= 1110 else 8
        seed = 100 + i
        q, k, v = inputs(seed, L, H)
        with torch.inference_mode():
            r = est(_t(q), _t(k), _t(v), page_size)
        cases.append((seed, L, H))
        out[f"est_{L}"] = r.numpy()
    out["est_cases"] = np.array(cases, dtype=np.int64)

    # --- sparse attention with the oracle's own top-k indices (test_approx_attention.py:112)
    approx = mods["test_approx_attention"]._ref_self_approx_attention
    cases = []
    i = 0
    for L in [27, 61, 113, 482, 577, 1011, 1541]:
        for B in [4, 7, 15, 31, 55, 71]:
            H = 8
            seed = 200 + i
            i += 1
            q, k, v = inputs(seed, L, H)
            with torch.inference_mode():
                o, idx = approx(_t(q), _t(k), _t(v), page_size, B)
            cases.append((seed, L, H, B, 0 if idx is None else 1))
            out[f"approx_o_{L}_{B}"] = o.contiguous().numpy()
            if idx is not None:
                out[f"approx_idx_{L}_{B}"] = idx.reshape(H, B - 1).numpy().astype(np.int32)
    # BASELINE config 1: L=4096, budget 64 pages, H=32 (CPU-runnable case)
    for (L, B, H) in [(4096, 64, 32), (4099, 64, 32)]:
        seed = 200 + i
        i += 1
        q, k, v = inputs(seed, L, H)
        with torch.inference_mode():
            o, idx = approx(_t(q), _t(k), _t(v), page_size, B)
        cases.append((seed, L, H, B, 1))
        out[f"approx_o_{L}_{B}"] = o.contiguous().numpy()
        out[f"approx_idx_{L}_{B}"] = idx.reshape(H, B - 1).numpy().astype(np.int32)
    out["approx_cases"] = np.array(cases, dtype=np.int64)

    # --- dense decode (test_decode_attention.py:46)
    dense = mods["test_decode_attention"]._ref_self_attention
    cases = []
    for i, L in enumerate([27, 61, 113, 482, 577, 1011]):
        H = 8
        seed = 300 + i
        q, k, v = inputs(seed, L, H)
        with torch.inference_mode():
            o = dense(_t(q), _t(k), _t(v))
        cases.append((seed, L, H))
        out[f"dense_o_{L}"] = o.contiguous().numpy()
    out["dense_cases"] = np.array(cases, dtype=np.int64)

    # --- top-k: what test_topk.py pins -- the selected VALUES of torch.topk on fp16 rows
    cases = []
    i = 0
    for n in [13, 24, 51, 77, 244, 311, 502, 1110]:
        for kk in [2, 5, 7, 19, 31, 69, 111, 251]:
            if kk > n:
                continue
            rows = 4
            seed = 400 + i
            i += 1
            qh = _t(normal_f16(seed * 3, (rows, 1, 128)))
            kh = _t(normal_f16(seed * 3 + 1, (rows, n, 128)))
            with torch.inference_mode():
                w = (torch.matmul(qh, kh.transpose(1, 2)) / (128 ** 0.5)).squeeze(1).contiguous()
                tv = torch.topk(w, k=kk, dim=-1).values
            cases.append((seed, n, kk, rows))
            out[f"topk_in_{n}_{kk}"] = w.numpy()
            out[f"topk_vals_{n}_{kk}"] = tv.numpy()
    out["topk_cases"] = np.array(cases, dtype=np.int64)

    # --- rope (test_rope.py:17-30) -- depends on the installed transformers API
    if "test_rope" in mods:
        rope = mods["test_rope"]._ref_apply_qk_rope
        cases = []
        try:
            for i, (past, n) in enumerate([(13, 2), (77, 19), (502, 69), (1110, 5)]):
                H = 4
                seed = 500 + i
                q = normal_f16(seed * 3, (n, H, 128))
                k = normal_f16(seed * 3 + 1, (n, H, 128))
                with torch.inference_mode():
                    qr, kr = rope(_t(q), _t(k), past)
                cases.append((seed, past, n, H))
                out[f"rope_q_{past}_{n}"] = qr.contiguous().numpy()
                out[f"rope_k_{past}_{n}"] = kr.contiguous().numpy()
            out["rope_cases"] = np.array(cases, dtype=np.int64)
        except Exception as e:
            print(f"[make_golden] rope oracle raised {type(e).__name__}: {e} -> no rope fixtures")

    path = os.path.join(HERE, "quest_ref_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
