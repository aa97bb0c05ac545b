"""The reference-side C++ binding (examples/pybind_binding.cpp: a PyBind/torch extension with the reference's
op signatures, bsk_ops.h:23-117 / bsk_ops.cu:4-20, whose bodies call the C ABI of libquest_hip.so) must give the
same bits as quest_amd._kernels for every op, raise like the reference, and carry a whole decode step when it
is registered under the reference's module name.  The extension is prebuilt by __graft_entry__.build(); it is
rebuilt here if stale and a host compiler is present, and the test skips cleanly when neither is possible."""
import os
import sys

import pytest
import torch

from _harness import cuda, fill, inputs, make_controller

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ext():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import build_binding

    mod = build_binding.load_module()
    if mod is None:
        pytest.skip("no prebuilt examples/pybind_binding.cpp extension and no host compiler to build it")
    return mod


@pytest.mark.parametrize("layout", [0, 2], ids=["NHD", "NHD_ROT"])
def test_every_op_of_the_binding_matches_the_python_shim(ext, layout):
    """(layout 2: the row-rotated pool of round 6 -- the binding only passes the layout id through to the C ABI)"""
    from quest_amd import _kernels
    import quest_amd.utils as qu

    dev = "cuda:0"
    L, Hq, Hkv, D, B = 613, 8, 4, 128, 9
    q, k, v = inputs(5, L, Hq, Hkv, D)
    results = []
    for mod in (_kernels, ext):
        ctl = make_controller(L, Hq, Hkv, D, 16, B, shuffle_seed=3, layout=layout)
        kc, vc = cuda(k), cuda(v)
        qd = cuda(q)
        qr, kr = qd.clone(), kc[-1:].clone()
        mod.apply_rope_in_place(qr, kr, L - 1, 1.0, 1e4)
        # prefill L-1 tokens, decode-append the last (append args in the reference's order, page.cu:6-19)
        def args():
            kv, meta = ctl.kv_cache, ctl.metadata_cache
            return (kv.buf_layer(0), ctl.kv_indices_with_last, ctl.kv_indptr_for_append, kv.last_page_len,
                    ctl.kv_last_page_idx, meta.buf_layer(0), ctl.metadata_indices, ctl.metadata_indptr_for_append,
                    meta.last_page_len, ctl.metadata_last_page_idx, ctl.layout)
        ctl.prepare_metadata(L - 1)
        ctl.begin_forward(L - 1)
        mod.append_kv_cache_prefill(kc[:-1], vc[:-1], *args())
        # prefill attention over the pages just written (bsk_ops.h:78-86): the whole prompt (causal), a 37-row chunk at the
        # end of the cache (causal, masked) and one row (full attention) -- GQA, K/V not repeated
        qp = cuda(inputs(21, L - 1, Hq, Hkv, D)[1][:, :1].repeat(Hq, axis=1))  # [L-1, Hq, D]
        pre = [mod.prefill_with_paged_kv_cache(x, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last, ctl.kv_cache.last_page_len,
                                               True, ctl.layout, False, 1.0, 1e4) for x in (qp, qp[-37:], qp[-1:])]
        ctl.end_forward()
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        mod.append_kv_cache_decode(kc[-1:], vc[-1:], *args())
        meta = ctl.metadata_cache
        n_out = meta.seqlen - 1
        est = torch.empty(Hq, n_out, dtype=torch.float16, device=dev)
        mod.estimate_attn_score(qd, est, meta.buf_layer(0), ctl.metadata_indices, ctl.metadata_indptr_for_append,
                                meta.last_page_len, ctl.metadata_last_page_idx, ctl.layout)
        dv = torch.zeros(Hq, B - 1, dtype=torch.float16, device=dev)
        di = torch.zeros(Hq, B - 1, dtype=torch.int32, device=dev)
        mod.topk_filtering(est, ctl.kv_indices_without_last, dv, di, ctl.topk_buf, B - 1)
        h = mod.BatchDecodeWithPagedKVCachePyTorchWrapper(ctl.layout)
        h.begin_forward(torch.tensor([0, B - 1], dtype=torch.int32), Hq, Hkv, D, 16, torch.empty(0, dtype=torch.float16))
        o = torch.empty_like(qd)
        h.forward(qd, o, ctl.kv_cache.buf_layer(0), di, ctl.kv_indptr_for_approx_decode, ctl.kv_cache.last_page_len,
                  ctl.kv_last_page_idx, 1.0, 1e4)
        h.end_forward()
        ctl.end_forward()
        x = cuda(inputs(9, 3, 1, 1, 1024)[1].reshape(1, 3, 1024))
        w = cuda(inputs(10, 1, 1, 1, 1024)[1].reshape(1024))
        y = torch.empty_like(x)
        mod.rms_norm_forward(x, w, y, 1e-5)
        table = torch.tensor(ctl.kv_cache.indicies, device=dev)
        mtable = torch.tensor(ctl.metadata_cache.indicies, device=dev)
        results.append(dict(rope_q=qr, rope_k=kr, kv=ctl.kv_cache.buf_layer(0)[table][:-1],
                            meta=ctl.metadata_cache.buf_layer(0)[mtable][:-1], est=est, topk_v=dv, topk_i=di, o=o, rms=y,
                            prefill_whole=pre[0], prefill_chunk=pre[1], prefill_row=pre[2]))
    for key in results[0]:
        assert torch.equal(results[0][key], results[1][key]), f"C++ binding differs from quest_amd._kernels: {key}"
    assert torch.isfinite(results[1]["o"].float()).all() and results[1]["o"].abs().sum() > 0
    # the prefill op itself against fp32 torch attention (causal over the cache of L - 1 tokens)
    kf = cuda(k)[:L - 1].float().repeat_interleave(Hq // Hkv, dim=1)
    vf = cuda(v)[:L - 1].float().repeat_interleave(Hq // Hkv, dim=1)
    qf = cuda(inputs(21, L - 1, Hq, Hkv, D)[1][:, :1].repeat(Hq, axis=1)).float()
    logits = torch.einsum("ihd,jhd->hij", qf, kf) / D ** 0.5
    logits = logits.masked_fill(torch.triu(torch.ones(L - 1, L - 1, dtype=torch.bool, device=dev), 1), float("-inf"))
    ref = torch.einsum("hij,jhd->ihd", torch.softmax(logits, -1), vf)
    assert results[1]["prefill_whole"].shape == (L - 1, Hq, D)
    torch.testing.assert_close(results[1]["prefill_whole"].float(), ref, rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(results[1]["prefill_chunk"].float(), ref[-37:], rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(results[1]["prefill_row"].float(), ref[-1:], rtol=5e-3, atol=5e-3)


def test_binding_error_behaviour(ext):
    dev = "cuda:0"
    est = torch.zeros(4, 8, dtype=torch.float16, device=dev)
    idx = torch.zeros(4, 8, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="topk_filtering failed"):  # CHECK_GE(num_pages, page_budget), topk.cu:26
        ext.topk_filtering(est, idx, torch.zeros(4, 9, dtype=torch.float16, device=dev),
                           torch.zeros(4, 9, dtype=torch.int32, device=dev), est, 9)
    h = ext.BatchDecodeWithPagedKVCachePyTorchWrapper(0)
    with pytest.raises(ValueError):  # num_qo_heads % num_kv_heads != 0 -> std::invalid_argument (decode_attn.cuh:1045-1050)
        h.begin_forward(torch.tensor([0, 3], dtype=torch.int32), 6, 4, 128, 16, torch.empty(0, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="dispatch with dtype"):
        h.begin_forward(torch.tensor([0, 3], dtype=torch.int32), 4, 4, 128, 16, torch.empty(0, dtype=torch.float32))
    q = torch.zeros(1, 4, 128, dtype=torch.float16, device=dev)
    with pytest.raises(RuntimeError, match="begin_forward"):  # forward before begin_forward (decode_handler.cuh:226-231)
        h.forward(q, torch.empty_like(q), torch.zeros(4, 2, 16, 4, 128, dtype=torch.float16, device=dev),
                  torch.zeros(4, 2, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev), 4, 0, 1.0, 1e4)


def test_quest_amd_utils_run_over_the_cpp_binding(ext, monkeypatch):
    """The host mirror (controller + wrappers) is written against the module surface, so it runs unchanged over the
    C++ binding: one decode step through quest_amd.utils with `_kernels` swapped == the same step over the shim."""
    import quest_amd.utils as qu
    import quest_amd.utils.decode_wrapper as dw
    from quest_amd import _kernels

    L, H, D, B = 900, 8, 128, 12
    q, k, v = inputs(15, L, H, H, D)

    def step():
        ctl = make_controller(L, H, H, D, 16, B, shuffle_seed=8)
        fill(ctl, k, v)
        est = qu.decode_estimate(cuda(q), ctl, 0)
        qu.decode_topk(est, ctl)
        o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.topk_dindices_buffer)
        ctl.end_forward()
        return est, ctl.topk_dindices_buffer.clone(), o

    a = step()

    class Swapped:  # the C++ module for the reference's ops, the shim for everything it does not define
        def __getattr__(self, name):
            return getattr(ext, name) if hasattr(ext, name) else getattr(_kernels, name)

    monkeypatch.setattr(qu, "_kernels", Swapped())
    monkeypatch.setattr(dw, "_kernels", Swapped())
    b = step()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
