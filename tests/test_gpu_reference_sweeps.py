"""The parameter sweeps of the reference's own C++ tests, run against the HIP path through the C ABI:

* append (prefill + decode) with fused min/max: page_size {1,3,7,16,31} x seq_len {17..4100} x NHD/HND x
  head_dim {64,128,256}, 32 heads (kernels/src/test/test_page.cu:620-660) -- pools bit-exact vs the oracle;
* sparse paged decode attention over RANDOM distinct pages per head (as the reference test picks them,
  test_batch_decode.cu:57-76): seq_len {63..28837} x page_size {1,3,7,16,32} x budget {16,32,64,128} x
  head_dim {64,128}, 32 heads (test_batch_decode.cu:236-257) -- vs the fp64 oracle at 2e-3 (the reference's own
  bar is 1e-3 on > 99 % of the elements);
* the chain estimate -> top-k -> attention at the page sizes the hand-picked cases did not have (1, 7, 31).
"""
import numpy as np
import pytest
import torch

import oracle
from _harness import gather_entries, make_controller

pytestmark = pytest.mark.gpu
U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)
DEV = "cuda:0"


def _randn(seed, *shape):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV, dtype=torch.float16)


def _fill(ctl, k, v, tail=1):
    """prefill-append all but the last `tail` tokens, decode-append the rest (stays inside the last begin_forward)."""
    import quest_amd.utils as qu

    L = k.shape[0]
    n0 = L - tail
    ctl.prepare_metadata(n0)
    ctl.begin_forward(n0)
    qu.append_kv(k[:n0], v[:n0], ctl, 0)
    ctl.end_forward()
    for t in range(n0, L):
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        qu.append_kv(k[t:t + 1], v[t:t + 1], ctl, 0)
        if t != L - 1:
            ctl.end_forward()


def _host_pool(cache, layout):
    """oracle.Paged over the DEVICE pool's bytes and the controller's page table."""
    return oracle.Paged(cache.buf_layer(0).cpu().numpy(), np.array(cache.indicies, np.int32), cache.last_page_len, layout)


@pytest.mark.parametrize("page_size", [1, 3, 7, 16, 31])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_append_reference_sweep_pools_bit_exact(page_size, layout):
    for seq_len in (17, 31, 71, 111, 330, 512, 1110, 4100):
        for D in (64, 128, 256):
            H = 32
            k, v = _randn(seq_len + D, seq_len, H, D), _randn(seq_len + D + 1, seq_len, H, D)
            ctl = make_controller(seq_len, H, H, D, page_size, 1 << 20, layout=layout, shuffle_seed=seq_len,
                                  max_seq_len=seq_len + 2 * page_size)
            _fill(ctl, k, v, tail=min(3, seq_len - 2))
            ctl.end_forward()
            kn, vn = k.cpu().numpy(), v.cpu().numpy()
            # oracle pools with the controller's physical page ids
            kvc, mc = ctl.kv_cache, ctl.metadata_cache
            kv_o = oracle.Paged(np.zeros(tuple(kvc.buf_layer(0).shape), np.float16), np.array(kvc.indicies, np.int32),
                                kvc.last_page_len, layout)
            meta_o = oracle.Paged(np.zeros(tuple(mc.buf_layer(0).shape), np.float16), np.array(mc.indicies, np.int32),
                                  mc.last_page_len, layout)
            oracle.append_prefill(kv_o, meta_o, kn, vn)
            n_pages = len(kvc.indicies)
            for dev_cache, ora, n in ((kvc, kv_o, seq_len), (mc, meta_o, n_pages)):
                dk, dv = gather_entries(dev_cache.buf_layer(0).cpu().numpy(), dev_cache.indicies, n, layout)
                ok_, ov_ = gather_entries(ora.data, ora.indices, n, layout)
                assert np.array_equal(U16(dk), U16(ok_)) and np.array_equal(U16(dv), U16(ov_)), (seq_len, D)
            # metadata == per-page extrema (exact in fp16)
            pad = n_pages * page_size - seq_len
            kp = np.concatenate([kn, np.repeat(kn[-1:], pad, 0)]).reshape(n_pages, page_size, H, D)
            mx, mn = gather_entries(mc.buf_layer(0).cpu().numpy(), mc.indicies, n_pages, layout)
            assert np.array_equal(mx, kp.max(1)) and np.array_equal(mn, kp.min(1))


@pytest.mark.parametrize("page_size", [1, 3, 7, 16, 32])
@pytest.mark.parametrize("seq_len", [63, 127, 213, 1110, 2000, 4099, 8192, 8222, 12345, 28837])
def test_sparse_attention_reference_sweep(seq_len, page_size):
    import quest_amd.utils as qu

    H = 32
    n_pages = (seq_len + page_size - 1) // page_size
    rng = np.random.default_rng(seq_len * 41 + page_size)
    for D in (64, 128):
        k, v = _randn(seq_len + D, seq_len, H, D), _randn(seq_len + D + 7, seq_len, H, D)
        q = _randn(seq_len + D + 13, 1, H, D)
        ctl = make_controller(seq_len, H, H, D, page_size, 1 << 20, shuffle_seed=seq_len + page_size,
                              max_seq_len=seq_len + 2 * page_size)
        _fill(ctl, k, v)
        ctl.end_forward()
        kv_o = _host_pool(ctl.kv_cache, 0)
        table = np.array(ctl.kv_cache.indicies, np.int32)
        qn = q.cpu().numpy()
        for budget in (16, 32, 64, 128):
            B = min(budget, n_pages)
            if B < 2:
                continue
            ctl.set_page_budget(B)
            ctl.begin_forward(1)
            # random distinct pages per head, never the last one (test_batch_decode.cu:57-76)
            logical = np.stack([rng.permutation(n_pages - 1)[: B - 1] for _ in range(H)]).astype(np.int32)
            phys = table[logical]
            o = qu.decode_sparse_attn(q, ctl, 0, torch.from_numpy(phys).to(DEV))
            ctl.end_forward()
            eo, _ = oracle.sparse_attn(qn, kv_o, phys, B - 1, int(table[-1]), kv_o.last_page_len)
            np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3,
                                       err_msg=f"L={seq_len} S={page_size} B={B} D={D}")


@pytest.mark.parametrize("page_size,L,B,Hq,Hkv,D,layout", [
    (1, 700, 33, 8, 8, 128, 0), (1, 333, 8, 8, 2, 64, 1), (7, 2000, 16, 32, 32, 128, 0), (7, 1234, 40, 16, 4, 128, 1),
    (31, 4100, 9, 8, 8, 128, 0), (3, 999, 64, 4, 4, 256, 0), (1, 5000, 128, 4, 4, 128, 0), (7, 28837, 128, 8, 8, 128, 0)])
def test_chain_at_odd_page_sizes(page_size, L, B, Hq, Hkv, D, layout):
    """estimate -> top-k -> attention, op by op and through the fused launches, vs the oracle chain."""
    import quest_amd.utils as qu

    k, v = _randn(L, L, Hkv, D), _randn(L + 1, L, Hkv, D)
    q = _randn(L + 2, 1, Hq, D)
    ctl = make_controller(L, Hq, Hkv, D, page_size, B, layout=layout, shuffle_seed=L, max_seq_len=L + 2 * page_size)
    _fill(ctl, k, v, tail=2)
    assert ctl.need_estimate()
    meta_o, kv_o = _host_pool(ctl.metadata_cache, layout), _host_pool(ctl.kv_cache, layout)
    qn = q.cpu().numpy()
    est = qu.decode_estimate(q, ctl, 0)
    e_est = oracle.estimate(qn, meta_o)
    assert np.array_equal(U16(est.cpu().numpy()), U16(e_est))
    qu.decode_topk(est, ctl)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    ev, ei = oracle.topk(e_est, np.tile(table[:-1], (Hq, 1)), B - 1)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    assert np.array_equal(U16(ctl.topk_dout_buffer.cpu().numpy()), U16(ev))
    o = qu.decode_sparse_attn(q, ctl, 0, ctl.topk_dindices_buffer)
    eo, _ = oracle.sparse_attn(qn, kv_o, ei, B - 1, int(table[-1]), kv_o.last_page_len)
    np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
    ctl.topk_dindices_buffer.zero_()
    o2 = qu.decode_topk_sparse_attn(q, est, ctl, 0, write_topk=True)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    assert torch.equal(o, o2)
    ctl.end_forward()


@pytest.mark.parametrize("kv_len", [33, 66, 129, 400, 700, 1110])
def test_prefill_attention_reference_sweep(kv_len):
    """quest/tests/test_prefill_attention.py:44-88: qo_len x kv_len of the reference's own sweep (H = 32, D = 128, page 16;
    the last qo_len rows of a kv_len cache, bottom-right causal) against its _ref_self_attention restated in fp32, at its
    tolerance.  qo_len == kv_len goes through SDPA's flash backend, qo_len < kv_len through the masked memory-efficient one."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    H, D = 32, 128
    g = torch.Generator(device=dev).manual_seed(kv_len)
    k = torch.randn(kv_len, H, D, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn(kv_len, H, D, generator=g, device=dev, dtype=torch.float16)
    ctl = make_controller(kv_len, H, H, D, 16, 1024, shuffle_seed=kv_len)
    ctl.prepare_metadata(kv_len)
    ctl.begin_forward(kv_len)
    qu.append_kv(k, v, ctl, 0)
    for qo_len in (13, 24, 51, 77, 244, 311, 502, kv_len):
        if qo_len > kv_len:
            continue
        q = torch.randn(qo_len, H, D, generator=g, device=dev, dtype=torch.float16)
        o = qu.prefill_forward(q, ctl, 0)
        s = torch.einsum("qhd,khd->hqk", q.float(), k.float()) / D ** 0.5
        mask = torch.ones(qo_len, kv_len, dtype=torch.bool, device=dev).tril(diagonal=kv_len - qo_len)
        ref = torch.einsum("hqk,khd->qhd", s.masked_fill(~mask, float("-inf")).softmax(-1), v.float())
        torch.testing.assert_close(o.float(), ref, rtol=5e-3, atol=5e-3)
    ctl.end_forward()
