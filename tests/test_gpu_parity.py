"""GPU parity: quest_amd (HIP, through the C ABI) vs the CPU oracle and the committed golden
fixtures.  Run on the MI355X box with ``pytest -m gpu``.

Bars (north_star): metadata, estimate and top-k are integer/bit-pattern work -> bit-exact vs the
oracle; attention output fp16 within rtol = atol = 5e-3 (the reference's own tolerance,
quest/tests/test_approx_attention.py:10-15), and we also assert the tighter 2e-3 against the
double-precision oracle.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import synth
from _harness import cuda, fill, inputs, make_controller, oracle_pools, pools_match

pytestmark = pytest.mark.gpu

PAGE = 16
U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)


def _qu():
    import quest_amd.utils as qu

    return qu


def _close(a, b, tol=5e-3):
    torch.testing.assert_close(torch.from_numpy(np.asarray(a, np.float32)), torch.from_numpy(np.asarray(b, np.float32)),
                               rtol=tol, atol=tol)


# ---------------------------------------------------------------------------- append / metadata

@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("L,split", [(27, None), (61, 17), (113, 100), (482, None), (1011, 1000), (33, 32), (17, 16)])
def test_append_bit_exact(layout, L, split):
    Hq = Hkv = 8
    q, k, v = inputs(7 + L, L, Hq)
    ctl = make_controller(L, Hq, Hkv, 128, PAGE, 64, layout=layout, shuffle_seed=L)
    fill(ctl, k, v, split=split)
    kv_o, meta_o = oracle_pools(ctl, k, v)
    assert pools_match(ctl, kv_o, meta_o, L)
    ctl.end_forward()


def test_append_chunked_prefill_bit_exact():
    """Two prefill appends (second one starts mid-page) == one (decode_page.cuh:487-504 start_entry_idx)."""
    qu = _qu()
    L, H = 200, 4
    q, k, v = inputs(3, L, H)
    ctl = make_controller(L, H, H, 128, PAGE, 64, shuffle_seed=5)
    kc, vc = cuda(k), cuda(v)
    for a, b in [(0, 75), (75, 200)]:
        ctl.prepare_metadata(b - a)
        ctl.begin_forward(b - a)
        qu.append_kv(kc[a:b], vc[a:b], ctl, 0)
        ctl.end_forward()
    kv_o, meta_o = oracle_pools(ctl, k, v)
    assert pools_match(ctl, kv_o, meta_o, L)


# ---------------------------------------------------------------------------- estimate

@pytest.mark.parametrize("layout", [0, 1, 2])
def test_estimate_bit_exact_and_golden(golden, layout):
    qu = _qu()
    for seed, L, H in golden["est_cases"]:
        seed, L, H = int(seed), int(L), int(H)
        q, k, v = inputs(seed, L, H)
        ctl = make_controller(L, H, H, 128, PAGE, 1024, layout=layout, shuffle_seed=seed)
        fill(ctl, k, v)
        got = qu.decode_estimate(cuda(q), ctl, 0).cpu().numpy()
        ctl.end_forward()
        kv_o, meta_o = oracle_pools(ctl, k, v)
        exp = oracle.estimate(q, meta_o)
        assert got.shape == exp.shape == golden[f"est_{L}"].shape
        assert np.array_equal(U16(got), U16(exp)), f"estimate not bit-exact at L={L}"
        _close(got, golden[f"est_{L}"])


@pytest.mark.parametrize("Hq,Hkv,D,page", [(32, 8, 128, 16), (8, 2, 64, 16), (8, 8, 256, 16), (8, 4, 128, 8),
                                            (6, 3, 128, 32), (16, 2, 128, 16)])
def test_estimate_gqa_dims_pages(Hq, Hkv, D, page):
    qu = _qu()
    L = 777
    q, k, v = inputs(50 + Hq + D, L, Hq, Hkv, D)
    for layout in (0, 1, 2):
        ctl = make_controller(L, Hq, Hkv, D, page, 1024, layout=layout, shuffle_seed=1)
        fill(ctl, k, v)
        got = qu.decode_estimate(cuda(q), ctl, 0).cpu().numpy()
        ctl.end_forward()
        _, meta_o = oracle_pools(ctl, k, v)
        assert np.array_equal(U16(got), U16(oracle.estimate(q, meta_o)))


def test_estimate_is_upper_bound():
    """Domain property: the page score bounds every token's q.k in that page from above."""
    qu = _qu()
    L, H = 2000, 8
    q, k, v = inputs(99, L, H)
    ctl = make_controller(L, H, H, 128, PAGE, 1024)
    fill(ctl, k, v)
    est = qu.decode_estimate(cuda(q), ctl, 0).float().cpu().numpy()
    ctl.end_forward()
    qk = np.einsum("hd,lhd->hl", q[0].astype(np.float64), k.astype(np.float64))
    n = est.shape[1]
    per_page = qk[:, : n * PAGE].reshape(H, n, PAGE).max(-1)
    assert np.all(est + 0.07 * np.maximum(1, np.abs(est) / 64) >= per_page)  # fp16 rounding slack


# ---------------------------------------------------------------------------- top-k

def _topk_dev(vals, in_idx, kk):
    from quest_amd import _kernels

    rows = vals.shape[0]
    ov = torch.zeros(rows, kk, dtype=torch.float16, device="cuda:0")
    oi = torch.zeros(rows, kk, dtype=torch.int32, device="cuda:0")
    buf = torch.zeros(rows, 8, dtype=torch.float16, device="cuda:0")
    _kernels.topk_filtering(cuda(vals), cuda(in_idx), ov, oi, buf, kk)
    return ov.cpu().numpy(), oi.cpu().numpy()


def test_topk_golden_values_and_oracle_indices(golden):
    for seed, n, kk, rows in golden["topk_cases"]:
        n, kk, rows = int(n), int(kk), int(rows)
        vals = golden[f"topk_in_{n}_{kk}"]
        in_idx = np.tile(np.arange(n, dtype=np.int32)[::-1].copy(), (rows, 1))
        ov, oi = _topk_dev(vals, in_idx, kk)
        ev, ei = oracle.topk(vals, in_idx, kk)
        assert np.array_equal(oi, ei) and np.array_equal(U16(ov), U16(ev))
        ref = golden[f"topk_vals_{n}_{kk}"]
        assert np.array_equal(np.sort(ov.astype(np.float32), axis=1)[:, ::-1], ref.astype(np.float32))


@pytest.mark.parametrize("n,kk", [(2047, 127), (8191, 255), (255, 63), (1024, 1024), (1025, 1), (16384, 511),
                                  (4000, 3999), (64, 64), (1, 1), (3, 2)])
def test_topk_ties_bit_exact(n, kk):
    rng = np.random.default_rng(n + kk)
    rows = 8
    vals = np.empty((rows, n), np.float16)
    vals[5] = rng.integers(0, 65536, n).astype(np.uint16).view(np.float16)   # every bit pattern, NaNs included
    vals[6] = (180 + 10 * rng.standard_normal(n)).astype(np.float16)         # one binade, like real page scores
    vals[7] = np.where(rng.random(n) < 0.02, 60000.0, 1e-4 * rng.random(n)).astype(np.float16)  # 2 clusters, wide gap
    vals[0] = (rng.integers(-8, 8, n) * 0.25).astype(np.float16)           # massive ties
    vals[1] = rng.standard_normal(n).astype(np.float16) * 64                 # cfg-like scores
    vals[2] = 1.0                                                            # all equal
    vals[3] = np.where(rng.random(n) < 0.5, 0.0, -0.0).astype(np.float16)   # signed zeros
    vals[4] = rng.standard_normal(n).astype(np.float16)
    vals[4, rng.integers(0, n, max(1, n // 50))] = np.float16(np.inf)
    vals[4, rng.integers(0, n, max(1, n // 50))] = np.float16(-np.inf)
    in_idx = rng.permutation(n).astype(np.int32)[None].repeat(rows, 0)
    ov, oi = _topk_dev(vals, in_idx, kk)
    ev, ei = oracle.topk(vals, in_idx, kk)
    assert np.array_equal(oi, ei)
    assert np.array_equal(U16(ov), U16(ev))


def test_topk_32_heads_cfg3_shape():
    rng = np.random.default_rng(0)
    vals = (rng.standard_normal((32, 2047)) * 40).astype(np.float16)
    in_idx = np.tile(rng.permutation(2048)[:2047].astype(np.int32), (32, 1))
    ov, oi = _topk_dev(vals, in_idx, 127)
    ev, ei = oracle.topk(vals, in_idx, 127)
    assert np.array_equal(oi, ei) and np.array_equal(U16(ov), U16(ev))


# ---------------------------------------------------------------------------- sparse attention

def test_sparse_attention_golden_indices(golden):
    """The reference's decoupled test: feed the ORACLE's top-k pages (test_approx_attention.py:178-196)."""
    qu = _qu()
    for seed, L, H, B, has_idx in golden["approx_cases"]:
        seed, L, H, B = int(seed), int(L), int(H), int(B)
        q, k, v = inputs(seed, L, H)
        for layout in (0, 1, 2):
            ctl = make_controller(L, H, H, 128, PAGE, B, layout=layout, shuffle_seed=seed)
            fill(ctl, k, v)
            table = np.array(ctl.kv_cache.indicies, np.int32)
            if has_idx:
                assert ctl.need_estimate()
                phys = table[golden[f"approx_idx_{L}_{B}"]]
                o = qu.decode_sparse_attn(cuda(q), ctl, 0, cuda(phys))
            else:
                assert not ctl.need_estimate()
                o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.kv_indices_without_last)
                phys = np.tile(table[:-1], (H, 1))
            ctl.end_forward()
            _close(o.cpu().numpy(), golden[f"approx_o_{L}_{B}"])
            kv_o, _ = oracle_pools(ctl, k, v)
            eo, _ = oracle.sparse_attn(q, kv_o, phys if phys.shape[1] else np.zeros((H, 1), np.int32), phys.shape[1],
                                       int(table[-1]), kv_o.last_page_len)
            _close(o.cpu().numpy(), eo, tol=2e-3)


def test_dense_decode_golden(golden):
    qu = _qu()
    for seed, L, H in golden["dense_cases"]:
        seed, L, H = int(seed), int(L), int(H)
        q, k, v = inputs(seed, L, H)
        ctl = make_controller(L, H, H, 128, PAGE, 1024, shuffle_seed=seed)
        fill(ctl, k, v)
        assert not ctl.need_estimate()
        o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.kv_indices_without_last)
        ctl.end_forward()
        _close(o.cpu().numpy(), golden[f"dense_o_{L}"])


@pytest.mark.parametrize("Hq,Hkv,D,page,B", [(32, 8, 128, 16, 12), (8, 2, 64, 16, 9), (8, 8, 256, 16, 5),
                                              (8, 4, 128, 8, 17), (6, 3, 128, 32, 4), (4, 4, 128, 3, 40)])
def test_chain_gqa_dims_pages(Hq, Hkv, D, page, B):
    """Full chain estimate -> top-k -> sparse attention vs the oracle chain (bit-exact up to the
    page selection, then 2e-3)."""
    qu = _qu()
    L = 613
    q, k, v = inputs(70 + Hq + D + page, L, Hq, Hkv, D)
    for layout in (0, 1, 2):
        ctl = make_controller(L, Hq, Hkv, D, page, B, layout=layout, shuffle_seed=2)
        fill(ctl, k, v)
        assert ctl.need_estimate()
        est = qu.decode_estimate(cuda(q), ctl, 0)
        qu.decode_topk(est, ctl)
        o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.topk_dindices_buffer)
        ctl.end_forward()
        kv_o, meta_o = oracle_pools(ctl, k, v)
        e_est = oracle.estimate(q, meta_o)
        assert np.array_equal(U16(est.cpu().numpy()), U16(e_est))
        table = np.array(ctl.kv_cache.indicies, np.int32)
        ev, ei = oracle.topk(e_est, np.tile(table[:-1], (Hq, 1)), B - 1)
        assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
        assert np.array_equal(U16(ctl.topk_dout_buffer.cpu().numpy()), U16(ev))
        eo, _ = oracle.sparse_attn(q, kv_o, ei, B - 1, int(table[-1]), kv_o.last_page_len)
        _close(o.cpu().numpy(), eo, tol=2e-3)


def test_sparse_attention_spiked_softmax():
    """Force the online-softmax rescale branch: one key row aligned with q dominates late in a chunk."""
    qu = _qu()
    L, H = 1000, 4
    q, k, v = inputs(123, L, H)
    k = k.copy()
    for h in range(H):
        k[777, h] = (q[0, h].astype(np.float32) * 3).astype(np.float16)
        k[13, h] = (q[0, h].astype(np.float32) * 2).astype(np.float16)
    ctl = make_controller(L, H, H, 128, PAGE, 1024)
    fill(ctl, k, v)
    o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.kv_indices_without_last)
    ctl.end_forward()
    kv_o, _ = oracle_pools(ctl, k, v)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    eo, _ = oracle.sparse_attn(q, kv_o, np.tile(table[:-1], (H, 1)), len(table) - 1, int(table[-1]), kv_o.last_page_len)
    _close(o.cpu().numpy(), eo, tol=2e-3)


@pytest.mark.parametrize("ppc", [1, 2, 3, 5, 8, 64, 1000])
def test_sparse_attention_any_split(ppc):
    """The result must not depend on how the planner cuts the page list into workgroups."""
    qu = _qu()
    L, H, B = 1541, 8, 55
    q, k, v = inputs(31, L, H)
    ctl = make_controller(L, H, H, 128, PAGE, B, shuffle_seed=4)
    ctl._decode_handler.set_pages_per_chunk(ppc)
    fill(ctl, k, v)
    est = qu.decode_estimate(cuda(q), ctl, 0)
    qu.decode_topk(est, ctl)
    o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.topk_dindices_buffer)
    assert ctl._decode_handler.plan_info()[0] == ppc
    ctl.end_forward()
    kv_o, _ = oracle_pools(ctl, k, v)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    eo, _ = oracle.sparse_attn(q, kv_o, ctl.topk_dindices_buffer.cpu().numpy(), B - 1, int(table[-1]),
                               kv_o.last_page_len)
    _close(o.cpu().numpy(), eo, tol=2e-3)


# ---------------------------------------------------------------------------- rope / rmsnorm

@pytest.mark.parametrize("past,n,Hq,Hkv,D", [(13, 2, 4, 4, 128), (502, 69, 8, 2, 128), (32767, 1, 32, 32, 128),
                                             (77, 5, 4, 4, 64), (1110, 3, 2, 2, 256)])
def test_rope_vs_oracle(past, n, Hq, Hkv, D):
    qu = _qu()
    q = synth.normal_f16(past, (n, Hq, D))
    k = synth.normal_f16(past + 1, (n, Hkv, D))
    qd, kd = cuda(q), cuda(k)
    qu.apply_rope_in_place(qd, kd, past)
    qe, ke = q.copy(), k.copy()
    oracle.rope_in_place(qe, past)
    oracle.rope_in_place(ke, past)
    _close(qd.cpu().numpy(), qe, tol=3e-3)
    _close(kd.cpu().numpy(), ke, tol=3e-3)
    # linear scaling (LongChat, QuestAttention.py:44-48)
    qd2 = cuda(q)
    kd2 = cuda(k)
    qu.apply_rope_in_place(qd2, kd2, past, rope_scale=8.0)
    qe2 = q.copy()
    oracle.rope_in_place(qe2, past, 8.0)
    _close(qd2.cpu().numpy(), qe2, tol=3e-3)


def test_rope_vs_hf_fixtures(rope_golden):
    """HIP rope against the oracle of the reference's rope test (HF functions; make_rope_golden.py), 5e-3."""
    qu = _qu()
    for seed, past, n, H in rope_golden["rope_cases"]:
        seed, past, n, H = int(seed), int(past), int(n), int(H)
        qd = cuda(synth.normal_f16(seed * 3, (n, H, 128)))
        kd = cuda(synth.normal_f16(seed * 3 + 1, (n, H, 128)))
        qu.apply_rope_in_place(qd, kd, past)
        _close(qd.cpu().numpy(), rope_golden[f"rope_q_{past}_{n}"])
        _close(kd.cpu().numpy(), rope_golden[f"rope_k_{past}_{n}"])


@pytest.mark.parametrize("rows,cols", [(1, 4096), (7, 4096), (3, 256), (2, 11008)])
def test_rms_norm_vs_oracle(rows, cols):
    qu = _qu()
    x = synth.normal_f16(cols, (1, rows, cols))
    w = synth.normal_f16(cols + 1, (cols,))
    got = qu.rms_norm_forward(cuda(x), cuda(w), 1e-5).cpu().numpy()
    _close(got, oracle.rms_norm(x, w, 1e-5), tol=2e-3)


# ---------------------------------------------------------------------------- full-size properties (cfg 3)

def test_cfg3_full_size_properties():
    """L=32768, budget 128 pages, H=32, D=128 (BASELINE cfg 3): size-independent properties."""
    qu = _qu()
    L, H, B = 32768, 32, 128
    g = torch.Generator(device="cuda:0").manual_seed(0)
    dev = torch.device("cuda:0")
    k = torch.randn(L, H, 128, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn(L, H, 128, generator=g, device=dev, dtype=torch.float16)
    q = torch.randn(1, H, 128, generator=g, device=dev, dtype=torch.float16)
    ctl = make_controller(L, H, H, 128, PAGE, B, shuffle_seed=9, max_seq_len=L)
    ctl.prepare_metadata(L - 1)
    ctl.begin_forward(L - 1)
    qu.append_kv(k[:-1], v[:-1], ctl, 0)
    ctl.end_forward()
    ctl.prepare_metadata(1)
    ctl.begin_forward(1)
    qu.append_kv(k[-1:], v[-1:], ctl, 0)
    n_pages = L // PAGE
    # (1) metadata == per-page extrema computed by torch on the device (exact: max/min of fp16)
    table = ctl.kv_indices_with_last.long()
    kp = k.view(n_pages, PAGE, H, 128)
    meta = ctl.metadata_cache.buf_layer(0)[ctl.metadata_indices.long()]  # [n_meta, 2, S, H, D]
    assert torch.equal(meta[:, 0].reshape(-1, H, 128)[:n_pages], kp.amax(1))
    assert torch.equal(meta[:, 1].reshape(-1, H, 128)[:n_pages], kp.amin(1))
    # (2) estimate is an upper bound of the true per-page max logit and close to the fp32 formula
    est = qu.decode_estimate(q, ctl, 0)
    ref = torch.maximum(q[0][None].float() * kp.amax(1).float(), q[0][None].float() * kp.amin(1).float()).sum(-1).t()
    torch.testing.assert_close(est.float(), ref[:, :-1].contiguous(), rtol=5e-3, atol=5e-3)
    # (3) top-k: ascending columns, threshold property, matches a stable sort of the same fp16 scores
    qu.decode_topk(est, ctl)
    idx = ctl.topk_dindices_buffer
    order = torch.sort(est.float(), dim=1, descending=True, stable=True).indices[:, : B - 1]
    exp_cols = torch.sort(order, dim=1).values
    assert torch.equal(idx.long(), table[:-1][exp_cols])
    assert torch.equal(ctl.topk_dout_buffer, torch.gather(est, 1, exp_cols))
    # (4) sparse attention == torch fp32 attention over exactly the selected tokens
    o = qu.decode_sparse_attn(q, ctl, 0, idx)
    pages = torch.cat([exp_cols, torch.full((H, 1), n_pages - 1, device=dev)], 1)  # logical
    tok = (pages[:, :, None] * PAGE + torch.arange(PAGE, device=dev)).reshape(H, -1)
    kh = k.transpose(0, 1).float()
    vh = v.transpose(0, 1).float()
    ks = torch.gather(kh, 1, tok[:, :, None].expand(-1, -1, 128))
    vs = torch.gather(vh, 1, tok[:, :, None].expand(-1, -1, 128))
    p = torch.softmax((ks @ q[0].float()[:, :, None]).squeeze(-1) / 128 ** 0.5, dim=-1)
    o_ref = (p[:, None] @ vs).squeeze(1)
    torch.testing.assert_close(o[0].float(), o_ref, rtol=5e-3, atol=5e-3)
    # (5) order invariance: permuting the selected pages changes nothing beyond fp32 rounding
    perm = torch.randperm(B - 1, device=dev)
    o2 = qu.decode_sparse_attn(q, ctl, 0, idx[:, perm].contiguous())
    torch.testing.assert_close(o2.float(), o.float(), rtol=2e-3, atol=2e-3)
    ctl.end_forward()
    # (6) dense path (budget >= pages) == torch fp32 full attention
    ctl.set_page_budget(1 << 20)
    ctl.begin_forward(1, updateTensor=False)
    assert not ctl.need_estimate()
    od = qu.decode_sparse_attn(q, ctl, 0, ctl.kv_indices_without_last)
    pd = torch.softmax((kh @ q[0].float()[:, :, None]).squeeze(-1) / 128 ** 0.5, dim=-1)
    torch.testing.assert_close(od[0].float(), (pd[:, None] @ vh).squeeze(1), rtol=5e-3, atol=5e-3)
    ctl.end_forward()


# ---------------------------------------------------------------------------- fused launches

@pytest.mark.parametrize("Hq,Hkv,D,page,B,L,layout", [(32, 32, 128, 16, 128, 4099, 0), (32, 8, 128, 16, 12, 613, 1),
                                                       (8, 8, 128, 16, 40, 1541, 0), (8, 2, 64, 16, 9, 777, 0),
                                                       (8, 8, 256, 16, 5, 300, 1), (4, 4, 128, 8, 17, 500, 0),
                                                       (8, 8, 128, 16, 200, 9000, 0), (4, 2, 128, 16, 300, 40000, 0)])
def test_fused_equals_unfused(Hq, Hkv, D, page, B, L, layout):
    """append+estimate in one launch and top-k+attention in one launch must reproduce the separate
    ops bit for bit (pools, scores, selected pages) and the attention output exactly as well (a launch forced onto the
    column-range front end, whose work split differs, would be held to <= 2e-3)."""
    qu = _qu()
    q, k, v = inputs(900 + Hq + D + page + B, L, Hq, Hkv, D)
    outs = []
    for fused in (False, True):
        ctl = make_controller(L, Hq, Hkv, D, page, B, layout=layout, shuffle_seed=11)
        kc, vc, qc = cuda(k), cuda(v), cuda(q)
        ctl.prepare_metadata(L - 1)
        ctl.begin_forward(L - 1)
        qu.append_kv(kc[:-1], vc[:-1], ctl, 0)
        ctl.end_forward()
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        assert ctl.need_estimate()
        if fused:
            est = qu.decode_append_estimate(qc, kc[-1:], vc[-1:], ctl, 0)
            o = qu.decode_topk_sparse_attn(qc, est, ctl, 0)
            variant = ctl._decode_handler.last_launch_info()["front_end_variant"]
        else:
            qu.append_kv(kc[-1:], vc[-1:], ctl, 0)
            est = qu.decode_estimate(qc, ctl, 0)
            qu.decode_topk(est, ctl)
            o = qu.decode_sparse_attn(qc, ctl, 0, ctl.topk_dindices_buffer)
        ctl.end_forward()
        n_pages = len(ctl.kv_cache.indicies)
        outs.append((est.cpu().numpy(), ctl.topk_dindices_buffer.cpu().numpy().copy(),
                     ctl.topk_dout_buffer.cpu().numpy().copy(), o.cpu().numpy(),
                     ctl.kv_cache.buf_layer(0)[ctl.kv_indices_with_last.long()].cpu().numpy(),
                     ctl.metadata_cache.buf_layer(0)[ctl.metadata_indices.long()].cpu().numpy(), n_pages))
    a, b = outs
    assert np.array_equal(U16(a[0]), U16(b[0])), "scores"
    assert np.array_equal(a[1], b[1]), "selected pages"
    assert np.array_equal(U16(a[2]), U16(b[2])), "selected values"
    assert variant in (0, 1, 2, 3)
    assert np.array_equal(U16(a[3]), U16(b[3])), "attention output"
    # pools: compare only valid entries (the tail of the last page / last metadata page is uninitialised)
    from _harness import gather_entries
    for x, y, n in ((a[4], b[4], L), (a[5], b[5], a[6])):
        idx = np.arange(x.shape[0])
        kx, vx = gather_entries(x, idx, n, layout)
        ky, vy = gather_entries(y, idx, n, layout)
        assert np.array_equal(U16(kx), U16(ky)) and np.array_equal(U16(vx), U16(vy))


# ---------------------------------------------------------------------------- edge cases

@pytest.mark.parametrize("L,B", [(17, 2), (32, 2), (33, 3), (18, 64), (4096 + 1, 64), (16 * 70, 70), (16 * 70, 69)])
def test_edge_lengths_and_budgets(L, B):
    """Two-page minimum, budget 2 (one selected page), last_page_len 1 and 16, budget == pages and
    budget == pages - 1 -- through both the fused and the op-by-op path, against the oracle chain."""
    qu = _qu()
    H = 4
    q, k, v = inputs(4000 + L + B, L, H)
    for fused in (False, True):
        ctl = make_controller(L, H, H, 128, PAGE, B, shuffle_seed=L)
        kc, vc, qc = cuda(k), cuda(v), cuda(q)
        ctl.prepare_metadata(L - 1)
        ctl.begin_forward(L - 1)
        qu.append_kv(kc[:-1], vc[:-1], ctl, 0) if L - 1 > 1 else qu.append_kv(kc[:1], vc[:1], ctl, 0)
        ctl.end_forward()
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        kv_o = None
        if not ctl.need_estimate():
            qu.append_kv(kc[-1:], vc[-1:], ctl, 0)
            o = qu.decode_sparse_attn(qc, ctl, 0, ctl.kv_indices_without_last)
            table = np.array(ctl.kv_cache.indicies, np.int32)
            sel = np.tile(table[:-1], (H, 1))
        elif fused:
            est = qu.decode_append_estimate(qc, kc[-1:], vc[-1:], ctl, 0)
            o = qu.decode_topk_sparse_attn(qc, est, ctl, 0)
            sel = ctl.topk_dindices_buffer.cpu().numpy()
        else:
            qu.append_kv(kc[-1:], vc[-1:], ctl, 0)
            est = qu.decode_estimate(qc, ctl, 0)
            qu.decode_topk(est, ctl)
            o = qu.decode_sparse_attn(qc, ctl, 0, ctl.topk_dindices_buffer)
            sel = ctl.topk_dindices_buffer.cpu().numpy()
        ctl.end_forward()
        kv_o, meta_o = oracle_pools(ctl, k, v)
        table = np.array(ctl.kv_cache.indicies, np.int32)
        if ctl.need_estimate():
            e_est = oracle.estimate(q, meta_o)
            _, e_sel = oracle.topk(e_est, np.tile(table[:-1], (H, 1)), ctl.inference_page_budget - 1)
            assert np.array_equal(sel, e_sel)
        eo, _ = oracle.sparse_attn(q, kv_o, sel, sel.shape[1], int(table[-1]), kv_o.last_page_len)
        _close(o.cpu().numpy(), eo, tol=2e-3)


def test_error_paths_raise_like_the_reference():
    """Argument errors surface as RuntimeError / ValueError from the op layer (TORCH_CHECK / invalid_argument)."""
    from quest_amd import _kernels

    qu = _qu()
    dev = torch.device("cuda:0")
    ctl = make_controller(100, 4, 4, 128, PAGE, 3)
    q = torch.zeros(1, 4, 128, dtype=torch.float16, device=dev)
    # forward before begin_forward
    ctl.prepare_metadata(100)
    with pytest.raises(RuntimeError, match="begin_forward"):
        ctl._decode_handler.forward(q, torch.empty_like(q), ctl.kv_cache.buf_layer(0),
                                    torch.zeros(4, 2, dtype=torch.int32, device=dev),
                                    torch.zeros(2, dtype=torch.int32, device=dev), 4, 0)
    # wrong dtype
    with pytest.raises(RuntimeError, match="dispatch with dtype"):
        _kernels.apply_rope_in_place(q.float(), q.float(), 0, 1.0, 1e4)
    # k larger than the row
    with pytest.raises(RuntimeError, match="CHECK_GE"):
        _kernels.topk_filtering(torch.zeros(4, 8, dtype=torch.float16, device=dev),
                                torch.zeros(4, 8, dtype=torch.int32, device=dev),
                                torch.zeros(4, 9, dtype=torch.float16, device=dev),
                                torch.zeros(4, 9, dtype=torch.int32, device=dev), None, 9)
    # non-contiguous input
    with pytest.raises(RuntimeError, match="contiguous"):
        _kernels.apply_rope_in_place(torch.zeros(2, 8, 128, dtype=torch.float16, device=dev)[:, ::2],
                                     torch.zeros(2, 4, 128, dtype=torch.float16, device=dev), 0, 1.0, 1e4)
    # head_dim outside the built set
    with pytest.raises(RuntimeError):
        bad = torch.zeros(1, 4, 96, dtype=torch.float16, device=dev)
        _kernels.apply_rope_in_place(bad, bad.clone(), 0, 1.0, 1e4)
    # rms_norm columns not a multiple of 8 (rms_norm.cu:164-166)
    with pytest.raises(RuntimeError):
        x = torch.zeros(1, 2, 100, dtype=torch.float16, device=dev)
        qu.rms_norm_forward(x, torch.zeros(100, dtype=torch.float16, device=dev), 1e-5)


def _rope_fp32(x, pos0, theta=1e4, scale=1.0):
    """Rotate-half RoPE in torch fp32 (the formula of HF's apply_rotary_pos_emb, which the reference's rope test
    uses as its oracle, quest/tests/test_rope.py:17-30); x: [N, H, D], row i at position pos0 + i."""
    N, H, D = x.shape
    inv = theta ** (-torch.arange(0, D, 2, device=x.device, dtype=torch.float32) / D)
    ang = (torch.arange(pos0, pos0 + N, device=x.device, dtype=torch.float32) / scale)[:, None] * inv[None]
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1)[:, None], torch.cat([ang.sin(), ang.sin()], -1)[:, None]
    xf = x.float()
    rot = torch.cat([-xf[..., D // 2:], xf[..., : D // 2]], -1)
    return xf * cos + rot * sin


@pytest.mark.parametrize("Hq,Hkv,L,B,rope_scaling", [(8, 8, 700, 9, None), (8, 2, 1500, 17, {"type": "linear", "factor": 4.0}),
                                                      (4, 4, 100, 64, None)])
def test_quest_attention_module_decode_vs_torch_reference(Hq, Hkv, L, B, rope_scaling):
    """QuestAttention (quest/models/QuestAttention.py:99-157) end to end: prefill L tokens, decode one.
    (1) the fused-launch module and its fused=False form (the reference's op order) give the same bits;
    (2) the cache holds rope(k_proj(x)) / v_proj(x);
    (3) the decode output equals o_proj(fp32 attention over the pages the ORACLE selects from the device's own
        metadata with the module's query) -- or over all pages when the budget covers the cache -- at 5e-3."""
    from types import SimpleNamespace
    from quest_amd.models import QuestAttention

    qu = _qu()
    dev = torch.device("cuda:0")
    hid, D = Hq * 128, 128
    cfg = SimpleNamespace(hidden_size=hid, num_attention_heads=Hq, num_key_value_heads=Hkv, max_position_embeddings=4096,
                          rope_scaling=rope_scaling)
    scale = 1.0 if rope_scaling is None else rope_scaling["factor"]
    outs = []
    for fused in (True, False):
        torch.manual_seed(1)
        m = QuestAttention(cfg, layer_idx=0, fused=fused).to(dev).half()
        ctl = qu.InferenceController(1, Hq, D, PAGE, B, L + 64, torch.float16, dev, num_kv_heads=Hkv, shuffle_seed=6)
        g = torch.Generator(device=dev).manual_seed(2)
        hs = torch.randn(1, L, hid, generator=g, device=dev, dtype=torch.float16) * 0.3
        h1 = torch.randn(1, 1, hid, generator=g, device=dev, dtype=torch.float16) * 0.3
        with torch.inference_mode():
            ctl.prepare_metadata(L)
            ctl.begin_forward(L)
            m(hs, iController=ctl)
            ctl.end_forward()
            ctl.prepare_metadata(1)
            ctl.begin_forward(1)
            sparse = ctl.need_estimate()
            assert sparse == (L // PAGE + 1 > B)
            out, _, _ = m(h1, iController=ctl)
            ctl.end_forward()
        outs.append(out.float().cpu())
    assert outs[0].shape == (1, 1, hid)
    assert torch.equal(outs[0], outs[1])

    # ---- reference for the last (fused=False) module / controller
    with torch.inference_mode():
        n_tok = L + 1
        n_pages = (n_tok + PAGE - 1) // PAGE
        table = torch.tensor(ctl.kv_cache.indicies, device=dev)
        pool = ctl.kv_cache.buf_layer(0)[table]  # [n_pages, 2, S, Hkv, D]
        k_cache = pool[:, 0].reshape(-1, Hkv, D)[:n_tok]
        v_cache = pool[:, 1].reshape(-1, Hkv, D)[:n_tok]
        x_all = torch.cat([hs, h1], 1)[0]
        k_ref = _rope_fp32(m.k_proj(x_all).view(n_tok, Hkv, D), 0, scale=scale)
        torch.testing.assert_close(k_cache.float(), k_ref, rtol=5e-3, atol=5e-3)
        torch.testing.assert_close(v_cache.float(), m.v_proj(x_all).view(n_tok, Hkv, D).float(), rtol=2e-3, atol=2e-3)
        q = m.q_proj(h1).view(1, Hq, D).clone()
        k_tmp = m.k_proj(h1).view(1, Hkv, D).clone()
        qu.apply_rope_in_place(q, k_tmp, L, rope_scale=scale)  # the module's query, bit for bit
        torch.testing.assert_close(q.float(), _rope_fp32(m.q_proj(h1).view(1, Hq, D), L, scale=scale), rtol=5e-3, atol=5e-3)
        if sparse:
            meta = oracle.Paged(ctl.metadata_cache.buf_layer(0).cpu().numpy(), np.array(ctl.metadata_cache.indicies, np.int32),
                                ctl.metadata_cache.last_page_len, 0)
            e_est = oracle.estimate(q.cpu().numpy(), meta)
            cols = np.tile(np.arange(n_pages - 1, dtype=np.int32), (Hq, 1))
            _, logical = oracle.topk(e_est, cols, B - 1)  # logical page numbers
            pages = torch.from_numpy(logical.astype(np.int64)).to(dev)
        else:
            pages = torch.arange(n_pages - 1, device=dev)[None].expand(Hq, -1)
        tok = (pages[:, :, None] * PAGE + torch.arange(PAGE, device=dev)).reshape(Hq, -1)
        tok = torch.cat([tok, torch.arange((n_pages - 1) * PAGE, n_tok, device=dev)[None].expand(Hq, -1)], 1)
        kvh = torch.arange(Hq, device=dev) // (Hq // Hkv)
        ks, vs = k_cache.float()[tok, kvh[:, None]], v_cache.float()[tok, kvh[:, None]]
        p = torch.softmax((ks @ q[0].float()[:, :, None]).squeeze(-1) / D ** 0.5, dim=-1)
        attn = (p[:, None] @ vs).squeeze(1)  # [Hq, D] fp32
        out_ref = attn.reshape(1, 1, hid) @ m.o_proj.weight.float().t()
    torch.testing.assert_close(outs[1].to(dev), out_ref, rtol=5e-3, atol=1e-3)  # |out| ~ 0.05: well below 5e-3 abs


def test_llama_model_decode_runs_with_layer_skip():
    """quest_amd.models.llama: quest_init / forward / quest_clear surface of the reference's model fork; the
    first two layers decode dense, the rest sparse, and the result equals the op-by-op (unfused) model."""
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM

    dev = torch.device("cuda:0")
    cfg = LlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                      num_attention_heads=4, num_key_value_heads=2)
    logits = []
    for fused in (True, False):
        torch.manual_seed(7)
        with torch.device(dev):
            m = LlamaForCausalLM(cfg, fused=fused).half()
        for p_ in m.parameters():
            p_.data.normal_(0, 0.05)
        m.quest_init(16, 1024, token_budget=64)
        ids = torch.arange(200, device=dev)[None] % 512
        with torch.inference_mode():
            m(input_ids=ids)                               # prefill 200 tokens (13 pages > 4-page budget)
            out = None
            for t in range(20):                            # decode across a page boundary
                out = m(input_ids=torch.tensor([[(7 * t) % 512]], device=dev))
        ctl = m.model.iController
        assert ctl.kv_cache.seqlen == 220 and ctl.inference_page_budget == 4
        logits.append(out.float().cpu())
        m.quest_clear()
        assert ctl.kv_cache.seqlen == 0
    assert logits[0].shape == (1, 1, 512) and torch.isfinite(logits[0]).all()
    assert torch.equal(logits[0], logits[1])


@pytest.mark.parametrize("Hq,Hkv,D,L,layout", [(8, 2, 128, 300, 0), (32, 8, 128, 2500, 1), (8, 1, 64, 777, 0),
                                                (4, 4, 128, 100, 0), (8, 2, 256, 200, 0), (16, 2, 128, 5000, 0)])
def test_dense_decode_shared_lists_gqa(Hq, Hkv, D, L, layout):
    """Budget >= pages: all heads attend every page.  quest_amd routes this through the group-shared
    kernel (K/V read once per kv head); D=256 exercises the fallback to per-head lists."""
    qu = _qu()
    q, k, v = inputs(600 + Hq + D + L, L, Hq, Hkv, D)
    ctl = make_controller(L, Hq, Hkv, D, PAGE, 1 << 20, layout=layout, shuffle_seed=L)
    fill(ctl, k, v)
    assert not ctl.need_estimate()
    o = qu.decode_sparse_attn(cuda(q), ctl, 0, ctl.kv_indices_without_last)
    ctl.end_forward()
    kv_o, _ = oracle_pools(ctl, k, v)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    eo, _ = oracle.sparse_attn(q, kv_o, np.tile(table[:-1], (Hq, 1)), len(table) - 1, int(table[-1]), kv_o.last_page_len)
    _close(o.cpu().numpy(), eo, tol=2e-3)


@pytest.mark.parametrize("Hq,Hkv,layout", [(4, 4, 0), (8, 2, 1)])
def test_prefill_attention_whole_and_chunked(Hq, Hkv, layout):
    """prefill_forward (not on the sparse path; torch SDPA over the paged cache): the whole prompt at once and
    the same prompt in two chunks over a growing cache both equal causal attention in fp32."""
    qu = _qu()
    dev = torch.device("cuda:0")
    L, D, split = 203, 128, 77
    g = torch.Generator(device=dev).manual_seed(4)
    q = torch.randn(L, Hq, D, generator=g, device=dev, dtype=torch.float16)
    k = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    kk = k.float().repeat_interleave(Hq // Hkv, 1)
    vv = v.float().repeat_interleave(Hq // Hkv, 1)
    s = torch.einsum("qhd,khd->hqk", q.float(), kk) / D ** 0.5
    s = s.masked_fill(~torch.ones(L, L, dtype=torch.bool, device=dev).tril(), float("-inf"))
    ref = torch.einsum("hqk,khd->qhd", s.softmax(-1), vv)

    ctl = make_controller(L, Hq, Hkv, D, PAGE, 4, layout=layout, shuffle_seed=2)
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    qu.append_kv(k, v, ctl, 0)
    o = qu.prefill_forward(q, ctl, 0)
    ctl.end_forward()
    torch.testing.assert_close(o.float(), ref, rtol=5e-3, atol=5e-3)

    ctl = make_controller(L, Hq, Hkv, D, PAGE, 4, layout=layout, shuffle_seed=3)
    outs = []
    for a, b in ((0, split), (split, L)):
        ctl.prepare_metadata(b - a)
        ctl.begin_forward(b - a)
        qu.append_kv(k[a:b], v[a:b], ctl, 0)
        outs.append(qu.prefill_forward(q[a:b], ctl, 0))
        ctl.end_forward()
    torch.testing.assert_close(torch.cat(outs).float(), ref, rtol=5e-3, atol=5e-3)


def test_topk_on_padded_score_rows_without_a_copy():
    """`decode_append_estimate` returns a `[:, :n]` view of rows padded to 8 columns; `decode_topk` takes it as
    is (the op passes the row stride on) and selects what it selects from the contiguous copy."""
    qu = _qu()
    L, H, B = 16 * 301 + 5, 8, 40  # 302 pages -> 301 columns: not a multiple of 8
    q, k, v = inputs(77, L, H)
    ctl = make_controller(L, H, H, 128, PAGE, B, shuffle_seed=2)
    kc, vc, qc = cuda(k), cuda(v), cuda(q)
    ctl.prepare_metadata(L - 1)
    ctl.begin_forward(L - 1)
    qu.append_kv(kc[:-1], vc[:-1], ctl, 0)
    ctl.end_forward()
    ctl.prepare_metadata(1)
    ctl.begin_forward(1)
    est = qu.decode_append_estimate(qc, kc[-1:], vc[-1:], ctl, 0)
    assert not est.is_contiguous() and est.stride(0) == 304
    qu.decode_topk(est, ctl)
    idx, val = ctl.topk_dindices_buffer.clone(), ctl.topk_dout_buffer.clone()
    ctl.topk_dindices_buffer.fill_(-1)
    qu.decode_topk(est.contiguous(), ctl)
    assert torch.equal(idx, ctl.topk_dindices_buffer) and torch.equal(val, ctl.topk_dout_buffer)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    ev, ei = oracle.topk(est.contiguous().cpu().numpy(), np.tile(table[:-1], (H, 1)), B - 1)
    assert np.array_equal(idx.cpu().numpy(), ei) and np.array_equal(U16(val.cpu().numpy()), U16(ev))
    ctl.end_forward()
