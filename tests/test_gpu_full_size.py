"""BASELINE configs[3] and configs[4] at FULL size on the MI355X (SURVEY cfg 4 and cfg 5):

* cfg 4 -- Llama-3.1-8B GQA shapes, L = 131072 (8192 pages), Hq = 32 / Hkv = 8, budget 4096 tokens = 256 pages.
  The 8191-long score rows take the routes no small test reaches: the eager path's stand-alone top-k at 8
  columns per thread + index-tensor attention (the fused front end is refused beyond 4096 pages), and the
  state-driven (graph) path's fused front end at 16-32 columns per thread.
* cfg 5 -- 8 independent 32K GQA sequences decoded by ONE batched launch per op over a shared pool.

Checks: metadata == per-page extrema (exact), page scores bit-exact vs the oracle run on the same metadata bytes,
selected pages (values + ids) bit-exact vs the oracle's top-k with the declared tie rule, attention output vs a
torch fp32 attention over exactly the selected tokens at the north-star tolerance (rtol = atol = 5e-3).
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
PAGE, D = 16, 128
DEV = "cuda:0"
U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)


def _torch_attention(q, k, v, logical_pages, L):
    """fp32 attention of q [Hq, D] over the tokens of `logical_pages` [Hq, B-1] (full pages) plus the tokens of
    the sequence's last page; k, v: [L, Hkv, D]."""
    Hq, Hkv = q.shape[0], k.shape[1]
    n_pages = (L + PAGE - 1) // PAGE
    pages = torch.as_tensor(logical_pages, device=q.device, dtype=torch.long)
    tok = (pages[:, :, None] * PAGE + torch.arange(PAGE, device=q.device)).reshape(Hq, -1)
    last = torch.arange((n_pages - 1) * PAGE, L, device=q.device)[None].expand(Hq, -1)
    tok = torch.cat([tok, last], 1)  # [Hq, T]
    kv_head = torch.arange(Hq, device=q.device) // (Hq // Hkv)
    ks = k.float()[tok, kv_head[:, None]]  # [Hq, T, D]
    vs = v.float()[tok, kv_head[:, None]]
    p = torch.softmax((ks @ q.float()[:, :, None]).squeeze(-1) / D ** 0.5, dim=-1)
    return (p[:, None] @ vs).squeeze(1)


def _expected(q_np, meta_pool_np, meta_table, meta_last_len, kv_table, budget, layout=0):
    """Oracle chain on the device's own metadata bytes: scores, then (values, physical ids) of the top budget-1."""
    meta = oracle.Paged(meta_pool_np, np.asarray(meta_table, np.int32), meta_last_len, layout)
    est = oracle.estimate(q_np, meta)
    Hq = q_np.shape[1]
    table = np.asarray(kv_table, np.int32)
    assert est.shape[1] == len(table) - 1
    ev, ei = oracle.topk(est, np.tile(table[:-1], (Hq, 1)), budget - 1)
    return est, ev, ei


def _logical(kv_table, phys):
    inv = np.full(int(max(kv_table)) + 1, -1, np.int64)
    inv[np.asarray(kv_table)] = np.arange(len(kv_table))
    out = inv[phys]
    assert (out >= 0).all()
    return out


def test_cfg4_llama31_gqa_131072_tokens_budget_256_pages():
    import quest_amd.utils as qu

    L, Hq, Hkv, B = 131072, 32, 8, 256
    n_pages = L // PAGE
    g = torch.Generator(device=DEV).manual_seed(4)
    k = torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(1, Hq, D, generator=g, device=DEV, dtype=torch.float16)
    ctl = qu.InferenceController(1, Hq, D, PAGE, B, L + 64, torch.float16, torch.device(DEV), num_kv_heads=Hkv,
                                 shuffle_seed=14)
    ctl.prepare_metadata(L - 1)
    ctl.begin_forward(L - 1)
    qu.append_kv(k[:-1], v[:-1], ctl, 0)
    ctl.end_forward()
    ctl.prepare_metadata(1)
    ctl.begin_forward(1)
    assert ctl.need_estimate() and ctl.inference_page_budget == B
    # ---- eager route through the drop-in API: fused append+estimate into 16-byte aligned score rows, then top-k +
    # attention in ONE launch (the second-generation front end serves the 8191-column rows; so it does, since round 5, on
    # the reference's contiguous [Hq, 8191] layout -- odd row stride, 2-byte aligned rows --, checked below)
    est = qu.decode_append_estimate(q, k[-1:], v[-1:], ctl, 0)
    assert est.shape == (Hq, n_pages - 1) and est.stride(0) % 8 == 0
    kp = k.view(n_pages, PAGE, Hkv, D)
    meta = ctl.metadata_cache.buf_layer(0)[ctl.metadata_indices.long()]  # [n_meta, 2, S, Hkv, D]
    assert torch.equal(meta[:, 0].reshape(-1, Hkv, D)[:n_pages], kp.amax(1))
    assert torch.equal(meta[:, 1].reshape(-1, Hkv, D)[:n_pages], kp.amin(1))
    q_np = q.cpu().numpy()
    meta_np = ctl.metadata_cache.buf_layer(0).cpu().numpy()
    kv_table = list(ctl.kv_cache.indicies)
    e_est, ev, ei = _expected(q_np, meta_np, ctl.metadata_cache.indicies, ctl.metadata_cache.last_page_len, kv_table, B)
    assert np.array_equal(U16(est.cpu().numpy()), U16(e_est)), "page scores not bit-exact"
    o = qu.decode_topk_sparse_attn(q, est, ctl, 0, write_topk=True)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei), "selected pages differ from the oracle"
    assert np.array_equal(U16(ctl.topk_dout_buffer.cpu().numpy()), U16(ev))
    o_chk = torch.empty_like(q)
    assert ctl._decode_handler.forward_fused_topk(q, o_chk, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last, est, None, None,
                                                  ctl.kv_cache.last_page_len, ctl.kv_last_page_idx), "fused launch refused"
    assert torch.equal(o_chk, o)
    est_ref_layout = est.contiguous()  # the reference's layout: rows of 8191 fp16, 2-byte aligned
    assert est_ref_layout.stride(0) == n_pages - 1 and est_ref_layout.stride(0) % 4 != 0
    # round 5: also ONE fused launch (second generation on the aligned stream below each row): same slots, same bits
    o_ref_layout = torch.empty_like(q)
    tv = torch.zeros(Hq, B - 1, dtype=torch.float16, device=DEV)
    ti = torch.full((Hq, B - 1), -1, dtype=torch.int32, device=DEV)
    assert ctl._decode_handler.forward_fused_topk(q, o_ref_layout, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last,
                                                  est_ref_layout, tv, ti, ctl.kv_cache.last_page_len, ctl.kv_last_page_idx)
    info = ctl._decode_handler.last_launch_info()
    assert (info["front_end_variant"], info["waves"]) == (2, 8), info
    assert np.array_equal(ti.cpu().numpy(), ei) and np.array_equal(U16(tv.cpu().numpy()), U16(ev))
    assert torch.equal(o_ref_layout, o)
    ctl.topk_dindices_buffer.fill_(-1)
    ctl._decode_handler.set_front_end(1)  # first generation only: refuses rows beyond 4096 pages -> the reference's own op
    assert not ctl._decode_handler.forward_fused_topk(q, o_chk, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last,  # sequence
                                                      est_ref_layout, None, None, ctl.kv_cache.last_page_len,
                                                      ctl.kv_last_page_idx)
    o_two = qu.decode_topk_sparse_attn(q, est_ref_layout, ctl, 0)  # stand-alone top-k + index-list attention
    ctl._decode_handler.set_front_end(0)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    torch.testing.assert_close(o_two.float(), o.float(), rtol=2e-3, atol=2e-3)
    ctl.end_forward()
    o_ref = _torch_attention(q[0], k, v, _logical(kv_table, ei), L)
    torch.testing.assert_close(o[0].float(), o_ref, rtol=5e-3, atol=5e-3)
    # GQA: the 4 query heads of a group score against the same metadata but select their own pages
    sel = np.sort(ei, axis=1)
    assert any(not np.array_equal(sel[0], sel[j]) for j in range(1, 4))

    # ---- state-driven (graph) route on the same cache: rewind one token on the device state, decode it again
    # (appending the same key/value is idempotent on the pools and on the metadata fold)
    ctl.enable_device_state()
    st = ctl.step_state.cpu().tolist()
    assert st[2] == PAGE  # L is a multiple of the page size: the last token did not open a page
    st[0] -= 1
    st[2] -= 1
    ctl.step_state.copy_(torch.tensor(st, dtype=torch.int32))
    ctl.begin_graph_decode()
    sel_v = torch.zeros(1, Hq, B - 1, dtype=torch.float16, device=DEV)
    sel_i = torch.full((1, Hq, B - 1), -1, dtype=torch.int32, device=DEV)
    ctl._decode_handler.set_selection_out(sel_v, sel_i)
    scores = qu.score_scratch(ctl).zero_()  # aligned rows: the only front end that serves 8191-column rows
    qu.step_advance_dyn(ctl)
    o2 = qu.decode_layer_dyn(q, k[-1:], v[-1:], ctl, 0, scores)
    ctl._decode_handler.set_selection_out(None, None)
    # the launch bench.py --config 4 times since round 5: the tiles front end (the estimate handed the rows' tile maxima
    # over: top-255 tiles, then the exact top-255 over their 2040 scores), slot ownership
    info = ctl._decode_handler.last_launch_info()
    assert (info["keys_per_thread"], info["front_end_variant"], info["waves"], info["specialised"]) == (8, 8, 8, True), info
    assert ctl.step_state.cpu().tolist()[:3] == [L, n_pages, PAGE]
    assert np.array_equal(U16(scores[:, : n_pages - 1].cpu().numpy()), U16(e_est))
    assert np.array_equal(sel_i[0].cpu().numpy(), ei) and np.array_equal(U16(sel_v[0].cpu().numpy()), U16(ev))
    torch.testing.assert_close(o2[0].float(), o_ref, rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(o2.float(), o.float(), rtol=2e-3, atol=2e-3)
    # the same step (the append is idempotent) through the whole-row front ends (round 4's launch: second generation, 24
    # keys per thread) without and with the histogram pre-filter (csrc/topk_bitmap.cuh): same pages, same slots, same bits
    for gen in (2, 3):
        ctl.step_state.copy_(torch.tensor(st, dtype=torch.int32))
        ctl._decode_handler.set_front_end(gen)
        sel_i2 = torch.full_like(sel_i, -1)
        ctl._decode_handler.set_selection_out(None, sel_i2)
        qu.step_advance_dyn(ctl)
        o3 = qu.decode_layer_dyn(q, k[-1:], v[-1:], ctl, 0, qu.score_scratch(ctl).zero_(), tiles=False)
        ctl._decode_handler.set_selection_out(None, None)
        assert torch.equal(sel_i2, sel_i), f"front end {gen}: page lists differ"
        info = ctl._decode_handler.last_launch_info()
        assert (info["keys_per_thread"], info["front_end_variant"]) == (24, 2), info
        assert torch.equal(o3, o2), f"front end {gen}"
    ctl._decode_handler.set_front_end(0)
    ctl.end_forward()

    # ---- dense decode of the same cache (the speed-up's baseline): group-shared kernel == torch fp32 full attention
    ctl.set_page_budget(1 << 20)
    ctl.begin_forward(1)
    od = qu.decode_sparse_attn(q, ctl, 0, ctl.kv_indices_without_last)
    ctl.end_forward()
    kv_head = torch.arange(Hq, device=DEV) // (Hq // Hkv)
    logits = torch.einsum("hd,lhd->hl", q[0].float(), k.float()[:, kv_head]) / D ** 0.5
    od_ref = torch.einsum("hl,lhd->hd", torch.softmax(logits, -1), v.float()[:, kv_head])
    torch.testing.assert_close(od[0].float(), od_ref, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("layout", [0, 2], ids=["NHD", "NHD_ROT"])
def test_cfg5_eight_batched_32k_gqa_sequences(layout):
    import quest_amd.utils as qu

    n, Hq, Hkv, B = 8, 32, 8, 128
    # 32K sequences (config 5); a few tokens shorter on some so that last_page_len / page counts differ in the batch
    lens = [32768, 32768, 32767, 32755, 32768, 32760, 32752, 32768]
    dev = torch.device(DEV)
    b = qu.BatchedInferenceController(n, 1, Hq, D, PAGE, B, 32768 + 64, torch.float16, dev, num_kv_heads=Hkv,
                                      shuffle_seed=55, layout=layout)
    g = torch.Generator(device=DEV).manual_seed(5)
    ks, vs = [], []
    for c, L in zip(b.seqs, lens):
        k = torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        v = torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        c.prepare_metadata(L - 1)
        c.begin_forward(L - 1)
        qu.append_kv(k[:-1], v[:-1], c, 0)
        c.end_forward()
        ks.append(k)
        vs.append(v)
    q = torch.randn(n, Hq, D, generator=g, device=DEV, dtype=torch.float16)
    k1 = torch.stack([k[-1] for k in ks])  # [n, Hkv, D]: the decode token of every sequence
    v1 = torch.stack([v[-1] for v in vs])
    b.enable_device_state()
    b.begin_graph_decode()
    sel_v = torch.zeros(n, Hq, B - 1, dtype=torch.float16, device=DEV)
    sel_i = torch.full((n, Hq, B - 1), -1, dtype=torch.int32, device=DEV)
    b._decode_handler.set_selection_out(sel_v, sel_i)
    scores = qu.score_scratch(b).zero_()
    qu.step_advance_batched(b)
    o = qu.decode_layer_batched(q, k1, v1, b, 0, scores)
    b._decode_handler.set_selection_out(None, None)
    # the launch bench.py --config 5 times: one workgroup per head, staging arrays fed by vector loads (needs the
    # stacked page tables' rows 16-byte aligned: BatchedInferenceController pads the stride to a multiple of 4)
    info = b._decode_handler.last_launch_info()
    assert (info["keys_per_thread"], info["waves"], info["front_end_variant"], info["specialised"]) == (8, 8, 1, True), info
    # the same step through the other front-end generation (the append is idempotent): same pages, same bits
    states_after = b.step_states.clone()
    b.step_states[:, 0] -= 1   # seq_len
    b.step_states[:, 2] -= 1   # kv_last_page_len (every sequence's last page holds >= 2 tokens here)
    assert bool((b.step_states[:, 2] >= 1).all())
    rewound = b.step_states.clone()
    for gen in (1, 2, 3):
        b.step_states.copy_(rewound)
        b._decode_handler.set_front_end(gen)
        sel_i2 = torch.full_like(sel_i, -1)
        b._decode_handler.set_selection_out(None, sel_i2)
        qu.step_advance_batched(b)
        o_gen = qu.decode_layer_batched(q, k1, v1, b, 0, qu.score_scratch(b).zero_())
        b._decode_handler.set_selection_out(None, None)
        assert torch.equal(b.step_states, states_after)
        assert torch.equal(sel_i2, sel_i), f"front end {gen}: page lists differ"
        assert torch.equal(o_gen, o), f"front end {gen}"
    b._decode_handler.set_front_end(0)
    b.prepare_metadata(1)  # host mirror
    torch.cuda.synchronize()
    ppc, chunks = b._decode_handler.plan_info()
    assert chunks == 1  # 8 x 32 (sequence, head) pairs fill the chip: one workgroup per head, no merge launch
    meta_np = b.metadata_layer(0).cpu().numpy()
    states = b.step_states.cpu().tolist()
    for i, (c, L) in enumerate(zip(b.seqs, lens)):
        n_pages = (L + PAGE - 1) // PAGE
        kv_table = list(c.kv_cache.indicies)
        assert states[i][:3] == [L, n_pages, (L - 1) % PAGE + 1] and len(kv_table) == n_pages
        e_est, ev, ei = _expected(q[i:i + 1].cpu().numpy(), meta_np, c.metadata_cache.indicies,
                                  c.metadata_cache.last_page_len, kv_table, B, layout)
        assert np.array_equal(U16(scores[i, :, : n_pages - 1].cpu().numpy()), U16(e_est)), f"sequence {i}: scores"
        assert np.array_equal(sel_i[i].cpu().numpy(), ei), f"sequence {i}: selected pages"
        assert np.array_equal(U16(sel_v[i].cpu().numpy()), U16(ev))
        o_ref = _torch_attention(q[i], ks[i], vs[i], _logical(kv_table, ei), L)
        torch.testing.assert_close(o[i].float(), o_ref, rtol=5e-3, atol=5e-3)
        # the new token landed in the cache and was folded into its page's metadata
        last = c.kv_cache.indicies[-1]
        slot = (L - 1) % PAGE
        page = qu.TensorLayout.to_logical(b.kv_layer(0)[last].view(torch.int16), layout).view(torch.float16)  # [2, S, Hkv, D]
        assert torch.equal(page[0, slot], ks[i][-1]) and torch.equal(page[1, slot], vs[i][-1])


def _prefill(ctl, k, v):
    import quest_amd.utils as qu

    L = k.shape[0]
    ctl.prepare_metadata(L - 1)
    ctl.begin_forward(L - 1)
    qu.append_kv(k[:-1], v[:-1], ctl, 0)
    ctl.end_forward()


@pytest.mark.parametrize("layout", [0, 1, 2], ids=["NHD", "HND", "NHD_ROT"])
def test_cfg3_headline_on_the_timed_path(layout):
    """BASELINE configs[2] (the headline: Hq = Hkv = 32, D = 128, L = 32768, budget 2048 tokens = 128 pages) through the
    launches bench.py TIMES -- device-resident step state, `step_advance_dyn` + `decode_layer_dyn`, pool capacity of the
    bench's default run (2179 pages -> 8 keys per thread, 8-wave workgroups, 16 per head) -- against the oracle: pools,
    page scores, selected pages (ids + values) bit-exact, attention within 5e-3 of fp32 torch over the selected tokens
    and 2e-3 of the eager fused launch.  Asserts the launch IS sparse_decode_kernel<128,16,8,8,3> and that the other
    front ends (column-range ownership, scalar-load staging) select the same pages.  Both pool layouts (the reference's
    default NHD is the bench's; HND puts a (page, head) tile in 4 contiguous KiB)."""
    import quest_amd.utils as qu

    def entries(pool_pages):  # [n, 2, ...] pages of a pool layer -> (K-slot rows, V-slot rows) as [n * S, H, D]
        x = qu.TensorLayout.to_logical(pool_pages.view(torch.int16), layout).view(torch.float16)  # -> [n, 2, S, H, D], head order
        return x[:, 0].reshape(-1, H, D), x[:, 1].reshape(-1, H, D)

    L, H, B = 32768, 32, 128
    n_pages = L // PAGE
    g = torch.Generator(device=DEV).manual_seed(3)
    k = torch.randn(L, H, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(L, H, D, generator=g, device=DEV, dtype=torch.float16)
    q = torch.randn(1, H, D, generator=g, device=DEV, dtype=torch.float16)
    ctl = qu.InferenceController(1, H, D, PAGE, B, L + 2084, torch.float16, torch.device(DEV), shuffle_seed=33, layout=layout)
    _prefill(ctl, k, v)
    # eager fused launches (host-planned entry points) first: they leave the cache at L tokens
    ctl.prepare_metadata(1)
    ctl.begin_forward(1)
    est = qu.decode_append_estimate(q, k[-1:], v[-1:], ctl, 0)
    o_eager = qu.decode_topk_sparse_attn(q, est, ctl, 0, write_topk=True)
    assert ctl._decode_handler.last_launch_info()["front_end_variant"] == 3
    ctl.end_forward()
    kp = k.view(n_pages, PAGE, H, D)
    mmax, mmin = entries(ctl.metadata_cache.buf_layer(0)[ctl.metadata_indices.long()])
    assert torch.equal(mmax[:n_pages], kp.amax(1)) and torch.equal(mmin[:n_pages], kp.amin(1))
    kv_table = list(ctl.kv_cache.indicies)
    e_est, ev, ei = _expected(q.cpu().numpy(), ctl.metadata_cache.buf_layer(0).cpu().numpy(), ctl.metadata_cache.indicies,
                              ctl.metadata_cache.last_page_len, kv_table, B, layout)
    assert np.array_equal(U16(est.cpu().numpy()), U16(e_est))
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    assert np.array_equal(U16(ctl.topk_dout_buffer.cpu().numpy()), U16(ev))
    o_ref = _torch_attention(q[0], k, v, _logical(kv_table, ei), L)
    torch.testing.assert_close(o_eager[0].float(), o_ref, rtol=5e-3, atol=5e-3)

    # ---- the timed path: rewind one token on the device state and decode it again (the append is idempotent)
    ctl.enable_device_state()
    assert ctl.max_pages == 2179
    st = ctl.step_state.cpu().tolist()
    assert st[:3] == [L, n_pages, PAGE]
    st[0] -= 1
    st[2] -= 1
    ctl.begin_graph_decode()
    outs = {}
    for gen in (0, 1):
        ctl.step_state.copy_(torch.tensor(st, dtype=torch.int32))
        ctl._decode_handler.set_front_end(gen)
        sel_v = torch.zeros(1, H, B - 1, dtype=torch.float16, device=DEV)
        sel_i = torch.full((1, H, B - 1), -1, dtype=torch.int32, device=DEV)
        ctl._decode_handler.set_selection_out(sel_v, sel_i)
        scores = qu.score_scratch(ctl).zero_()
        qu.step_advance_dyn(ctl)
        o = qu.decode_layer_dyn(q, k[-1:], v[-1:], ctl, 0, scores)
        ctl._decode_handler.set_selection_out(None, None)
        info = ctl._decode_handler.last_launch_info()
        # 0 = the timed kernel: keys straight into registers; forced generation 1 = scalar-load staging in the generic kernel
        assert info == {"keys_per_thread": 8, "waves": 8, "front_end_variant": {0: 3, 1: 0}[gen],
                        "specialised": gen != 1, "workgroups_per_head": 16, "n_seqs": 1}, info
        assert ctl.step_state.cpu().tolist()[:4] == [L, n_pages, PAGE, kv_table[-1]]
        assert np.array_equal(U16(scores[:, : n_pages - 1].cpu().numpy()), U16(e_est)), "page scores"
        assert np.array_equal(sel_i[0].cpu().numpy(), ei), f"front end {gen}: selected pages"
        assert np.array_equal(U16(sel_v[0].cpu().numpy()), U16(ev)), f"front end {gen}: selected values"
        torch.testing.assert_close(o[0].float(), o_ref, rtol=5e-3, atol=5e-3)
        torch.testing.assert_close(o.float(), o_eager.float(), rtol=2e-3, atol=2e-3)
        outs[gen] = o
    ctl._decode_handler.set_front_end(0)
    assert torch.equal(outs[0], o_eager) and torch.equal(outs[1], o_eager)  # same slot split: same bits
    mmax, mmin = entries(ctl.metadata_cache.buf_layer(0)[ctl.metadata_indices.long()])
    assert torch.equal(mmax[:n_pages], kp.amax(1)) and torch.equal(mmin[:n_pages], kp.amin(1))
    kk, vv = entries(ctl.kv_cache.buf_layer(0)[torch.tensor(kv_table[-1:], device=DEV)])
    slot = (L - 1) % PAGE
    assert torch.equal(kk[slot], k[-1]) and torch.equal(vv[slot], v[-1])


@pytest.mark.parametrize("extra_tokens,stride,one_launch,layout", [(280, 2068, True, 0), (280, 2068, False, 0), (264, 2068, True, 0),
                                                                   (328, 2072, True, 0), (328, 2072, False, 0),
                                                                   # the row-rotated pool (QUEST_LAYOUT_NHD_ROT), the shape it is for
                                                                   (280, 2068, True, 2), (328, 2072, False, 2)])
def test_cfg3_eight_batched_mha_sequences_on_the_timed_path(extra_tokens, stride, one_launch, layout):
    """The `batched_8seq` side measurement of bench.py (8 x cfg-3 MHA sequences) through its timed entry points at the
    bench's own pool capacities -- 2066 / 2065 / 2069 pages per sequence, none a multiple of 4, which used to push every
    batched launch onto the scalar-load front end.  one_launch: the layer as ONE launch (layer_decode_kernel: what the
    bench times since round 5) or as the two launches append+estimate | top-k+attention (the stacked tables' stride is
    padded, the launch is the vector-fed one-variant kernel).  Either way results match the oracle per sequence; the
    one-launch run also repeats the layer with the two launches and requires identical bits."""
    import quest_amd.utils as qu

    n, H, B, L = 8, 32, 128, 32768
    dev = torch.device(DEV)
    b = qu.BatchedInferenceController(n, 1, H, D, PAGE, B, L + extra_tokens, torch.float16, dev, shuffle_seed=77, layout=layout)
    g = torch.Generator(device=DEV).manual_seed(8)
    lens = [L, L - 1, L, L - 13, L, L, L - 8, L]
    ks, vs = [], []
    for c, Li in zip(b.seqs, lens):
        k = torch.randn(Li, H, D, generator=g, device=DEV, dtype=torch.float16)
        v = torch.randn(Li, H, D, generator=g, device=DEV, dtype=torch.float16)
        _prefill(c, k, v)
        ks.append(k)
        vs.append(v)
    q = torch.randn(n, H, D, generator=g, device=DEV, dtype=torch.float16)
    k1 = torch.stack([k[-1] for k in ks])
    v1 = torch.stack([v[-1] for v in vs])
    b.enable_device_state()
    assert b.kv_tables.stride(0) == stride and b.kv_tables.size(1) % 4 != 0
    b.begin_graph_decode()
    sel_v = torch.zeros(n, H, B - 1, dtype=torch.float16, device=DEV)
    sel_i = torch.full((n, H, B - 1), -1, dtype=torch.int32, device=DEV)
    b._decode_handler.set_selection_out(sel_v, sel_i)
    scores = qu.score_scratch(b).zero_()
    qu.step_advance_batched(b)
    assert b.one_launch_layers  # MHA: the default of the timed path
    o = qu.decode_layer_batched(q, k1, v1, b, 0, scores, one_launch=one_launch, write_scores=True)
    info = b._decode_handler.last_launch_info()
    assert info == {"keys_per_thread": 8, "waves": 8, "front_end_variant": 7 if one_launch else 1, "specialised": True,
                    "workgroups_per_head": 1, "n_seqs": 8}, info
    if one_launch:  # the same layer again as two launches (the append is idempotent): identical bits
        sel_v2, sel_i2, scores2 = torch.zeros_like(sel_v), torch.full_like(sel_i, -1), torch.zeros_like(scores)
        b._decode_handler.set_selection_out(sel_v2, sel_i2)
        o2 = qu.decode_layer_batched(q, k1, v1, b, 0, scores2, one_launch=False)
        assert b._decode_handler.last_launch_info()["front_end_variant"] == 1
        assert torch.equal(o, o2) and torch.equal(sel_i, sel_i2) and torch.equal(sel_v, sel_v2)
        for i, Li in enumerate(lens):
            n_out = (Li + PAGE - 1) // PAGE - 1
            assert torch.equal(scores[i, :, :n_out], scores2[i, :, :n_out])
    b._decode_handler.set_selection_out(None, None)
    b.prepare_metadata(1)
    meta_np = b.metadata_layer(0).cpu().numpy()
    for i, (c, Li) in enumerate(zip(b.seqs, lens)):
        n_pages = (Li + PAGE - 1) // PAGE
        kv_table = list(c.kv_cache.indicies)
        e_est, ev, ei = _expected(q[i:i + 1].cpu().numpy(), meta_np, c.metadata_cache.indicies,
                                  c.metadata_cache.last_page_len, kv_table, B, layout)
        assert np.array_equal(U16(scores[i, :, : n_pages - 1].cpu().numpy()), U16(e_est)), f"sequence {i}: scores"
        assert np.array_equal(sel_i[i].cpu().numpy(), ei), f"sequence {i}: selected pages"
        assert np.array_equal(U16(sel_v[i].cpu().numpy()), U16(ev))
        if extra_tokens == 280 or i < 2:  # the fp32 reference of every sequence once; two per further capacity
            o_ref = _torch_attention(q[i], ks[i], vs[i], _logical(kv_table, ei), Li)
            torch.testing.assert_close(o[i].float(), o_ref, rtol=5e-3, atol=5e-3)


def test_cfg2_longchat_4096_tokens_budget_512_pages_full_kv():
    """BASELINE configs[1] at its exact shape (SURVEY cfg 2): Hq = Hkv = 32, D = 128, L = 4096 (256 pages), page budget
    512 >= 256 -> the reference's full-KV branch (QuestAttention.py:123-132, bench_batch_decode.cu's published row).
    The state-driven dense layer (`decode_layer_dense_dyn`: append + group-shared attention + merge) against fp32
    torch attention over the whole context, for the token that fills the 256th page and the one that opens the 257th."""
    import quest_amd.utils as qu

    dev = torch.device(DEV)
    H, L, budget = 32, 4096, 512
    g = torch.Generator(device=dev).manual_seed(21)
    k = torch.randn(L + 1, H, D, generator=g, device=dev, dtype=torch.float16)
    v = torch.randn(L + 1, H, D, generator=g, device=dev, dtype=torch.float16)
    q = torch.randn(2, 1, H, D, generator=g, device=dev, dtype=torch.float16)
    ctl = qu.InferenceController(1, H, D, PAGE, budget, L + 4 * PAGE, torch.float16, dev, shuffle_seed=6)
    ctl.prepare_metadata(L - 1)
    ctl.begin_forward(L - 1)
    qu.append_kv(k[:L - 1], v[:L - 1], ctl, 0)
    ctl.end_forward()
    ctl.enable_device_state()
    ctl.begin_graph_decode(dense_layers=True)
    for t, n_tok in enumerate((L, L + 1)):
        qu.step_advance_dyn(ctl)
        o = qu.decode_layer_dense_dyn(q[t], k[n_tok - 1:n_tok], v[n_tok - 1:n_tok], ctl, 0)
        ctl.prepare_metadata(1)
        assert not ctl.need_estimate()  # 256 / 257 pages <= budget 512: full KV
        kf, vf = k[:n_tok].float(), v[:n_tok].float()
        p = torch.softmax(torch.einsum("hd,lhd->hl", q[t, 0].float(), kf) / D ** 0.5, -1)
        ref = torch.einsum("hl,lhd->hd", p, vf)
        torch.testing.assert_close(o[0].float(), ref, rtol=5e-3, atol=5e-3)
        assert (o[0].float() - ref).abs().max() < 2e-3
    assert ctl.kv_cache.seqlen == L + 1 and len(ctl.kv_cache.indicies) == 257


def test_cfg3_graph_replay_growth_at_full_size():
    """The timed path while the sequence GROWS at the headline shape: one captured step (device-side reservation + fused
    append/estimate + fused top-k/attention + merge) replayed for 90 tokens from 32724 tokens on, across five KV-page
    boundaries and the metadata-page boundary at page 2048.  Every 30 tokens and at the end: page scores and selected
    pages bit-exact vs the oracle on the device's own metadata, attention within 5e-3 of fp32 torch over the selected tokens;
    at the end every appended token is in the pool and every page's metadata is the extrema of its keys."""
    import quest_amd.utils as qu

    H, B, steps = 32, 128, 90
    L0 = 2045 * PAGE + 4
    g = torch.Generator(device=DEV).manual_seed(11)
    k0 = torch.randn(L0, H, D, generator=g, device=DEV, dtype=torch.float16)
    v0 = torch.randn(L0, H, D, generator=g, device=DEV, dtype=torch.float16)
    new_q = torch.randn(steps, 1, H, D, generator=g, device=DEV, dtype=torch.float16)
    new_k = torch.randn(steps, 1, H, D, generator=g, device=DEV, dtype=torch.float16)
    new_v = torch.randn(steps, 1, H, D, generator=g, device=DEV, dtype=torch.float16)
    ctl = qu.InferenceController(1, H, D, PAGE, B, L0 + 2084, torch.float16, torch.device(DEV), shuffle_seed=44)
    ctl.prepare_metadata(L0)
    ctl.begin_forward(L0)
    qu.append_kv(k0, v0, ctl, 0)
    ctl.end_forward()
    ctl.enable_device_state()
    ctl.begin_graph_decode()
    qbuf, kbuf, vbuf = torch.empty_like(new_q[0]), torch.empty_like(new_k[0]), torch.empty_like(new_v[0])
    scores = qu.score_scratch(ctl).zero_()
    sel_v = torch.zeros(1, H, B - 1, dtype=torch.float16, device=DEV)
    sel_i = torch.full((1, H, B - 1), -1, dtype=torch.int32, device=DEV)
    ctl._decode_handler.set_selection_out(sel_v, sel_i)
    out = [None]

    def step():
        qu.step_advance_dyn(ctl)
        out[0] = qu.decode_layer_dyn(qbuf, kbuf, vbuf, ctl, 0, scores)

    qbuf.copy_(new_q[0]); kbuf.copy_(new_k[0]); vbuf.copy_(new_v[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()  # warm-up with token 0's inputs (the first replay appends the same token again: idempotent)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ctl.sync_device_state()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    info = ctl._decode_handler.last_launch_info()
    assert (info["keys_per_thread"], info["waves"], info["front_end_variant"], info["workgroups_per_head"]) == (8, 8, 3, 16), info
    for t in range(steps):
        qbuf.copy_(new_q[t]); kbuf.copy_(new_k[t]); vbuf.copy_(new_v[t])
        graph.replay()
        ctl.prepare_metadata(1)
        if t % 30 == 29 or t == steps - 1:
            L = L0 + t + 1
            n_pages = (L + PAGE - 1) // PAGE
            kv_table = list(ctl.kv_cache.indicies)
            assert len(kv_table) == n_pages and ctl.step_state.cpu().tolist()[:2] == [L, n_pages]
            e_est, ev, ei = _expected(new_q[t].cpu().numpy(), ctl.metadata_cache.buf_layer(0).cpu().numpy(),
                                      ctl.metadata_cache.indicies, ctl.metadata_cache.last_page_len, kv_table, B)
            assert np.array_equal(U16(scores[:, : n_pages - 1].cpu().numpy()), U16(e_est)), f"token {t}: page scores"
            assert np.array_equal(sel_i[0].cpu().numpy(), ei), f"token {t}: selected pages"
            assert np.array_equal(U16(sel_v[0].cpu().numpy()), U16(ev))
            k_all = torch.cat([k0, new_k[: t + 1, 0]])
            v_all = torch.cat([v0, new_v[: t + 1, 0]])
            o_ref = _torch_attention(new_q[t, 0], k_all, v_all, _logical(kv_table, ei), L)
            torch.testing.assert_close(out[0][0].float(), o_ref, rtol=5e-3, atol=5e-3)
    ctl._decode_handler.set_selection_out(None, None)
    L = L0 + steps
    n_pages = (L + PAGE - 1) // PAGE
    assert n_pages == 2051 and len(ctl.metadata_cache.indicies) == 129  # crossed page 2048 = a new metadata page
    k_all = torch.cat([k0, new_k[:, 0]])
    v_all = torch.cat([v0, new_v[:, 0]])
    pool = ctl.kv_cache.buf_layer(0)[torch.tensor(ctl.kv_cache.indicies, device=DEV)]  # [n_pages, 2, S, H, D]
    assert torch.equal(pool[:, 0].reshape(-1, H, D)[:L], k_all) and torch.equal(pool[:, 1].reshape(-1, H, D)[:L], v_all)
    meta = ctl.metadata_cache.buf_layer(0)[torch.tensor(ctl.metadata_cache.indicies, device=DEV)]
    pad = n_pages * PAGE - L
    kp_max = torch.cat([k_all, torch.full((pad, H, D), -65504.0, dtype=torch.float16, device=DEV)]).view(n_pages, PAGE, H, D).amax(1)
    kp_min = torch.cat([k_all, torch.full((pad, H, D), 65504.0, dtype=torch.float16, device=DEV)]).view(n_pages, PAGE, H, D).amin(1)
    assert torch.equal(meta[:, 0].reshape(-1, H, D)[:n_pages], kp_max)
    assert torch.equal(meta[:, 1].reshape(-1, H, D)[:n_pages], kp_min)
