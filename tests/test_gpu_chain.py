"""Chained launch (quest_chain_decode_dyn: append + estimate + top-k + attention of a decode step in ONE grid, the
attention workgroups of a head group waiting on a device counter for that group's scores) must reproduce the
two-launch state-driven chain bit for bit -- outputs, page scores, pools -- eagerly and under hipGraph replay, for
every dispatch order (`lead` = estimate head groups ahead of the attention), live lengths below the capacity the grid
is sized for, MHA / GQA and both pool layouts."""
import pytest
import torch

pytestmark = pytest.mark.gpu
PAGE, D = 16, 128
DEV = "cuda:0"


def _prefilled(k0, v0, L0, Hq, Hkv, layout, budget, layers, cap_tokens):
    import quest_amd.utils as qu

    ctl = qu.InferenceController(layers, Hq, D, PAGE, budget, cap_tokens, torch.float16, torch.device(DEV),
                                 num_kv_heads=Hkv, layout=layout, shuffle_seed=5)
    ctl.prepare_metadata(L0)
    ctl.begin_forward(L0)
    for l in range(layers):
        qu.append_kv(k0[l], v0[l], ctl, l)
    ctl.end_forward()
    ctl.enable_device_state()
    ctl.begin_graph_decode()
    return ctl


@pytest.mark.parametrize("Hq,Hkv,layout,lead,budget,L0,graph", [
    (32, 32, 0, 0, 128, 16 * 1100 + 5, True),    # cfg-3 shapes (4 head groups), default order, graph replay
    (32, 32, 0, 1, 128, 16 * 1100 + 5, False),   # attention of group g right behind its own estimate
    (32, 32, 1, 4, 64, 16 * 1290 + 9, False),    # all estimates first; HND; live length close to the capacity
    (32, 8, 0, 0, 128, 16 * 1100 + 16, True),    # GQA-4: one head group
    (32, 8, 1, 0, 33, 16 * 4200 + 1, False),     # GQA-4, > 4096 pages: second-generation front end, odd budget
    (8, 8, 0, 3, 128, 16 * 1500 + 3, False),     # one head group, many chunks per head
    (64, 64, 0, 2, 128, 16 * 1100 + 5, False),   # 8 head groups
])
def test_chained_launch_equals_the_two_launch_chain(Hq, Hkv, layout, lead, budget, L0, graph):
    import quest_amd.utils as qu

    layers, steps = 2, 24
    cap_tokens = max(16 * 1320, L0 + 16 * 24)
    g = torch.Generator(device=DEV).manual_seed(L0 + Hq)
    k0 = torch.randn(layers, L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    v0 = torch.randn(layers, L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    new_q = torch.randn(steps, layers, 1, Hq, D, generator=g, device=DEV, dtype=torch.float16)
    new_k = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    new_v = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)

    ref = _prefilled(k0, v0, L0, Hq, Hkv, layout, budget, layers, cap_tokens)
    ch = _prefilled(k0, v0, L0, Hq, Hkv, layout, budget, layers, cap_tokens)
    ch._decode_handler.set_chain_lead(lead)
    sc_ref = [qu.score_scratch(ref) for _ in range(layers)]
    sc_ch = [qu.score_scratch(ch) for _ in range(layers)]
    for s in sc_ref + sc_ch:
        s.fill_(float("nan"))
    qbuf = torch.empty(layers, 1, Hq, D, device=DEV, dtype=torch.float16)
    kbuf = torch.empty(layers, 1, Hkv, D, device=DEV, dtype=torch.float16)
    vbuf = torch.empty(layers, 1, Hkv, D, device=DEV, dtype=torch.float16)
    obuf = [None] * layers

    def chained_step():
        qu.step_advance_dyn(ch)
        for l in range(layers):
            obuf[l] = qu.decode_layer_dyn(qbuf[l], kbuf[l], vbuf[l], ch, l, sc_ch[l], apply_rope=True, chain=True)

    replay = chained_step
    if graph:
        qbuf.copy_(new_q[0]); kbuf.copy_(new_k[0]); vbuf.copy_(new_v[0])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chained_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ch.sync_device_state()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            chained_step()
        ch.sync_device_state()
        replay = cg.replay

    pages0 = len(ref.kv_cache.indicies)
    for t in range(steps):
        # reference: the two state-driven launches per layer, eagerly
        qu.step_advance_dyn(ref)
        want = []
        for l in range(layers):
            want.append(qu.decode_layer_dyn(new_q[t, l].clone(), new_k[t, l].clone(), new_v[t, l], ref, l, sc_ref[l],
                                            apply_rope=True, chain=False))
        ref.prepare_metadata(1)
        qbuf.copy_(new_q[t]); kbuf.copy_(new_k[t]); vbuf.copy_(new_v[t])
        replay()
        ch.prepare_metadata(1)
        n = len(ref.kv_cache.indicies) - 1
        for l in range(layers):
            assert torch.equal(sc_ch[l][:, :n], sc_ref[l][:, :n]), f"token {t} layer {l}: page scores differ"
            assert torch.equal(obuf[l], want[l]), f"token {t} layer {l}: chained output differs"
    assert ch._decode_handler.chain_error() == 0
    assert len(ref.kv_cache.indicies) > pages0, "the run must cross a page boundary"
    assert ch.kv_cache.indicies == ref.kv_cache.indicies
    # pools: every full page, and the valid entries of the last one (the rest of it is uninitialised pool memory)
    def valid(buf, page, n):
        return buf[page][:, :n] if layout == 0 else buf[page][:, :, :n]

    for cache_c, cache_r in ((ch.kv_cache, ref.kv_cache), (ch.metadata_cache, ref.metadata_cache)):
        assert cache_c.indicies == cache_r.indicies and cache_c.last_page_len == cache_r.last_page_len
        full = torch.tensor(cache_r.indicies[:-1], device=DEV)
        for l in range(layers):
            assert torch.equal(cache_c.buf_layer(l)[full], cache_r.buf_layer(l)[full])
            assert torch.equal(valid(cache_c.buf_layer(l), cache_r.indicies[-1], cache_r.last_page_len),
                               valid(cache_r.buf_layer(l), cache_r.indicies[-1], cache_r.last_page_len))


def test_chained_launch_declines_shapes_outside_its_set():
    """Short rows (4-wave plans) and page sizes other than 16 fall back to the two launches, silently and correctly."""
    import quest_amd.utils as qu

    Hq = Hkv = 4
    L0 = 16 * 40 + 3
    g = torch.Generator(device=DEV).manual_seed(1)
    k0 = torch.randn(1, L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    v0 = torch.randn(1, L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    a = _prefilled(k0, v0, L0, Hq, Hkv, 0, 8, 1, L0 + 64)
    b = _prefilled(k0, v0, L0, Hq, Hkv, 0, 8, 1, L0 + 64)
    q = torch.randn(1, Hq, D, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    v = torch.randn(1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    sa, sb = qu.score_scratch(a), qu.score_scratch(b)
    o = torch.empty_like(q)
    qu.step_advance_dyn(a)
    assert not a._decode_handler.chain_decode_dyn(k, v, q, o, a.kv_cache.buf_layer(0), a.kv_table_full,
                                                  a.metadata_cache.buf_layer(0), a.meta_table_full, sa, a.step_state,
                                                  a.max_pages - 1)
    got = qu.decode_layer_dyn(q.clone(), k.clone(), v, a, 0, sa, chain=True)
    qu.step_advance_dyn(b)
    want = qu.decode_layer_dyn(q.clone(), k.clone(), v, b, 0, sb, chain=False)
    assert torch.equal(got, want)
