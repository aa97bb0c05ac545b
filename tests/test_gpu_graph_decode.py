"""Device-resident step state: ONE captured hipGraph replayed token after token while the sequence grows
must reproduce the eager per-token path bit for bit -- across KV-page and metadata-page boundaries."""
import numpy as np
import pytest
import torch

from _harness import cuda, inputs, make_controller, oracle_pools, pools_match

pytestmark = pytest.mark.gpu
PAGE = 16


@pytest.mark.parametrize("Hq,Hkv,layout,L0,steps", [(4, 4, 0, 16 * 31 + 10, 45), (8, 2, 1, 16 * 15 + 16, 40),
                                                  (32, 8, 2, 16 * 15 + 16, 40), (32, 32, 2, 16 * 31 + 10, 40)])
def test_graph_replay_matches_eager_over_growing_sequence(Hq, Hkv, layout, L0, steps):
    import quest_amd.utils as qu
    from quest_amd import _kernels

    dev = torch.device("cuda:0")
    layers, D, B = 2, 128, 7
    q0, k0, v0 = inputs(77, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(3)
    new_q = torch.randn(steps, layers, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    def prefilled():
        ctl = make_controller(L0 + steps + 40, Hq, Hkv, D, PAGE, B, layout=layout, shuffle_seed=21, num_layers=layers,
                              max_seq_len=L0 + steps + 40)
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        for l in range(layers):
            qu.append_kv(cuda(k0), cuda(v0), ctl, l)
        ctl.end_forward()
        return ctl

    # ---- eager path, one token at a time (controller re-plans on the host every token)
    ea = prefilled()
    eager_out = []
    for t in range(steps):
        ea.prepare_metadata(1)
        ea.begin_forward(1)
        assert ea.need_estimate()
        outs = []
        for l in range(layers):
            q, k = new_q[t, l].clone(), new_k[t, l].clone()
            qu.apply_rope_in_place(q, k, ea.kv_cache.seqlen - 1)
            est = qu.decode_append_estimate(q, k, new_v[t, l], ea, l)
            outs.append(qu.decode_topk_sparse_attn(q, est, ea, l, write_topk=False))
        ea.end_forward()
        eager_out.append(torch.stack(outs))

    # ---- graph path: capture once, replay `steps` times
    gr = prefilled()
    gr.enable_device_state()
    gr.begin_graph_decode()
    qbuf = torch.empty(layers, 1, Hq, D, device=dev, dtype=torch.float16)
    kbuf = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
    vbuf = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
    # first case: 16-byte aligned score rows (second-generation front end), second: odd stride (first generation)
    scores = qu.score_scratch(gr) if layout != 1 else torch.empty(Hq, gr.max_pages | 1, device=dev, dtype=torch.float16)
    gr._decode_handler.set_front_end(2 if layout == 0 else 1 if layout == 1 else 0)
    obuf = [None] * layers

    def step():
        qu.step_advance_dyn(gr)
        for l in range(layers):
            obuf[l] = qu.decode_layer_dyn(qbuf[l], kbuf[l], vbuf[l], gr, l, scores, apply_rope=True)

    # warm-up on a side stream would advance the state: run it, then restore the state before capture
    qbuf.copy_(new_q[0]); kbuf.copy_(new_k[0]); vbuf.copy_(new_v[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr.sync_device_state()  # back to "before the first decode token" (pool bytes written by the warm-up are rewritten)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    gr.sync_device_state()  # capture does not execute, but keep the invariant explicit

    for t in range(steps):
        qbuf.copy_(new_q[t]); kbuf.copy_(new_k[t]); vbuf.copy_(new_v[t])
        graph.replay()
        gr.prepare_metadata(1)  # host mirror of what the graph's first node did on the device
        got = torch.stack([o.clone() for o in obuf])
        assert torch.equal(got, eager_out[t]), f"token {t}: graph replay differs from eager"
        st = gr.step_state.cpu().tolist()
        kv, meta = gr.kv_cache, gr.metadata_cache
        assert st[:7] == [kv.seqlen, len(kv.indicies), kv.last_page_len, kv.indicies[-1], len(meta.indicies),
                          meta.last_page_len, meta.indicies[-1]]
    gr.end_forward()
    # pools identical at the end
    n_pages = len(ea.kv_cache.indicies)
    assert gr.kv_cache.indicies == ea.kv_cache.indicies and gr.metadata_cache.indicies == ea.metadata_cache.indicies
    assert n_pages > (L0 + PAGE - 1) // PAGE + 1, "the run must cross page boundaries"
    for l in range(layers):
        ia = torch.tensor(ea.kv_cache.indicies[:-1], device=dev)
        assert torch.equal(ea.kv_cache.buf_layer(l)[ia], gr.kv_cache.buf_layer(l)[ia])


def test_model_generation_graph_replay_equals_eager():
    """A whole Llama-architecture model (2 dense layers + sparse layers, GQA) generating greedily: one
    captured hipGraph replayed per token must produce exactly the logits of the eager, host-planned path."""
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM

    dev = torch.device("cuda:0")
    cfg = LlamaConfig(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                      num_attention_heads=4, num_key_value_heads=2)
    prompt = (torch.arange(300, device=dev)[None] * 7) % 256
    n_new = 40  # crosses two page boundaries and (at 320 tokens = 20 pages -> 21) a metadata-page boundary

    def build():
        torch.manual_seed(11)
        with torch.device(dev):
            m = LlamaForCausalLM(cfg).half()
        for p_ in m.parameters():
            p_.data.normal_(0, 0.05)
        m.quest_init(16, 512, token_budget=96)
        with torch.inference_mode():
            first = m(input_ids=prompt)
        return m, first

    eager, first_e = build()
    toks_e, logits_e = [], []
    tok = first_e.argmax(-1)
    with torch.inference_mode():
        for _ in range(n_new):
            toks_e.append(int(tok))
            out = eager(input_ids=tok.view(1, 1))
            logits_e.append(out.float().clone())
            tok = out.argmax(-1)

    graph, first_g = build()
    assert torch.equal(first_e, first_g)
    graph.capture_decode_graph(fused_layers=False)  # the module-by-module step: bit-comparable with the eager path
    tok = first_g.argmax(-1)
    with torch.inference_mode():
        for t in range(n_new):
            assert int(tok) == toks_e[t]
            out = graph.decode_graph_step(input_ids=tok.view(1, 1))
            assert torch.equal(out.float(), logits_e[t]), f"token {t}"
            tok = out.argmax(-1)
    assert graph.model.iController.kv_cache.seqlen == 300 + n_new


def test_state_advance_stops_at_pool_capacity():
    """A replayed graph must never index past the page tables: at capacity the device state stays put and
    flags it (the host mirror raises 'KvPool exhausted')."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    ctl = qu.InferenceController(1, 2, 128, PAGE, 3, 64, torch.float16, dev)  # 4 pages of capacity
    k = torch.zeros(60, 2, 128, dtype=torch.float16, device=dev)
    ctl.prepare_metadata(60)
    ctl.begin_forward(60)
    qu.append_kv(k, k, ctl, 0)
    ctl.end_forward()
    ctl.enable_device_state()
    for _ in range(4):
        qu.step_advance_dyn(ctl)
    st = ctl.step_state.cpu().tolist()
    assert st[0] == 64 and st[1] == 4 and st[2] == 16 and st[7] == 0
    qu.step_advance_dyn(ctl)  # would need a 5th page
    st = ctl.step_state.cpu().tolist()
    assert st[0] == 64 and st[1] == 4 and st[7] == 1
    for _ in range(4):
        ctl.prepare_metadata(1)
    with pytest.raises(RuntimeError, match="KvPool exhausted"):
        ctl.prepare_metadata(1)


@pytest.mark.parametrize("L0,cap", [(5, 0), (16, 0), (40, 0), (23, 72000)])
def test_graph_decode_from_short_context_into_the_sparse_regime(L0, cap):
    """ONE graph captured while the sequence is still shorter than the page budget (down to a single page):
    it attends all pages (the reference's full-attention branch) and slides into the sparse regime as the
    sequence grows.  Reference: the eager host-planned path, which switches branches by itself."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    Hq, Hkv, D, B, layers, steps = 4, 2, 128, 5, 2, 150  # 5-page budget; 150 tokens cross it at page 6
    _, k0, v0 = inputs(31, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(12)
    new_q = torch.randn(steps, layers, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    def prefilled():
        # cap: a pool sized for a long request (4500 pages of capacity -> the state-driven launch is dispatched for
        # 4499-column rows, i.e. the second-generation front end by default) holding a very short live sequence
        ctl = make_controller(L0 + steps + 40, Hq, Hkv, D, PAGE, B, num_layers=layers,
                              max_seq_len=max(cap, L0 + steps + 40))
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        for l in range(layers):
            qu.append_kv(cuda(k0), cuda(v0), ctl, l)
        ctl.end_forward()
        return ctl

    ea = prefilled()
    eager_out, regimes = [], set()
    for t in range(steps):
        ea.prepare_metadata(1)
        ea.begin_forward(1)
        regimes.add(ea.need_estimate())
        outs = []
        for l in range(layers):
            q, k = new_q[t, l].clone(), new_k[t, l].clone()
            qu.apply_rope_in_place(q, k, ea.kv_cache.seqlen - 1)
            if ea.need_estimate():
                est = qu.decode_append_estimate(q, k, new_v[t, l], ea, l)
                outs.append(qu.decode_topk_sparse_attn(q, est, ea, l, write_topk=False))
            else:
                qu.append_kv(k, new_v[t, l], ea, l)
                outs.append(qu.decode_sparse_attn(q, ea, l, ea.kv_indices_without_last))
        ea.end_forward()
        eager_out.append(torch.stack(outs))
    assert regimes == {False, True}, "the run must start dense and end sparse"

    gr = prefilled()
    gr.enable_device_state()
    gr.begin_graph_decode()
    qbuf = torch.empty(layers, 1, Hq, D, device=dev, dtype=torch.float16)
    kbuf = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
    vbuf = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
    scores = qu.score_scratch(gr)
    # second-generation front end, from a one-page sequence into the sparse regime (forced for the small pools; the
    # long-capacity case takes it by default)
    gr._decode_handler.set_front_end(0 if cap else 2)
    obuf = [None] * layers

    def step():
        qu.step_advance_dyn(gr)
        for l in range(layers):
            obuf[l] = qu.decode_layer_dyn(qbuf[l], kbuf[l], vbuf[l], gr, l, scores, apply_rope=True)

    qbuf.copy_(new_q[0]); kbuf.copy_(new_k[0]); vbuf.copy_(new_v[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr.sync_device_state()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    gr.sync_device_state()
    for t in range(steps):
        qbuf.copy_(new_q[t]); kbuf.copy_(new_k[t]); vbuf.copy_(new_v[t])
        graph.replay()
        gr.prepare_metadata(1)
        got = torch.stack([o.clone() for o in obuf])
        # dense regime: same pages, different partial-merge order than the eager full-attention kernel -> tolerance
        torch.testing.assert_close(got.float(), eager_out[t].float(), rtol=2e-3, atol=2e-3)
    for l in range(layers):
        ia = torch.tensor(ea.kv_cache.indicies, device=dev)
        L = ea.kv_cache.seqlen
        assert gr.kv_cache.indicies == ea.kv_cache.indicies and gr.kv_cache.seqlen == L
        a = ea.kv_cache.buf_layer(l)[ia].reshape(-1, 2, PAGE, Hkv, D).transpose(0, 1).reshape(2, -1, Hkv, D)[:, :L]
        b = gr.kv_cache.buf_layer(l)[ia].reshape(-1, 2, PAGE, Hkv, D).transpose(0, 1).reshape(2, -1, Hkv, D)[:, :L]
        assert torch.equal(a, b)


def test_captured_graph_survives_replanning_of_its_handler():
    """A captured step holds the handler's partial-state workspace BY VALUE.  An eager begin_forward with a
    larger plan on the same handler (what bench.py and a server mixing graph and eager steps do) must not
    free that buffer under the graph: replays before and after the re-plan give the same bits."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    Hq, Hkv, D, B, L0, steps = 4, 4, 128, 6, 16 * 60 + 5, 12
    _, k0, v0 = inputs(5, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(8)
    new_q = torch.randn(steps, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    def run(disturb):
        ctl = make_controller(L0 + steps + 40, Hq, Hkv, D, PAGE, B, shuffle_seed=3, max_seq_len=L0 + steps + 40)
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        qu.append_kv(cuda(k0), cuda(v0), ctl, 0)
        ctl.end_forward()
        ctl.enable_device_state()
        ctl.begin_graph_decode()  # small plan: 6 slots
        qb, kb, vb = new_q[0].clone(), new_k[0].clone(), new_v[0].clone()
        scores = torch.empty(Hq, ctl.max_pages, device=dev, dtype=torch.float16)
        ob = [None]

        def step():
            qu.step_advance_dyn(ctl)
            ob[0] = qu.decode_layer_dyn(qb, kb, vb, ctl, 0, scores)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ctl.sync_device_state()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        outs = []
        for t in range(steps):
            qb.copy_(new_q[t]); kb.copy_(new_k[t]); vb.copy_(new_v[t])
            graph.replay()
            ctl.prepare_metadata(1)
            outs.append(ob[0].clone())
            if disturb and t % 3 == 1:
                # eager full-KV decode on the SAME handler: a plan over all ~62 pages needs a far larger workspace
                ctl.end_forward()
                ctl.set_page_budget(1 << 20)
                ctl.begin_forward(1)
                for ppc in (1, 2):  # per-head-list kernel with many chunks, then the shared kernel
                    ctl._decode_handler.set_pages_per_chunk(ppc)
                    ctl.begin_forward(1, updateTensor=False)
                    junk = qu.decode_sparse_attn(new_q[t], ctl, 0, ctl.kv_indices_without_last.clone())
                ctl._decode_handler.set_pages_per_chunk(0)
                ctl.end_forward()
                ctl.set_page_budget(B)
                assert torch.isfinite(junk.float()).all()
        torch.cuda.synchronize()
        return torch.stack(outs)

    a, b = run(False), run(True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("D,page", [(256, 16), (128, 8), (64, 32)])
def test_dense_state_driven_layer_outside_the_shared_kernels_set(D, page):
    """decode_layer_dense_dyn on shapes the group-shared kernel does not cover (head_dim 256, page_size != 16)
    runs the per-head-list kernel over the one page table with the live length from the state; result ==
    the eager full-KV decode, token after token while the sequence grows."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    Hq, Hkv, L0, steps = 4, 2, 5 * page + 3, 2 * page + 3
    _, k0, v0 = inputs(17, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(2)
    new_q = torch.randn(steps, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    def prefilled():
        ctl = make_controller(L0 + steps + 2 * page, Hq, Hkv, D, page, 1 << 20, shuffle_seed=1,
                              max_seq_len=L0 + steps + 2 * page)
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        qu.append_kv(cuda(k0), cuda(v0), ctl, 0)
        ctl.end_forward()
        return ctl

    ea, gr = prefilled(), prefilled()
    gr.enable_device_state()
    gr.begin_graph_decode(dense_layers=True)
    for t in range(steps):
        ea.prepare_metadata(1)
        ea.begin_forward(1)
        assert not ea.need_estimate()
        qu.append_kv(new_k[t], new_v[t], ea, 0)
        o_e = qu.decode_sparse_attn(new_q[t], ea, 0, ea.kv_indices_without_last)
        ea.end_forward()
        qu.step_advance_dyn(gr)
        o_g = qu.decode_layer_dense_dyn(new_q[t], new_k[t], new_v[t], gr, 0)
        gr.prepare_metadata(1)
        torch.testing.assert_close(o_g.float(), o_e.float(), rtol=2e-3, atol=2e-3)
    assert gr.kv_cache.seqlen == ea.kv_cache.seqlen == L0 + steps


def test_stale_decode_graph_is_refused_after_quest_clear():
    """quest_clear() ends the request: the captured step addressed its pages, so replaying it must fail loudly
     until capture_decode_graph() runs again; and a released single-sequence cache hands its pages
    out in the same order for the next request."""
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM

    dev = torch.device("cuda:0")
    cfg = LlamaConfig(vocab_size=128, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                      num_attention_heads=2, num_key_value_heads=2)
    torch.manual_seed(3)
    with torch.device(dev):
        m = LlamaForCausalLM(cfg).half()
    for p_ in m.parameters():
        p_.data.normal_(0, 0.05)
    m.quest_init(16, 256, token_budget=64)
    prompt = (torch.arange(90, device=dev)[None] * 5) % 128

    def generate(n):
        with torch.inference_mode():
            tok = m(input_ids=prompt).argmax(-1)
            m.capture_decode_graph()
            toks = []
            for _ in range(n):
                toks.append(int(tok))
                tok = m.decode_graph_step(input_ids=tok.view(1, 1)).argmax(-1)
        return toks

    first = generate(20)
    order_1 = list(m.model.iController.kv_cache.indicies)
    m.quest_clear()
    with pytest.raises(RuntimeError, match="capture_decode_graph"):
        m.decode_graph_step(input_ids=torch.zeros(1, 1, dtype=torch.long, device=dev))
    second = generate(20)  # same prompt, same weights: same tokens, same physical pages in the same order
    assert second == first
    assert list(m.model.iController.kv_cache.indicies) == order_1


@pytest.mark.parametrize("Hq,Hkv,D,layout,L0", [(8, 8, 128, 0, 70), (8, 2, 128, 1, 33), (4, 1, 64, 0, 16 * 17), (16, 2, 128, 0, 300),
                                                 (32, 32, 128, 0, 4096 - 5), (32, 32, 128, 2, 16 * 40 - 5), (32, 8, 128, 2, 300), (8, 4, 64, 2, 100)])
def test_dense_layer_with_the_append_folded_into_the_attention_launch(Hq, Hkv, D, layout, L0):
    """A full-KV layer of a captured step is TWO launches (attention with the decode append folded in + merge) instead of
    three: same KV pool bytes, same metadata bytes and the same output bits as the separate append launch followed by the
    attention launch, token after token across page boundaries (tokens that open a KV page start its metadata entry from
    the sentinels) and metadata-page boundaries; MHA and GQA, both layouts; and both match the oracle's pools."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    page, steps = 16, 2 * 16 + 5
    _, k0, v0 = inputs(31 + Hq + L0, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(7)
    new_q = torch.randn(steps, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    def prefilled():
        ctl = make_controller(L0 + steps + 2 * page, Hq, Hkv, D, page, 1 << 20, layout=layout, shuffle_seed=3,
                              max_seq_len=L0 + steps + 2 * page)
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        qu.append_kv(cuda(k0), cuda(v0), ctl, 0)
        ctl.end_forward()
        ctl.enable_device_state()
        ctl.begin_graph_decode(dense_layers=True)
        return ctl

    two, three = prefilled(), prefilled()
    for t in range(steps):
        outs = []
        for ctl, fuse in ((two, True), (three, False)):
            qu.step_advance_dyn(ctl)
            outs.append(qu.decode_layer_dense_dyn(new_q[t], new_k[t], new_v[t], ctl, 0, fuse_append=fuse))
            ctl.prepare_metadata(1)
        assert torch.equal(outs[0], outs[1]), f"token {t}: output bits"
    assert torch.equal(two.step_state, three.step_state)
    L = L0 + steps
    k_all = np.concatenate([k0, new_k[:, 0].cpu().numpy()])
    v_all = np.concatenate([v0, new_v[:, 0].cpu().numpy()])
    kv_o, meta_o = oracle_pools(two, k_all, v_all)
    assert pools_match(two, kv_o, meta_o, L) and pools_match(three, kv_o, meta_o, L)
    # the fused launch really took the group-shared kernel (one launch + merge): its handler never saw a per-head list
    kvp = two.kv_cache.buf_layer(0)[torch.as_tensor(two.kv_cache.indicies, device=dev).long()]
    kvq = three.kv_cache.buf_layer(0)[torch.as_tensor(three.kv_cache.indicies, device=dev).long()]
    n_full = (L // page) if L % page == 0 else (L // page)
    assert torch.equal(kvp[:n_full], kvq[:n_full])


@pytest.mark.parametrize("Hq,Hkv,D,layout,L0,B,dense,ppc", [
    (8, 8, 128, 0, 16 * 31 + 10, 7, False, 0),      # sparse chain, several workgroups per head: rides in the merge launch
    (32, 8, 128, 2, 16 * 15 + 16, 6, False, 0),     # GQA, row-rotated pool
    (4, 4, 64, 0, 16 * 40 + 3, 9, False, 9),        # one workgroup per head (no merge launch): issued as its own launch
    (8, 2, 128, 0, 16 * 20 + 5, 1 << 20, True, 0),  # full-KV layers: rides in the dense handler's merge launch
    (8, 8, 128, 1, 16 * 18 + 16, 5, False, 0),      # HND
])
def test_next_tokens_reservation_riding_in_the_last_launch_equals_the_advance_launch(Hq, Hkv, D, layout, L0, B, dense, ppc):
    """Folded stepping (round 6, quest_decode_arm_step_advance): the step_state_advance launch at the head of every captured
    step against the same reservation riding in the LAST layer's merge launch (first token reserved once, up front).  Same
    outputs token after token across KV-page and metadata-page boundaries, same pools; between steps the folded run's device
    state == its host mirror == the other run's state one reservation ahead."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    layers, steps = 3, 45
    q0, k0, v0 = inputs(91, L0, Hq, Hkv, D)
    g = torch.Generator(device=dev).manual_seed(7)
    new_q = torch.randn(steps, layers, 1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    class Run:
        def __init__(self, fold):
            self.fold = fold
            ctl = self.ctl = make_controller(L0 + steps + 40, Hq, Hkv, D, PAGE, B, layout=layout, shuffle_seed=21,
                                             num_layers=layers, max_seq_len=L0 + steps + 40)
            ctl.prepare_metadata(L0)
            ctl.begin_forward(L0)
            for l in range(layers):
                qu.append_kv(cuda(k0), cuda(v0), ctl, l)
            ctl.end_forward()
            ctl.enable_device_state()
            if ppc:
                ctl._decode_handler.set_pages_per_chunk(ppc)
            ctl.begin_graph_decode(dense_layers=dense)
            self.scores = qu.score_scratch(ctl)
            self.q = torch.empty(layers, 1, Hq, D, device=dev, dtype=torch.float16)
            self.k = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
            self.v = torch.empty(layers, 1, Hkv, D, device=dev, dtype=torch.float16)
            self.o = [None] * layers
            if fold:  # the first token's reservation, once
                qu.step_advance_dyn(ctl)
                ctl.prepare_metadata(1)
            self.set_inputs(0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.step()  # warm-up (re-decoded by the first replay: the append is idempotent)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ctl.sync_device_state()  # device <- host mirror (prefilled, + the first reservation when folded)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.step()
            ctl.sync_device_state()

        def set_inputs(self, t):
            self.q.copy_(new_q[t]); self.k.copy_(new_k[t]); self.v.copy_(new_v[t])

        def step(self):
            ctl = self.ctl
            if not self.fold:
                qu.step_advance_dyn(ctl)
            for l in range(layers):
                last = self.fold and l == layers - 1
                if dense:
                    self.o[l] = qu.decode_layer_dense_dyn(self.q[l], self.k[l], self.v[l], ctl, l, apply_rope=True, advance_after=last)
                else:
                    self.o[l] = qu.decode_layer_dyn(self.q[l], self.k[l], self.v[l], ctl, l, self.scores, apply_rope=True,
                                                    advance_after=last)

    sep, fold = Run(False), Run(True)
    if ppc:
        assert fold.ctl._decode_handler.plan_info()[1] == 1  # no merge launch in this plan
    for t in range(steps):
        for r in (sep, fold):
            r.set_inputs(t)
            r.graph.replay()
            r.ctl.prepare_metadata(1)
        for l in range(layers):
            assert torch.equal(sep.o[l], fold.o[l]), f"token {t} layer {l}"
        # device state == host mirror in both runs; the folded one is one reservation ahead
        for r in (sep, fold):
            st = r.ctl.step_state.cpu().tolist()
            kv, meta = r.ctl.kv_cache, r.ctl.metadata_cache
            assert st[:8] == [kv.seqlen, len(kv.indicies), kv.last_page_len, kv.indicies[-1], len(meta.indicies),
                              meta.last_page_len, meta.indicies[-1], 0], (t, r.fold, st)
        assert fold.ctl.kv_cache.seqlen == sep.ctl.kv_cache.seqlen + 1
    L = sep.ctl.kv_cache.seqlen
    assert L == L0 + steps and len(sep.ctl.kv_cache.indicies) > (L0 + PAGE - 1) // PAGE + 1, "the run must cross page boundaries"
    # every appended token and every page's metadata identical (the folded run's extra reserved token holds nothing yet)
    from quest_amd.utils import TensorLayout

    def valid(ctl, cache, n, l):
        pages = cache.buf_layer(l)[torch.tensor(list(cache.indicies), device=dev)]
        x = TensorLayout.to_logical(pages.view(torch.int16), ctl.layout)
        return x.transpose(0, 1).reshape(2, -1, x.shape[-2], x.shape[-1])[:, :n]

    n_pages = (L + PAGE - 1) // PAGE
    for l in range(layers):
        assert torch.equal(valid(sep.ctl, sep.ctl.kv_cache, L, l), valid(fold.ctl, fold.ctl.kv_cache, L, l))
        assert torch.equal(valid(sep.ctl, sep.ctl.metadata_cache, n_pages, l), valid(fold.ctl, fold.ctl.metadata_cache, n_pages, l))
