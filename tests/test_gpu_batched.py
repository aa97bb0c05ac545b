"""Batched state-driven decode (n sequences per launch, one shared pool) must reproduce, per sequence, what
the single-sequence state-driven path computes: states, pool bytes, estimates and selections bit for bit,
attention outputs bit for bit when the work split is the same and within the attention tolerance otherwise."""
import os

import pytest
import torch

from _harness import cuda, inputs

pytestmark = pytest.mark.gpu
PAGE = 16


def _prefill(ctl, k, v, layers):
    import quest_amd.utils as qu

    L = k.shape[0]
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    for l in range(layers):
        qu.append_kv(k, v, ctl, l)
    ctl.end_forward()


def _gather(buf_layer, indices, n_entries, layout):
    """Logical [n_entries, H, D] K-slot / V-slot rows of a pool layer (device tensors)."""
    idx = torch.tensor(list(indices), device=buf_layer.device)
    pages = buf_layer[idx]
    if layout == 1:
        pages = pages.permute(0, 1, 3, 2, 4)
    if layout == 2:  # the row-rotated pool (QUEST_LAYOUT_NHD_ROT): back to head order
        from quest_amd.utils.utils import TensorLayout

        pages = TensorLayout.to_logical(pages.view(torch.int16), 2).view(pages.dtype)
    n, _, S, H, D = pages.shape
    return pages[:, 0].reshape(n * S, H, D)[:n_entries], pages[:, 1].reshape(n * S, H, D)[:n_entries]


@pytest.mark.parametrize("Hq,Hkv,layout,lens,same_split,D", [
    (4, 4, 0, (16 * 31 + 10, 16 * 20 + 16, 16 * 40 + 1), True, 128),
    (8, 4, 0, (16 * 12 + 3, 16 * 26 + 16), True, 64),
    # mixed regimes in one batch: shorter than the budget (all pages attended), at the budget, far beyond it
    (4, 4, 0, (19, 16 * 30 + 5, 16 * 6 + 16, 7), True, 128),
    (2, 1, 1, (16 * 18 + 7, 16 * 11 + 1, 16 * 30 + 16), True, 256),
    (8, 2, 1, (16 * 15 + 16, 16 * 33 + 5), True, 128),
    (8, 8, 0, (16 * 70 + 3, 16 * 24 + 15, 16 * 24 + 16, 16 * 50 + 8, 16 * 9 + 9), False, 128),
    # cfg-5 head shapes, 8 sequences: one workgroup per head -> the XCD-aware grid order (heads of a kv-head group on one XCD)
    (32, 8, 0, (16 * 20 + 5, 16 * 9 + 16, 16 * 31 + 1, 16 * 12 + 7, 16 * 25 + 3, 16 * 8 + 2, 16 * 17 + 9, 16 * 30 + 16), False, 128),
    # the row-rotated pool (QUEST_LAYOUT_NHD_ROT): GQA 32 / 8 (rot 3, flip 4), MHA with 32 heads (rot 3, flip 28), head_dim 64
    (32, 8, 2, (16 * 20 + 5, 16 * 9 + 16, 16 * 31 + 1, 16 * 12 + 7, 16 * 25 + 3, 16 * 8 + 2, 16 * 17 + 9, 16 * 30 + 16), False, 128),
    (32, 32, 2, (16 * 31 + 10, 16 * 20 + 16, 16 * 40 + 1), True, 128),
    (8, 4, 2, (16 * 12 + 3, 16 * 26 + 16, 7), True, 64),
])
def test_batched_decode_matches_single_sequence(Hq, Hkv, layout, lens, same_split, D):
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    layers, B, steps = 2, 7, int(os.environ.get("QUEST_SOAK_STEPS", "36"))  # soak: QUEST_SOAK_STEPS=400
    n = len(lens)
    cap = max(lens) + steps + 40
    ks = [cuda(inputs(100 + i, L, Hq, Hkv, D)[1]) for i, L in enumerate(lens)]
    vs = [cuda(inputs(100 + i, L, Hq, Hkv, D)[2]) for i, L in enumerate(lens)]
    g = torch.Generator(device=dev).manual_seed(5)
    new_q = torch.randn(steps, layers, n, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    # ---- reference: every sequence alone (own pools), single-sequence state-driven ops
    singles, single_out, single_scores = [], [], []
    for i, L in enumerate(lens):
        c = qu.InferenceController(layers, Hq, D, PAGE, B, cap, torch.float16, dev, num_kv_heads=Hkv, layout=layout,
                                   shuffle_seed=40 + i)
        _prefill(c, ks[i], vs[i], layers)
        c.enable_device_state()
        if same_split:
            c._decode_handler.set_pages_per_chunk(2)
        c.begin_graph_decode()
        singles.append(c)
    for t in range(steps):
        outs, scs = [], []
        for i, c in enumerate(singles):
            qu.step_advance_dyn(c)
            sc = torch.zeros(layers, Hq, c.max_pages, device=dev, dtype=torch.float16)
            o = []
            for l in range(layers):
                q, k = new_q[t, l, i:i + 1].clone(), new_k[t, l, i:i + 1].clone()
                o.append(qu.decode_layer_dyn(q, k, new_v[t, l, i:i + 1], c, l, sc[l], apply_rope=True))
            c.prepare_metadata(1)
            outs.append(torch.stack(o))  # [layers, 1, Hq, D]
            scs.append(sc)
        single_out.append(torch.cat(outs, dim=1))  # [layers, n, Hq, D]
        single_scores.append(scs)

    # ---- batched: one shared pool, one launch per op for all sequences, captured once and replayed
    b = qu.BatchedInferenceController(n, layers, Hq, D, PAGE, B, cap, torch.float16, dev, num_kv_heads=Hkv,
                                      layout=layout, shuffle_seed=9)
    for i in range(n):
        _prefill(b.seqs[i], ks[i], vs[i], layers)
    b.enable_device_state()
    if same_split:
        b._decode_handler.set_pages_per_chunk(2)
    b._decode_handler.set_front_end(2)  # second-generation front end (aligned score rows below) vs the reference's first
    b.begin_graph_decode()
    qbuf = torch.empty(layers, n, Hq, D, device=dev, dtype=torch.float16)
    kbuf = torch.empty(layers, n, Hkv, D, device=dev, dtype=torch.float16)
    vbuf = torch.empty(layers, n, Hkv, D, device=dev, dtype=torch.float16)
    # rows padded to 16 bytes -> second-generation front end; the single-sequence reference above uses its pool's own
    # page count as the stride (first generation unless that happens to be a multiple of 4): a cross-check of the two
    scores = torch.zeros(layers, n, Hq, (b.max_pages + 7) // 8 * 8, device=dev, dtype=torch.float16)
    obuf = torch.empty(layers, n, Hq, D, device=dev, dtype=torch.float16)

    def step():
        qu.step_advance_batched(b)
        for l in range(layers):
            qu.decode_layer_batched(qbuf[l], kbuf[l], vbuf[l], b, l, scores[l], apply_rope=True, out=obuf[l])

    qbuf.copy_(new_q[0]); kbuf.copy_(new_k[0]); vbuf.copy_(new_v[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()  # warm-up; the same token is folded again by the first replay (idempotent)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    b.sync_device_state()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    b.sync_device_state()

    for t in range(steps):
        qbuf.copy_(new_q[t]); kbuf.copy_(new_k[t]); vbuf.copy_(new_v[t])
        graph.replay()
        b.prepare_metadata(1)
        st = b.step_states.cpu().tolist()
        for i, c in enumerate(b.seqs):
            kv, meta = c.kv_cache, c.metadata_cache
            assert st[i][:7] == [kv.seqlen, len(kv.indicies), kv.last_page_len, kv.indicies[-1], len(meta.indicies),
                                 meta.last_page_len, meta.indicies[-1]], f"token {t} seq {i}: state"
            assert st[i][7] == 0
            n_out = len(kv.indicies) - 1
            for l in range(layers):  # estimates are bit-exact whatever the work split
                assert torch.equal(scores[l, i, :, :n_out], single_scores[t][i][l][:, :n_out]), \
                    f"token {t} seq {i} layer {l}: estimate"
        if same_split:
            assert torch.equal(obuf, single_out[t]), f"token {t}: batched output differs from single-sequence"
        else:
            torch.testing.assert_close(obuf.float(), single_out[t].float(), rtol=2e-3, atol=2e-3)

    # pools: every sequence's logical KV rows and metadata entries are the same bits as in its own pool
    for i, (c, s) in enumerate(zip(b.seqs, singles)):
        L = c.kv_cache.seqlen
        assert L == lens[i] + steps == s.kv_cache.seqlen
        for l in range(layers):
            bk, bv = _gather(b.kv_layer(l), c.kv_cache.indicies, L, layout)
            sk, sv = _gather(s.kv_cache.buf_layer(l), s.kv_cache.indicies, L, layout)
            assert torch.equal(bk, sk) and torch.equal(bv, sv), f"seq {i} layer {l}: KV pool"
            np_ = len(c.kv_cache.indicies)
            bmx, bmn = _gather(b.metadata_layer(l), c.metadata_cache.indicies, np_, layout)
            smx, smn = _gather(s.metadata_cache.buf_layer(l), s.metadata_cache.indicies, np_, layout)
            assert torch.equal(bmx, smx) and torch.equal(bmn, smn), f"seq {i} layer {l}: metadata pool"
    # no page is owned by two sequences
    owned = [p for c in b.seqs for p in c.kv_cache.indicies]
    assert len(owned) == len(set(owned))


def test_batched_dense_layer_matches_single_sequence():
    """Full-KV (dense) layers of a batched step: append + group-shared attention over all pages."""
    import quest_amd.utils as qu
    from quest_amd.utils.decode_wrapper import BatchDecodeWithPagedKVCacheWrapper

    dev = torch.device("cuda:0")
    Hq, Hkv, D, B, layers, steps = 8, 2, 128, 6, 1, 20
    lens = (16 * 12 + 16, 16 * 30 + 7, 16 * 21 + 1)
    n, cap = len(lens), max(lens) + steps + 30
    ks = [cuda(inputs(200 + i, L, Hq, Hkv, D)[1]) for i, L in enumerate(lens)]
    vs = [cuda(inputs(200 + i, L, Hq, Hkv, D)[2]) for i, L in enumerate(lens)]
    g = torch.Generator(device=dev).manual_seed(8)
    new_q = torch.randn(steps, n, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    singles = []
    for i, L in enumerate(lens):
        c = qu.InferenceController(layers, Hq, D, PAGE, B, cap, torch.float16, dev, num_kv_heads=Hkv)
        _prefill(c, ks[i], vs[i], layers)
        c.enable_device_state()
        c._dense_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout="NHD")
        c._dense_handler.set_pages_per_chunk(4)  # same work split as the batched plan -> bit-identical merge
        c.begin_graph_decode(dense_layers=True)
        singles.append(c)
    b = qu.BatchedInferenceController(n, layers, Hq, D, PAGE, B, cap, torch.float16, dev, num_kv_heads=Hkv)
    for i in range(n):
        _prefill(b.seqs[i], ks[i], vs[i], layers)
    b.enable_device_state()
    b._dense_handler = BatchDecodeWithPagedKVCacheWrapper(kv_layout="NHD")
    b._dense_handler.set_pages_per_chunk(4)
    b.begin_graph_decode(dense_layers=True)
    for t in range(steps):
        ref = []
        for i, c in enumerate(singles):
            qu.step_advance_dyn(c)
            q, k = new_q[t, i:i + 1].clone(), new_k[t, i:i + 1].clone()
            ref.append(qu.decode_layer_dense_dyn(q, k, new_v[t, i:i + 1], c, 0, apply_rope=True))
            c.prepare_metadata(1)
        qu.step_advance_batched(b)
        q, k = new_q[t].clone(), new_k[t].clone()
        got = qu.decode_layer_dense_batched(q, k, new_v[t], b, 0, apply_rope=True)
        b.prepare_metadata(1)
        assert torch.equal(got, torch.cat(ref)), f"token {t}"


@pytest.mark.skipif(os.environ.get("QUEST_TORCH_CHECK") == "0", reason="argument validation is switched off")
def test_batched_argument_errors():
    import quest_amd.utils as qu
    from quest_amd import _kernels

    dev = torch.device("cuda:0")
    b = qu.BatchedInferenceController(2, 1, 4, 128, PAGE, 4, 400, torch.float16, dev)
    for c in b.seqs:
        k = torch.zeros(200, 4, 128, device=dev, dtype=torch.float16)
        _prefill(c, k, k, 1)
    b.enable_device_state()
    q = torch.zeros(2, 4, 128, device=dev, dtype=torch.float16)
    o = torch.empty_like(q)
    scores = torch.zeros(2, 4, b.max_pages, device=dev, dtype=torch.float16)
    # handler planned for one sequence only -> refuses a batch of two
    b._decode_handler.begin_forward(torch.tensor([0, 3], dtype=torch.int32), 4, 4, 128, PAGE, torch.float16)
    with pytest.raises(RuntimeError, match="begin_forward"):
        b._decode_handler.forward_fused_topk_batched(q, o, b.kv_layer(0), b.kv_tables, scores, b.step_states,
                                                     b.max_pages - 1)
    b.begin_graph_decode()
    # q batch size disagrees with the states
    with pytest.raises(RuntimeError):
        b._decode_handler.forward_fused_topk_batched(q[:1], o[:1], b.kv_layer(0), b.kv_tables, scores, b.step_states,
                                                     b.max_pages - 1)
    with pytest.raises(RuntimeError):
        _kernels.append_estimate_batched(q[:1], q[:1], b.kv_layer(0), b.kv_tables, q, scores, b.metadata_layer(0),
                                         b.meta_tables, b.step_states, b.max_pages - 1, 0)
    with pytest.raises(ValueError):
        b._decode_handler.set_batch(0)


@pytest.mark.parametrize("fused,lens", [(False, (300, 215, 330)), (True, (300, 215, 330)),
                                        (True, (300, 215, 330, 180, 257, 199, 310, 222, 176))])  # 9 sequences: 3 token groups
def test_model_batched_generation_matches_single_sequence(fused, lens):
    """A Llama-architecture model (2 dense layers + sparse layers, GQA) decoding three prompts of different
    lengths together -- one hipGraph replay per token for the whole batch -- against the same model decoding
    each prompt alone (teacher-forced with the single-sequence greedy tokens).  fused: the decoder layers' projections
    as the fused launches of csrc/decode_layer.hip (batch-1 kernel for the single sequences, the n-token MFMA kernel for
    the batch) instead of nn.Linear + separate norm / RoPE / activation launches."""
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM

    dev = torch.device("cuda:0")
    cfg = LlamaConfig(vocab_size=256, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                      num_attention_heads=4, num_key_value_heads=2)
    prompts = [((torch.arange(L, device=dev)[None] * (5 + 2 * i)) + i) % 256 for i, L in enumerate(lens)]
    n_new = 24 if len(lens) <= 3 else 10

    def build():
        torch.manual_seed(11)
        with torch.device(dev):
            m = LlamaForCausalLM(cfg).half()
        for p_ in m.parameters():
            p_.data.normal_(0, 0.05)
        return m

    # each prompt alone: graph-replayed single-sequence generation
    toks, logits = [], []
    for pr in prompts:
        m = build()
        m.quest_init(16, 512, token_budget=96)
        with torch.inference_mode():
            first = m(input_ids=pr)
        m.capture_decode_graph(fused_layers=fused)
        tk, lg = [int(first.argmax(-1))], [first.float().clone()]
        with torch.inference_mode():
            for _ in range(n_new):
                out = m.decode_graph_step(input_ids=torch.tensor([[tk[-1]]], device=dev))
                lg.append(out.float().clone())
                tk.append(int(out.argmax(-1)))
        toks.append(tk)
        logits.append(lg)
        del m

    mb = build()
    mb.quest_init_batched(len(prompts), 16, 512, token_budget=96)
    with torch.inference_mode():
        for i, pr in enumerate(prompts):
            first = mb.prefill_sequence(i, pr)
            assert torch.equal(first.float(), logits[i][0]), "prefill over the shared pool differs"
    mb.capture_decode_graph_batched(fused_layers=fused)
    assert mb.fused_layers == fused
    errs = []
    with torch.inference_mode():
        for t in range(n_new):
            ids = torch.tensor([toks[i][t] for i in range(len(prompts))], device=dev)
            out = mb.decode_graph_step_batched(ids).float()
            for i in range(len(prompts)):
                # GEMMs run with M = 3 instead of M = 1 (different kernels / summation order), so the comparison is a
                # tolerance, not bits; logits are O(1).  The sparse layers' page selection is a discrete function of
                # q: a rounding-level difference can swap a page at a few positions, which moves those logits more
                errs.append(float((out[i].reshape(-1) - logits[i][t + 1][0].reshape(-1)).abs().max()))
    errs = torch.tensor(errs)
    assert float(errs.median()) < 3e-2 and float((errs < 3e-2).float().mean()) >= 0.8 and float(errs.max()) < 0.5, errs.tolist()
    for i, c in enumerate(mb.model.bController.seqs):
        assert c.kv_cache.seqlen == lens[i] + n_new


@pytest.mark.parametrize("Hq,Hkv,layout,D,lens,budgets", [
    # ragged lengths, a budget per sequence: far below the page count, above it (-> all pages), exactly it, one page
    (8, 8, 0, 128, (16 * 40 + 3, 16 * 9 + 16, 16 * 25 + 1, 5), (7, 30, 26, 4)),
    (8, 2, 1, 128, (16 * 33 + 9, 16 * 12 + 2, 16 * 20 + 16), (12, 5, 21)),
    (4, 4, 0, 64, (16 * 18 + 7, 16 * 50 + 16), (9, 3)),
    (32, 8, 2, 128, (16 * 33 + 9, 16 * 12 + 2, 16 * 20 + 16), (12, 5, 21)),
    (16, 16, 2, 128, (16 * 40 + 3, 16 * 9 + 16, 5), (7, 30, 4)),
])
def test_batched_eager_ops_with_per_sequence_budgets_match_single_sequence_ops(Hq, Hkv, layout, D, lens, budgets):
    """The four operators of a decode step one by one for a whole batch, WITHOUT a captured graph and with a page budget
    per sequence (replaces the reference's per-request loop, quest/utils/controller.py:80-129 + utils/__init__.py:141-276):
    append_kv_batched / decode_estimate_batched / decode_topk_batched / decode_sparse_attn_batched must give, per sequence,
    the bits of the single-sequence reference-signature operators run on that sequence alone with ITS budget; the fused
    batched layer with budgets must select the same pages."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    layers, steps, n = 2, 20, len(lens)
    cap = max(lens) + steps + 40
    ks = [cuda(inputs(300 + i, L, Hq, Hkv, D)[1]) for i, L in enumerate(lens)]
    vs = [cuda(inputs(300 + i, L, Hq, Hkv, D)[2]) for i, L in enumerate(lens)]
    g = torch.Generator(device=dev).manual_seed(12)
    new_q = torch.randn(steps, layers, n, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)

    singles = []
    for i, L in enumerate(lens):
        c = qu.InferenceController(layers, Hq, D, PAGE, budgets[i], cap, torch.float16, dev, num_kv_heads=Hkv, layout=layout,
                                   shuffle_seed=70 + i)
        _prefill(c, ks[i], vs[i], layers)
        c._decode_handler.set_pages_per_chunk(2)
        singles.append(c)
    b = qu.BatchedInferenceController(n, layers, Hq, D, PAGE, max(budgets), cap, torch.float16, dev, num_kv_heads=Hkv,
                                      layout=layout, shuffle_seed=5)
    for i in range(n):
        _prefill(b.seqs[i], ks[i], vs[i], layers)
    b.set_page_budgets(budgets)
    b.enable_device_state()
    b._decode_handler.set_pages_per_chunk(2)

    for t in range(steps):
        # ---- single-sequence reference flow (host-planned, reference op signatures), one request after the other
        ref_o, ref_sel = [], []
        for i, c in enumerate(singles):
            c.prepare_metadata(1)
            c.begin_forward(1)
            oo, ss = [], []
            for l in range(layers):
                q, k, v = new_q[t, l, i:i + 1], new_k[t, l, i:i + 1], new_v[t, l, i:i + 1]
                qu.append_kv(k, v, c, l)
                if c.need_estimate():
                    est = qu.decode_estimate(q, c, l)
                    qu.decode_topk(est, c)
                    oo.append(qu.decode_sparse_attn(q, c, l, c.topk_dindices_buffer))
                    ss.append(c.topk_dindices_buffer.clone())
                else:  # within its budget: every page (QuestAttention.py:123-132)
                    oo.append(qu.decode_sparse_attn(q, c, l, c.kv_indices_without_last))
                    ss.append(None)
            c.end_forward()
            ref_o.append(torch.cat(oo))   # [layers, Hq, D]
            ref_sel.append(ss)
        # ---- the batch: one host -> device copy of the lengths, then one launch per operator for all sequences
        b.prepare_metadata(1)
        b.begin_forward()
        for l in range(layers):
            q, k, v = new_q[t, l], new_k[t, l], new_v[t, l]
            qu.append_kv_batched(k, v, b, l)
            est = qu.decode_estimate_batched(q, b, l)
            qu.decode_topk_batched(est, b)
            o = qu.decode_sparse_attn_batched(q, b, l, b.topk_dindices_buffer)
            sel_i = torch.full((n, Hq, b.inference_page_budget - 1), -1, dtype=torch.int32, device=dev)
            b._decode_handler.set_selection_out(None, sel_i)
            o_fused = qu.decode_layer_batched(q, k, v, b, l, qu.score_scratch(b).zero_())  # (the append is idempotent)
            b._decode_handler.set_selection_out(None, None)
            for i, c in enumerate(singles):
                pages = len(c.kv_cache.indicies)
                k_i = min(budgets[i] - 1, pages - 1)
                if ref_sel[i][l] is not None:
                    assert k_i == budgets[i] - 1
                    assert torch.equal(_logical_pages(b.seqs[i], b.topk_dindices_buffer[i, :, :k_i]),
                                       _logical_pages(c, ref_sel[i][l])), f"token {t} seq {i} layer {l}: selected pages"
                    assert torch.equal(_logical_pages(b.seqs[i], sel_i[i, :, :k_i]), _logical_pages(c, ref_sel[i][l]))
                else:  # all pages, in table order
                    want = torch.arange(pages - 1).expand(Hq, -1)
                    assert torch.equal(_logical_pages(b.seqs[i], b.topk_dindices_buffer[i, :, :k_i]), want)
                assert torch.equal(o[i], ref_o[i][l]) or ref_sel[i][l] is None, f"token {t} seq {i} layer {l}: output bits"
                torch.testing.assert_close(o[i].float(), ref_o[i][l].float(), rtol=2e-3, atol=2e-3)
                torch.testing.assert_close(o_fused[i].float(), ref_o[i][l].float(), rtol=2e-3, atol=2e-3)
        b.end_forward()
    for i, (c, s) in enumerate(zip(b.seqs, singles)):
        assert c.kv_cache.seqlen == s.kv_cache.seqlen == lens[i] + steps


def _logical_pages(ctl, phys):
    """Physical page ids [H, k] -> logical page numbers of the controller's sequence (CPU int64)."""
    table = ctl.kv_cache.indicies
    inv = {p: j for j, p in enumerate(table)}
    return torch.tensor([[inv[int(x)] for x in row] for row in phys.cpu().tolist()], dtype=torch.int64).reshape(phys.shape[0], -1)


@pytest.mark.parametrize("Hq,Hkv,layout,D,B,lens,budgets,odd_q", [
    # MHA, lengths across page / metadata-page boundaries during the replay, one sequence within the budget, one with 1 page
    (4, 4, 0, 128, 7, (16 * 31 + 10, 16 * 15 + 16, 16 * 40 + 1, 19, 7), None, False),
    (8, 8, 1, 128, 12, (16 * 47 + 3, 16 * 12 + 15, 16 * 16 + 16), None, False),
    (4, 4, 0, 64, 9, (16 * 33 + 9, 16 * 50 + 16), None, False),
    # per-sequence budgets below / above the page count
    (4, 4, 0, 128, 30, (16 * 40 + 3, 16 * 9 + 16, 16 * 25 + 1, 5), (7, 30, 26, 4), False),
    # zero / infinite query elements: the literal form of the estimate
    (4, 4, 0, 128, 7, (16 * 21 + 5, 16 * 35 + 16), None, True),
    # GQA: the query heads of a group are separate workgroups that re-read their kv head's metadata; one of them appends
    (8, 2, 0, 128, 7, (16 * 22 + 3, 16 * 30 + 16, 16 * 17 + 1), None, False),
    (32, 8, 0, 128, 6, (16 * 20 + 5, 16 * 9 + 16, 16 * 31 + 1, 16 * 12 + 7, 16 * 25 + 3, 16 * 8 + 2, 16 * 17 + 9, 16 * 30 + 16), None, False),
    # round 6: up to 255 selected pages in one workgroup (token budget 4096 = 256 pages, the reference's largest:
    # scripts/passkey.sh:11); one sequence still shorter than the budget
    (4, 4, 0, 128, 256, (16 * 300 + 10, 16 * 280 + 16, 16 * 100 + 3), None, False),
    (8, 8, 2, 128, 200, (16 * 260 + 1, 16 * 199 + 16), None, False),
    (4, 4, 0, 64, 256, (16 * 270 + 7, 16 * 400 + 16), (256, 130), False),
    # the row-rotated pool: MHA with 32 heads (the shape the layout is for), 4 heads, head_dim 64, GQA, odd q
    (32, 32, 2, 128, 9, (16 * 31 + 10, 16 * 15 + 16, 16 * 40 + 1, 19), None, False),
    (4, 4, 2, 128, 7, (16 * 31 + 10, 16 * 15 + 16, 16 * 40 + 1, 19, 7), None, True),
    (8, 8, 2, 64, 9, (16 * 33 + 9, 16 * 50 + 16), None, False),
    (32, 8, 2, 128, 6, (16 * 20 + 5, 16 * 9 + 16, 16 * 31 + 1, 16 * 12 + 7), None, False),
])
def test_one_launch_layer_equals_two_launches(Hq, Hkv, layout, D, B, lens, budgets, odd_q):
    """The one-launch layer of a batched step (csrc/layer_device.cuh: a workgroup per (sequence, head) appends, scores its
    head's pages into LDS, selects from LDS and gathers) against the two launches it replaces (append+estimate |
    top-k+attention), both captured and replayed over the same tokens with one workgroup per head: page scores, selected
    values and pages, outputs, step states and -- at the end -- every pool byte must be IDENTICAL (torch.equal)."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    layers, steps, n = 2, int(os.environ.get("QUEST_SOAK_STEPS", "40")), len(lens)  # soak: QUEST_SOAK_STEPS=400
    # a capacity beyond 1024 pages: the two-launch form then gathers with 8-wave workgroups like the one-launch kernel (a
    # head's pages are dealt over the waves, so another wave count is another fp32 fold order)
    cap = 16 * 1040
    ks = [cuda(inputs(400 + i, L, Hq, Hkv, D)[1]) for i, L in enumerate(lens)]
    vs = [cuda(inputs(400 + i, L, Hq, Hkv, D)[2]) for i, L in enumerate(lens)]
    g = torch.Generator(device=dev).manual_seed(15)
    new_q = torch.randn(steps, layers, n, Hq, D, generator=g, device=dev, dtype=torch.float16)
    new_k = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    new_v = torch.randn(steps, layers, n, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    if odd_q:
        new_q[::2, :, :, 0, 5] = 0.0
        new_q[1::3, :, :, 1, 17] = float("inf")
        new_q[::5, :, 0, 2, 40] = float("-inf")

    class Run:
        def __init__(self, one_launch):
            self.one = one_launch
            b = self.b = qu.BatchedInferenceController(n, layers, Hq, D, PAGE, B, cap, torch.float16, dev, num_kv_heads=Hkv,
                                                       layout=layout, shuffle_seed=9)
            for i in range(n):
                _prefill(b.seqs[i], ks[i], vs[i], layers)
            if budgets is not None:
                b.set_page_budgets(budgets)
            b.enable_device_state()
            b._decode_handler.set_pages_per_chunk(B)  # one workgroup per head, whatever the batch size
            b.begin_graph_decode()
            assert b._decode_handler.plan_info() == (B, 1)
            self.q = torch.empty(layers, n, Hq, D, device=dev, dtype=torch.float16)
            self.k = torch.empty(layers, n, Hkv, D, device=dev, dtype=torch.float16)
            self.v = torch.empty(layers, n, Hkv, D, device=dev, dtype=torch.float16)
            self.scores = torch.zeros(layers, n, Hq, (b.max_pages + 7) // 8 * 8, device=dev, dtype=torch.float16)
            self.o = torch.empty(layers, n, Hq, D, device=dev, dtype=torch.float16)
            # the selection of the LAST layer of a step (one inspection buffer per handler)
            self.sel_v = torch.zeros(n, Hq, B - 1, dtype=torch.float16, device=dev)
            self.sel_i = torch.full((n, Hq, B - 1), -1, dtype=torch.int32, device=dev)
            b._decode_handler.set_selection_out(self.sel_v, self.sel_i)
            self.set_inputs(0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.step()  # warm-up; the first replay folds the same token again (idempotent)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            info = b._decode_handler.last_launch_info()
            assert info["front_end_variant"] == (7 if one_launch else info["front_end_variant"]) and info["workgroups_per_head"] == 1
            assert (info["front_end_variant"] == 7) == one_launch and info["waves"] == 8, info
            b.sync_device_state()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.step()
            b.sync_device_state()

        def set_inputs(self, t):
            self.q.copy_(new_q[t]); self.k.copy_(new_k[t]); self.v.copy_(new_v[t])

        def step(self):
            qu.step_advance_batched(self.b)
            for l in range(layers):
                qu.decode_layer_batched(self.q[l], self.k[l], self.v[l], self.b, l, self.scores[l], apply_rope=True,
                                        out=self.o[l], one_launch=self.one, write_scores=True)

    def same(a, b):  # bit patterns (outputs / scores of non-finite queries are NaN: NaN != NaN under torch.equal)
        return torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16))

    two, one = Run(False), Run(True)
    for t in range(steps):
        for r in (two, one):
            r.set_inputs(t)
            r.graph.replay()
            r.b.prepare_metadata(1)
        assert torch.equal(one.b.step_states, two.b.step_states), f"token {t}: step states"
        for i, c in enumerate(one.b.seqs):
            n_out = len(c.kv_cache.indicies) - 1
            assert same(one.scores[:, i, :, :n_out], two.scores[:, i, :, :n_out]), f"token {t} seq {i}: page scores"
            k_i = min((budgets[i] if budgets else B) - 1, n_out)
            assert torch.equal(one.sel_i[i, :, :k_i], two.sel_i[i, :, :k_i]), f"token {t} seq {i}: selected pages"
            assert same(one.sel_v[i, :, :k_i], two.sel_v[i, :, :k_i]), f"token {t} seq {i}: selected values"
        assert same(one.o, two.o), f"token {t}: outputs"
    for r in (two, one):
        r.b._decode_handler.set_selection_out(None, None)
    for l in range(layers):  # same shuffle seed -> same physical pages: the pools must agree byte for byte on every page in use
        for c1, c2 in zip(one.b.seqs, two.b.seqs):
            assert list(c1.kv_cache.indicies) == list(c2.kv_cache.indicies)
            L = c1.kv_cache.seqlen
            k1, v1 = _gather(one.b.kv_layer(l), c1.kv_cache.indicies, L, layout)
            k2, v2 = _gather(two.b.kv_layer(l), c2.kv_cache.indicies, L, layout)
            assert torch.equal(k1, k2) and torch.equal(v1, v2), f"layer {l}: KV pool"
            np_ = len(c1.kv_cache.indicies)
            a1, b1 = _gather(one.b.metadata_layer(l), c1.metadata_cache.indicies, np_, layout)
            a2, b2 = _gather(two.b.metadata_layer(l), c2.metadata_cache.indicies, np_, layout)
            assert torch.equal(a1, a2) and torch.equal(b1, b2), f"layer {l}: metadata pool"
