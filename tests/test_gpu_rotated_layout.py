"""QUEST_LAYOUT_NHD_ROT (include/quest_hip.h; round 6) may only change WHERE a vector lives: for the same inputs and the same
page tables, every operator must give the bits it gives on the plain NHD pool -- page scores, selected pages, attention
outputs -- and the pools must be the NHD pools permuted by the definition (`TensorLayout.to_logical` restates it in torch).

The other GPU suites run the rotated pool against the ORACLE (their layout parameters include 2: append sweeps, estimate
semantics, fuzz, prefill, batched, graph replay, the full-size cfg 3 / cfg 5 tests); this file pins it against NHD itself,
path by path: the reference's operator sequence, the fused launches, the state-driven launches (whole-row and tiles front
ends), the dense layer with the append folded in, prefill attention, and a batched layer as two launches and as one.
"""
import numpy as np
import pytest
import torch

from _harness import cuda, fill, inputs, make_controller

pytestmark = pytest.mark.gpu
PAGE = 16
DEV = "cuda:0"
NHD, ROT = 0, 2


def _logical(ctl, cache, n_entries, layer=0):
    """The first n_entries valid (K slot, V slot) rows of a cache in head order, whatever its layout: [2, n_entries, H, D]
    (entries past the valid ones hold whatever the allocation held)."""
    from quest_amd.utils import TensorLayout

    pages = cache.buf_layer(layer)[torch.tensor(list(cache.indicies), device=DEV)]
    x = TensorLayout.to_logical(pages.view(torch.int16), ctl.layout)  # [n, 2, S, H, D]
    n, _, S, H, D = x.shape
    return x.transpose(0, 1).reshape(2, n * S, H, D)[:, :n_entries]


def _pools_equal(a, b):
    L, n_pages = a.kv_cache.seqlen, len(a.kv_cache.indicies)
    assert b.kv_cache.seqlen == L and list(a.kv_cache.indicies) == list(b.kv_cache.indicies)
    return (torch.equal(_logical(a, a.kv_cache, L), _logical(b, b.kv_cache, L))
            and torch.equal(_logical(a, a.metadata_cache, n_pages), _logical(b, b.metadata_cache, n_pages)))


@pytest.mark.parametrize("Hq,Hkv,D,page,L,B", [
    (32, 32, 128, 16, 16 * 70 + 5, 20),   # rot 3, flip 28: the shape the layout is for
    (32, 8, 128, 16, 16 * 50 + 16, 9),    # GQA: rot 3, flip 4
    (16, 16, 64, 16, 16 * 40 + 1, 12),    # head_dim 64 (8 rows per wave load), flip 12
    (8, 8, 256, 16, 16 * 30 + 9, 7),      # head_dim 256 (2 rows per wave load: the rotation crosses load rounds)
    (8, 2, 128, 16, 333, 6),              # rot 1, no flip
    (12, 12, 128, 7, 7 * 31 + 3, 8),      # generic page walk, rot 3
    (4, 4, 128, 3, 100, 9),               # page size 3
    (3, 3, 128, 16, 200, 5),              # odd head count: the layout degenerates to NHD
])
def test_reference_operator_sequence_and_fused_launches_give_the_nhd_bits(Hq, Hkv, D, page, L, B):
    import quest_amd.utils as qu

    q, k, v = inputs(40 + Hq + D + page, L, Hq, Hkv, D)
    qd = cuda(q)
    res = {}
    for layout in (NHD, ROT):
        ctl = make_controller(L, Hq, Hkv, D, page, B, layout=layout, shuffle_seed=L)
        fill(ctl, k, v, split=L - 3)  # prefill append + three decode appends
        est = qu.decode_estimate(qd, ctl, 0)
        qu.decode_topk(est, ctl)
        o = qu.decode_sparse_attn(qd, ctl, 0, ctl.topk_dindices_buffer)
        idx, val = ctl.topk_dindices_buffer.clone(), ctl.topk_dout_buffer.clone()
        o_full = qu.decode_sparse_attn(qd, ctl, 0, ctl.kv_indices_without_last)  # every page: the per-head-list kernel
        ctl.end_forward()
        # the fused pair on one more token
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        k1, v1 = cuda(k[5:6]), cuda(v[5:6])
        est2 = qu.decode_append_estimate(qd, k1, v1, ctl, 0)
        o2 = qu.decode_topk_sparse_attn(qd, est2, ctl, 0, write_topk=True)
        idx2 = ctl.topk_dindices_buffer.clone()
        ctl.end_forward()
        # prefill attention of the last 37 rows against the cache
        from quest_amd import _kernels

        qp = torch.randn(37, Hq, D, generator=torch.Generator(device=DEV).manual_seed(L), device=DEV, dtype=torch.float16)
        op = _kernels.prefill_with_paged_kv_cache(qp, ctl.kv_cache.buf_layer(0),
                                                  torch.tensor(list(ctl.kv_cache.indicies), dtype=torch.int32, device=DEV),
                                                  ctl.kv_cache.last_page_len, True, ctl.layout, False, 1.0, 1e4)
        res[layout] = (ctl, est, idx, val, o, o_full, est2, idx2, o2, op)
    a, b = res[NHD], res[ROT]
    assert list(a[0].kv_cache.indicies) == list(b[0].kv_cache.indicies)  # same seed -> same physical pages
    for i, name in enumerate(("ctl", "page scores", "selected pages", "selected values", "sparse attention", "full attention",
                              "fused append+estimate scores", "fused selection", "fused attention", "prefill attention")):
        if i:
            assert torch.equal(a[i], b[i]), name
    assert _pools_equal(a[0], b[0])
    from quest_amd.utils import TensorLayout

    rot, flip = TensorLayout.rotation(Hkv)
    if (rot and page > 1) or flip:  # the pools really are laid out differently
        pa = a[0].kv_cache.buf_layer(0)[torch.tensor(list(a[0].kv_cache.indicies), device=DEV)]
        pb = b[0].kv_cache.buf_layer(0)[torch.tensor(list(b[0].kv_cache.indicies), device=DEV)]
        assert not torch.equal(pa, pb)


@pytest.mark.parametrize("Hq,Hkv,D,L0,B,tiles", [(32, 32, 128, 16 * 40 + 10, 9, False), (32, 8, 128, 16 * 130 + 16, 33, True),
                                                 (8, 8, 64, 16 * 75 + 1, 9, True), (16, 8, 256, 16 * 33 + 5, 12, False)])
def test_state_driven_layer_and_dense_layer_give_the_nhd_bits_while_the_sequence_grows(Hq, Hkv, D, L0, B, tiles):
    """decode_layer_dyn (whole-row or tiles front end) and decode_layer_dense_dyn (append folded into the attention launch),
    30 tokens across KV-page boundaries, on both pools side by side."""
    import quest_amd.utils as qu

    steps = 30
    g = torch.Generator(device=DEV).manual_seed(L0 + B)
    kc = torch.randn(L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    vc = torch.randn(L0, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
    ctls = {}
    for layout in (NHD, ROT):
        ctl = make_controller(L0, Hq, Hkv, D, PAGE, B, layout=layout, shuffle_seed=3, max_seq_len=L0 + steps + 40, num_layers=2)
        ctl.prepare_metadata(L0)
        ctl.begin_forward(L0)
        for l in range(2):
            qu.append_kv(kc, vc, ctl, l)
        ctl.end_forward()
        ctl.enable_device_state()
        ctl.begin_graph_decode(dense_layers=True)
        ctls[layout] = (ctl, qu.score_scratch(ctl).zero_())
    for t in range(steps):
        q = torch.randn(1, Hq, D, generator=g, device=DEV, dtype=torch.float16)
        k1 = torch.randn(1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        v1 = torch.randn(1, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        outs = {}
        for layout, (ctl, sc) in ctls.items():
            qu.step_advance_dyn(ctl)
            o_dense = qu.decode_layer_dense_dyn(q, k1, v1, ctl, 0)          # layer 0: full KV, append folded in
            o = qu.decode_layer_dyn(q, k1, v1, ctl, 1, sc, tiles=tiles)      # layer 1: the sparse chain
            assert (ctl._decode_handler.last_launch_info()["front_end_variant"] == 8) == tiles
            ctl.prepare_metadata(1)
            outs[layout] = (o_dense, o, sc[:, :len(ctl.kv_cache.indicies) - 1].clone())
        for i, name in enumerate(("dense layer", "sparse layer", "page scores")):
            assert torch.equal(outs[NHD][i], outs[ROT][i]), f"token {t}: {name}"
    a, b = ctls[NHD][0], ctls[ROT][0]
    L, n_pages = a.kv_cache.seqlen, len(a.kv_cache.indicies)
    assert L == L0 + steps and b.kv_cache.seqlen == L
    for l in range(2):
        assert torch.equal(_logical(a, a.kv_cache, L, l), _logical(b, b.kv_cache, L, l))
        assert torch.equal(_logical(a, a.metadata_cache, n_pages, l), _logical(b, b.metadata_cache, n_pages, l))


@pytest.mark.parametrize("Hq,Hkv,D,B,lens", [(32, 32, 128, 9, (16 * 31 + 10, 16 * 15 + 16, 16 * 40 + 1, 19)),
                                             (32, 8, 128, 6, (16 * 20 + 5, 16 * 9 + 16, 16 * 31 + 1)),
                                             (8, 8, 64, 9, (16 * 33 + 9, 16 * 50 + 16))])
def test_batched_layer_as_two_launches_and_as_one_gives_the_nhd_bits(Hq, Hkv, D, B, lens):
    import quest_amd.utils as qu

    n, steps = len(lens), 20
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(sum(lens))
    ks = [torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16) for L in lens]
    vs = [torch.randn(L, Hkv, D, generator=g, device=DEV, dtype=torch.float16) for L in lens]
    bs = {}
    for layout in (NHD, ROT):
        for form in (1, 2):
            # (a capacity beyond 1024 pages: the two-launch form then gathers with 8-wave workgroups like the one-launch
            # kernel -- a head's pages are dealt over the waves, so another wave count is another fp32 fold order)
            b = qu.BatchedInferenceController(n, 1, Hq, D, PAGE, B, 16 * 1040, torch.float16, dev,
                                              num_kv_heads=Hkv, layout=layout, shuffle_seed=9)
            for c, k, v in zip(b.seqs, ks, vs):
                c.prepare_metadata(k.shape[0])
                c.begin_forward(k.shape[0])
                qu.append_kv(k, v, c, 0)
                c.end_forward()
            b.enable_device_state()
            b._decode_handler.set_pages_per_chunk(B)  # one workgroup per head whatever the batch size (the one-launch form's plan)
            b.begin_graph_decode()
            assert b._decode_handler.plan_info() == (B, 1)
            bs[layout, form] = (b, qu.score_scratch(b).zero_())
    for t in range(steps):
        q = torch.randn(n, Hq, D, generator=g, device=DEV, dtype=torch.float16)
        k1 = torch.randn(n, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        v1 = torch.randn(n, Hkv, D, generator=g, device=DEV, dtype=torch.float16)
        outs = {}
        for key, (b, sc) in bs.items():
            qu.step_advance_batched(b)
            outs[key] = qu.decode_layer_batched(q, k1, v1, b, 0, sc, one_launch=key[1] == 1, write_scores=True)
            assert b.last_layer_launches == key[1]
            b.prepare_metadata(1)
        for key in outs:
            assert torch.equal(outs[key], outs[NHD, 2]), f"token {t}: layout {key[0]}, {key[1]} launch(es)"
        for i, c in enumerate(bs[NHD, 2][0].seqs):
            n_out = len(c.kv_cache.indicies) - 1
            for key, (b, sc) in bs.items():
                assert torch.equal(sc[i, :, :n_out], bs[NHD, 2][1][i, :, :n_out]), f"token {t}: scores of sequence {i}, {key}"
    ref = bs[NHD, 2][0]
    for key, (b, _) in bs.items():
        for ca, cb in zip(ref.seqs, b.seqs):
            assert list(ca.kv_cache.indicies) == list(cb.kv_cache.indicies)
            assert _pools_equal(ca, cb), key


def test_argument_errors_of_the_rotated_layout():
    """Layout ids beyond NHD_ROT are malformed, at the Python surface and at the C ABI."""
    import quest_amd.utils as qu
    from quest_amd import _kernels

    with pytest.raises(KeyError):
        qu.TensorLayout.parse(3)
    with pytest.raises(KeyError):
        qu.TensorLayout.parse("NHD_ROTATED")
    assert qu.TensorLayout.parse("NHD_ROT") == 2
    with pytest.raises((RuntimeError, ValueError)):
        _kernels.BatchDecodeWithPagedKVCachePyTorchWrapper(3)
    q, k, v = inputs(5, 100, 4, 4, 128)
    ctl = make_controller(100, 4, 4, 128, PAGE, 4, layout=ROT)
    fill(ctl, k, v)
    with pytest.raises((RuntimeError, ValueError)):  # a pool view with layout 3
        _kernels.estimate_attn_score(cuda(q), torch.empty(4, 6, dtype=torch.float16, device=DEV), ctl.metadata_cache.buf_layer(0),
                                     ctl.metadata_indices, ctl.metadata_indptr_for_append, ctl.metadata_cache.last_page_len,
                                     ctl.metadata_last_page_idx, 3)
    ctl.end_forward()
