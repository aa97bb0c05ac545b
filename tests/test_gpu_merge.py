"""Where the per-workgroup partial states of a head are merged must not change a bit of the output: by the last-arriving
workgroup inside the attention launch (default; arrival tickets in the handler's workspace) or by the separate merge
launch (`set_merge_mode(1)` / QUEST_MERGE=launch) -- flashinfer's VariableLengthMergeStates role,
kernels/include/decode/decode_attn.cuh:992-1001.  Repeated launches check that the tickets re-arm themselves."""
import numpy as np
import pytest
import torch

import oracle
from _harness import cuda, fill, inputs, make_controller, oracle_pools

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("Hq,Hkv,D,page,L,B,ppc,layout", [
    (8, 8, 128, 16, 16 * 40 + 5, 24, 2, 0),    # 12 chunks per head, 4-wave workgroups
    (8, 2, 128, 16, 16 * 70 + 16, 33, 1, 1),   # 33 chunks: more than the in-kernel merge takes -> merge launch either way
    (4, 4, 64, 16, 16 * 50 + 1, 32, 1, 0),     # 32 chunks (the in-kernel limit), D = 64
    (4, 2, 256, 16, 16 * 20 + 9, 9, 3, 0),     # D = 256
    (6, 3, 128, 7, 7 * 30 + 3, 11, 2, 0),      # run-time page size
])
def test_in_kernel_merge_equals_merge_launch(Hq, Hkv, D, page, L, B, ppc, layout):
    import quest_amd.utils as qu

    q, k, v = inputs(11, L, Hq, Hkv, D)
    outs = {}
    for mode in (0, 1):
        ctl = make_controller(L, Hq, Hkv, D, page, B, layout=layout, shuffle_seed=3)
        h = ctl._decode_handler
        h.set_pages_per_chunk(ppc)
        h.set_merge_mode(mode)
        fill(ctl, k, v)
        assert ctl.need_estimate()
        qd = cuda(q)
        est = qu.decode_estimate(qd, ctl, 0)
        qu.decode_topk(est, ctl)
        res = []
        for _ in range(5):  # every launch draws and re-arms the same tickets
            res.append(qu.decode_sparse_attn(qd, ctl, 0, ctl.topk_dindices_buffer))
        fused = qu.decode_topk_sparse_attn(qd, est, ctl, 0)
        _, chunks = h.plan_info()
        assert chunks > 1
        torch.cuda.synchronize()
        for r in res[1:]:
            assert np.array_equal(_bits(r), _bits(res[0]))
        assert np.array_equal(_bits(fused), _bits(res[0]))
        outs[mode] = res[0]
        if mode == 0:
            kv_o, _ = oracle_pools(ctl, k, v)
            idx = ctl.topk_dindices_buffer.cpu().numpy()
            e_o, _ = oracle.sparse_attn(q, kv_o, idx, B - 1, int(ctl.kv_cache.indicies[-1]), kv_o.last_page_len)
            err = np.abs(res[0].cpu().numpy().astype(np.float32) - e_o.astype(np.float32)).max()
            assert err < 2e-3, err
        ctl.end_forward()
    assert np.array_equal(_bits(outs[0]), _bits(outs[1]))


@pytest.mark.parametrize("Hq,Hkv,D,L", [(32, 32, 128, 4096), (32, 8, 128, 16 * 300 + 7), (8, 1, 64, 16 * 90 + 16)])
def test_in_kernel_merge_full_kv_group_shared_kernel(Hq, Hkv, D, L):
    """Full-KV decode (the reference's branch for contexts within the budget, QuestAttention.py:123-132; cfg 2 is the
    first case): the group-shared kernel's last workgroup of a KV head merges all query heads of the group."""
    import quest_amd.utils as qu

    q, k, v = inputs(5, L, Hq, Hkv, D)
    outs = {}
    for mode in (0, 1):
        ctl = make_controller(L, Hq, Hkv, D, 16, 1 << 20, shuffle_seed=9)
        ctl._decode_handler.set_merge_mode(mode)
        fill(ctl, k, v)
        assert not ctl.need_estimate()
        qd = cuda(q)
        res = [qu.decode_sparse_attn(qd, ctl, 0, ctl.kv_indices_without_last) for _ in range(4)]
        _, chunks = ctl._decode_handler.plan_info()
        torch.cuda.synchronize()
        for r in res[1:]:
            assert np.array_equal(_bits(r), _bits(res[0]))
        outs[mode] = res[0]
        ctl.end_forward()
    assert np.array_equal(_bits(outs[0]), _bits(outs[1]))
    # fp32 torch attention over the whole context
    G = Hq // Hkv
    kf = torch.from_numpy(k.astype(np.float32)).repeat_interleave(G, 1)
    vf = torch.from_numpy(v.astype(np.float32)).repeat_interleave(G, 1)
    qf = torch.from_numpy(q.astype(np.float32))[0]
    w = torch.softmax(torch.einsum("hd,lhd->hl", qf, kf) / D ** 0.5, -1)
    ref = torch.einsum("hl,lhd->hd", w, vf)
    assert (outs[0].cpu().float()[0] - ref).abs().max() < 2e-3


def test_in_kernel_merge_under_graph_replay_and_batch():
    """State-driven launches replayed from a hipGraph while the sequences grow (tickets live in the handler's
    workspace, which captured launches hold by value), single sequence and batched, both merge modes."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    Hq, Hkv, D, B, steps, layers = 8, 4, 128, 9, 24, 2
    lens = (16 * 21 + 3, 16 * 35 + 16, 16 * 12 + 9)
    g = torch.Generator(device=dev).manual_seed(1)
    nq = torch.randn(steps, layers, len(lens), Hq, D, generator=g, device=dev, dtype=torch.float16)
    nk = torch.randn(steps, layers, len(lens), Hkv, D, generator=g, device=dev, dtype=torch.float16)
    nv = torch.randn(steps, layers, len(lens), Hkv, D, generator=g, device=dev, dtype=torch.float16)
    results = {}
    for mode in (0, 1):
        b = qu.BatchedInferenceController(len(lens), layers, Hq, D, 16, B, max(lens) + steps + 40, torch.float16, dev,
                                          num_kv_heads=Hkv, shuffle_seed=2)
        for i, c in enumerate(b.seqs):
            _, k, v = inputs(60 + i, lens[i], Hq, Hkv, D)
            c.prepare_metadata(lens[i])
            c.begin_forward(lens[i])
            for l in range(layers):
                qu.append_kv(cuda(k), cuda(v), c, l)
            c.end_forward()
        b.enable_device_state()
        b._decode_handler.set_pages_per_chunk(2)
        b._decode_handler.set_merge_mode(mode)
        b.begin_graph_decode()
        scores = qu.score_scratch(b)
        q_in, k_in, v_in = nq[0].clone(), nk[0].clone(), nv[0].clone()
        o = torch.empty_like(q_in)

        def step():
            qu.step_advance_batched(b)
            for l in range(layers):
                qu.decode_layer_batched(q_in[l], k_in[l], v_in[l], b, l, scores, out=o[l])

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()  # warm-up; the device state is reset below
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        b.sync_device_state()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        b.sync_device_state()
        outs = []
        for t in range(steps):
            q_in.copy_(nq[t]), k_in.copy_(nk[t]), v_in.copy_(nv[t])
            graph.replay()
            b.prepare_metadata(1)
            outs.append(o.clone())
        torch.cuda.synchronize()
        _, chunks = b._decode_handler.plan_info()
        assert chunks > 1
        results[mode] = torch.stack(outs)
    assert np.array_equal(_bits(results[0]), _bits(results[1]))
    assert torch.isfinite(results[0].float()).all()
