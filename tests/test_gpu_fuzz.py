"""Seeded random sweep of shapes through the whole chain -- unfused ops, fused launches and the state-driven
(graph-replayable) launches -- against the CPU oracle: pools / estimates / selections bit-exact, attention
within the attention tolerance.  Complements the hand-picked cases of test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

import oracle
from _harness import cuda, fill, inputs, make_controller, oracle_pools, pools_match

pytestmark = pytest.mark.gpu
U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        D = int(rng.choice([64, 128, 128, 128, 256]))
        Hkv = int(rng.choice([1, 2, 3, 4, 8]))
        G = int(rng.choice([1, 1, 2, 4, 8]))
        if Hkv * G > 16:
            continue
        page = int(rng.choice([16, 16, 16, 4, 8, 32, 5, 1, 7]))  # incl. the reference sweep's 1 and 7
        n_pages = int(rng.integers(2, 60))
        L = (n_pages - 1) * page + int(rng.integers(1, page + 1))
        B = int(rng.integers(2, n_pages + 3))  # sometimes >= pages: the full-attention branch
        layout = int(rng.integers(0, 2))  # (the draw of earlier rounds: the other parameters keep their sequence)
        if len(out) % 3 == 2:
            layout = 2  # every third case on the row-rotated pool (QUEST_LAYOUT_NHD_ROT)
        out.append((Hkv * G, Hkv, D, page, L, B, layout, int(rng.integers(0, 1 << 20))))
    return out


# soak: QUEST_FUZZ_CASES=600 QUEST_FUZZ_SEED=7 pytest tests/test_gpu_fuzz.py
@pytest.mark.parametrize("case", _cases(int(os.environ.get("QUEST_FUZZ_CASES", "48")), int(os.environ.get("QUEST_FUZZ_SEED", "2025"))), ids=lambda c: "Hq%d_Hkv%d_D%d_S%d_L%d_B%d_lay%d" % c[:7])
def test_random_shape_chain_matches_oracle(case):
    import quest_amd.utils as qu

    Hq, Hkv, D, page, L, B, layout, seed = case
    q, k, v = inputs(seed % 9973, L, Hq, Hkv, D)
    ctl = make_controller(L, Hq, Hkv, D, page, B, layout=layout, shuffle_seed=seed)
    fill(ctl, k, v, split=max(1, L - 1 - seed % 3))  # last 1-3 tokens through the decode append
    kv_o, meta_o = oracle_pools(ctl, k, v)
    assert pools_match(ctl, kv_o, meta_o, L)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    qd = cuda(q)
    if not ctl.need_estimate():  # budget covers the cache: full attention over every page
        o = qu.decode_sparse_attn(qd, ctl, 0, ctl.kv_indices_without_last)
        ctl.end_forward()
        idx = np.tile(table[:-1], (Hq, 1)) if len(table) > 1 else np.zeros((Hq, 1), np.int32)
        eo, _ = oracle.sparse_attn(q, kv_o, idx, len(table) - 1, int(table[-1]), kv_o.last_page_len)
        np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
        return
    # reference op sequence
    est = qu.decode_estimate(qd, ctl, 0)
    e_est = oracle.estimate(q, meta_o)
    assert np.array_equal(U16(est.cpu().numpy()), U16(e_est))
    qu.decode_topk(est, ctl)
    budget = ctl.inference_page_budget
    ev, ei = oracle.topk(e_est, np.tile(table[:-1], (Hq, 1)), budget - 1)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    assert np.array_equal(U16(ctl.topk_dout_buffer.cpu().numpy()), U16(ev))
    o = qu.decode_sparse_attn(qd, ctl, 0, ctl.topk_dindices_buffer)
    eo, _ = oracle.sparse_attn(q, kv_o, ei, budget - 1, int(table[-1]), kv_o.last_page_len)
    np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
    # fused top-k + attention: same selection, same output bits
    ctl.topk_dindices_buffer.zero_()
    o2 = qu.decode_topk_sparse_attn(qd, est, ctl, 0, write_topk=True)
    assert np.array_equal(ctl.topk_dindices_buffer.cpu().numpy(), ei)
    assert torch.equal(o, o2)
    eager_waves = ctl._decode_handler.last_launch_info()["waves"]
    ctl.end_forward()
    # state-driven launches on the same cache: rewind one token on the device state and decode it again
    # (append of the same key/value is idempotent on the pools)
    ctl.enable_device_state()
    st = ctl.step_state.cpu().tolist()
    # undo the last token: the state must describe the cache BEFORE it
    if st[2] > 1:
        st[0] -= 1; st[2] -= 1
        ctl.step_state.copy_(torch.tensor(st, dtype=torch.int32))
        ctl.begin_graph_decode()
        n_out = len(table) - 1
        # both top-k front ends of the fused launch: first generation (csrc/topk_select.cuh; any row stride) and
        # second (csrc/topk_bitmap.cuh; 8-byte aligned rows), then the default choice
        for gen, scores in ((2, qu.score_scratch(ctl).zero_()),
                            (3, qu.score_scratch(ctl).zero_()),  # second generation with its histogram pre-filter
                            (1, torch.zeros(Hq, ctl.max_pages | 1, dtype=torch.float16, device="cuda:0")),
                            ("tiles", qu.score_scratch(ctl).zero_()),  # tile maxima from the estimate (sparse_decode_tiles_body)
                            (0, qu.score_scratch(ctl).zero_())):
            tiles = gen == "tiles" and page == 16 and budget - 1 <= 256
            ctl._decode_handler.set_front_end(gen if isinstance(gen, int) else 0)
            ctl.step_state.copy_(torch.tensor(st, dtype=torch.int32))
            sel_i = torch.full((1, Hq, budget - 1), -1, dtype=torch.int32, device="cuda:0")
            sel_v = torch.zeros(1, Hq, budget - 1, dtype=torch.float16, device="cuda:0")
            ctl._decode_handler.set_selection_out(sel_v, sel_i)
            qu.step_advance_dyn(ctl)
            o3 = qu.decode_layer_dyn(qd, cuda(k[L - 1:L]), cuda(v[L - 1:L]), ctl, 0, scores, tiles=tiles)
            ctl._decode_handler.set_selection_out(None, None)
            assert np.array_equal(sel_i[0].cpu().numpy(), ei) and np.array_equal(U16(sel_v[0].cpu().numpy()), U16(ev))
            assert np.array_equal(U16(scores[:, :n_out].cpu().numpy()), U16(e_est))
            assert pools_match(ctl, kv_o, meta_o, L)
            np.testing.assert_allclose(o3.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
            info = ctl._decode_handler.last_launch_info()
            assert info["front_end_variant"] == 8 if tiles else info["front_end_variant"] in (0, 1, 2, 3), info
            if info["waves"] == eager_waves:  # same pages in the same order, dealt over the same waves: same bits as the
                assert torch.equal(o3, o)      # eager fused launch (the tiles launch always runs 8-wave workgroups)
