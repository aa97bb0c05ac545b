"""The long-row top-k front ends of the fused attention launch (score rows of 4097 .. 16384 pages: 16-32 keys per thread)
on CRAFTED score rows -- the fused launch takes any fp16 scores, not only the estimate's -- against the oracle's top-k
(declared tie rule) and its fp64 attention over the selected pages:

* third generation (csrc/topk_prefilter.cuh; forced -- measured slower than the second, not the default): candidates above
  a per-wave lower bound of the threshold are compacted, then selected with <= 4 keys per thread;
* its fallbacks into the second generation (csrc/topk_bitmap.cuh): a wave with more than 256 candidates (rows of many
  equal scores, high scores clustered in one wave's columns) and k > 512;
* the second generation forced, with and without its histogram pre-filter, and the column-range variant on top of it.

All of them must produce the oracle's page list (values + ids, ascending column order)."""
import numpy as np
import pytest
import torch

import oracle
from _harness import cuda, inputs, make_controller, oracle_pools

pytestmark = pytest.mark.gpu
PAGE = 16
U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)


def _row_patterns(rng, n, kind):
    if kind == "normal":
        return rng.standard_normal(n).astype(np.float16) * 4 + 60
    if kind == "ties":  # few distinct values: the threshold key is shared by hundreds of columns
        return (rng.integers(0, 12, n) * 0.5 + 50).astype(np.float16)
    if kind == "equal":  # every column a tie: the selection is the k lowest columns
        return np.full(n, 3.25, np.float16)
    if kind == "clustered":  # the best pages sit next to each other (one wave's columns): > 256 candidates in a wave
        x = rng.standard_normal(n).astype(np.float32)
        lo = int(rng.integers(0, n - 700))
        x[lo:lo + 600] += 8.0
        return x.astype(np.float16)
    if kind == "mixed":  # negatives, zeros of both signs, infinities
        x = rng.standard_normal(n).astype(np.float16)
        x[rng.integers(0, n, 40)] = np.float16(0.0)
        x[rng.integers(0, n, 40)] = np.float16(-0.0)
        x[rng.integers(0, n, 5)] = np.float16(np.inf)
        x[rng.integers(0, n, 5)] = np.float16(-np.inf)
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("n_pages,k,kind", [(8192, 255, "normal"), (8192, 255, "ties"), (8192, 255, "equal"), (8192, 255, "clustered"),
                                             (8192, 255, "mixed"), (4100, 63, "normal"), (4100, 1, "ties"), (6000, 512, "normal"),
                                             (6000, 513, "normal"), (9000, 300, "ties"), (12288, 255, "normal"),
                                             (12300, 127, "clustered"), (16384, 255, "normal"), (16384, 500, "mixed")])
def test_long_score_rows_select_the_oracles_pages(n_pages, k, kind):
    import quest_amd.utils as qu

    Hq, Hkv, D = 4, 2, 64
    L = n_pages * PAGE - 5
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n_pages + k)
    kc = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    vc = torch.randn(L, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    q = torch.randn(1, Hq, D, generator=g, device=dev, dtype=torch.float16)
    ctl = make_controller(L, Hq, Hkv, D, PAGE, k + 1, shuffle_seed=5)
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    qu.append_kv(kc, vc, ctl, 0)
    ctl.end_forward()
    ctl.prepare_metadata(0)
    ctl.begin_forward(1)  # plans the decode handler for budget k + 1; the cache already holds the "current" token
    assert ctl.inference_page_budget == k + 1
    n = n_pages - 1
    rng = np.random.default_rng(n_pages * 7 + k)
    scores_np = np.stack([_row_patterns(rng, n, kind) for _ in range(Hq)])
    stride = (n + 7) // 8 * 8
    scores = torch.zeros(Hq, stride, dtype=torch.float16, device=dev)
    scores[:, :n] = torch.from_numpy(scores_np).to(dev)
    scores = scores[:, :n]
    table = np.array(ctl.kv_cache.indicies, np.int32)
    ev, ei = oracle.topk(scores_np, np.tile(table[:-1], (Hq, 1)), k)
    kv_o, _ = oracle_pools(ctl, kc.cpu().numpy(), vc.cpu().numpy())
    eo, _ = oracle.sparse_attn(q.cpu().numpy(), kv_o, ei, k, int(table[-1]), kv_o.last_page_len)
    h = ctl._decode_handler
    for gen, want in ((0, 2), (2, 2), (3, 2), (4, 5), (6, 6 if k <= 512 else 2)):
        h.set_front_end(gen)
        val = torch.zeros(Hq, k, dtype=torch.float16, device=dev)
        idx = torch.full((Hq, k), -1, dtype=torch.int32, device=dev)
        o = torch.empty_like(q)
        assert h.forward_fused_topk(q, o, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last, scores, val, idx,
                                    ctl.kv_cache.last_page_len, ctl.kv_last_page_idx)
        info = h.last_launch_info()
        if gen == 4 and info["front_end_variant"] != 5:
            continue  # the plan's column ranges do not fit a wave (few heads -> few, long ranges): not launched as such
        assert info["front_end_variant"] == want and info["waves"] == 8 and info["specialised"], (gen, info)
        assert np.array_equal(idx.cpu().numpy(), ei), f"front end {gen}: page ids"
        assert np.array_equal(U16(val.cpu().numpy()), U16(ev)), f"front end {gen}: values"
        np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
    h.set_front_end(0)
    ctl.end_forward()
