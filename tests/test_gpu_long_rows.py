"""The long-row top-k front ends of the fused attention launch (score rows of 4097 .. 16384 pages) on CRAFTED score rows
-- the fused launch takes any fp16 scores, not only the estimate's -- against the oracle's top-k (declared tie rule) and
its fp64 attention over the selected pages:

* the second generation (csrc/topk_bitmap.cuh: 16-32 keys per thread, every workgroup of a head selects over the whole row),
  automatic and forced, with and without its histogram pre-filter;
* the tiles front end (csrc/decode_device.cuh sparse_decode_tiles_body: the rows carry per-8-page maxima; top-k tiles, then
  the exact top-k over their scores), incl. while the sequence grows, against the whole-row launches bit for bit.

All of them must produce the oracle's page list (values + ids, ascending column order).  (Round 4's third generation and
column-range variants, measured slower, were removed with their tests in round 5.)"""
import numpy as np
import pytest
import torch

import oracle
from _harness import make_controller, oracle_pools

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
    # rows padded to 16 bytes (what decode_append_estimate writes), then the reference's own layout: contiguous [Hq, n]
    # rows, 2-byte aligned (round 5: the second generation reads the aligned stream below each row)
    layouts = [("padded", scores)]
    if n % 4 and n + 3 <= 16384:  # (the row + its up to 3 skipped leading columns must fit the 16384-position bitmaps)
        ref_layout = scores.contiguous()
        assert any((ref_layout[hh].data_ptr() % 8) != 0 for hh in range(Hq))
        layouts.append(("reference", ref_layout))
    for (lname, scores), (gen, want) in [(lay, gw) for lay in layouts for gw in ((0, 2), (2, 2), (3, 2))]:
        h.set_front_end(gen)
        val = torch.zeros(Hq, k, dtype=torch.float16, device=dev)
        idx = torch.full((Hq, k), -1, dtype=torch.int32, device=dev)
        o = torch.empty_like(q)
        assert h.forward_fused_topk(q, o, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last, scores, val, idx,
                                    ctl.kv_cache.last_page_len, ctl.kv_last_page_idx)
        info = h.last_launch_info()
        assert info["front_end_variant"] == want and info["waves"] == 8 and info["specialised"], (gen, info)
        assert np.array_equal(idx.cpu().numpy(), ei), f"{lname} rows, front end {gen}: page ids"
        assert np.array_equal(U16(val.cpu().numpy()), U16(ev)), f"{lname} rows, front end {gen}: values"
        np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
    h.set_front_end(0)
    ctl.end_forward()


def _half_keys(bits_u16):
    """Order-preserving 16-bit keys of fp16 bit patterns (csrc/quest_common.cuh half_key)."""
    b = bits_u16.astype(np.uint32)
    return np.where(b & 0x8000, (~b) & 0xffff, b | 0x8000).astype(np.uint16)


@pytest.mark.parametrize("n_pages,k,kind", [(8192, 255, "normal"), (8192, 255, "ties"), (8192, 255, "equal"), (8192, 255, "clustered"),
                                             (8192, 255, "mixed"), (4100, 63, "normal"), (4100, 1, "ties"), (6000, 256, "normal"),
                                             (9000, 200, "ties"), (12300, 127, "clustered"), (16380, 255, "mixed"),
                                             (301, 255, "normal"), (2041, 255, "ties"), (1000, 129, "equal"), (40, 7, "normal")])
def test_tiles_front_end_selects_the_oracles_pages(n_pages, k, kind):
    """The tiles front end (csrc/decode_device.cuh sparse_decode_tiles_body: top-k tiles by their maxima, then the exact
    top-k over those tiles' scores) on crafted score rows with host-computed tile maxima: the oracle's page list (values
    + ids, ascending column order) for every pattern -- ties at the threshold inside and across tiles, all-equal rows
    (the k lowest columns), rows shorter than 8 k columns (every tile is a candidate), the last tile partly valid."""
    import quest_amd.utils as qu
    from quest_amd import _kernels

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
    ctl.enable_device_state()  # the cache already holds the "current" token: no advance
    ctl.begin_graph_decode()
    n, max_n = n_pages - 1, ctl.max_pages - 1
    rng = np.random.default_rng(n_pages * 7 + k)
    scores_np = np.stack([_row_patterns(rng, n, kind) for _ in range(Hq)])
    off, stride = _kernels.tile_max_offset(max_n), _kernels.tiles_row_stride(max_n)
    buf = np.zeros((Hq, stride), np.uint16)
    buf[:, :n] = U16(scores_np)
    n_tiles = (n + 7) // 8
    padded = np.zeros((Hq, n_tiles * 8), np.uint16)  # key 0 = below every real key: the invalid columns of the last tile
    padded[:, :n] = _half_keys(U16(scores_np))
    buf[:, off:off + n_tiles] = padded.reshape(Hq, n_tiles, 8).max(axis=2)
    buf[:, off + n_tiles:] = 0xffff  # stale maxima of tiles past the live row must be ignored
    scores = torch.from_numpy(buf.view(np.float16)).to(dev)
    table = np.array(ctl.kv_cache.indicies, np.int32)
    kk = min(k, n)
    ev, ei = oracle.topk(scores_np, np.tile(table[:-1], (Hq, 1)), kk)
    kv_o, _ = oracle_pools(ctl, kc.cpu().numpy(), vc.cpu().numpy())
    eo, _ = oracle.sparse_attn(q.cpu().numpy(), kv_o, ei, kk, int(table[-1]), kv_o.last_page_len)
    h = ctl._decode_handler
    val = torch.zeros(1, Hq, k, dtype=torch.float16, device=dev)
    idx = torch.full((1, Hq, k), -1, dtype=torch.int32, device=dev)
    h.set_selection_out(val, idx)
    o = torch.empty_like(q)
    assert h.forward_fused_topk_dyn(q, o, ctl.kv_cache.buf_layer(0), ctl.kv_table_full, scores, ctl.step_state, max_n, tiles=True)
    h.set_selection_out(None, None)
    info = h.last_launch_info()
    assert info["front_end_variant"] == 8 and info["waves"] == 8 and info["specialised"], info
    assert np.array_equal(idx[0, :, :kk].cpu().numpy(), ei), "page ids"
    assert np.array_equal(U16(val[0, :, :kk].cpu().numpy()), U16(ev)), "values"
    np.testing.assert_allclose(o.cpu().numpy().astype(np.float32), eo.astype(np.float32), rtol=2e-3, atol=2e-3)
    # the same row through the launch without tile maxima: identical slots -> identical bits
    o2 = torch.empty_like(q)
    h.forward_fused_topk_dyn(q, o2, ctl.kv_cache.buf_layer(0), ctl.kv_table_full, scores, ctl.step_state, max_n)
    info2 = h.last_launch_info()
    assert info2["front_end_variant"] != 8
    if info2["waves"] == 8:
        assert torch.equal(o, o2)


@pytest.mark.parametrize("Hq,Hkv,D,L0,B", [(8, 2, 128, 16 * 300 + 9, 20), (4, 4, 128, 16 * 130 + 16, 33), (8, 8, 64, 16 * 75 + 1, 9),
                                           # head_dim 256 with kv heads in multiples of 8: the estimate's 32-row tile
                                           # used to be 8 heads x 4 pages there and the tile maxima mixed two heads' scores
                                           (8, 8, 256, 16 * 90 + 5, 17), (16, 8, 256, 16 * 70 + 16, 12), (16, 16, 256, 16 * 40 + 3, 7)])
def test_tiles_launches_equal_the_whole_row_launches_while_the_sequence_grows(Hq, Hkv, D, L0, B):
    """decode_layer_dyn with the tiles launches forced (the estimate writes the tile maxima, the attention launch selects
    from them) against the same step through the whole-row launches, token after token across KV-page and metadata-page
    boundaries (the append is idempotent: both run on the same cache): page scores, selections and outputs identical."""
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    steps = 70
    g = torch.Generator(device=dev).manual_seed(L0 + B)
    kc = torch.randn(L0, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    vc = torch.randn(L0, Hkv, D, generator=g, device=dev, dtype=torch.float16)
    ctl = make_controller(L0, Hq, Hkv, D, PAGE, B, shuffle_seed=3, max_seq_len=L0 + steps + 40)
    ctl.prepare_metadata(L0)
    ctl.begin_forward(L0)
    qu.append_kv(kc, vc, ctl, 0)
    ctl.end_forward()
    ctl.enable_device_state()
    ctl.begin_graph_decode()
    h = ctl._decode_handler
    sel = [(torch.zeros(1, Hq, B - 1, dtype=torch.float16, device=dev), torch.full((1, Hq, B - 1), -1, dtype=torch.int32, device=dev))
           for _ in range(2)]
    sc = [qu.score_scratch(ctl).zero_() for _ in range(2)]
    for t in range(steps):
        q = torch.randn(1, Hq, D, generator=g, device=dev, dtype=torch.float16)
        k1 = torch.randn(1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
        v1 = torch.randn(1, Hkv, D, generator=g, device=dev, dtype=torch.float16)
        qu.step_advance_dyn(ctl)
        outs = []
        for i, tiles in enumerate((True, False)):
            h.set_selection_out(*sel[i])
            outs.append(qu.decode_layer_dyn(q, k1, v1, ctl, 0, sc[i], tiles=tiles))
            assert (h.last_launch_info()["front_end_variant"] == 8) == tiles
        h.set_selection_out(None, None)
        ctl.prepare_metadata(1)
        n = len(ctl.kv_cache.indicies) - 1
        assert torch.equal(sc[0][:, :n], sc[1][:, :n]), f"token {t}: scores"
        assert torch.equal(sel[0][1], sel[1][1]) and torch.equal(sel[0][0], sel[1][0]), f"token {t}: selection"
        assert torch.equal(outs[0], outs[1]), f"token {t}: outputs"
