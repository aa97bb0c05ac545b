"""Pin the CPU oracle (oracle/) against fixtures produced by the REFERENCE's own pure-torch
oracles (tests/golden/make_golden.py).  CPU only.

Tolerances are the reference's own: rtol = atol = 5e-3 for fp16
(quest/tests/test_estimate.py:10-15, test_approx_attention.py:10-15).
"""
import numpy as np
import torch

import oracle
from oracle import synth, torch_ref

PAGE = 16
RTOL = ATOL = 5e-3


def _inputs(seed, L, H, D=128):
    return (synth.normal_f16(seed * 3, (1, H, D)), synth.normal_f16(seed * 3 + 1, (L, H, D)),
            synth.normal_f16(seed * 3 + 2, (L, H, D)))


def _close(a, b):
    torch.testing.assert_close(torch.from_numpy(np.asarray(a, dtype=np.float32)),
                               torch.from_numpy(np.asarray(b, dtype=np.float32)), rtol=RTOL, atol=ATOL)


def test_half_conversions_match_numpy():
    lib = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in list(range(0, 65536, 97)) + [0, 1, 0x3ff, 0x400, 0x7bff, 0x7c00, 0x8000, 0xfbff, 0xfc00]:
        got = lib.qo_h2f(int(h))
        exp = float(f[h])
        assert (np.isnan(got) and np.isnan(exp)) or got == exp, h
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * 100,
        rng.standard_normal(5000).astype(np.float32) * 1e-5,
        rng.standard_normal(5000).astype(np.float32) * 1e-7,
        np.array([65504, 65519.99, 65520, 1e9, -1e9, 0.0, -0.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                  6.1e-5, 6.0975e-5], dtype=np.float32),
    ])
    with np.errstate(over="ignore"):
        exp = xs.astype(np.float16).view(np.uint16)
    for x, e in zip(xs, exp):
        assert lib.qo_f2h(float(x)) == int(e), (x, e)


def test_estimate_matches_reference_oracle(golden):
    worst = 0.0
    for seed, L, H in golden["est_cases"]:
        q, k, v = _inputs(int(seed), int(L), int(H))
        for layout in (oracle.NHD, oracle.HND, oracle.NHD_ROT):
            kv, meta = synth.build_sequence(k, v, PAGE, layout=layout, perm_seed=int(seed))
            got = oracle.estimate(q, meta)
            ref = golden[f"est_{L}"]
            assert got.shape == ref.shape
            _close(got, ref)
            ulp = np.abs(got.view(np.int16).astype(np.int32) - ref.view(np.int16).astype(np.int32))
            worst = max(worst, float(ulp.max()))
    # same maths, different fp32 summation order: a few fp16 ulps at most
    assert worst <= 4


def test_torch_ref_scores_match_reference_oracle(golden):
    for seed, L, H in golden["est_cases"]:
        q, k, v = _inputs(int(seed), int(L), int(H))
        got = torch_ref.page_scores(torch.from_numpy(q), torch.from_numpy(k), PAGE).numpy()
        _close(got, golden[f"est_{L}"])


def test_sparse_attention_matches_reference_oracle(golden):
    for seed, L, H, B, has_idx in golden["approx_cases"]:
        seed, L, H, B = int(seed), int(L), int(H), int(B)
        q, k, v = _inputs(seed, L, H)
        kv, meta = synth.build_sequence(k, v, PAGE, perm_seed=seed)
        n_pages = len(kv.indices)
        ref_o = golden[f"approx_o_{L}_{B}"]
        if has_idx:
            logical = golden[f"approx_idx_{L}_{B}"]
            assert logical.shape == (H, B - 1)
            n_sel = B - 1
        else:  # n_pages <= budget: full attention through kv_indices_without_last (controller.py:106)
            logical = np.tile(np.arange(n_pages - 1, dtype=np.int32), (H, 1))
            n_sel = n_pages - 1
        phys = kv.indices[logical] if n_sel > 0 else np.zeros((H, 1), dtype=np.int32)
        o, _ = oracle.sparse_attn(q, kv, phys, n_sel, int(kv.indices[-1]), kv.last_page_len)
        _close(o, ref_o)


def test_torch_ref_sparse_matches_reference_oracle(golden):
    for seed, L, H, B, has_idx in golden["approx_cases"]:
        seed, L, H, B = int(seed), int(L), int(H), int(B)
        q, k, v = map(torch.from_numpy, _inputs(seed, L, H))
        o, idx = torch_ref.sparse_decode(q, k, v, PAGE, B)
        _close(o.numpy(), golden[f"approx_o_{L}_{B}"])
        if has_idx:
            ref = golden[f"approx_idx_{L}_{B}"]
            # same fp32 scores up to summation order: the selected SETS agree except at near-ties
            same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(idx.numpy(), ref))
            assert same >= 0.97 * ref.size


def test_dense_attention_matches_reference_oracle(golden):
    for seed, L, H in golden["dense_cases"]:
        seed, L, H = int(seed), int(L), int(H)
        q, k, v = _inputs(seed, L, H)
        kv, meta = synth.build_sequence(k, v, PAGE, perm_seed=seed)
        n_pages = len(kv.indices)
        phys = np.tile(kv.indices[:-1], (H, 1))
        o, _ = oracle.sparse_attn(q, kv, phys, n_pages - 1, int(kv.indices[-1]), kv.last_page_len)
        _close(o, golden[f"dense_o_{L}"])
        ot = torch_ref.dense_decode(*map(torch.from_numpy, (q, k, v))).numpy()
        _close(ot, golden[f"dense_o_{L}"])


def test_topk_selected_values_match_torch_topk(golden):
    """What the reference pins for top-k (test_topk.py:45,62-64) is the selected values; here the
    multiset must be bit-identical to torch.topk's, and the declared tie rule must hold."""
    for seed, n, kk, rows in golden["topk_cases"]:
        n, kk, rows = int(n), int(kk), int(rows)
        vals = golden[f"topk_in_{n}_{kk}"]
        ref = golden[f"topk_vals_{n}_{kk}"]
        in_idx = np.tile(np.arange(n, dtype=np.int32) * 3 + 1, (rows, 1))  # arbitrary ids
        ov, oi = oracle.topk(vals, in_idx, kk)
        for r in range(rows):
            got_sorted = np.sort(ov[r].astype(np.float32))[::-1]
            assert np.array_equal(got_sorted, ref[r].astype(np.float32))
            cols = (oi[r] - 1) // 3
            assert np.all(np.diff(cols) > 0), "ascending column order"
            assert np.array_equal(vals[r][cols].view(np.uint16), ov[r].view(np.uint16))
            # tie rule: among equals at the threshold the lowest columns win
            key = oracle.half_key(vals[r]).astype(np.int32)
            thr = key[cols].min()
            assert set(np.nonzero(key > thr)[0]) <= set(cols.tolist())
            eq_cols = np.nonzero(key == thr)[0]
            need = kk - int((key > thr).sum())
            assert np.array_equal(np.sort(cols[key[cols] == thr]), eq_cols[:need])


def test_topk_matches_stable_sort_statement():
    """T-tie rule == stable descending sort: heavy-tie inputs."""
    rng = np.random.default_rng(5)
    for n, kk in [(64, 7), (255, 63), (2047, 127), (8191, 255), (100, 100), (5, 1)]:
        vals = (rng.integers(-6, 6, size=(3, n)) * 0.5).astype(np.float16)  # many duplicates
        idx = np.tile(np.arange(n, dtype=np.int32), (3, 1))
        ov, oi = oracle.topk(vals, idx, kk)
        t = torch.from_numpy(vals.astype(np.float32))
        order = torch.sort(t, dim=-1, descending=True, stable=True).indices[:, :kk].numpy()
        for r in range(3):
            assert np.array_equal(np.sort(order[r]), oi[r])


def test_append_decode_equals_prefill_and_extrema():
    """Metadata = element-wise max (K slot) / min (V slot) per page (decode_page.cuh:443-446);
    appending token by token must give the same pools as one prefill (test_page.cu:96-114)."""
    H, D, L = 3, 128, 83
    k = synth.normal_f16(1, (L, H, D))
    v = synth.normal_f16(2, (L, H, D))
    for layout in (oracle.NHD, oracle.HND):
        kv_a, meta_a = synth.build_sequence(k, v, PAGE, layout=layout, perm_seed=3)
        # token-by-token: prefill the first 21, then decode-append the rest
        n0 = 21
        kv_b, meta_b = synth.build_sequence(k[:n0], v[:n0], PAGE, layout=layout, perm_seed=3)
        cap, mcap = kv_a.data.shape[0], meta_a.data.shape[0]
        kv_b = oracle.Paged(np.zeros_like(kv_a.data), kv_a.indices[: len(kv_b.indices)], kv_b.last_page_len, layout)
        meta_b = oracle.Paged(np.zeros_like(meta_a.data), meta_a.indices[: len(meta_b.indices)], meta_b.last_page_len, layout)
        oracle.append_prefill(kv_b, meta_b, k[:n0], v[:n0])
        for t in range(n0, L):
            n_pages = t // PAGE + 1
            kv_b = oracle.Paged(kv_b.data, kv_a.indices[:n_pages], t % PAGE + 1, layout)
            n_meta = (n_pages - 1) // PAGE + 1
            meta_b = oracle.Paged(meta_b.data, meta_a.indices[:n_meta], (n_pages - 1) % PAGE + 1, layout)
            oracle.append_decode(kv_b, meta_b, k[t:t + 1], v[t:t + 1])
        assert np.array_equal(kv_a.data.view(np.uint16), kv_b.data.view(np.uint16))
        assert np.array_equal(meta_a.data.view(np.uint16), meta_b.data.view(np.uint16))
        # extrema vs numpy
        n_pages = len(kv_a.indices)
        for p in range(n_pages):
            blk = k[p * PAGE:(p + 1) * PAGE].astype(np.float32)
            mp, slot = meta_a.indices[p // PAGE], p % PAGE
            if layout == oracle.NHD:
                mx, mn = meta_a.data[mp, 0, slot], meta_a.data[mp, 1, slot]
            else:
                mx, mn = meta_a.data[mp, 0, :, slot], meta_a.data[mp, 1, :, slot]
            assert np.array_equal(mx.astype(np.float32), blk.max(0))
            assert np.array_equal(mn.astype(np.float32), blk.min(0))


def test_rope_oracle_vs_hf_fixtures(rope_golden):
    """The C restatement against the oracle of the reference's own rope test (HF LlamaRotaryEmbedding +
    apply_rotary_pos_emb, fixtures from tests/golden/make_rope_golden.py), at the reference's tolerance
    (test_rope.py:9-14: 5e-3)."""
    for seed, past, n, H in rope_golden["rope_cases"]:
        seed, past, n, H = int(seed), int(past), int(n), int(H)
        q, k = synth.normal_f16(seed * 3, (n, H, 128)), synth.normal_f16(seed * 3 + 1, (n, H, 128))
        oracle.rope_in_place(q, past, 1.0, 1e4)
        oracle.rope_in_place(k, past, 1.0, 1e4)
        _close(q, rope_golden[f"rope_q_{past}_{n}"])
        _close(k, rope_golden[f"rope_k_{past}_{n}"])


def test_rope_and_rmsnorm_oracle_vs_torch():
    """The rope restatement also against the HF rotate-half formula written out in torch fp32 (tighter than the
    fp16-table fixtures above), and rmsnorm against numpy."""
    n, H, D, past = 9, 4, 128, 37
    x = synth.normal_f16(11, (n, H, D))
    got = x.copy()
    oracle.rope_in_place(got, past, 1.0, 1e4)
    xt = torch.from_numpy(x).float()
    inv = 1.0 / (1e4 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(past, past + n).float()[:, None] * inv[None]
    cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None]
    sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None]
    rot = torch.cat([-xt[..., D // 2:], xt[..., :D // 2]], -1)
    _close(got, (xt * cos + rot * sin).numpy())
    w = synth.normal_f16(12, (256,))
    xx = synth.normal_f16(13, (1, 5, 256))
    ref = xx.astype(np.float32)
    ref = ref / np.sqrt((ref ** 2).mean(-1, keepdims=True) + 1e-5) * w.astype(np.float32)
    _close(oracle.rms_norm(xx, w, 1e-5), ref)


def test_prefill_oracle_matches_reference_oracle():
    """torch_ref.prefill_attention (fp32 restatement) vs the outputs of the reference's own `_ref_self_attention`
    (test_prefill_attention.py:17-44; tests/golden/make_prefill_golden.py) on pairs of the reference's sweep, at the
    reference's tolerance; the fp64 form agrees with the fp32 one far inside it."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "prefill_ref_golden.npz"))
    for seed, qo, kv, H in g["cases"]:
        q, k, v = (torch.from_numpy(synth.normal_f16(int(seed) * 3 + i, (n, int(H), 128)))
                   for i, n in ((0, int(qo)), (1, int(kv)), (2, int(kv))))
        o = torch_ref.prefill_attention(q, k, v)
        _close(o.numpy(), g[f"o_{qo}_{kv}"])
        o64 = torch_ref.prefill_attention(q, k, v, dtype=torch.float64)
        assert float((o64 - o.double()).abs().max()) < 1e-5
    # non-causal = every key for every row; one row over the whole cache = dense decode
    q, k, v = (torch.from_numpy(synth.normal_f16(900 + i, (n, 2, 128))) for i, n in ((0, 5), (1, 70), (2, 70)))
    full = torch_ref.prefill_attention(q, k, v, causal=False)
    last = torch_ref.prefill_attention(q[-1:], k, v, causal=True)
    torch.testing.assert_close(full[-1:], last, rtol=1e-6, atol=1e-6)


def test_row_rotated_layout_is_a_permutation_of_nhd_and_gives_the_same_results():
    """Layout 2 (NHD_ROT, include/quest_hip.h: head h's K / max vector of entry e in head slot h ^ (e & rot), its V / min
    vector in that slot ^ flip) is this build's extension, not the reference's: it may only change WHERE a vector lives.
    For head counts with every (rot, flip) combination, page sizes 16 / 7 / 1, prefill + decode appends: the pools are the
    NHD pools permuted by the definition (quest_amd.utils.TensorLayout.to_logical restates it in torch), and estimate,
    top-k and attention give the NHD results bit for bit."""
    from quest_amd.utils.utils import TensorLayout

    for H, S, L in ((32, 16, 16 * 19 + 5), (8, 16, 16 * 20), (4, 16, 300), (6, 7, 200), (12, 16, 257), (16, 16, 16 * 17 + 1),
                    (2, 1, 40), (3, 16, 100), (64, 16, 16 * 16 + 3)):
        D = 64
        rot, flip = TensorLayout.rotation(H)
        low = H & -H
        assert rot == min(low, 4) - 1 and flip == ((min(low, 32) - 1) & ~3) and (rot | flip) < max(low, 1)
        q = synth.normal_f16(H * 7 + 1, (1, 2 * H, D))  # GQA 2: per-query-head scores against the kv head's metadata
        k, v = synth.normal_f16(H * 7 + 2, (L, H, D)), synth.normal_f16(H * 7 + 3, (L, H, D))
        res = {}
        for layout in (oracle.NHD, oracle.NHD_ROT):
            n0 = L - 9
            kv, meta = synth.build_sequence(k[:n0], v[:n0], S, layout=layout, perm_seed=5, slack_pages=4)
            # build_sequence sized the pools for n0 tokens + slack: grow token by token through the decode append
            full_kv, full_meta = synth.build_sequence(k, v, S, layout=layout, perm_seed=5, slack_pages=4)
            kv = oracle.Paged(np.zeros_like(full_kv.data), full_kv.indices, 1, layout)
            meta = oracle.Paged(np.zeros_like(full_meta.data), full_meta.indices, 1, layout)
            n_pages0, n_meta0 = (n0 + S - 1) // S, ((n0 + S - 1) // S + S - 1) // S
            kv = oracle.Paged(kv.data, full_kv.indices[:n_pages0], (n0 - 1) % S + 1, layout)
            meta = oracle.Paged(meta.data, full_meta.indices[:n_meta0], (n_pages0 - 1) % S + 1, layout)
            oracle.append_prefill(kv, meta, k[:n0], v[:n0])
            for t in range(n0, L):
                n_pages, n_meta = t // S + 1, (t // S) // S + 1
                kv = oracle.Paged(kv.data, full_kv.indices[:n_pages], t % S + 1, layout)
                meta = oracle.Paged(meta.data, full_meta.indices[:n_meta], (n_pages - 1) % S + 1, layout)
                oracle.append_decode(kv, meta, k[t:t + 1], v[t:t + 1])
            assert np.array_equal(kv.data.view(np.uint16), full_kv.data.view(np.uint16)), (H, S, "append paths disagree")
            assert np.array_equal(meta.data.view(np.uint16), full_meta.data.view(np.uint16))
            est = oracle.estimate(q, meta)
            n_sel = min(5, est.shape[1])
            vals, idx = oracle.topk(est, np.tile(kv.indices[:-1], (2 * H, 1)), n_sel)
            o, _ = oracle.sparse_attn(q, kv, idx, n_sel, int(kv.indices[-1]), kv.last_page_len)
            res[layout] = (kv, meta, est, vals, idx, o)
        a, b = res[oracle.NHD], res[oracle.NHD_ROT]
        for i in (2, 3, 4, 5):
            assert np.array_equal(np.asarray(a[i]).view(np.uint16 if a[i].dtype == np.float16 else a[i].dtype),
                                  np.asarray(b[i]).view(np.uint16 if b[i].dtype == np.float16 else b[i].dtype)), (H, S, i)
        for pa, pb in ((a[0], b[0]), (a[1], b[1])):  # pools: rotated == NHD permuted by the definition
            used = pa.indices
            back = TensorLayout.to_logical(torch.from_numpy(pb.data[used].view(np.int16)), 2).numpy()
            assert np.array_equal(back, pa.data[used].view(np.int16)), (H, S)
            if (rot and S > 1) or flip:  # (a one-entry page has nothing to rotate by)
                assert not np.array_equal(pb.data[used].view(np.int16), pa.data[used].view(np.int16))
