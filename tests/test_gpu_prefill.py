"""prefill_with_paged_kv_cache (batch_prefill.cu:27-117) on the MFMA flash kernel of csrc/prefill.hip, through the C ABI:

* against the fixtures of the reference's own `_ref_self_attention` (test_prefill_attention.py:17-44) at its tolerance;
* against the fp32 oracle restatement (oracle/torch_ref.py prefill_attention) at 2e-3 over the shapes that stress the
  kernel's structure: query blocks of 128 with ragged tails, 64-key tiles with ragged tails, the causal diagonal inside
  a tile, one query row, one key, chunked prefill with a long prefix, GQA groups, the HND pool, page sizes other than 16
  (the generic page walk), non-causal;
* at a size no CPU oracle finishes (32 heads x 4096 and a 2048-row chunk after a 6144-token prefix) through
  size-independent properties: a chunk of a prompt == the same rows of the whole prompt, and rows whose keys all hold
  the same V return that V.
"""
import os

import numpy as np
import pytest
import torch

from oracle import synth, torch_ref
from _harness import make_controller

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _randn(seed, *shape):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV, dtype=torch.float16)


def _cache(k, v, Hq, page_size=16, layout=0, seed=1):
    import quest_amd.utils as qu

    L, Hkv, D = k.shape
    ctl = make_controller(L, Hq, Hkv, D, page_size, 4, layout=layout, shuffle_seed=seed)
    ctl.prepare_metadata(L)
    ctl.begin_forward(L)
    qu.append_kv(k, v, ctl, 0)
    return ctl


def _prefill(q, ctl, causal=True):
    from quest_amd import _kernels

    return _kernels.prefill_with_paged_kv_cache(q, ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last,
                                                ctl.kv_cache.last_page_len, causal, ctl.layout, False, 1.0, 1e4)


def test_prefill_vs_reference_fixtures():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "prefill_ref_golden.npz"))
    for seed, qo, kv, H in g["cases"]:
        q, k, v = (torch.from_numpy(synth.normal_f16(int(seed) * 3 + i, (n, int(H), 128))).to(DEV)
                   for i, n in ((0, int(qo)), (1, int(kv)), (2, int(kv))))
        ctl = _cache(k, v, int(H), seed=int(seed))
        o = _prefill(q, ctl)
        ctl.end_forward()
        torch.testing.assert_close(o.cpu().float(), torch.from_numpy(g[f"o_{qo}_{kv}"]).float(), rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("qo,kv,Hq,Hkv,page,layout,causal", [
    (1, 1, 2, 2, 16, 0, True),          # one key
    (1, 777, 4, 4, 16, 0, True),        # one query row = dense decode
    (64, 64, 2, 2, 16, 0, True),        # exactly one tile
    (65, 65, 2, 2, 16, 1, True),        # one key into the second tile
    (128, 128, 2, 1, 16, 0, True),
    (129, 300, 4, 2, 16, 1, True),      # second query block holds one row
    (200, 1000, 8, 2, 16, 0, True),     # diagonal crosses tiles mid-wave
    (333, 333, 3, 3, 16, 0, True),      # head count not a multiple of 8
    (500, 2049, 8, 8, 16, 0, True),
    (300, 515, 4, 4, 16, 0, False),     # non-causal, ragged last tile
    (7, 1024, 2, 2, 16, 1, False),
    (150, 411, 4, 2, 7, 0, True),       # generic page walk
    (90, 200, 2, 2, 1, 1, True),
    (257, 640, 4, 1, 31, 0, True),
    (100, 260, 2, 2, 3, 0, False),
    (200, 1000, 8, 2, 16, 2, True),     # the row-rotated pool (QUEST_LAYOUT_NHD_ROT): GQA, MHA with 32 heads, generic page walk
    (300, 515, 32, 32, 16, 2, False),
    (150, 411, 4, 4, 7, 2, True),
    (129, 300, 16, 8, 16, 2, True),
])
def test_prefill_vs_oracle(qo, kv, Hq, Hkv, page, layout, causal):
    q, k, v = _randn(qo * 7 + kv, qo, Hq, 128), _randn(kv + 1, kv, Hkv, 128), _randn(kv + 2, kv, Hkv, 128)
    ctl = _cache(k, v, Hq, page, layout, seed=qo)
    o = _prefill(q, ctl, causal)
    ctl.end_forward()
    ref = torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu(), causal)
    torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("D", [64, 256])
@pytest.mark.parametrize("qo,kv,Hq,Hkv,page,layout,causal", [
    (1, 1, 2, 2, 16, 0, True),
    (65, 65, 2, 2, 16, 1, True),
    (200, 1000, 8, 2, 16, 0, True),
    (333, 333, 3, 3, 16, 0, True),
    (300, 515, 4, 4, 16, 1, False),
    (150, 411, 4, 2, 7, 0, True),
    (257, 640, 4, 1, 31, 1, True),
    (200, 700, 8, 4, 16, 2, True),
    (120, 333, 8, 8, 5, 2, True),
])
def test_prefill_other_head_dims_vs_oracle(D, qo, kv, Hq, Hkv, page, layout, causal):
    """head_dim 64 (two LDS buffers of 32 KiB, 32-row staging passes) and 256 (one wave per SIMD, ONE buffer and a second
    barrier per tile, 8-row staging passes): the reference's prefill dispatches 64 / 128 / 256 too (prefill.cuh:1073 SWITCH_HEAD_DIM_PREFILL)."""
    q, k, v = _randn(qo * 7 + kv + D, qo, Hq, D), _randn(kv + 1 + D, kv, Hkv, D), _randn(kv + 2 + D, kv, Hkv, D)
    ctl = _cache(k, v, Hq, page, layout, seed=qo)
    o = _prefill(q, ctl, causal)
    ctl.end_forward()
    ref = torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu(), causal)
    torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)


def test_prefill_fuzz_vs_oracle():
    """60 seeded random shapes: head_dim 64 / 128 / 256, query rows 1..700, cached tokens up to 2500, 1-9 kv heads x groups
    1/2/4, page sizes 1..33, both layouts, causal or not -- every combination of ragged query blocks, ragged key tiles and page walks."""
    rng = np.random.default_rng(int(os.environ.get("QUEST_FUZZ_SEED", "20250705")))  # soak: QUEST_FUZZ_CASES=600 QUEST_FUZZ_SEED=7
    for case in range(int(os.environ.get("QUEST_FUZZ_CASES", "60"))):
        kv = int(rng.integers(1, 2500))
        qo = int(rng.integers(1, min(kv, 700) + 1))
        Hkv, group = int(rng.integers(1, 10)), int(rng.choice([1, 1, 2, 4]))
        page = int(rng.choice([16, 16, 16, 1, 2, 5, 8, 24, 33]))
        layout, causal = int(rng.integers(0, 2)), bool(rng.integers(0, 4))
        if case % 3 == 2:
            layout = 2  # the row-rotated pool
        D = int(rng.choice([128, 128, 64, 256]))
        q = _randn(3 * case, qo, Hkv * group, D)
        k, v = _randn(3 * case + 1, kv, Hkv, D), _randn(3 * case + 2, kv, Hkv, D)
        ctl = _cache(k, v, Hkv * group, page, layout, seed=case)
        o = _prefill(q, ctl, causal)
        ctl.end_forward()
        ref = torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu(), causal)
        torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3,
                                   msg=lambda m: f"case {case}: D={D} qo={qo} kv={kv} Hkv={Hkv} group={group} page={page} "
                                                 f"layout={layout} causal={causal}: {m}")


def test_prefill_scale_of_scores_and_masked_keys_never_leak():
    """Large scores (|s| ~ 100: softmax close to one-hot) and garbage in the masked future: keys a row must not see hold
    NaN / inf K and huge V; the visible result is untouched (a masked score is replaced, never multiplied)."""
    qo, kv, H = 96, 400, 2
    q, k, v = _randn(1, qo, H, 128) * 4, _randn(2, kv, H, 128) * 4, _randn(3, kv, H, 128)
    ref = torch_ref.prefill_attention(q[:40].cpu(), k[:kv - 56].cpu(), v[:kv - 56].cpu())
    k2, v2 = k.clone(), v.clone()
    k2[kv - 56:] = float("nan")
    k2[kv - 30:] = float("inf")
    v2[kv - 56:] = 60000.0
    ctl = _cache(k2, v2, H)
    o = _prefill(q, ctl)  # rows 0..39 see keys <= kv - 96 + i <= kv - 57
    ctl.end_forward()
    torch.testing.assert_close(o[:40].cpu().float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("pattern", ["ramp", "jump", "late_one_hot"])
def test_prefill_row_maximum_that_keeps_growing(pattern):
    """The kernel defers the softmax's reference maximum (it follows a row's maximum only when that runs ahead by more
    than 8 log2 units) -- a branch that bounded random data almost never takes after the first tiles.  Crafted keys take
    it on purpose: `ramp` grows every row's maximum a little per tile (deferred: probabilities above 1 for a while, then a
    rescale), `jump` plants keys ten times larger late in the sequence (a rescale by 2^-100 and more in the middle of a
    row's life, for some rows of a wave only -- the causal mask hides the jump from the others), `late_one_hot` makes the
    very last visible key of each row dominate."""
    qo, kv, H = 300, 1500, 2
    q, k, v = _randn(21, qo, H, 128), _randn(22, kv, H, 128), _randn(23, kv, H, 128)
    if pattern == "ramp":
        k = (k.float() * torch.linspace(0.2, 3.0, kv, device=DEV).view(kv, 1, 1)).half()
        k = (k.float().abs() * q[-1:].float().sign()).half()  # aligned with the last query: scores grow along the keys
    elif pattern == "jump":
        k = (k.float() * 0.3).half()
        for j in (1210, 1290, 1377, 1499):
            k[j] = (q[j - (kv - qo)].float() * 3.0).half()  # key j is the diagonal of row j - shift: score ~ 3 |q|^2 / sqrt(D)
    else:
        k = (k.float() * 0.2).half()
        rows = torch.arange(qo, device=DEV)
        k[kv - qo + rows] = (q.float() * 2.0).half()
    ctl = _cache(k, v, H)
    o = _prefill(q, ctl)
    ctl.end_forward()
    ref = torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu())
    torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)


def test_prefill_first_tile_with_strongly_negative_scores():
    """the first key tile used to rescale from the initial reference 0 to the tile's maximum g with
    alpha = exp2((0 - g) c); a row whose largest first-tile score is below about -128 log2 units (head_dim 128: q.k < -1004)
    overflowed alpha to +inf and 0 x inf made the whole row NaN.  Every query here is anti-aligned with the first 64+ keys
    (q.k ~ -2000 ... -4000); causal row 0 sees key 0 only."""
    qo = kv = 200
    H = 2
    b = _randn(31, 1, H, 128).float() * 4
    q = (b + _randn(32, qo, H, 128).float() * 0.2).half()
    k = _randn(33, kv, H, 128)
    k[:100] = (-b * torch.linspace(1.0, 2.0, 100, device=DEV).view(100, 1, 1)).half()
    v = _randn(34, kv, H, 128)
    for causal in (True, False):
        ctl = _cache(k, v, H)
        o = _prefill(q, ctl, causal)
        ctl.end_forward()
        assert torch.isfinite(o.float()).all(), f"causal={causal}: non-finite output rows"
        ref = torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu(), causal)
        torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)
    # a chunk whose rows all see ONLY strongly negative keys (the whole sequence is anti-aligned)
    k2 = (-b * torch.linspace(1.0, 1.5, kv, device=DEV).view(kv, 1, 1)).half()
    ctl = _cache(k2, v, H)
    o = _prefill(q[-70:], ctl)
    ctl.end_forward()
    ref = torch_ref.prefill_attention(q[-70:].cpu(), k2.cpu(), v.cpu())
    torch.testing.assert_close(o.cpu().float(), ref, rtol=2e-3, atol=2e-3)


def test_prefill_full_size_properties():
    """Llama-2-7B head shapes at 4096 and a 2048-row chunk after a 6144-token prefix: (i) chunked == whole on the same
    rows (different tile counts per row block, different dispatch order); (ii) spot rows against the fp32 oracle;
    (iii) with V constant per head, every row returns that constant whatever the softmax did (rows sum to one)."""
    import quest_amd.utils as qu

    H, D, L, split = 32, 128, 8192, 6144
    q, k, v = _randn(11, L, H, D), _randn(12, L, H, D), _randn(13, L, H, D)
    ctl = _cache(k, v, H)
    whole = _prefill(q, ctl)
    chunk = _prefill(q[split:], ctl)
    ctl.end_forward()
    torch.testing.assert_close(chunk.float(), whole[split:].float(), rtol=1e-3, atol=1e-3)
    rows = torch.tensor([0, 1, 63, 64, 127, 128, 4095, 4096, 6143, 6144, 8191], device=DEV)
    s = torch.einsum("qhd,khd->hqk", q[rows].float(), k.float()) / D ** 0.5
    s = s.masked_fill(torch.arange(L, device=DEV)[None, None, :] > rows[None, :, None], float("-inf"))
    ref = torch.einsum("hqk,khd->qhd", s.softmax(-1), v.float())
    torch.testing.assert_close(whole[rows].float(), ref, rtol=2e-3, atol=2e-3)

    vc = torch.arange(H, device=DEV, dtype=torch.float16).view(1, H, 1).expand(4096, H, D).contiguous() / 8
    ctl = _cache(k[:4096], vc, H, seed=5)
    o = _prefill(q[:4096], ctl)
    ctl.end_forward()
    torch.testing.assert_close(o.float(), vc.float(), rtol=1e-3, atol=1e-3)


def test_prefill_argument_errors():
    from quest_amd import _kernels

    q, k, v = _randn(1, 40, 2, 128), _randn(2, 30, 2, 128), _randn(3, 30, 2, 128)
    ctl = _cache(k, v, 2)
    with pytest.raises(ValueError):  # more causal query rows than cached tokens (test_prefill_attention.py:50-51)
        _prefill(q, ctl)
    assert _prefill(q[:0], ctl).shape == (0, 2, 128)  # no query rows: an empty result, nothing launched
    o = _prefill(q, ctl, causal=False)  # without the mask any number of rows may look at the cache
    torch.testing.assert_close(o.cpu().float(), torch_ref.prefill_attention(q.cpu(), k.cpu(), v.cpu(), causal=False),
                               rtol=2e-3, atol=2e-3)
    with pytest.raises(RuntimeError):  # head_dim mismatch (batch_prefill.cu:58)
        _kernels.prefill_with_paged_kv_cache(q[:10, :, :64].contiguous(), ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last,
                                             ctl.kv_cache.last_page_len, True, ctl.layout, False, 1.0, 1e4)
    with pytest.raises(RuntimeError):  # fp32 query (DISPATCH_PYTORCH_DTYPE_TO_CTYPE has only Half)
        _kernels.prefill_with_paged_kv_cache(q[:10].float(), ctl.kv_cache.buf_layer(0), ctl.kv_indices_with_last,
                                             ctl.kv_cache.last_page_len, True, ctl.layout, False, 1.0, 1e4)
    ctl.end_forward()
