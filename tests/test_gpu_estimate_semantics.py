"""estimate_attn_score on CALLER-SUPPLIED metadata (not built by append): the reference computes
``sum_d max(q*Kslot, q*Vslot)`` with no assumption that the K slot >= the V slot
(kernels/include/decode/decode_attn.cuh:152-156) and its own gtest fills both slots with N(0,1)
(kernels/src/test/test_max_possible.cu:50-51, sweep :158-170).  Through the drop-in operator the tensor
is the caller's, so the HIP kernel must reproduce that formula -- bit for bit against the oracle -- on
unordered slots, +-inf / NaN entries, the +-65504 sentinels of a freshly opened page
(decode_page.cuh:424-432) and query vectors containing zeros / infinities.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import synth

pytestmark = pytest.mark.gpu

U16 = lambda a: np.ascontiguousarray(a).view(np.uint16)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _run(q, meta: oracle.Paged):
    """quest_amd._kernels.estimate_attn_score through the C ABI on the oracle's pool bytes."""
    from quest_amd import _kernels

    n_meta = len(meta.indices)
    n_out = (n_meta - 1) * meta.page_size + meta.last_page_len - 1
    Hq = q.shape[1]
    o = torch.full((Hq, n_out), 7.0, dtype=torch.float16, device="cuda:0")
    _kernels.estimate_attn_score(_dev(q), o, _dev(meta.data), _dev(meta.indices),
                                 torch.tensor([0, n_meta], dtype=torch.int32, device="cuda:0"),
                                 meta.last_page_len, int(meta.indices[-1]), meta.layout)
    torch.cuda.synchronize()
    return o.cpu().numpy()


def _raw_meta(seed, n_pages, S, H, D, layout, slack=2):
    """A metadata pool whose BOTH slots are N(0,1) (test_max_possible.cu:50-51), shuffled page table."""
    n_meta = (n_pages + S - 1) // S
    cap = n_meta + slack
    shape = (cap, 2, H, S, D) if layout == oracle.HND else (cap, 2, S, H, D)
    # numpy's generator (same image on the GPU box): the libm-free synth generator takes minutes at the sweep's
    # largest pools, and no golden file depends on these bytes
    data = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32).astype(np.float16)
    idx = synth.page_permutation(seed, cap)[:n_meta]
    return oracle.Paged(data, idx, (n_pages - 1) % S + 1, layout)


def _same(got, exp):
    """Bit-exact where the oracle is not NaN; NaN where it is (NaN payloads are not compared)."""
    nan = np.isnan(exp)
    assert np.array_equal(np.isnan(got), nan), "NaN pattern differs"
    assert np.array_equal(U16(got)[~nan], U16(exp)[~nan]), f"{np.count_nonzero(U16(got)[~nan] != U16(exp)[~nan])} scores differ"


@pytest.mark.parametrize("page_size", [1, 3, 7, 16, 32])
@pytest.mark.parametrize("seq_len", [65, 127, 213, 1110, 2000, 4099, 8192, 8222, 12345, 28837])
def test_unordered_slots_reference_sweep(seq_len, page_size):
    """The reference gtest's sweep (test_max_possible.cu:158-170): H=32, D in {64,128}, both slots N(0,1)."""
    n_pages = (seq_len + page_size - 1) // page_size
    if n_pages < 2:
        pytest.skip("one page: nothing to score")
    for D in (64, 128):
        meta = _raw_meta(seq_len * 37 + page_size + D, n_pages, page_size, 32, D, oracle.NHD)
        q = synth.normal_f16(seq_len + D, (1, 32, D))
        got, exp = _run(q, meta), oracle.estimate(q, meta)
        assert got.shape == exp.shape == (32, n_pages - 1)
        _same(got, exp)
        # and against the gtest's own fp32 formula at its tolerance (1e-3 on > 99 % of the elements)
        if D == 128 and page_size == 16 and seq_len <= 4099:
            kmax, kmin = _slots(meta, n_pages - 1)
            qf = q[0].astype(np.float32)
            ref = np.maximum(qf[None] * kmax, qf[None] * kmin).sum(-1).T  # [H, n]
            ok = np.isclose(got.astype(np.float32), ref, rtol=1e-3, atol=1e-3)
            assert ok.mean() > 0.99


def _slots(meta: oracle.Paged, n):
    """(K-slot, V-slot) rows [n, H, D] in logical entry order, fp32."""
    pages = meta.data[meta.indices]
    if meta.layout == oracle.HND:
        pages = pages.transpose(0, 1, 3, 2, 4)
    if meta.layout == oracle.NHD_ROT:  # the row-rotated pool: back to head order
        from quest_amd.utils.utils import TensorLayout

        pages = TensorLayout.to_logical(torch.from_numpy(np.ascontiguousarray(pages).view(np.int16)), 2).numpy().view(np.float16)
    m, _, S, H, D = pages.shape
    return (pages[:, 0].reshape(m * S, H, D)[:n].astype(np.float32),
            pages[:, 1].reshape(m * S, H, D)[:n].astype(np.float32))


@pytest.mark.parametrize("Hq,Hkv,D,S,layout", [(32, 8, 128, 16, 0), (8, 2, 64, 16, 1), (8, 8, 256, 16, 0), (16, 2, 128, 8, 1),
                                                (32, 4, 128, 16, 0), (6, 3, 128, 7, 0), (4, 4, 64, 1, 1),
                                                (32, 8, 128, 16, 2), (32, 32, 128, 16, 2), (8, 4, 64, 16, 2), (6, 6, 128, 7, 2), (16, 8, 256, 16, 2)])
def test_unordered_slots_gqa_layouts_dims(Hq, Hkv, D, S, layout):
    n_pages = 777
    meta = _raw_meta(11 + Hq + D + S, n_pages, S, Hkv, D, layout)
    q = synth.normal_f16(5 + Hq, (1, Hq, D))
    _same(_run(q, meta), oracle.estimate(q, meta))


@pytest.mark.parametrize("Hq,Hkv", [(8, 8), (16, 4)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_nonfinite_and_sentinel_metadata(Hq, Hkv, layout):
    """+-inf, NaN and the +-65504 open-page sentinels in either slot; the reference's max() keeps the finite
    product where one product is NaN or -inf."""
    D, S, n_pages = 128, 16, 403
    meta = _raw_meta(900 + Hq, n_pages, S, Hkv, D, layout)
    rng = np.random.default_rng(Hq + layout)
    flat = meta.data.reshape(-1)
    for val, frac in ((np.inf, 0.004), (-np.inf, 0.004), (np.nan, 0.002), (65504.0, 0.01), (-65504.0, 0.01), (0.0, 0.01),
                      (-0.0, 0.01)):
        flat[rng.integers(0, flat.size, int(flat.size * frac))] = np.float16(val)
    # a freshly opened page exactly as append leaves it: K slot (max) = -65504, V slot (min) = +65504
    k_slot = meta.data[meta.indices[3], 0]
    v_slot = meta.data[meta.indices[3], 1]
    k_slot[...] = np.float16(-65504.0)
    v_slot[...] = np.float16(65504.0)
    q = synth.normal_f16(77, (1, Hq, D))
    got, exp = _run(q, meta), oracle.estimate(q, meta)
    assert np.isinf(exp).any() and np.isfinite(exp).any()
    _same(got, exp)


@pytest.mark.parametrize("Hq,Hkv", [(8, 8), (32, 8)])
def test_zero_and_nonfinite_query_elements(Hq, Hkv):
    """Query vectors with +-0, +-inf and NaN elements take the kernel's literal max-of-products form:
    0 * inf = NaN must lose against the other (finite) product exactly as in the reference."""
    D, S, n_pages = 128, 16, 300
    meta = _raw_meta(31 + Hq, n_pages, S, Hkv, D, 0)
    rng = np.random.default_rng(Hq)
    flat = meta.data.reshape(-1)
    flat[rng.integers(0, flat.size, flat.size // 200)] = np.float16(np.inf)
    flat[rng.integers(0, flat.size, flat.size // 200)] = np.float16(-np.inf)
    q = synth.normal_f16(3, (1, Hq, D)).copy()
    q[0, 0, 5] = 0.0            # only head 0's tile is affected ...
    q[0, 1, 9] = -0.0
    q[0, Hq - 1, 100] = np.inf  # ... and the last head's
    q[0, Hq - 1, 17] = 0.0
    got, exp = _run(q, meta), oracle.estimate(q, meta)
    _same(got, exp)
    q[...] = 0.0  # all-zero query: every score is 0 unless both slots of a feature are non-finite (then NaN)
    _same(_run(q, meta), oracle.estimate(q, meta))


def test_ordered_metadata_unchanged_vs_append_built():
    """On metadata built by append (K slot >= V slot) the result is what it always was: bit-exact vs oracle."""
    from _harness import cuda, fill, inputs, make_controller, oracle_pools
    import quest_amd.utils as qu

    L, H = 2000, 8
    q, k, v = inputs(99, L, H)
    ctl = make_controller(L, H, H, 128, 16, 1024, shuffle_seed=3)
    fill(ctl, k, v)
    got = qu.decode_estimate(cuda(q), ctl, 0).cpu().numpy()
    ctl.end_forward()
    _, meta_o = oracle_pools(ctl, k, v)
    assert np.array_equal(U16(got), U16(oracle.estimate(q, meta_o)))


@pytest.mark.parametrize("Hq,Hkv", [(8, 8), (16, 4)])
def test_fp16_denormal_metadata_and_queries(Hq, Hkv):
    """fp16 subnormals (|x| < 6.1e-5) in the metadata and in q must not be flushed anywhere on the way (packed fp16
    max/min, fp16 -> fp32 operands of the mixed FMA): pages whose entries are all subnormal give scores around
    1e-4 that differ bit-wise if an input is flushed to zero."""
    D, S, n_pages = 128, 16, 200
    meta = _raw_meta(55 + Hq, n_pages, S, Hkv, D, 0)
    rng = np.random.default_rng(Hq)
    sub = (rng.integers(1, 1024, meta.data.shape).astype(np.uint16)            # subnormal magnitudes, random sign
           | (rng.integers(0, 2, meta.data.shape).astype(np.uint16) << 15)).view(np.float16)
    meta.data[meta.indices[::2]] = sub[meta.indices[::2]]                       # every other metadata page: all subnormal
    q = np.random.default_rng(1).standard_normal((1, Hq, D), dtype=np.float32).astype(np.float16)
    q[0, 1, ::3] = sub.reshape(-1)[: len(q[0, 1, ::3])]                         # subnormal (non-zero) query elements
    got, exp = _run(q, meta), oracle.estimate(q, meta)
    tiny = np.abs(exp.astype(np.float32)) < 1e-2
    assert tiny.any() and (exp[tiny] != 0).any(), "the case must produce small non-zero scores"
    _same(got, exp)
