"""Shared helpers for the GPU parity tests: run the reference's canonical single-layer harness
(quest/tests/test_approx_attention.py:139-197: prefill-append kv_len-1 tokens, decode-append 1)
through quest_amd on cuda:0 and rebuild the same pools with the CPU oracle."""
import numpy as np
import torch

import oracle
from oracle import synth


def inputs(seed, L, Hq, Hkv=None, D=128):
    Hkv = Hq if Hkv is None else Hkv
    return (synth.normal_f16(seed * 3, (1, Hq, D)), synth.normal_f16(seed * 3 + 1, (L, Hkv, D)),
            synth.normal_f16(seed * 3 + 2, (L, Hkv, D)))


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def make_controller(L, Hq, Hkv, D, page_size, page_budget, layout=0, shuffle_seed=None, max_seq_len=None,
                    num_layers=1):
    import quest_amd.utils as qu

    return qu.InferenceController(num_layers, Hq, D, page_size, page_budget,
                                  max_seq_len or (L + 4 * page_size), torch.float16, torch.device("cuda:0"),
                                  num_kv_heads=Hkv, layout=layout, shuffle_seed=shuffle_seed)


def fill(ctl, k, v, layer=0, split=None):
    """Prefill-append all but the last token, then decode-append the last one (reference harness)."""
    import quest_amd.utils as qu

    L = k.shape[0]
    n0 = L - 1 if split is None else split
    kc, vc = cuda(k), cuda(v)
    ctl.prepare_metadata(n0)
    ctl.begin_forward(n0)
    qu.append_kv(kc[:n0], vc[:n0], ctl, layer)
    ctl.end_forward()
    for t in range(n0, L):
        ctl.prepare_metadata(1)
        ctl.begin_forward(1)
        qu.append_kv(kc[t:t + 1], vc[t:t + 1], ctl, layer)
        if t != L - 1:
            ctl.end_forward()
    # left inside begin_forward(1) of the last token, like the reference tests


def oracle_pools(ctl, k, v):
    """The same pools built by the oracle with the controller's physical page ids."""
    kvc, mc = ctl.kv_cache, ctl.metadata_cache
    shape = tuple(kvc.buf_layer(0).shape)
    mshape = tuple(mc.buf_layer(0).shape)
    kv = oracle.Paged(np.zeros(shape, np.float16), np.array(kvc.indicies, np.int32), kvc.last_page_len, ctl.layout)
    meta = oracle.Paged(np.zeros(mshape, np.float16), np.array(mc.indicies, np.int32), mc.last_page_len, ctl.layout)
    oracle.append_prefill(kv, meta, k, v)
    return kv, meta


def used_pages_equal(dev_buf, ora, indices):
    d = dev_buf.cpu().numpy().view(np.uint16)
    o = ora.data.view(np.uint16)
    idx = np.asarray(indices)
    return np.array_equal(d[idx], o[idx])


def gather_entries(buf: np.ndarray, indices, n_entries: int, layout: int):
    """(K-slot, V-slot) rows ``[n_entries, H, D]`` of a pool layer, in logical order."""
    S = buf.shape[3] if layout == 1 else buf.shape[2]
    idx = np.asarray(indices)
    pages = buf[idx]  # [n, 2, ...]
    if layout == 1:
        pages = pages.transpose(0, 1, 3, 2, 4)  # -> [n, 2, S, H, D]
    if layout == 2:  # row-rotated NHD: head h's K vector of entry e sits in slot h ^ (e & rot), V in that slot ^ flip
        import torch
        from quest_amd.utils.utils import TensorLayout

        pages = TensorLayout.to_logical(torch.from_numpy(np.ascontiguousarray(pages).view(np.int16)), 2).numpy().view(pages.dtype)
    n, _, _, H, D = pages.shape
    k = pages[:, 0].reshape(n * S, H, D)[:n_entries]
    v = pages[:, 1].reshape(n * S, H, D)[:n_entries]
    return k, v


def pools_match(ctl, kv_o, meta_o, L, layer=0):
    """Bit-compare every valid KV token and every valid metadata entry of the device pools with the oracle's."""
    n_pages = len(ctl.kv_cache.indicies)
    dk, dv = gather_entries(ctl.kv_cache.buf_layer(layer).cpu().numpy(), ctl.kv_cache.indicies, L, ctl.layout)
    ok_, ov_ = gather_entries(kv_o.data, kv_o.indices, L, ctl.layout)
    dmx, dmn = gather_entries(ctl.metadata_cache.buf_layer(layer).cpu().numpy(), ctl.metadata_cache.indicies,
                              n_pages, ctl.layout)
    omx, omn = gather_entries(meta_o.data, meta_o.indices, n_pages, ctl.layout)
    u = lambda a: np.ascontiguousarray(a).view(np.uint16)
    return (np.array_equal(u(dk), u(ok_)) and np.array_equal(u(dv), u(ov_)) and np.array_equal(u(dmx), u(omx))
            and np.array_equal(u(dmn), u(omn)))
