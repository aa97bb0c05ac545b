"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the
``_kernels`` shim validates arguments like the reference's CHECK_* macros, and the controller / KV-cache
bookkeeping follows quest/utils/controller.py and kv_cache.py."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from quest_amd import _lib

    header = open(os.path.join(ROOT, "include", "quest_hip.h")).read()
    declared = set(re.findall(r"\b(quest_[a-z0-9_]+)\s*\(", header))
    declared -= {"quest_stream_t"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert b"gfx950" in _lib.lib.quest_build_info()
    assert b"begin_forward" in _lib.lib.quest_error_string(-3)


def test_pool_slot_of_the_library_is_the_layout_definition_python_restates():
    """QUEST_LAYOUT_NHD_ROT is defined in ONE place per language: csrc/quest_common.cuh (exported as the host function
    quest_pool_slot), `TensorLayout.rotation` / `to_logical` in Python and qo_k_off / qo_v_off in the C oracle.  The
    library's function against the Python restatement for every head count up to 72 and every entry of a 16-entry page, and the
    properties the kernels rely on: a slot stays inside the row, K and V slots are permutations of the heads, the rotation
    touches slot bits 0-1 only and the V flip bits 2-4 only (so the K -> V distance is a per-head constant)."""
    import torch

    from quest_amd._lib import lib
    from quest_amd.utils.utils import TensorLayout

    for H in range(1, 73):
        rot, flip = TensorLayout.rotation(H)
        assert rot in (0, 1, 3) and flip in (0, 4, 12, 28) and (rot & flip) == 0
        for e in range(16):
            k_slots = [lib.quest_pool_slot(2, H, h, e, 0) for h in range(H)]
            v_slots = [lib.quest_pool_slot(2, H, h, e, 1) for h in range(H)]
            assert k_slots == [h ^ (e & rot) for h in range(H)] and v_slots == [s ^ flip for s in k_slots]
            assert sorted(k_slots) == list(range(H)) and sorted(v_slots) == list(range(H))
            assert all(((h ^ k) & ~3) == 0 and ((k ^ v) & ~28) == 0 for h, (k, v) in enumerate(zip(k_slots, v_slots)))
            for layout in (0, 1):
                assert [lib.quest_pool_slot(layout, H, h, e, 1) for h in range(H)] == list(range(H))
        # to_logical on a tagged pool page: element [kv, e, h] of the NHD view is what slot (kv, e, h) of the rotated page holds
        S = 16
        page = torch.zeros(1, 2, S, H, 1, dtype=torch.int32)
        for kv in range(2):
            for e in range(S):
                for h in range(H):
                    page[0, kv, e, lib.quest_pool_slot(2, H, h, e, kv), 0] = 10000 * kv + 100 * e + h
        want = torch.tensor([[[10000 * kv + 100 * e + h for h in range(H)] for e in range(S)] for kv in range(2)], dtype=torch.int32)
        assert torch.equal(TensorLayout.to_logical(page, 2)[0, ..., 0], want), H
    assert lib.quest_pool_slot(3, 8, 0, 0, 0) == 0xffffffff and lib.quest_pool_slot(2, 8, 8, 0, 0) == 0xffffffff


def test_c_abi_argument_errors_without_gpu():
    """Argument validation returns error codes before any launch (no GPU needed)."""
    import ctypes

    from quest_amd._lib import PagedKV, lib

    empty = PagedKV()
    assert lib.quest_append_kv_cache_decode(None, None, empty, empty, None) == -1
    assert lib.quest_estimate_attn_score(None, None, 32, 10, empty, None) == -1
    assert lib.quest_topk_filtering(None, None, None, None, None, 32, 10, 5, None) == -1
    h = ctypes.c_void_p()
    assert lib.quest_decode_handler_create(ctypes.byref(h), 7) == -1
    assert lib.quest_decode_handler_create(ctypes.byref(h), 0) == 0
    one = ctypes.c_void_p(16)
    kv = PagedKV(data=16, indices=16, indptr=16, num_heads=32, page_size=16, head_dim=128, page_budget=4,
                 last_page_len=1, last_page_idx=0, layout=0)
    assert lib.quest_decode_forward(h, one, one, kv, 32, None, None) == -3  # forward before begin_forward
    assert lib.quest_decode_begin_forward(h, 3, 32, 5, 128, 16, None) == -1   # heads not divisible
    assert lib.quest_decode_begin_forward(h, 3, 32, 32, 100, 16, None) == -2  # unsupported head_dim
    lib.quest_decode_handler_destroy(h)


def test_kernels_shim_rejects_cpu_tensors_like_check_cuda():
    from quest_amd import _kernels

    q = torch.zeros(1, 4, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _kernels.apply_rope_in_place(q, q, 0, 1.0, 1e4)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _kernels.topk_filtering(torch.zeros(4, 8, dtype=torch.float16), torch.zeros(4, 8, dtype=torch.int32),
                                torch.zeros(4, 2, dtype=torch.float16), torch.zeros(4, 2, dtype=torch.int32), None, 2)
    names = ["apply_rope_in_place", "rms_norm_forward", "topk_filtering", "estimate_attn_score",
             "append_kv_cache_prefill", "append_kv_cache_decode", "prefill_with_paged_kv_cache",
             "BatchDecodeWithPagedKVCachePyTorchWrapper"]  # bsk_ops.cu:4-20
    for n in names:
        assert hasattr(_kernels, n)
    w = _kernels.BatchDecodeWithPagedKVCachePyTorchWrapper(0)
    for m in ("begin_forward", "end_forward", "forward"):
        assert hasattr(w, m)
    with pytest.raises(RuntimeError, match="dispatch with dtype"):
        w.begin_forward(torch.tensor([0, 3], dtype=torch.int32), 32, 32, 128, 16, torch.empty(0, dtype=torch.bfloat16))


def test_utils_surface_matches_reference():
    import quest_amd.utils as qu

    ref_all = ["TensorLayout", "KvCache", "InferenceController", "BatchDecodeWithPagedKVCacheWrapper", "append_kv",
               "prefill_forward", "decode_estimate", "decode_topk", "decode_sparse_attn", "rms_norm_forward",
               "apply_rope_in_place"]  # quest/utils/__init__.py:11-23
    for n in ref_all:
        assert hasattr(qu, n) and n in qu.__all__
    assert qu.TensorLayout.NHD == 0 and qu.TensorLayout.HND == 1
    with pytest.raises(KeyError):
        qu.BatchDecodeWithPagedKVCacheWrapper("XYZ")


def test_kv_cache_bookkeeping():
    """append_seq / last_page_len / release follow kv_cache.py:104-132."""
    from quest_amd.utils import KvCache

    c = KvCache(2, 4, 128, 100, 16, torch.float16, "cpu")
    assert c.pool.capacity == 7 and c.pool.block_len == 16 and c.pool.num_layers == 2
    assert c.buf_layer(1).shape == (7, 2, 16, 4, 128)
    assert c.append_seq(0) == 0
    assert c.append_seq(17) == 2 and c.seqlen == 17 and c.last_page_len == 1 and len(c.indicies) == 2
    assert c.append_seq(15) == 0 and c.last_page_len == 16
    assert c.append_seq(1) == 1 and c.last_page_len == 1 and len(c.indicies) == 3
    assert c.pool.num_free_blocks == 4
    with pytest.raises(RuntimeError):
        c.append_seq(16 * 5)
    hnd = KvCache(1, 4, 128, 32, 16, torch.float16, "cpu", layout=1)
    assert hnd.buf_layer(0).shape == (2, 2, 4, 16, 128)
    c.release()
    assert c.seqlen == 0 and c.indicies == [] and c.pool.num_free_blocks == 7
    shuf = KvCache(1, 1, 64, 160, 16, torch.float16, "cpu", shuffle_seed=3)
    shuf.append_seq(160)
    assert sorted(shuf.indicies) == list(range(10)) and shuf.indicies != list(range(10))
    assert shuf.device_table().tolist() == shuf.indicies


def test_shared_pool_reservation_and_batched_controller_host_logic():
    """Sequences sharing one pool reserve disjoint page sets whose use order is known in advance (what the
    batched device-resident page tables rely on); release keeps the reservation and the order."""
    from quest_amd.utils import BatchedInferenceController, KvCache
    from quest_amd.utils.kv_cache import KvPool

    pool = KvPool(1, 2, 64, 12, 16, torch.float16, "cpu", shuffle_seed=5)
    a = KvCache(1, 2, 64, 64, 16, torch.float16, "cpu", pool=pool)   # 4 pages
    b = KvCache(1, 2, 64, 100, 16, torch.float16, "cpu", pool=pool)  # 7 pages
    assert pool.num_free_blocks == 1 and a.capacity_pages == 4 and b.capacity_pages == 7
    plan_a, plan_b = a.full_device_table().tolist(), b.full_device_table().tolist()
    assert len(plan_a) == 4 and len(plan_b) == 7 and not set(plan_a) & set(plan_b)
    b.append_seq(20)
    a.append_seq(40)
    b.append_seq(60)
    assert a.indicies == plan_a[:3] and b.indicies == plan_b[:5]
    assert a.full_device_table().tolist() == plan_a and b.full_device_table().tolist() == plan_b
    with pytest.raises(RuntimeError, match="KvPool exhausted"):
        a.append_seq(64)  # would need a 5th page: the free block of the pool is not this sequence's
    a.release()
    assert a.seqlen == 0 and pool.num_free_blocks == 1 and a.full_device_table().tolist() == plan_a
    a.append_seq(64)
    assert a.indicies == plan_a

    bc = BatchedInferenceController(3, 2, 4, 128, 16, 5, 200, torch.float16, "cpu", num_kv_heads=2, shuffle_seed=1)
    assert bc.kv_pool.capacity == 3 * 13 and bc.metadata_pool.capacity == 3 * 1
    assert bc.kv_layer(1).shape == (39, 2, 16, 2, 128)
    for i, c in enumerate(bc.seqs):
        c.prepare_metadata(100 + 16 * i)
    bc.enable_device_state()
    assert bc.kv_tables.shape == (3, 13) and bc.meta_tables.shape == (3, 1) and bc.max_pages == 13
    # rows padded to a multiple of 4 entries: every sequence's table starts 16-byte aligned (vector loads of page ids in
    # the fused attention launch); the views keep the true capacity as their width
    assert bc.kv_tables.stride(0) == 16 and bc.meta_tables.stride(0) == 4 and bc.kv_tables.stride(1) == 1
    assert all((bc.kv_tables.data_ptr() + 4 * 16 * i) % 16 == 0 for i in range(3))
    rows = bc.kv_tables.tolist()
    assert len({p for r in rows for p in r}) == 39  # disjoint reservations cover the pool
    st = bc.step_states.tolist()
    assert [s[0] for s in st] == [100, 116, 132] and [s[1] for s in st] == [7, 8, 9]
    assert [s[2] for s in st] == [4, 4, 4] and [s[3] for s in st] == [bc.seqs[i].kv_cache.indicies[-1] for i in range(3)]
    bc.prepare_metadata(1)
    assert [c.kv_cache.seqlen for c in bc.seqs] == [101, 117, 133]
    with pytest.raises(ValueError):
        BatchedInferenceController(0, 1, 4, 128, 16, 5, 200, torch.float16, "cpu")


def test_batched_controller_budgets_and_graph_plan(monkeypatch):
    """(1) per-sequence budgets live in ONE persistent device buffer (a captured graph holds its address), new
    budgets are copied in place, and budgets a captured plan cannot honour are refused instead of silently ignored;
    (2) an eager begin_forward() after begin_graph_decode() with the same budget still allocates the top-k buffers."""
    import quest_amd.utils.controller as ctl_mod

    class _Stub:
        def __init__(self, kv_layout="NHD"):
            self.planned = []

        def set_batch(self, n):
            pass

        def begin_forward(self, indptr, hq, hkv, d, s, dt):
            self.planned.append(indptr.tolist()[1])

    monkeypatch.setattr(ctl_mod, "BatchDecodeWithPagedKVCacheWrapper", _Stub)
    bc = ctl_mod.BatchedInferenceController(3, 1, 4, 128, 16, 5, 400, torch.float16, "cpu")
    for c in bc.seqs:
        c.prepare_metadata(300)
    bc.enable_device_state()
    bc.set_page_budgets([3, 5, 4])
    buf = bc.page_budgets
    assert buf.tolist() == [3, 5, 4] and bc.max_page_budget() == 5
    bc.begin_graph_decode()
    assert bc._decode_handler.planned == [4]
    bc.set_page_budgets([2, 2, 5])
    assert bc.page_budgets is buf and buf.tolist() == [2, 2, 5]  # same storage: a captured graph sees the new values
    with pytest.raises(RuntimeError, match="exceeds"):
        bc.set_page_budgets([2, 2, 6])  # beyond the captured plan
    bc.set_page_budgets(None)  # "no per-sequence budget" for a graph that reads the buffer = the sentinel, in place
    assert bc.page_budgets is buf and buf.tolist() == [bc.kNoBudget] * 3 and bc.max_page_budget() == 5
    # eager step after a graph plan with the same budget: the top-k buffers must exist (they used to stay None)
    bc.begin_forward()
    assert bc.topk_dout_buffer.shape == (3, 4, 4) and bc.topk_dindices_buffer.dtype == torch.int32
    assert bc._decode_handler.planned == [4]  # same budget: not re-planned

    # a plan captured with per-sequence budgets BELOW the constructor's budget cannot serve "the constructor's
    # budget for all" -- the kernels clamp to the captured plan: refused like a too-large list, until a re-plan
    bc3 = ctl_mod.BatchedInferenceController(3, 1, 4, 128, 16, 5, 400, torch.float16, "cpu")
    for c in bc3.seqs:
        c.prepare_metadata(300)
    bc3.enable_device_state()
    bc3.set_page_budgets([3, 3, 3])
    bc3.begin_graph_decode()
    assert bc3._decode_handler.planned == [2]
    with pytest.raises(RuntimeError, match="captured plan"):
        bc3.set_page_budgets(None)
    assert bc3.page_budgets.tolist() == [3, 3, 3]  # untouched: the captured graph keeps attending 3 pages, as reported
    assert bc3.max_page_budget() == 3

    bc2 = ctl_mod.BatchedInferenceController(2, 1, 4, 128, 16, 5, 400, torch.float16, "cpu")
    for c in bc2.seqs:
        c.prepare_metadata(300)
    bc2.enable_device_state()
    bc2.begin_graph_decode()  # planned WITHOUT per-sequence budgets: the captured launches pass no budget pointer
    with pytest.raises(RuntimeError, match="before begin_graph_decode"):
        bc2.set_page_budgets([3, 3])


def test_controller_budget_logic(monkeypatch):
    """need_estimate / inference_page_budget / index tensors (controller.py:80-142), with the handler stubbed
    (it needs the GPU only for its workspace)."""
    import quest_amd.utils.controller as ctl_mod

    class _Stub:
        def __init__(self, kv_layout="NHD"):
            self.calls = []

        def begin_forward(self, indptr, hq, hkv, d, s, dt):
            self.calls.append((indptr.tolist(), hq, hkv, d, s))

        def end_forward(self):
            self.calls.append("end")

    monkeypatch.setattr(ctl_mod, "BatchDecodeWithPagedKVCacheWrapper", _Stub)
    c = ctl_mod.InferenceController(2, 8, 128, 16, 4, 1000, torch.float16, "cpu", num_kv_heads=2)
    assert c.page_budget_pages == 4 and c.token_budget == 64
    assert c.kv_cache.buf_layer(0).shape[3] == 2  # pools hold kv heads
    assert not c.need_estimate()
    c.prepare_metadata(40)
    c.begin_forward(40)
    assert c.kv_indptr_for_append.tolist() == [0, 3] and c.metadata_indptr_for_append.tolist() == [0, 1]
    assert c.kv_indices_with_last.tolist() == c.kv_cache.indicies
    c.end_forward()
    c.prepare_metadata(1)
    c.begin_forward(1)
    assert c.inference_page_budget == 3 and not c.need_estimate()  # 3 pages <= budget 4
    assert c.kv_indices_without_last.shape == (8, 2)
    assert c.kv_indptr_for_approx_decode.tolist() == [0, 2]
    assert c._decode_handler.calls[-1] == ([0, 2], 8, 2, 128, 16)
    c.end_forward()
    c.prepare_metadata(60)
    c.begin_forward(60)
    c.end_forward()
    c.prepare_metadata(1)
    c.begin_forward(1)
    n_pages = len(c.kv_cache.indicies)
    assert n_pages == 7 and c.inference_page_budget == 4 and c.need_estimate()
    assert c.topk_dindices_buffer.shape == (8, 3) and c.topk_dout_buffer.dtype == torch.float16
    assert c.kv_indices_without_last.shape == (8, 6)
    assert c.kv_last_page_idx == c.kv_cache.indicies[-1]
    # layer-skip re-plan: a huge budget without touching index tensors (llama.py:428-439)
    c.end_forward()
    c.set_page_budget(1 << 20)
    c.begin_forward(1, updateTensor=False)
    assert c.inference_page_budget == 7 and not c.need_estimate()
    c.clean_states()
    assert c.kv_cache.seqlen == 0 and c.metadata_cache.seqlen == 0


def test_quest_attention_module_api():
    from types import SimpleNamespace

    from quest_amd.models import QuestAttention

    cfg = SimpleNamespace(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=4096,
                          rope_scaling={"type": "linear", "factor": 8.0})
    m = QuestAttention(cfg, layer_idx=3)
    assert m.head_dim == 128 and m.rope_scale == 8.0 and m.layer_idx == 3
    assert sorted(n for n, _ in m.named_parameters()) == ["k_proj.weight", "o_proj.weight", "q_proj.weight",
                                                           "v_proj.weight"]
    assert m.k_proj.weight.shape == (256, 512)
    with pytest.raises(ValueError):
        QuestAttention(SimpleNamespace(hidden_size=512, num_attention_heads=4, rope_scaling={"type": "yarn"}), 0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/quest"), reason="reference tree only exists in the build container")
def test_reference_python_stack_runs_on_our_kernels_module():
    """INTEGRATION.md route A: register quest_amd._kernels as `quest._kernels`; the reference's own
    quest.utils (controller, kv cache, wrappers) must import and drive our handler unchanged."""
    import importlib
    import sys

    import quest_amd._kernels as ours

    saved = {k: v for k, v in sys.modules.items() if k == "quest" or k.startswith("quest.")}
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    try:
        sys.modules["quest._kernels"] = ours
        qutils = importlib.import_module("quest.utils")
        assert qutils._kernels is ours
        ctl = qutils.InferenceController(2, 32, 128, 16, 8, 1024, torch.float16, torch.device("cpu"))
        ctl.prepare_metadata(300)
        ctl.begin_forward(300)
        ctl.end_forward()
        ctl.prepare_metadata(1)
        # builds the reference's index tensors and calls OUR begin_forward with the reference's arguments;
        # without a GPU the call gets as far as the workspace hipMalloc inside libquest_hip.so
        if torch.cuda.is_available():
            ctl.begin_forward(1)
            assert ctl._decode_handler._wrapper.plan_info()[0] >= 1
            ctl.end_forward()
        else:
            with pytest.raises(RuntimeError, match="BatchDecodeWithPagedKVCache failed with error code 100"):
                ctl.begin_forward(1)
        assert ctl.need_estimate() and ctl.inference_page_budget == 8
        # the op wrappers reach our validation (CPU tensors are rejected exactly like CHECK_CUDA)
        q = torch.zeros(1, 32, 128, dtype=torch.float16)
        with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
            qutils.decode_estimate(q, ctl, 0)
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k == "quest" or k.startswith("quest.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_bench_byte_accounting_matches_survey():
    """bench.py's algorithmic bytes per layer-step are SURVEY.md 8(d)'s (the reference benches' accounting)."""
    import argparse
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    MiB = 1 << 20
    a = argparse.Namespace(page_size=16, head_dim=128, heads=32, kv_heads=32, seqlen=32768, token_budget=2048)
    b = bench.bytes_per_layer(a)
    assert abs(b["estimate"] / MiB - 32.13) < 0.05      # 32 MiB metadata + q + table + 128 KiB scores
    assert abs(b["attn"] / MiB - 32.03) < 0.02          # 128 pages x 16 x 2 x 32 heads x 256 B (+ idx, q, o)
    assert abs(b["dense"] / MiB - 512.3) < 0.1
    assert abs(b["chain"] / MiB - 64.65) < 0.1          # the "~64.6 MiB" of BASELINE.md section 3
    g = argparse.Namespace(page_size=16, head_dim=128, heads=32, kv_heads=8, seqlen=131072, token_budget=4096)
    bg = bench.bytes_per_layer(g)
    assert abs(bg["estimate"] / MiB - 32.5) < 0.1 and abs(bg["attn"] / MiB - 64.05) < 0.05
    assert abs(bg["dense"] / MiB - 512) < 2             # GQA: every kv head read once


def test_hf_checkpoint_round_trip_and_hf_interop(tmp_path):
    """quest_amd.models.llama reads and writes the Hugging Face checkpoint layout (config.json + safetensors shards
    + index, tensor names of quest/models/llama.py's HF fork): our export loads back bit for bit, transformers
    reads it, and a checkpoint written BY transformers loads into our model tensor for tensor.  (CPU: IO only.)"""
    import torch

    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                      num_key_value_heads=1, rope_scaling={"type": "linear", "factor": 2.0}, rope_theta=5e5)
    torch.manual_seed(0)
    m = LlamaForCausalLM(cfg).half()
    d = str(tmp_path / "ours")
    m.save_pretrained(d, max_shard_bytes=600_000)  # forces several shards + model.safetensors.index.json
    assert os.path.exists(os.path.join(d, "model.safetensors.index.json"))
    m2 = LlamaForCausalLM.from_pretrained(d, device=torch.device("cpu"))
    assert m2.config == cfg
    for (ka, a), (kb, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    transformers = pytest.importorskip("transformers")
    hf = transformers.LlamaForCausalLM.from_pretrained(d)
    assert torch.equal(hf.state_dict()["model.layers.2.self_attn.k_proj.weight"].half(),
                       m.state_dict()["model.layers.2.self_attn.k_proj.weight"])
    # the other direction: a checkpoint exported by transformers itself (single file, rope_parameters nesting)
    hcfg = transformers.LlamaConfig(vocab_size=96, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                    num_attention_heads=1, num_key_value_heads=1, max_position_embeddings=512,
                                    tie_word_embeddings=True)
    torch.manual_seed(1)
    h2 = transformers.LlamaForCausalLM(hcfg).half()
    d2 = str(tmp_path / "theirs")
    h2.save_pretrained(d2, safe_serialization=True)
    m3 = LlamaForCausalLM.from_pretrained(d2, device=torch.device("cpu"))
    assert m3.config.num_hidden_layers == 2 and m3.config.rope_theta == 10000.0 and m3.config.rope_scaling is None
    assert m3.lm_head.weight.data_ptr() == m3.model.embed_tokens.weight.data_ptr()  # tied embeddings honoured
    hs = h2.state_dict()
    for k, v in m3.state_dict().items():
        assert torch.equal(v, hs[k]), k
    # a tensor the decoder does not have is an error, not silently dropped
    from safetensors.torch import load_file, save_file
    sd = load_file(os.path.join(d2, "model.safetensors"))
    sd["model.layers.0.self_attn.q_proj.bias"] = torch.zeros(128, dtype=torch.float16)
    save_file(sd, os.path.join(d2, "model.safetensors"), metadata={"format": "pt"})
    with pytest.raises(KeyError, match="unexpected tensor"):
        LlamaForCausalLM.from_pretrained(d2, device=torch.device("cpu"))


def test_round3_entry_points_validate_arguments_without_gpu():
    """The round-3 C-ABI entries (batched operators, strided fused launches, fused decoder-layer launches) return
    argument errors before any launch (no GPU needed)."""
    import ctypes

    from quest_amd._lib import Batch, PagedKV, lib

    empty = PagedKV()
    one = ctypes.c_void_p(16)
    b1 = Batch(1, 0, 0, 0, None)
    assert lib.quest_estimate_attn_score_batched(one, one, 32, 8, 16, empty, one, b1, None) == -1   # o_stride < max_n_out
    assert lib.quest_estimate_attn_score_batched(one, one, 32, 16, 8, empty, None, b1, None) == -1  # no state
    assert lib.quest_topk_filtering_batched(None, 16, 8, one, one, one, 4, 32, 5, one, b1, None) == -1
    assert lib.quest_topk_filtering_batched(one, 4, 8, one, one, one, 4, 32, 5, one, b1, None) == -1   # stride < max pages
    assert lib.quest_topk_filtering_batched(one, 20000, 20000, one, one, one, 4, 32, 5, one, b1, None) == -4  # > QUEST_TOPK_MAX_ROW
    b2 = Batch(2, 4, 0, 0, None)
    assert lib.quest_topk_filtering_batched(one, 16, 8, one, one, one, 4, 32, 5, one, b2, None) == -1  # table stride < pages
    assert lib.quest_decode_forward_batched(None, one, one, empty, 32, one, 4, one, b1, None, None) == -1
    assert lib.quest_append_estimate_strided(None, None, empty, one, one, 32, 10, 16, empty, None) == -1
    kv = PagedKV(data=16, indices=16, indptr=16, num_heads=32, page_size=16, head_dim=128, page_budget=0,
                 last_page_len=1, last_page_idx=0, layout=0)
    assert lib.quest_append_estimate_strided(one, one, kv, one, one, 32, 10, 4, kv, None) == -1     # o_stride < n_out
    h = ctypes.c_void_p()
    assert lib.quest_decode_handler_create(ctypes.byref(h), 0) == 0
    assert lib.quest_decode_forward_fused_topk_strided(h, one, one, kv, 32, one, 10, 4, None, None, None, None) == -1
    assert lib.quest_decode_forward_fused_topk_strided(h, one, one, kv, 32, None, 10, 16, None, None, None, None) == -1
    assert lib.quest_decode_set_front_end(h, 3) == 0 and lib.quest_decode_set_front_end(h, 4) == -1  # (4-6: removed in round 5)
    assert lib.quest_decode_set_front_end(h, 0) == 0
    # round-5 entries: the tiles launches and the one-launch layer refuse malformed arguments before anything runs
    assert lib.quest_append_estimate_tiles_dyn(one, one, kv, one, one, 32, 16, 10, 0, kv, one, None) == -1   # no offset
    assert lib.quest_append_estimate_tiles_dyn(one, one, kv, one, one, 32, 16, 10, 16, kv, one, None) == -1  # stride has no room
    assert lib.quest_decode_forward_fused_topk_tiles_dyn(h, one, one, kv, 32, one, 32, 10, 0, one, None, None) == -1
    assert lib.quest_decode_layer_fused_batched(h, None, one, kv, one, one, kv, 32, 10, one, b1, None, 0, None, None) == -1
    assert lib.quest_decode_layer_fused_batched(h, one, one, kv, one, one, kv, 32, 10, one, b1, None, 0, None, None) == -3  # no plan
    # the prefill kernel's entry: null tensors, no pages, more query rows than cached tokens, a group that does not divide,
    # a head_dim the MFMA kernel is not built for
    pf = lib.quest_prefill_with_paged_kv_cache
    assert pf(None, one, 8, 32, kv, 4, 1, None) == -1 and pf(one, one, 0, 32, kv, 4, 1, None) == -1
    assert pf(one, one, 8, 32, kv, 0, 1, None) == -1
    assert pf(one, one, 50, 32, kv, 4, 1, None) == -1          # 3 * 16 + 1 = 49 cached tokens < 50 rows
    assert pf(one, one, 8, 48, kv, 4, 1, None) == -1           # 48 query heads over 32 kv heads
    kv96 = PagedKV(data=16, indices=16, indptr=16, num_heads=32, page_size=16, head_dim=96, page_budget=0,
                   last_page_len=1, last_page_idx=0, layout=0)
    assert pf(one, one, 8, 32, kv96, 4, 1, None) == -2         # 64 / 128 / 256 are built
    info = (ctypes.c_uint32 * 6)()
    assert lib.quest_decode_last_launch_info(h, info) == 0 and list(info) == [0] * 6  # nothing launched yet
    assert lib.quest_decode_last_launch_info(None, info) == -1
    lib.quest_decode_handler_destroy(h)
    # fused decoder-layer launches
    assert lib.quest_decode_norm_gemv(None, None, 0.0, one, one, 64, 8, None) == -1
    assert lib.quest_decode_norm_gemv(one, None, 0.0, one, one, 60, 8, None) == -2          # in_dim % 8
    assert lib.quest_decode_gemv_residual(one, one, None, 64, 8, None) == -1
    assert lib.quest_decode_mlp_gate_up(one, None, 1e-5, one, one, one, 64, 128, None) == -1  # the MLP launch needs gamma
    assert lib.quest_decode_qkv_rope(one, one, 1e-5, one, one, one, one, one, one, 512, 4, 4, 128, 1.0, 1e4, None, None) == -1
    assert lib.quest_decode_qkv_rope(one, one, 1e-5, one, one, one, one, one, one, 512, 4, 4, 128, 0.0, 1e4, one, None) == -1
    # ... and their n-token forms: 1..16 tokens, same argument rules (checked before anything is launched)
    assert lib.quest_decode_norm_gemv_batched(None, None, 0.0, one, one, 64, 8, 8, None) == -1
    assert lib.quest_decode_norm_gemv_batched(one, None, 0.0, one, one, 64, 8, 0, None) == -1
    assert lib.quest_decode_norm_gemv_batched(one, None, 0.0, one, one, 64, 8, 17, None) == -1
    assert lib.quest_decode_norm_gemv_batched(one, None, 0.0, one, one, 60, 8, 8, None) == -2   # in_dim % 8
    assert lib.quest_decode_gemv_residual_batched(one, one, None, 64, 8, 8, None) == -1
    assert lib.quest_decode_mlp_gate_up_batched(one, None, 1e-5, one, one, one, 64, 128, 8, None) == -1
    assert lib.quest_decode_qkv_rope_batched(one, one, 1e-5, one, one, one, one, one, one, 512, 4, 4, 128, 1.0, 1e4, None, 8,
                                             None) == -1
    assert lib.quest_decode_qkv_rope_batched(one, one, 1e-5, one, one, one, one, one, one, 512, 4, 4, 72, 1.0, 1e4, one, 8,
                                             None) == -2   # head_dim % 16
    assert lib.quest_build_info().startswith(b"quest_hip gfx950 src=")


def test_batched_controller_page_budgets_host_logic():
    from quest_amd.utils import BatchedInferenceController

    bc = BatchedInferenceController(3, 1, 4, 128, 16, 5, 200, torch.float16, "cpu", num_kv_heads=2)
    assert bc.max_page_budget() == 5 and bc.page_budgets is None
    bc.set_page_budgets([3, 9, 4])
    assert bc.max_page_budget() == 9 and bc.page_budgets.tolist() == [3, 9, 4] and bc.page_budgets.dtype == torch.int32
    with pytest.raises(ValueError):
        bc.set_page_budgets([3, 9])       # one per sequence
    with pytest.raises(ValueError):
        bc.set_page_budgets([3, 0, 4])    # >= 1: the current page is always attended
    bc.set_page_budgets(None)
    assert bc.max_page_budget() == 5 and bc.page_budgets is None
    with pytest.raises(RuntimeError, match="enable_device_state"):
        bc.begin_forward()                # eager batched steps need the device-resident tables / states


def test_kernel_sweep_table_and_side_flags():
    import importlib.util
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse(["--kernel-sweep"])
    assert a.kernel_sweep and not a.no_side and a.side_steps == 100
    assert bench.parse(["--no-side"]).no_side
    sys.path.insert(0, os.path.join(root, "scripts"))
    import kbench_reference_rows as sweep

    # the byte accounting of the reference bench (bench_batch_decode.cu:82-86) for its published 128-page row
    r = sweep.row("attention seqlen=4096 page_budget=128", 10.0, 32 * 128 * 2 + 128 * 2 * 32 * 16 * 128 * 2 + 8 + 32 * 127 * 4,
                  32 * 128 * 2, published_rtx6000ada={"us": 63.551, "GBps": 528.5, "pct_of_its_peak": 55.05})
    assert abs(r["read_MiB"] - 32.023) < 1e-3          # the figure the reference's plot prints
    md = sweep.markdown({"rows": [r]})
    assert "63.551 us" in md and md.count("\n") == 2
    assert sweep.LONGBENCH_LEN == (5819, 15370, 11984, 14101, 24723, 8154) and sweep.LONGBENCH_BUDGET[4] == 4096


def test_n_token_launch_planner_host_logic():
    """quest_decode_batched_plan (no GPU needed): Llama-2-7B launches at 8 tokens take the persistent kernel with the
    inputs in LDS -- except down_proj, whose 8 x 11008 inputs do not fit -- one workgroup per CU (256 when no device is
    visible), at most 4 weight fragments per task, every quad of rows covered by the rounds."""
    import ctypes

    from quest_amd._lib import lib

    def plan(in_dim, rows, n, rope=0):
        info = (ctypes.c_uint32 * 6)()
        rc = lib.quest_decode_batched_plan(in_dim, rows, n, rope, info)
        return rc, list(info)

    for name, in_dim, rows, rope in (("qkv", 4096, 3 * 4096, 1), ("o_proj", 4096, 4096, 0), ("gate_up", 4096, 2 * 11008, 0),
                                     ("lm_head", 4096, 32000, 0)):
        for n in (1, 4, 8, 16):
            rc, (ks, rounds, lds, grid, spp, cm) = plan(in_dim, rows, n, rope)
            assert rc == 1, (name, n)
            assert grid >= 1 and spp == 32 and cm == 4 and ks in (1, 2, 4, 8, 16) and (spp + ks - 1) // ks <= cm
            assert rounds * grid * (16 // ks) * 4 >= rows              # every row has a task
            assert (rounds - 1) * grid * (16 // ks) * 4 < rows         # and no round is empty
            tokens = 4 if n <= 4 else (8 if n <= 8 else 16)
            assert lds >= tokens * (in_dim + 32) * 2 and lds > 80 * 1024  # inputs fit; one workgroup per CU
    assert plan(11008, 4096, 8)[0] == 0          # down_proj at 8 tokens: inputs-from-L2 kernel
    assert plan(11008, 4096, 4)[0] == 0          # ... and at 4: rows longer than 16 slices x 4 steps x 128
    assert plan(8192, 4096, 4)[0] == 1
    assert plan(256, 40, 2)[0] == 0              # rows shorter than a wave sweep
    assert plan(4096, 4096, 0)[0] == -1 and plan(4096, 4096, 17)[0] == -1 and plan(4100, 4096, 8)[0] == -2


def test_fused_decoder_layer_shape_gate():
    """models outside the fused decoder-layer launches' shapes must fall back to the module path BEFORE a graph
    capture (the launches would return QUEST_EUNSUPPORTED inside it)."""
    from types import SimpleNamespace as NS

    from quest_amd.models.llama import fused_layer_launches_supported as ok

    llama7b = NS(hidden_size=4096, intermediate_size=11008, num_attention_heads=32)
    assert ok(llama7b, 1) and ok(llama7b, 8) and ok(NS(hidden_size=4096, intermediate_size=14336, num_attention_heads=32), 16)
    assert not ok(NS(hidden_size=4096, intermediate_size=11004, num_attention_heads=32), 1)   # not a multiple of 8 halves
    assert not ok(NS(hidden_size=4100, intermediate_size=11008, num_attention_heads=41), 1)
    assert not ok(NS(hidden_size=8192, intermediate_size=32768, num_attention_heads=64), 1)   # input vector beyond the LDS stage
    assert ok(NS(hidden_size=1536, intermediate_size=4096, num_attention_heads=16, head_dim=72), 1)
    assert not ok(NS(hidden_size=1536, intermediate_size=4096, num_attention_heads=16, head_dim=72), 4)  # n-token RoPE blocks
