"""Drop-in at the model level: a Llama checkpoint written by Hugging Face transformers loads into
quest_amd.models.llama and -- with a page budget that covers the cache, i.e. the reference's full-attention
branch -- reproduces transformers' own logits, prefill and decode, through the HIP kernels (RMSNorm, RoPE, paged
append with min/max metadata, paged decode attention) and the one-graph-per-token path.  With a small budget the
same model runs the Quest sparse path (no HF counterpart: checked for agreement between eager and graph modes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kv_heads,rope", [(4, None), (2, {"rope_type": "linear", "factor": 4.0, "rope_theta": 10000.0})])
def test_hf_checkpoint_logits_match_transformers(tmp_path, kv_heads, rope):
    transformers = pytest.importorskip("transformers")
    from quest_amd.models.llama import LlamaForCausalLM

    dev = torch.device("cuda:0")
    kw = dict(vocab_size=320, hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4,
              num_key_value_heads=kv_heads, max_position_embeddings=1024, rms_norm_eps=1e-5)
    if rope is not None:
        kw["rope_parameters"] = rope
    hcfg = transformers.LlamaConfig(**kw)
    torch.manual_seed(3)
    hf = transformers.LlamaForCausalLM(hcfg)
    for p in hf.parameters():  # twice the 0.02 init so that attention patterns are not flat
        p.data.mul_(2.0)
    for n, p in hf.named_parameters():
        if "norm" in n:
            p.data.fill_(1.0)
    d = str(tmp_path / "ckpt")
    hf.half().save_pretrained(d, safe_serialization=True)
    hf32 = hf.float().eval()  # fp32 reference on the CPU, from the SAME fp16 weights
    # transformers' own fp16 run on the GPU: the yardstick for how far an fp16 pipeline sits from fp32 on this model
    hf16 = transformers.LlamaForCausalLM.from_pretrained(d, dtype=torch.float16, attn_implementation="eager").to(dev).eval()

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    def check(got, r16, r32, what):
        got, r16 = got.float().cpu(), r16.float().cpu()
        assert rel(got, r32) <= 1.5 * rel(r16, r32) + 2e-3, f"{what}: {rel(got, r32):.4f} vs HF fp16 {rel(r16, r32):.4f}"
        assert rel(got, r16) <= 1e-2, f"{what}: {rel(got, r16):.4f} from transformers' fp16 logits"

    m = LlamaForCausalLM.from_pretrained(d, device=dev)
    assert m.config.num_key_value_heads == kv_heads
    assert m.model.layers[0].self_attn.rope_scale == (4.0 if rope else 1.0)
    m.quest_init(16, 512, token_budget=16 * 1024)  # budget >= pages: full attention, comparable with HF
    prompt = (torch.arange(150)[None] * 37 + 11) % 320
    n_new = 20  # crosses the 160-token page boundary
    with torch.no_grad():
        ref32 = hf32(input_ids=prompt, use_cache=True)
        ref16 = hf16(input_ids=prompt.to(dev), use_cache=True)
    with torch.inference_mode():
        hidden = m.model(input_ids=prompt.to(dev))
        got_all = m.lm_head(hidden)[0]
    check(got_all, ref16.logits[0], ref32.logits[0], "prefill, all positions")
    check(got_all[-1], ref16.logits[0, -1], ref32.logits[0, -1], "prefill, last position")
    past32, past16, tok = ref32.past_key_values, ref16.past_key_values, ref32.logits[0, -1].argmax()
    m.capture_decode_graph()
    for t in range(n_new):
        with torch.no_grad():
            ref32 = hf32(input_ids=tok.view(1, 1), past_key_values=past32, use_cache=True)
            ref16 = hf16(input_ids=tok.view(1, 1).to(dev), past_key_values=past16, use_cache=True)
        past32, past16 = ref32.past_key_values, ref16.past_key_values
        with torch.inference_mode():
            got = m.decode_graph_step(input_ids=tok.view(1, 1).to(dev))
        check(got[0, -1], ref16.logits[0, -1], ref32.logits[0, -1], f"decode token {t}")
        tok = ref32.logits[0, -1].argmax()
    assert m.model.iController.kv_cache.seqlen == 150 + n_new

    # the same checkpoint on the Quest sparse path (budget 4 pages of 16 tokens): eager == graph replay of the same
    # module-by-module step (bit for bit), finite.  (The fused decoder layers of the default graph differ from the
    # module path by fp16 rounding: tests/test_gpu_decode_layer.py.)
    outs = []
    for graph, kv_layout in ((False, "NHD"), (True, "NHD"), (True, "NHD_ROT")):
        q = LlamaForCausalLM.from_pretrained(d, device=dev)
        q.quest_init(16, 512, token_budget=64, kv_layout=kv_layout)
        with torch.inference_mode():
            lg = q(input_ids=prompt.to(dev))
            if graph:
                q.capture_decode_graph(fused_layers=False)
            seq = []
            for t in range(12):
                tk = lg[0, -1].argmax().view(1, 1)
                lg = q.decode_graph_step(input_ids=tk) if graph else q(input_ids=tk)
                seq.append(lg.float().clone())
        assert q.model.iController.need_estimate() or graph
        outs.append(torch.stack(seq))
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    # ... and on the row-rotated pool (quest_init(kv_layout="NHD_ROT"), round 6): where a vector lives does not change a logit
    assert torch.equal(outs[0], outs[2])
