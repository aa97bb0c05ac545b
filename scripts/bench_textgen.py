#!/usr/bin/env python3
"""End-to-end decode latency of a Llama-architecture model with Quest attention vs full-KV attention
(the shape of the reference's scripts/bench_textgen.py:75-97 and README Fig. 10).

Random weights (no checkpoints offline) of Llama-2-7B shape by default; the KV cache of `--ctx` tokens is
filled with synthetic keys/values through append_kv -- or, with --real-prefill, by the reference harness's own prefill: one
model forward over `--ctx` random hidden states, timed (attention on the MFMA kernel of csrc/prefill.hip) --, then one
decode token -- RMSNorm, QKV GEMV, RoPE, [append+estimate | top-k+sparse attention | merge], o_proj, MLP,
lm_head -- is captured in a hipGraph and replayed.  First `skip` = 2 layers run dense like the reference
(quest/models/llama.py:428-439).

    python scripts/bench_textgen.py --ctx 32768 --token-budget 2048
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def time_decode(model, ctl, emb, reps):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.inference_mode():
        model(inputs_embeds=emb)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.inference_mode(), torch.cuda.graph(g):
        out = model(inputs_embeds=emb)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    return e0.elapsed_time(e1) / reps


def time_batched(cfg, a, dense, dev):
    """`--seqs` > 1: all sequences decode together, one hipGraph replay per token for the batch
    (LlamaForCausalLM.capture_decode_graph_batched).  dense = every layer full-KV (group-shared kernel)."""
    from quest_amd.models.llama import LlamaForCausalLM
    import quest_amd.utils as qu

    with torch.device(dev):
        model = LlamaForCausalLM(cfg).half()
    for p in model.parameters():
        p.data.normal_(0, 0.02)
    for m in model.modules():
        if hasattr(m, "variance_epsilon"):
            m.weight.data.fill_(1.0)
    model.quest_init_batched(a.seqs, 16, a.ctx + 256, a.token_budget, kv_layout=a.layout)
    if dense:
        model.model._quest_skip_layer = a.layers
    b = model.model.bController
    g = torch.Generator(device=dev).manual_seed(1)
    D = a.hidden // a.heads
    k = torch.empty(a.ctx, a.kv_heads, D, dtype=torch.float16, device=dev)
    v = torch.empty_like(k)
    for c in b.seqs:
        c.prepare_metadata(a.ctx)
        c.begin_forward(a.ctx)
        for l in range(a.layers):
            k.normal_(generator=g)
            v.normal_(generator=g)
            qu.append_kv(k, v, c, l)
        c.end_forward()
    del k, v
    model.capture_decode_graph_batched()
    time_batched.fused_layers = bool(model.fused_layers)
    model.graph_input.copy_(torch.randn(a.seqs, 1, a.hidden, generator=g, device=dev, dtype=torch.float16) * 0.1)
    for _ in range(3):
        model._graph.replay()
        b.prepare_metadata(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        model._graph.replay()
        b.prepare_metadata(1)
    e1.record()
    torch.cuda.synchronize()
    assert torch.isfinite(model.graph_logits.float()).all()
    ms = e0.elapsed_time(e1) / a.reps
    del model, b
    torch.cuda.empty_cache()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--token-budget", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--layout", choices=["NHD", "HND", "NHD_ROT"], default="NHD_ROT",
                    help="pool layout (quest_init's kv_layout): the row-rotated NHD of round 6 by default, NHD = the reference's")
    ap.add_argument("--seqs", type=int, default=1, help="sequences decoded together (batched launches)")
    ap.add_argument("--checkpoint", default=None,
                    help="Hugging Face Llama checkpoint directory (config.json + safetensors) to load instead of "
                         "random-initialising; shapes come from its config.json")
    ap.add_argument("--make-checkpoint", default=None,
                    help="first WRITE a random-weight checkpoint of the given shape in HF layout to this directory "
                         "(save_pretrained), then load it back with from_pretrained and bench that model")
    ap.add_argument("--generate", type=int, default=0,
                    help="also generate this many tokens greedily by graph replay from the synthetic cache and report "
                         "ms/token over the whole run (host loop included)")
    ap.add_argument("--real-prefill", action="store_true",
                    help="fill the cache the way the reference's harness does (scripts/bench_textgen.py:77-84): ONE model "
                         "forward over --ctx random hidden states -- projections, RoPE, append with page metadata, the MFMA "
                         "prefill attention kernel, MLP in every layer -- and report its latency (time to first token)")
    a = ap.parse_args()
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM
    import quest_amd.utils as qu

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=a.vocab, hidden_size=a.hidden, intermediate_size=a.inter, num_hidden_layers=a.layers,
                      num_attention_heads=a.heads, num_key_value_heads=a.kv_heads, max_position_embeddings=a.ctx + 1024)
    ckpt_info = None
    if a.make_checkpoint:
        import time
        t0 = time.perf_counter()
        with torch.device(dev):
            tmp = LlamaForCausalLM(cfg).half()
        for p in tmp.parameters():
            p.data.normal_(0, 0.02)
        for m in tmp.modules():
            if hasattr(m, "variance_epsilon"):
                m.weight.data.fill_(1.0)
        tmp.save_pretrained(a.make_checkpoint)
        del tmp
        torch.cuda.empty_cache()
        a.checkpoint = a.make_checkpoint
        ckpt_info = {"written_s": time.perf_counter() - t0}
    if a.checkpoint:
        with open(os.path.join(a.checkpoint, "config.json")) as f:
            cfg = LlamaConfig.from_hf_dict(json.load(f))
        a.layers, a.hidden, a.heads, a.kv_heads = (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads,
                                                   cfg.num_key_value_heads)
        ckpt_info = dict(ckpt_info or {}, path=a.checkpoint,
                         files=sorted(f for f in os.listdir(a.checkpoint) if f.endswith((".safetensors", ".json"))),
                         bytes=sum(os.path.getsize(os.path.join(a.checkpoint, f)) for f in os.listdir(a.checkpoint)))
    if a.seqs > 1:
        q_ms = time_batched(cfg, a, False, dev)
        d_ms = time_batched(cfg, a, True, dev)
        print(json.dumps({"bench": "e2e batched decode, random-weight Llama", "sequences": a.seqs, "ctx": a.ctx, "kv_layout": a.layout,
                          "token_budget": a.token_budget, "layers": a.layers, "hidden": a.hidden, "heads": a.heads,
                          "kv_heads": a.kv_heads, "intermediate": a.inter, "vocab": a.vocab, "dense_first_layers": 2,
                          "fused_decoder_layer_launches": getattr(time_batched, "fused_layers", None), "ms_per_step_quest": q_ms,
                          "ms_per_step_full_kv": d_ms, "tokens_per_s_quest": a.seqs / (q_ms * 1e-3),
                          "tokens_per_s_full_kv": a.seqs / (d_ms * 1e-3), "speedup": d_ms / q_ms}))
        return
    results = {}
    gen_ms = {}
    prefill_s = {}
    for name, budget in (("quest", a.token_budget), ("dense", 1 << 24)):
        if a.checkpoint:
            import time
            t0 = time.perf_counter()
            model = LlamaForCausalLM.from_pretrained(a.checkpoint, device=dev)
            ckpt_info["load_s"] = time.perf_counter() - t0
        else:
            with torch.device(dev):
                model = LlamaForCausalLM(cfg).half()
            for p in model.parameters():
                p.data.normal_(0, 0.02)
            for m in model.modules():
                if hasattr(m, "variance_epsilon"):
                    m.weight.data.fill_(1.0)
        model.quest_init(16, a.ctx + 256 + a.generate, budget, kv_layout=a.layout)
        ctl = model.model.iController
        g = torch.Generator(device=dev).manual_seed(1)
        D = a.hidden // a.heads
        if a.real_prefill:
            import time
            x = torch.randn(1, a.ctx, a.hidden, generator=g, device=dev, dtype=torch.float16)
            with torch.inference_mode():
                model(inputs_embeds=x[:, :256])  # warm the libraries' kernels on a short request, then start over
                model.quest_clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first = model(inputs_embeds=x)
                torch.cuda.synchronize()
            prefill_s[name] = time.perf_counter() - t0
            assert torch.isfinite(first.float()).all() and ctl.kv_cache.seqlen == a.ctx
            del x, first
        else:
            ctl.prepare_metadata(a.ctx)
            ctl.begin_forward(a.ctx)
            k = torch.empty(a.ctx, a.kv_heads, D, dtype=torch.float16, device=dev)
            v = torch.empty_like(k)
            for l in range(a.layers):
                k.normal_(generator=g)
                v.normal_(generator=g)
                qu.append_kv(k, v, ctl, l)
            ctl.end_forward()
            del k, v
        emb = torch.randn(1, 1, a.hidden, generator=g, device=dev, dtype=torch.float16) * 0.1
        results[name] = time_decode(model, ctl, emb, a.reps)
        if a.generate:
            # greedy generation by one graph replay per token (device-resident lengths; the host only feeds the token)
            import time
            model.capture_decode_graph()
            tok = torch.zeros(1, 1, dtype=torch.long, device=dev)
            with torch.inference_mode():
                for _ in range(3):
                    tok = model.decode_graph_step(input_ids=tok).argmax(-1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.generate):
                    tok = model.decode_graph_step(input_ids=tok).argmax(-1)
                torch.cuda.synchronize()
            gen_ms[name] = (time.perf_counter() - t0) * 1e3 / a.generate
        del model, ctl
        torch.cuda.empty_cache()
    out = {"bench": "e2e decode latency, random-weight Llama", "ctx": a.ctx, "token_budget": a.token_budget,
           "page_budget_pages": a.token_budget // 16, "layers": a.layers, "hidden": a.hidden, "heads": a.heads,
           "kv_heads": a.kv_heads, "dense_first_layers": 2, "kv_layout": a.layout,
           "ms_per_token_quest": results["quest"], "ms_per_token_full_kv": results["dense"],
           "speedup": results["dense"] / results["quest"],
           "reference_published": "RTX 6000 Ada, ctx 32768 FP16: 36.8 ms -> 21.2 ms @ budget 2048 (1.74x)"}
    if gen_ms:
        out["generation_fused_decoder_layers"] = os.environ.get("QUEST_FUSED_LAYER", "1") != "0"
        out["generated_tokens"] = a.generate
        out["ms_per_generated_token_quest"] = gen_ms["quest"]
        out["ms_per_generated_token_full_kv"] = gen_ms["dense"]
    if prefill_s:
        out["prefill_s_quest"] = prefill_s["quest"]  # the same work in both runs (prefill is dense): two samples
        out["prefill_s_full_kv"] = prefill_s["dense"]
        out["prefill"] = ("one model forward over ctx random hidden states, as the reference harness "
                          "(scripts/bench_textgen.py:77-84); attention = csrc/prefill.hip")
    if ckpt_info:
        out["checkpoint"] = ckpt_info
        out["bench"] = "e2e decode latency, Llama checkpoint in Hugging Face layout loaded with from_pretrained"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
