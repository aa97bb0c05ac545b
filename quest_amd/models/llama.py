"""Llama-architecture decoder that threads an ``InferenceController`` through ``QuestAttention``.

The reference ships a fork of HF's modeling_llama (quest/models/llama.py, 835 lines) whose only
Quest-specific parts are: ``quest_init`` / ``quest_clear`` (:520-560), the per-forward controller
sequence in ``LlamaModel.forward`` (:424-439, :486: prepare_metadata -> begin_forward with a huge budget
for the first ``_quest_skip_layer`` = 2 layers -> re-plan with the real budget -> end_forward) and the
fused RMSNorm call (:72).  This module restates exactly those parts around a minimal decoder (embedding,
RMSNorm, QuestAttention, SwiGLU MLP, lm_head) with HF parameter names, so a HF Llama state_dict loads
with ``load_state_dict``; everything else of the HF class hierarchy (generation mixin, attention-mask
plumbing, gradient checkpointing) is out of scope.  Batch size 1, like the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from .. import utils as qutils
from .QuestAttention import QuestAttention


@dataclass
class LlamaConfig:
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    rms_norm_eps: float = 1e-5
    max_position_embeddings: int = 32768
    rope_scaling: Optional[dict] = None
    rope_theta: float = 1e4

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads

    # ---- Hugging Face config.json <-> LlamaConfig (the fields the Quest path reads; quest/models/llama.py takes the
    # whole HF LlamaConfig, of which it uses exactly these)
    @classmethod
    def from_hf_dict(cls, d: dict) -> "LlamaConfig":
        if d.get("model_type", "llama") != "llama":
            raise ValueError(f"not a Llama checkpoint: model_type = {d.get('model_type')!r}")
        rope = d.get("rope_parameters") or {}        # transformers >= 5 nests theta / scaling here
        scaling = d.get("rope_scaling")
        if scaling is None and rope.get("rope_type", "default") not in ("default", None):
            scaling = {k: v for k, v in rope.items() if k != "rope_theta"}
        if d.get("attention_bias") or d.get("mlp_bias"):
            raise ValueError("bias terms are not part of the reference's Llama fork")
        head_dim = d.get("head_dim")
        if head_dim is not None and head_dim * d["num_attention_heads"] != d["hidden_size"]:
            raise ValueError("head_dim * num_attention_heads != hidden_size is not supported (QuestAttention.py:24-29)")
        return cls(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], intermediate_size=d["intermediate_size"],
                   num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
                   num_key_value_heads=d.get("num_key_value_heads"), rms_norm_eps=d.get("rms_norm_eps", 1e-5),
                   max_position_embeddings=d.get("max_position_embeddings", 32768), rope_scaling=scaling,
                   rope_theta=float(d.get("rope_theta", rope.get("rope_theta", 1e4))))

    def to_hf_dict(self, dtype: str = "float16", tie_word_embeddings: bool = False) -> dict:
        return {"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_act": "silu",
                "attention_bias": False, "mlp_bias": False, "torch_dtype": dtype, "dtype": dtype,
                "vocab_size": self.vocab_size, "hidden_size": self.hidden_size,
                "intermediate_size": self.intermediate_size, "num_hidden_layers": self.num_hidden_layers,
                "num_attention_heads": self.num_attention_heads, "num_key_value_heads": self.num_key_value_heads,
                "rms_norm_eps": self.rms_norm_eps, "max_position_embeddings": self.max_position_embeddings,
                "rope_scaling": self.rope_scaling, "rope_theta": self.rope_theta,
                "tie_word_embeddings": tie_word_embeddings}


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return qutils.rms_norm_forward(hidden_states, self.weight, self.variance_epsilon)  # llama.py:72


class LlamaMLP(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(torch.nn.functional.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config: LlamaConfig, layer_idx: int, fused: bool = True):
        super().__init__()
        self.self_attn = QuestAttention(config, layer_idx, fused=fused)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, hidden_states: torch.Tensor, iController) -> torch.Tensor:
        h, _, _ = self.self_attn(self.input_layernorm(hidden_states), iController=iController)
        hidden_states = hidden_states + h
        return hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))

    def forward_dyn(self, hidden_states: torch.Tensor, iController, scores, dense: bool) -> torch.Tensor:
        h = self.self_attn.forward_dyn(self.input_layernorm(hidden_states), iController, scores, dense)
        hidden_states = hidden_states + h
        return hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))


    def forward_batched(self, hidden_states: torch.Tensor, bController, scores, dense: bool) -> torch.Tensor:
        h = self.self_attn.forward_batched(self.input_layernorm(hidden_states), bController, scores, dense)
        hidden_states = hidden_states + h
        return hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))

    def forward_dyn_fused(self, h: torch.Tensor, iController, scores, dense: bool, ws: "DecodeWorkspace") -> torch.Tensor:
        """``forward_dyn`` in 4 launches besides the attention's: RMSNorm + q/k/v projections + RoPE, o_proj + residual,
        RMSNorm + gate/up + SiLU*up, down_proj + residual (csrc/decode_layer.hip; EXTENSION).  ``h`` ``[hidden]`` is the
        residual stream, updated IN PLACE; batch 1, fp16."""
        from .. import _kernels

        a, m = self.self_attn, self.mlp
        _kernels.decode_qkv_rope(h, self.input_layernorm.weight, self.input_layernorm.variance_epsilon, a.q_proj.weight,
                                 a.k_proj.weight, a.v_proj.weight, ws.q, ws.k, ws.v, a.head_dim, a.rope_scale,
                                 a.rope_theta, iController.step_state)
        if dense:
            attn = qutils.decode_layer_dense_dyn(ws.q, ws.k, ws.v, iController, a.layer_idx)
        else:
            attn = qutils.decode_layer_dyn(ws.q, ws.k, ws.v, iController, a.layer_idx, scores)
        _kernels.decode_gemv_residual(attn, a.o_proj.weight, h)
        _kernels.decode_mlp_gate_up(h, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon,
                                    m.gate_proj.weight, m.up_proj.weight, ws.act)
        _kernels.decode_gemv_residual(ws.act, m.down_proj.weight, h)
        return h


    def forward_batched_fused(self, h: torch.Tensor, bController, scores, dense: bool, ws: "DecodeWorkspace") -> torch.Tensor:
        """``forward_batched`` with the same 4 fused launches per layer as ``forward_dyn_fused``, each for all ``n``
        tokens at once (the weights are read once per batch: csrc/decode_layer.hip persist_kernel).  ``h`` ``[n, hidden]``
        is the residual stream, updated IN PLACE; fp16, n <= 16."""
        from .. import _kernels

        a, m = self.self_attn, self.mlp
        _kernels.decode_qkv_rope_batched(h, self.input_layernorm.weight, self.input_layernorm.variance_epsilon,
                                         a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, ws.q, ws.k, ws.v, a.head_dim,
                                         a.rope_scale, a.rope_theta, bController.step_states)
        if dense:
            attn = qutils.decode_layer_dense_batched(ws.q, ws.k, ws.v, bController, a.layer_idx, out=ws.attn)
        else:
            attn = qutils.decode_layer_batched(ws.q, ws.k, ws.v, bController, a.layer_idx, scores, out=ws.attn)
        _kernels.decode_gemv_residual_batched(attn, a.o_proj.weight, h)
        _kernels.decode_mlp_gate_up_batched(h, self.post_attention_layernorm.weight,
                                            self.post_attention_layernorm.variance_epsilon, m.gate_proj.weight,
                                            m.up_proj.weight, ws.act)
        _kernels.decode_gemv_residual_batched(ws.act, m.down_proj.weight, h)
        return h


def fused_layer_launches_supported(config, n_tokens: int) -> bool:
    """Shapes the fused decoder-layer launches (csrc/decode_layer.hip) serve -- checked BEFORE a decode graph is captured,
    so that a model outside them falls back to the module path (nn.Linear + separate norm / RoPE / activation launches)
    instead of failing inside the capture with QUEST_EUNSUPPORTED: input dimensions a multiple of 8 halves (16-byte
    vectors); the batch-1 kernels stage an input vector of at most 30720 halves in LDS (24576 with the RMSNorm prologue
    of the n-token fallback kernel); RoPE pairs need head_dim % 4 == 0 (batch 1) / % 16 == 0 (n tokens)."""
    hidden, inter = config.hidden_size, config.intermediate_size
    head_dim = getattr(config, "head_dim", None) or hidden // config.num_attention_heads
    if hidden % 8 or inter % 8 or head_dim % 4:
        return False
    if max(hidden, inter) > 30720:
        return False
    if n_tokens > 1 and (head_dim % 16 or hidden > 24576):
        return False
    return True


class DecodeWorkspace:
    """Scratch of the fused decode layer (one set for the whole model: layers run one after the other); ``n`` tokens
    (sequences of a batch) per step."""

    def __init__(self, config: LlamaConfig, dtype, device, n: int = 1):
        d = config.hidden_size // config.num_attention_heads
        self.q = torch.empty(n, config.num_attention_heads, d, dtype=dtype, device=device)
        self.k = torch.empty(n, config.num_key_value_heads, d, dtype=dtype, device=device)
        self.v = torch.empty(n, config.num_key_value_heads, d, dtype=dtype, device=device)
        self.attn = torch.empty(n, config.num_attention_heads, d, dtype=dtype, device=device)
        self.act = torch.empty((n, config.intermediate_size) if n > 1 else (config.intermediate_size,), dtype=dtype, device=device)
        self.h = torch.empty((n, config.hidden_size) if n > 1 else (config.hidden_size,), dtype=dtype, device=device)


class LlamaModel(nn.Module):
    def __init__(self, config: LlamaConfig, fused: bool = True):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, i, fused) for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)
        self._quest_skip_layer = 2            # llama.py:538
        self._quest_max_page_limit = 1 << 20  # llama.py:537
        self._quest_page_budget = None
        self.iController: Optional[qutils.InferenceController] = None

    def forward(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None):
        ctl = self.iController
        assert ctl is not None, "call quest_init() first"
        h = self.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        q_len = h.shape[1]
        ctl.prepare_metadata(q_len)                                   # llama.py:425
        ctl.set_page_budget(self._quest_max_page_limit)               # llama.py:428-430: dense first layers
        ctl.begin_forward(q_len)
        for idx, layer in enumerate(self.layers):
            if idx == self._quest_skip_layer:                         # llama.py:434-439
                ctl.end_forward()
                ctl.set_page_budget(self._quest_page_budget)
                ctl.begin_forward(q_len, updateTensor=False)
            h = layer(h, ctl)
        ctl.end_forward()                                             # llama.py:486
        return self.norm(h)


    def forward_decode_dyn(self, h: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        """One decode token with no host-side planning: the device-side state advance replaces
        prepare_metadata / begin_forward / end_forward (capturable, replayable)."""
        ctl = self.iController
        qutils.step_advance_dyn(ctl)
        for idx, layer in enumerate(self.layers):
            h = layer.forward_dyn(h, ctl, scores, dense=idx < self._quest_skip_layer)
        return self.norm(h)


    def forward_decode_dyn_fused(self, h: torch.Tensor, scores: torch.Tensor, ws: DecodeWorkspace) -> torch.Tensor:
        """``forward_decode_dyn`` with the fused decode layer; returns the residual stream ``[hidden]`` BEFORE the final
        norm (the caller fuses that into its lm_head launch)."""
        ctl = self.iController
        qutils.step_advance_dyn(ctl)
        ws.h.copy_(h.reshape(-1))
        for idx, layer in enumerate(self.layers):
            layer.forward_dyn_fused(ws.h, ctl, scores, idx < self._quest_skip_layer, ws)
        return ws.h

    def forward_decode_batched_fused(self, h: torch.Tensor, scores: torch.Tensor, ws: DecodeWorkspace) -> torch.Tensor:
        """``forward_decode_batched`` with the fused n-token decoder layers; returns the residual stream ``[n, hidden]``
        BEFORE the final norm (fused into the caller's lm_head launch)."""
        b = self.bController
        qutils.step_advance_batched(b)
        ws.h.copy_(h.reshape(ws.h.shape))
        for idx, layer in enumerate(self.layers):
            layer.forward_batched_fused(ws.h, b, scores, idx < self._quest_skip_layer, ws)
        return ws.h

    def forward_decode_batched(self, h: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        """One decode token of every sequence of ``self.bController`` (``h``: ``[n, 1, hidden]``)."""
        b = self.bController
        qutils.step_advance_batched(b)
        for idx, layer in enumerate(self.layers):
            h = layer.forward_batched(h, b, scores, dense=idx < self._quest_skip_layer)
        return self.norm(h)


class LlamaForCausalLM(nn.Module):
    def __init__(self, config: LlamaConfig, fused: bool = True):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, fused)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    # ------------------------------------------------------------------ Hugging Face checkpoint format
    @classmethod
    def from_pretrained(cls, path: str, device=torch.device("cuda:0"), dtype=torch.float16,
                        fused: bool = True) -> "LlamaForCausalLM":
        """Load a Hugging Face Llama checkpoint directory -- ``config.json`` + ``model.safetensors`` or the sharded
        ``model-0000x-of-0000y.safetensors`` + ``model.safetensors.index.json`` -- the way the reference does with
        ``LlamaForCausalLM.from_pretrained(...)`` (scripts/bench_textgen.py:53-58, evaluation/*).  Parameter names
        are HF's (``model.layers.N.self_attn.q_proj.weight`` ...), so tensors are copied by name, shard by shard,
        straight to ``device``."""
        import json
        import os

        from safetensors import safe_open

        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        cfg = LlamaConfig.from_hf_dict(hf)
        # parameters are created directly in `dtype` on `device` and left uninitialised (every one of them is overwritten
        # from the checkpoint below, or reported missing): no fp32 copy of the model, no random-init pass
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device("meta"):
                model = cls(cfg, fused=fused)
            model = model.to_empty(device=device)
        finally:
            torch.set_default_dtype(prev)
        index = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = ["model.safetensors"]
        params = dict(model.state_dict())
        seen = set()
        with torch.no_grad():
            for fn in files:
                with safe_open(os.path.join(path, fn), framework="pt", device=str(device)) as sf:
                    for name in sf.keys():
                        if name.endswith("rotary_emb.inv_freq"):  # buffer of older HF exports; recomputed by the rope kernel
                            continue
                        if name not in params:
                            raise KeyError(f"{fn}: unexpected tensor {name!r} (not a Llama decoder parameter)")
                        t = sf.get_tensor(name)
                        if t.shape != params[name].shape:
                            raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != model {tuple(params[name].shape)}")
                        params[name].copy_(t)
                        if t.dtype != params[name].dtype and t.dtype in (torch.bfloat16, torch.float32) \
                                and not bool(torch.isfinite(params[name]).all()):
                            # e.g. a bf16 checkpoint whose values exceed fp16's range (the kernels are fp16 like the reference's)
                            raise ValueError(f"{name}: {t.dtype} values overflow {params[name].dtype}")
                        seen.add(name)
        missing = set(params) - seen
        if missing == {"lm_head.weight"} and hf.get("tie_word_embeddings", False):
            model.lm_head.weight = model.model.embed_tokens.weight
            missing = set()
        if missing:
            raise KeyError(f"checkpoint lacks {sorted(missing)[:5]}{' ...' if len(missing) > 5 else ''}")
        return model

    def save_pretrained(self, path: str, max_shard_bytes: int = 4 << 30) -> None:
        """Write ``config.json`` + safetensors shard(s) in the Hugging Face layout (tensor names = HF's), so that
        ``transformers.LlamaForCausalLM.from_pretrained(path)`` -- or ``from_pretrained`` above -- reads it."""
        import json
        import os

        from safetensors.torch import save_file

        os.makedirs(path, exist_ok=True)
        sd = self.state_dict()
        tied = self.lm_head.weight.data_ptr() == self.model.embed_tokens.weight.data_ptr()
        if tied:
            sd.pop("lm_head.weight")
        dt = {torch.float16: "float16", torch.bfloat16: "bfloat16", torch.float32: "float32"}[self.lm_head.weight.dtype]
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_hf_dict(dt, tied), f, indent=1)
        shards, cur, size = [], {}, 0
        for name, t in sd.items():
            nbytes = t.numel() * t.element_size()
            if cur and size + nbytes > max_shard_bytes:
                shards.append(cur)
                cur, size = {}, 0
            cur[name] = t
            size += nbytes
        shards.append(cur)
        if len(shards) == 1:
            save_file({k: v.detach().contiguous().cpu() for k, v in shards[0].items()},
                      os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
            return
        weight_map, total = {}, 0
        for i, shard in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file({k: v.detach().contiguous().cpu() for k, v in shard.items()}, os.path.join(path, fn),
                      metadata={"format": "pt"})
            for k, v in shard.items():
                weight_map[k] = fn
                total += v.numel() * v.element_size()
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)

    def quest_init(self, page_size: int, max_seq_len: int, token_budget: int = 512, dtype=torch.float16,
                   device=torch.device("cuda:0"), kv_layout="NHD") -> None:
        """llama.py:520-552: build the controller; ``token_budget`` is in TOKENS, the controller's page
        budget is ``token_budget // page_size`` pages (llama.py:536).  ``kv_layout`` (EXTENSION; the reference hard-wires
        NHD, controller.py:38): "NHD", "HND" or this build's row-rotated "NHD_ROT" (same results, faster batched launches)."""
        assert self.model.iController is None, "Can't init Quest Controller twice."
        cfg = self.config
        self.model._quest_page_budget = token_budget // page_size
        self.model.iController = qutils.InferenceController(
            cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads, page_size,
            self.model._quest_page_budget, max_seq_len, dtype, device, num_kv_heads=cfg.num_key_value_heads,
            layout=qutils.TensorLayout.parse(kv_layout))
        print(f"Quest allocates KV-Cache of {max_seq_len} tokens; token_budget {token_budget} = "
              f"{self.model._quest_page_budget} pages of {page_size}")

    def quest_clear(self) -> None:
        """llama.py:554-560: release the pages for the next request."""
        assert self.model.iController is not None, "Must quest_init() before quest_clear()."
        self.model.iController.clean_states()
        self._graph = None  # the captured step addressed the finished request's pages (re-capture after prefill)

    def forward(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None):
        h = self.model(input_ids=input_ids, inputs_embeds=inputs_embeds)
        return self.lm_head(h[:, -1:, :])  # decode only needs the last position

    # ------------------------------------------------------------------ one hipGraph per generated token
    def capture_decode_graph(self, fused_layers: Optional[bool] = None) -> None:
        """Capture ONE decode step (all layers + lm_head) and keep it for ``decode_graph_step``.  ``fused_layers``
        (default: on for fp16 models unless ``QUEST_FUSED_LAYER=0``): the decoder layers' projections, norms, RoPE,
        activation and residual adds run as 4 fused HIP launches per layer (csrc/decode_layer.hip) instead of ~14
        rocBLAS / PyTorch launches.  Call after
        the prompt has been processed (any length: while the cache holds fewer pages than the budget the
        sparse layers attend all of them, like the reference's full-attention branch).  The
        graph reads its input from ``self.graph_input`` ``[1, 1, hidden]`` and leaves the logits in
        ``self.graph_logits``; sequence lengths live on the device (EXTENSION)."""
        m, ctl = self.model, self.model.iController
        dev = next(self.parameters()).device
        ctl.set_page_budget(m._quest_page_budget)
        ctl.enable_device_state()
        ctl.begin_graph_decode(dense_layers=m._quest_skip_layer > 0)
        self.graph_input = torch.zeros(1, 1, self.config.hidden_size, dtype=self.lm_head.weight.dtype, device=dev)
        self._graph_scores = qutils.score_scratch(ctl)
        if fused_layers is None:
            import os

            fused_layers = os.environ.get("QUEST_FUSED_LAYER", "1") != "0"
        fused_layers = (fused_layers and self.lm_head.weight.dtype == torch.float16
                        and fused_layer_launches_supported(self.config, 1))
        self.fused_layers = fused_layers
        if fused_layers:
            from .. import _kernels

            # (kept on self: the captured launches hold these buffers' addresses)
            ws = self._graph_ws = DecodeWorkspace(self.config, torch.float16, dev)
            logits = torch.empty(1, 1, self.config.vocab_size, dtype=torch.float16, device=dev)

            def step():
                hs = m.forward_decode_dyn_fused(self.graph_input, self._graph_scores, ws)
                _kernels.decode_norm_gemv(hs, m.norm.weight, m.norm.variance_epsilon, self.lm_head.weight, logits)
                return logits
        else:
            def step():
                return self.lm_head(m.forward_decode_dyn(self.graph_input, self._graph_scores))

        # Warm-up (allocator, rocBLAS workspaces) runs a real step on a dummy input: it advances the device
        # state and folds the dummy key into the current page's (max, min) metadata entry.  K/V bytes it wrote
        # are overwritten by the first real token, but the metadata fold is not idempotent -> snapshot the
        # current metadata page of every layer and restore it, then rewind the state.
        meta_page = ctl.metadata_cache.indicies[-1]
        saved = ctl.metadata_cache.pool.buf[:, meta_page].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.inference_mode():
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ctl.metadata_cache.pool.buf[:, meta_page].copy_(saved)
        ctl.sync_device_state()
        self._graph = torch.cuda.CUDAGraph()
        with torch.inference_mode(), torch.cuda.graph(self._graph):
            self.graph_logits = step()
        self._graph_epoch = ctl.state_epoch

    def decode_graph_step(self, inputs_embeds: Optional[torch.Tensor] = None,
                          input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Generate one token's logits by replaying the captured step; returns ``self.graph_logits``."""
        if getattr(self, "_graph", None) is None or self._graph_epoch != self.model.iController.state_epoch:
            raise RuntimeError("no decode graph for the current request: call capture_decode_graph() after the "
                               "prompt (quest_clear() / enable_device_state() invalidate a captured graph)")
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        self.graph_input.copy_(inputs_embeds.view(1, 1, -1))
        self._graph.replay()
        self.model.iController.prepare_metadata(1)  # host mirror of the device-side reservation
        return self.graph_logits


    # ------------------------------------------------------------------ batched serving (EXTENSION)
    def quest_init_batched(self, n_seqs: int, page_size: int, max_seq_len: int, token_budget: int = 512,
                           dtype=torch.float16, device=torch.device("cuda:0"), kv_layout="NHD") -> None:
        """``quest_init`` for ``n_seqs`` sequences decoded together: one shared KV pool and metadata pool
        (``BatchedInferenceController``).  Prompts are processed one sequence at a time with
        ``prefill_sequence``; decode steps then run all sequences in one hipGraph replay."""
        assert self.model.iController is None and getattr(self.model, "bController", None) is None
        cfg = self.config
        self.model._quest_page_budget = token_budget // page_size
        self.model.bController = qutils.BatchedInferenceController(
            n_seqs, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads,
            page_size, self.model._quest_page_budget, max_seq_len, dtype, device,
            num_kv_heads=cfg.num_key_value_heads, layout=qutils.TensorLayout.parse(kv_layout))

    def prefill_sequence(self, seq: int, input_ids: torch.Tensor) -> torch.Tensor:
        """Run the prompt of sequence ``seq`` (``[1, L]``) through the ordinary single-sequence path over the
        shared pools; returns the logits of its last position."""
        m = self.model
        m.iController = m.bController.seqs[seq]
        try:
            return self.forward(input_ids=input_ids)
        finally:
            m.iController = None

    def capture_decode_graph_batched(self, fused_layers: Optional[bool] = None) -> None:
        """Batched ``capture_decode_graph``: input ``self.graph_input`` ``[n, 1, hidden]``, logits
        ``self.graph_logits`` ``[n, 1, vocab]``.  ``fused_layers`` as in ``capture_decode_graph`` (default on for fp16
        models and n <= 16 unless ``QUEST_FUSED_LAYER=0``): 4 fused launches per layer for the whole batch."""
        m, b = self.model, self.model.bController
        dev = next(self.parameters()).device
        b.enable_device_state()
        b.begin_graph_decode(dense_layers=m._quest_skip_layer > 0)
        n = b.n_seqs
        self.graph_input = torch.zeros(n, 1, self.config.hidden_size, dtype=self.lm_head.weight.dtype, device=dev)
        self._graph_scores = qutils.score_scratch(b)
        if fused_layers is None:
            import os

            fused_layers = os.environ.get("QUEST_FUSED_LAYER", "1") != "0"
        from .. import _kernels

        fused_layers = (fused_layers and self.lm_head.weight.dtype == torch.float16 and n <= _kernels.MAX_BATCHED_TOKENS
                        and fused_layer_launches_supported(self.config, n))
        self.fused_layers = fused_layers
        if fused_layers:
            ws = self._graph_ws = DecodeWorkspace(self.config, torch.float16, dev, n)  # (kept: the graph holds its addresses)
            logits = torch.empty(n, 1, self.config.vocab_size, dtype=torch.float16, device=dev)

            def step():
                hs = m.forward_decode_batched_fused(self.graph_input, self._graph_scores, ws)
                _kernels.decode_norm_gemv_batched(hs, m.norm.weight, m.norm.variance_epsilon, self.lm_head.weight, logits)
                return logits
        else:
            def step():
                return self.lm_head(m.forward_decode_batched(self.graph_input, self._graph_scores))

        # warm-up folds a dummy key into every sequence's current metadata entry: snapshot / restore those
        # pages (see capture_decode_graph)
        pages = torch.tensor([c.metadata_cache.indicies[-1] for c in b.seqs], device=dev)
        saved = b.metadata_pool.buf[:, pages].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.inference_mode():
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        b.metadata_pool.buf[:, pages] = saved
        b.sync_device_state()
        self._graph = torch.cuda.CUDAGraph()
        with torch.inference_mode(), torch.cuda.graph(self._graph):
            self.graph_logits = step()
        self._graph_epoch = b.state_epoch

    def decode_graph_step_batched(self, input_ids: torch.Tensor) -> torch.Tensor:
        """One token for every sequence: ``input_ids`` ``[n]`` -> logits ``[n, 1, vocab]`` (the graph's buffer)."""
        if getattr(self, "_graph", None) is None or self._graph_epoch != self.model.bController.state_epoch:
            raise RuntimeError("no decode graph for the current batch: call capture_decode_graph_batched()")
        self.graph_input.copy_(self.model.embed_tokens(input_ids.view(-1, 1)))
        self._graph.replay()
        self.model.bController.prepare_metadata(1)
        return self.graph_logits
