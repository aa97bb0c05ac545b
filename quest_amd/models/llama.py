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


class LlamaForCausalLM(nn.Module):
    def __init__(self, config: LlamaConfig, fused: bool = True):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, fused)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def quest_init(self, page_size: int, max_seq_len: int, token_budget: int = 512, dtype=torch.float16,
                   device=torch.device("cuda:0")) -> None:
        """llama.py:520-552: build the controller; ``token_budget`` is in TOKENS, the controller's page
        budget is ``token_budget // page_size`` pages (llama.py:536)."""
        assert self.model.iController is None, "Can't init Quest Controller twice."
        cfg = self.config
        self.model._quest_page_budget = token_budget // page_size
        self.model.iController = qutils.InferenceController(
            cfg.num_hidden_layers, cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads, page_size,
            self.model._quest_page_budget, max_seq_len, dtype, device, num_kv_heads=cfg.num_key_value_heads)
        print(f"Quest allocates KV-Cache of {max_seq_len} tokens; token_budget {token_budget} = "
              f"{self.model._quest_page_budget} pages of {page_size}")

    def quest_clear(self) -> None:
        """llama.py:554-560: release the pages for the next request."""
        assert self.model.iController is not None, "Must quest_init() before quest_clear()."
        self.model.iController.clean_states()

    def forward(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None):
        h = self.model(input_ids=input_ids, inputs_embeds=inputs_embeds)
        return self.lm_head(h[:, -1:, :])  # decode only needs the last position
