"""Attention module that runs Quest's query-aware sparse decode on MI355X.

Module API of the reference kept (quest/models/QuestAttention.py:15-181): same constructor
``QuestAttention(config, layer_idx)``, same parameter names (``q_proj``/``k_proj``/``v_proj``/``o_proj``, so
HF checkpoints load unchanged), same ``forward(hidden_states, ..., iController)`` returning
``(attn_output, None, past_key_value)``.  ``config`` only needs the ``LlamaConfig`` attributes read
below, so the module has no import-time dependency on transformers.

Per decode token the reference issues RoPE, append, estimate, top-k, attention (+merge) as separate
launches (QuestAttention.py:99-157); here append+estimate and top-k+attention are single launches
(``quest_amd.utils.decode_append_estimate`` / ``decode_topk_sparse_attn``), bit-identical to the pairs.
The projections stay ``nn.Linear`` (rocBLAS/hipBLASLt GEMV) -- they are not part of this path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from .. import utils as qutils


class QuestAttention(nn.Module):
    def __init__(self, config, layer_idx: int, fused: bool = True):
        super().__init__()
        self.layer_idx = layer_idx
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = getattr(config, "num_key_value_heads", self.num_heads)
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = getattr(config, "max_position_embeddings", None)
        self.fused = fused
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.num_heads}).")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        # rope: default theta 1e4; only linear position interpolation is supported, as in the
        # reference kernel path (QuestAttention.py:40-51)
        scaling = getattr(config, "rope_scaling", None)
        if scaling is None:
            self.rope_scale = 1.0
        elif scaling.get("type", scaling.get("rope_type")) == "linear":
            self.rope_scale = float(scaling["factor"])
        else:
            raise ValueError(f"Unknown RoPE scaling type {scaling}")
        self.rope_theta = float(getattr(config, "rope_theta", 1e4))

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value: Optional[Tuple[torch.Tensor]] = None,
                output_attentions: bool = False, use_cache: bool = False,
                iController: Optional[qutils.InferenceController] = None,
                ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Tuple[torch.Tensor]]]:
        bsz, q_len, _ = hidden_states.size()
        assert bsz == 1, "QuestAttention only supports batch size 1."
        assert iController is not None, "QuestAttention requires an InferenceController."
        nvtx = torch.cuda.nvtx  # range names of the reference (roctx on ROCm)

        nvtx.range_push("qkv_proj")
        q = self.q_proj(hidden_states).view(q_len, self.num_heads, self.head_dim)
        k = self.k_proj(hidden_states).view(q_len, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(hidden_states).view(q_len, self.num_key_value_heads, self.head_dim)
        nvtx.range_pop()

        nvtx.range_push("RoPE")
        qutils.apply_rope_in_place(q, k, iController.kv_cache.seqlen - q_len, rope_scale=self.rope_scale,
                                   rope_theta=self.rope_theta)
        nvtx.range_pop()

        if q_len > 1:
            nvtx.range_push("append_kv")
            qutils.append_kv(k, v, iController, self.layer_idx)
            nvtx.range_pop()
            nvtx.range_push("prefill_attn")
            attn = qutils.prefill_forward(q, iController, self.layer_idx)
            nvtx.range_pop()
        elif not iController.need_estimate():
            # budget covers the whole cache (or a layer the model keeps dense): plain paged decode
            nvtx.range_push("append_kv")
            qutils.append_kv(k, v, iController, self.layer_idx)
            nvtx.range_pop()
            nvtx.range_push("full_attn")
            attn = qutils.decode_sparse_attn(q, iController, self.layer_idx, iController.kv_indices_without_last)
            nvtx.range_pop()
        elif self.fused:
            nvtx.range_push("append_kv+estimate")
            scores = qutils.decode_append_estimate(q, k, v, iController, self.layer_idx)
            nvtx.range_pop()
            nvtx.range_push("topk+approx_attn")
            # the selected page ids stay inside the kernel (nothing downstream reads topk_dindices_buffer)
            attn = qutils.decode_topk_sparse_attn(q, scores, iController, self.layer_idx, write_topk=False)
            nvtx.range_pop()
        else:
            nvtx.range_push("append_kv")
            qutils.append_kv(k, v, iController, self.layer_idx)
            nvtx.range_pop()
            nvtx.range_push("estimate")
            scores = qutils.decode_estimate(q, iController, self.layer_idx)
            nvtx.range_pop()
            nvtx.range_push("topk")
            qutils.decode_topk(scores, iController)
            nvtx.range_pop()
            nvtx.range_push("approx_attn")
            attn = qutils.decode_sparse_attn(q, iController, self.layer_idx, iController.topk_dindices_buffer)
            nvtx.range_pop()

        attn = attn.unsqueeze(0)
        if attn.size() != (bsz, q_len, self.num_heads, self.head_dim):
            raise ValueError(f"`attn_output` should be of size {(bsz, q_len, self.num_heads, self.head_dim)}, but is"
                             f" {attn.size()}")
        nvtx.range_push("o_proj")
        out = self.o_proj(attn.reshape(bsz, q_len, self.hidden_size))
        nvtx.range_pop()
        return out, None, past_key_value

    def forward_dyn(self, hidden_states: torch.Tensor, iController: qutils.InferenceController, scores: torch.Tensor,
                    dense: bool) -> torch.Tensor:
        """Decode-token forward whose every length comes from the controller's device-resident step state
        (``enable_device_state`` + ``begin_graph_decode(dense_layers=True)``), so a whole model step can be
        captured in one hipGraph and replayed as the sequence grows.  ``dense``: full-KV layer (the model's
        first layers) vs Quest sparse layer.  EXTENSION of the reference module."""
        bsz, q_len, _ = hidden_states.size()
        assert bsz == 1 and q_len == 1
        q = self.q_proj(hidden_states).view(1, self.num_heads, self.head_dim)
        k = self.k_proj(hidden_states).view(1, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(hidden_states).view(1, self.num_key_value_heads, self.head_dim)
        if dense:
            attn = qutils.decode_layer_dense_dyn(q, k, v, iController, self.layer_idx, self.rope_scale, self.rope_theta,
                                                 apply_rope=True)
        else:
            attn = qutils.decode_layer_dyn(q, k, v, iController, self.layer_idx, scores, self.rope_scale,
                                           self.rope_theta, apply_rope=True)
        return self.o_proj(attn.reshape(1, 1, self.hidden_size))


    def forward_batched(self, hidden_states: torch.Tensor, bController: "qutils.BatchedInferenceController",
                        scores: torch.Tensor, dense: bool) -> torch.Tensor:
        """``forward_dyn`` for ``n`` sequences at once: ``hidden_states`` is ``[n, 1, hidden]`` (one decode token
        per sequence), all sequences share the controller's pools and every op is ONE launch for the batch
        (``quest_amd.utils.decode_layer_batched``).  EXTENSION: the reference module asserts batch size 1."""
        n, q_len, _ = hidden_states.size()
        assert q_len == 1 and n == bController.n_seqs
        q = self.q_proj(hidden_states).view(n, self.num_heads, self.head_dim)
        k = self.k_proj(hidden_states).view(n, self.num_key_value_heads, self.head_dim)
        v = self.v_proj(hidden_states).view(n, self.num_key_value_heads, self.head_dim)
        if dense:
            attn = qutils.decode_layer_dense_batched(q, k, v, bController, self.layer_idx, self.rope_scale,
                                                     self.rope_theta, apply_rope=True)
        else:
            attn = qutils.decode_layer_batched(q, k, v, bController, self.layer_idx, scores, self.rope_scale,
                                               self.rope_theta, apply_rope=True)
        return self.o_proj(attn.reshape(n, 1, self.hidden_size))
