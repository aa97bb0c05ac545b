#!/usr/bin/env python3
"""RoPE fixtures from the SAME third-party oracle the reference's rope test uses.

quest/tests/test_rope.py:17-30 checks the kernel against Hugging Face's ``LlamaRotaryEmbedding`` +
``apply_rotary_pos_emb``; its wrapper ``_ref_apply_qk_rope`` is written for an older transformers API and
raises an ordinary AttributeError under the installed 5.x (see make_golden.py).  This script calls the two HF
functions the reference test imports through the installed API instead -- default rope parameters (theta 1e4,
no scaling), positions ``past .. past + n``, fp16 in/out exactly like the reference wrapper -- on the
reference's own (past_kv_len, seq_len) grid points.  Inputs come from oracle.synth seeds; only outputs are stored.

    python tests/golden/make_rope_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from transformers.models.llama.configuration_llama import LlamaConfig  # noqa: E402
from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding, apply_rotary_pos_emb  # noqa: E402

from oracle.synth import normal_f16  # noqa: E402

CASES = [(13, 2), (24, 19), (77, 69), (244, 5), (502, 31), (1110, 7), (311, 111)]  # subset of test_rope.py:32
H, D = 4, 128


def main():
    cfg = LlamaConfig(hidden_size=H * D, num_attention_heads=H, num_key_value_heads=H, max_position_embeddings=4096)
    rot = LlamaRotaryEmbedding(cfg)
    out, cases = {}, []
    for i, (past, n) in enumerate(CASES):
        seed = 700 + i
        q = torch.from_numpy(normal_f16(seed * 3, (n, H, D)))
        k = torch.from_numpy(normal_f16(seed * 3 + 1, (n, H, D)))
        pos = torch.arange(past, past + n, dtype=torch.long).unsqueeze(0)
        with torch.inference_mode():
            cos, sin = rot(q, pos)  # [1, n, D] in q's dtype
            qr, kr = apply_rotary_pos_emb(q.transpose(0, 1).unsqueeze(0), k.transpose(0, 1).unsqueeze(0), cos, sin)
        out[f"rope_q_{past}_{n}"] = qr.squeeze(0).transpose(0, 1).contiguous().numpy()
        out[f"rope_k_{past}_{n}"] = kr.squeeze(0).transpose(0, 1).contiguous().numpy()
        cases.append((seed, past, n, H))
    out["rope_cases"] = np.array(cases, dtype=np.int64)
    path = os.path.join(HERE, "hf_rope_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
