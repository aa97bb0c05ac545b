"""Fused decode-token projections of a Llama decoder layer (csrc/decode_layer.hip; EXTENSION for SURVEY 8f-4): each launch
against a plain PyTorch fp32 reference of the same op, and the whole graph-replayed model step against the unfused
module path (nn.Linear + rms_norm_forward + apply_rope_in_place + PyTorch SiLU / adds)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(x, g, eps):
    x = x.float()
    return (x * torch.rsqrt(x.pow(2).mean() + eps) * g.float()).half().float()  # the stand-alone op rounds to fp16


@pytest.mark.parametrize("in_dim,out_dim", [(4096, 4096), (11008, 4096), (1376, 512), (512, 1000), (8, 3)])
def test_gemv_residual_and_norm_gemv_vs_fp32(in_dim, out_dim):
    from quest_amd import _kernels

    g = torch.Generator(device=DEV).manual_seed(in_dim + out_dim)
    x = torch.randn(in_dim, generator=g, device=DEV, dtype=torch.float16)
    w = (torch.randn(out_dim, in_dim, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    h = torch.randn(out_dim, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(in_dim, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    tol = dict(rtol=4e-3, atol=4e-3 * (in_dim ** 0.5) * 0.05 + 2e-3)
    h2 = h.clone()
    _kernels.decode_gemv_residual(x, w, h2)
    torch.testing.assert_close(h2.float(), h.float() + w.float() @ x.float(), **tol)
    out = torch.empty(out_dim, device=DEV, dtype=torch.float16)
    _kernels.decode_norm_gemv(x, None, 0.0, w, out)
    torch.testing.assert_close(out.float(), w.float() @ x.float(), **tol)
    _kernels.decode_norm_gemv(x, gamma, 1e-5, w, out)
    torch.testing.assert_close(out.float(), w.float() @ _rms(x, gamma, 1e-5), **tol)


@pytest.mark.parametrize("hidden,inter", [(4096, 11008), (512, 1376), (256, 40)])
def test_mlp_gate_up_vs_fp32(hidden, inter):
    from quest_amd import _kernels

    g = torch.Generator(device=DEV).manual_seed(hidden)
    h = torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    wg = (torch.randn(inter, hidden, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    wu = (torch.randn(inter, hidden, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    act = torch.empty(inter, device=DEV, dtype=torch.float16)
    _kernels.decode_mlp_gate_up(h, gamma, 1e-6, wg, wu, act)
    n = _rms(h, gamma, 1e-6)
    ref = torch.nn.functional.silu(wg.float() @ n) * (wu.float() @ n)
    torch.testing.assert_close(act.float(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()) + 1e-3)


@pytest.mark.parametrize("Hq,Hkv,D,pos,scale,theta", [(32, 32, 128, 32767, 1.0, 1e4), (8, 2, 128, 123, 4.0, 1e4),
                                                      (4, 4, 64, 7, 1.0, 5e5), (2, 1, 256, 4000, 2.0, 1e4)])
def test_qkv_rope_vs_fp32(Hq, Hkv, D, pos, scale, theta):
    from quest_amd import _kernels

    hidden = Hq * D
    g = torch.Generator(device=DEV).manual_seed(pos)
    h = torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    s = 1.0 / hidden ** 0.5
    wq = (torch.randn(Hq * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    wk = (torch.randn(Hkv * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    wv = (torch.randn(Hkv * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    q = torch.empty(1, Hq, D, device=DEV, dtype=torch.float16)
    k = torch.empty(1, Hkv, D, device=DEV, dtype=torch.float16)
    v = torch.empty(1, Hkv, D, device=DEV, dtype=torch.float16)
    state = torch.tensor([pos + 1, 1, 1, 0, 1, 1, 0, 0], dtype=torch.int32, device=DEV)  # seq_len = pos + 1
    _kernels.decode_qkv_rope(h, gamma, 1e-5, wq, wk, wv, q, k, v, D, scale, theta, state)
    n = _rms(h, gamma, 1e-5)

    def rope(x):  # rotate-half, linear position scaling (decode_page.cuh:644-692)
        x = x.view(-1, D)
        i = torch.arange(D // 2, device=DEV, dtype=torch.float32)
        ang = (pos / scale) * theta ** (-2 * i / D)
        c, sn = torch.cos(ang), torch.sin(ang)
        a, b = x[:, : D // 2], x[:, D // 2:]
        return torch.cat([a * c - b * sn, b * c + a * sn], 1)

    tol = dict(rtol=5e-3, atol=8e-3)
    torch.testing.assert_close(q.float().view(-1, D), rope(wq.float() @ n), **tol)
    torch.testing.assert_close(k.float().view(-1, D), rope(wk.float() @ n), **tol)
    torch.testing.assert_close(v.float().view(-1), wv.float() @ n, **tol)


def _rms_rows(x, g, eps):
    x = x.float()
    return (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * g.float()).half().float()


@pytest.mark.parametrize("n", [1, 3, 8, 16])
@pytest.mark.parametrize("in_dim,out_dim", [(4096, 4096), (11008, 4096), (1376, 512), (512, 1000), (8, 3), (40, 18)])
def test_batched_gemv_residual_and_norm_gemv_vs_fp32(n, in_dim, out_dim):
    """The n-token launches (MFMA row-dots, weights read once for the batch): every token against fp32, including rows
    whose length is not a multiple of the 32-wide k step and row counts that are not a multiple of the 16-row block."""
    from quest_amd import _kernels

    g = torch.Generator(device=DEV).manual_seed(in_dim + out_dim + n)
    x = torch.randn(n, in_dim, generator=g, device=DEV, dtype=torch.float16)
    w = (torch.randn(out_dim, in_dim, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    h = torch.randn(n, out_dim, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(in_dim, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    tol = dict(rtol=4e-3, atol=4e-3 * (in_dim ** 0.5) * 0.05 + 2e-3)
    h2 = h.clone()
    _kernels.decode_gemv_residual_batched(x, w, h2)
    torch.testing.assert_close(h2.float(), h.float() + x.float() @ w.float().T, **tol)
    out = torch.full((n, out_dim), float("nan"), device=DEV, dtype=torch.float16)
    _kernels.decode_norm_gemv_batched(x, None, 0.0, w, out)
    torch.testing.assert_close(out.float(), x.float() @ w.float().T, **tol)
    _kernels.decode_norm_gemv_batched(x, gamma, 1e-5, w, out)
    torch.testing.assert_close(out.float(), _rms_rows(x, gamma, 1e-5) @ w.float().T, **tol)
    # token i of the batch == the batch-1 launch on token i, up to the accumulation order
    one = torch.empty(out_dim, device=DEV, dtype=torch.float16)
    _kernels.decode_norm_gemv(x[n - 1], gamma, 1e-5, w, one)
    torch.testing.assert_close(out[n - 1].float(), one.float(), **tol)


@pytest.mark.parametrize("n,hidden,inter", [(8, 4096, 11008), (5, 512, 1376), (2, 256, 40), (16, 512, 1024)])
def test_batched_mlp_gate_up_vs_fp32(n, hidden, inter):
    from quest_amd import _kernels

    g = torch.Generator(device=DEV).manual_seed(hidden + n)
    h = torch.randn(n, hidden, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    wg = (torch.randn(inter, hidden, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    wu = (torch.randn(inter, hidden, generator=g, device=DEV, dtype=torch.float16) * 0.05).contiguous()
    act = torch.full((n, inter), float("nan"), device=DEV, dtype=torch.float16)
    _kernels.decode_mlp_gate_up_batched(h, gamma, 1e-6, wg, wu, act)
    nrm = _rms_rows(h, gamma, 1e-6)
    ref = torch.nn.functional.silu(nrm @ wg.float().T) * (nrm @ wu.float().T)
    torch.testing.assert_close(act.float(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()) + 1e-3)


@pytest.mark.parametrize("n,Hq,Hkv,D,scale,theta", [(8, 32, 32, 128, 1.0, 1e4), (3, 8, 2, 128, 4.0, 1e4),
                                                   (16, 4, 4, 64, 1.0, 5e5), (2, 2, 1, 256, 2.0, 1e4)])
def test_batched_qkv_rope_vs_fp32(n, Hq, Hkv, D, scale, theta):
    """Every token is rotated at ITS sequence's position (states[i].seq_len - 1)."""
    from quest_amd import _kernels

    hidden = Hq * D
    g = torch.Generator(device=DEV).manual_seed(n * 131 + D)
    h = torch.randn(n, hidden, generator=g, device=DEV, dtype=torch.float16)
    gamma = (1 + 0.1 * torch.randn(hidden, generator=g, device=DEV, dtype=torch.float16)).contiguous()
    s = 1.0 / hidden ** 0.5
    wq = (torch.randn(Hq * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    wk = (torch.randn(Hkv * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    wv = (torch.randn(Hkv * D, hidden, generator=g, device=DEV, dtype=torch.float16) * s).contiguous()
    q = torch.full((n, Hq, D), float("nan"), device=DEV, dtype=torch.float16)
    k = torch.full((n, Hkv, D), float("nan"), device=DEV, dtype=torch.float16)
    v = torch.full((n, Hkv, D), float("nan"), device=DEV, dtype=torch.float16)
    pos = [(7919 * (i + 1)) % 32768 for i in range(n)]
    states = torch.tensor([[p + 1, 1, 1, 0, 1, 1, 0, 0] for p in pos], dtype=torch.int32, device=DEV)
    _kernels.decode_qkv_rope_batched(h, gamma, 1e-5, wq, wk, wv, q, k, v, D, scale, theta, states)
    nrm = _rms_rows(h, gamma, 1e-5)
    i = torch.arange(D // 2, device=DEV, dtype=torch.float32)
    tol = dict(rtol=5e-3, atol=8e-3)
    for t in range(n):
        ang = (pos[t] / scale) * theta ** (-2 * i / D)
        c, sn = torch.cos(ang), torch.sin(ang)

        def rope(x):
            x = x.view(-1, D)
            a, b = x[:, : D // 2], x[:, D // 2:]
            return torch.cat([a * c - b * sn, b * c + a * sn], 1)

        torch.testing.assert_close(q[t].float().view(-1, D), rope(wq.float() @ nrm[t]), **tol)
        torch.testing.assert_close(k[t].float().view(-1, D), rope(wk.float() @ nrm[t]), **tol)
        torch.testing.assert_close(v[t].float().view(-1), wv.float() @ nrm[t], **tol)


@pytest.mark.parametrize("kv_heads,inter,budget_pages", [(4, 1376, 64), (2, 1024, 64), (4, 1376, 6)])
def test_fused_decode_graph_matches_the_unfused_module_path(kv_heads, inter, budget_pages):
    """One hipGraph replay per token with the fused decoder layers vs the same with nn.Linear / rms_norm_forward /
    apply_rope_in_place / PyTorch activation and adds: logits over 24 generated positions.  With a budget that covers the
    cache every layer attends all pages and the two paths differ by fp16 rounding only; with 6 of 21 pages the sparse
    layers' page selection is a discrete function of q, so a rounding-level difference can swap a page at a few
    positions -- there the bulk of the positions must agree and none may be far off."""
    from quest_amd.models.llama import LlamaConfig, LlamaForCausalLM
    import quest_amd.utils as qu

    dev = torch.device(DEV)
    cfg = LlamaConfig(vocab_size=1000, hidden_size=512, intermediate_size=inter, num_hidden_layers=4, num_attention_heads=4,
                      num_key_value_heads=kv_heads, max_position_embeddings=4096)
    torch.manual_seed(3)
    with torch.device(dev):
        ref_model = LlamaForCausalLM(cfg).half()
    for p in ref_model.parameters():
        p.data.normal_(0, 0.05)
    for m in ref_model.modules():
        if hasattr(m, "variance_epsilon"):
            m.weight.data.uniform_(0.8, 1.2)
    ctx, steps = 16 * 20 + 5, 24
    g = torch.Generator(device=dev).manual_seed(4)
    D = 128
    kc = torch.randn(cfg.num_hidden_layers, ctx, kv_heads, D, generator=g, device=dev, dtype=torch.float16)
    vc = torch.randn(cfg.num_hidden_layers, ctx, kv_heads, D, generator=g, device=dev, dtype=torch.float16)
    embs = torch.randn(steps, 1, 1, 512, generator=g, device=dev, dtype=torch.float16) * 0.3
    logits = {}
    for fused in (False, True):
        with torch.device(dev):
            model = LlamaForCausalLM(cfg).half()
        model.load_state_dict(ref_model.state_dict())
        model.quest_init(16, ctx + steps + 64, token_budget=16 * budget_pages)
        ctl = model.model.iController
        ctl.prepare_metadata(ctx)
        ctl.begin_forward(ctx)
        for l in range(cfg.num_hidden_layers):
            qu.append_kv(kc[l], vc[l], ctl, l)
        ctl.end_forward()
        model.capture_decode_graph(fused_layers=fused)
        assert model.fused_layers == fused
        outs = []
        with torch.inference_mode():
            for t in range(steps):
                outs.append(model.decode_graph_step(inputs_embeds=embs[t]).float().clone())
        torch.cuda.synchronize()
        logits[fused] = torch.stack(outs)
        assert ctl.kv_cache.seqlen == ctx + steps
    assert torch.isfinite(logits[True]).all()
    scale = float(logits[False].abs().max())
    err = (logits[True] - logits[False]).abs().flatten(1).max(1).values  # per generated position
    if budget_pages >= 64:
        assert float(err.max()) < 1e-2 * scale, (err.tolist(), scale)
    else:
        assert float(err.median()) < 1e-2 * scale and float(err.max()) < 0.2 * scale, (err.tolist(), scale)
