"""Decode attention with the qkv GEMM epilogue + RoPE + KV write in its prologue
(sgl_mi355_decode_attention_qkv_partials) against the two-launch sequence it replaces
(sgl_mi355_rotary_embedding_set_kv_from_partials, then sgl_mi355_decode_attention): both pools bit-identical; the
output within the per-element decode tolerance (1e-3 + 1 ulp of each 16-bit output, tests/conftest.py) of an fp32 evaluation
of the same attention, and within twice that of the two-launch result (the new token enters the softmax as a separate
partial state instead of through the last streamed tile, so the roundings differ).  The two-launch sequence itself is
pinned to the oracle in test_backend_gpu.py / test_decode_gpu.py."""
import pytest
import torch

from conftest import assert_elem_close, p_rounding_term

pytestmark = [pytest.mark.gpu, pytest.mark.optin_fusions]  # (skipped unless the loaded library was built with
# -DSGLM_OPTIN_FUSIONS=1: tests/conftest.py; the default library returns UNSUPPORTED from these entry points)
DEV = "cuda"


def _setup(B, Hq, Hk, D, dtype, lens, with_bias, seed, K=1024, pool_rows=None):
    from sglang_npu_amd import ops
    g = torch.Generator(device=DEV).manual_seed(seed)
    N = (Hq + 2 * Hk) * D
    a = ((torch.rand(B, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(B, 1, device=DEV, generator=g) * 4e-3 + 1e-3
    sb = torch.rand(N, 1, device=DEV, generator=g) * 4e-3 + 1e-3
    bias = torch.randn(N, device=DEV, generator=g).to(dtype) if with_bias else None
    lens = torch.as_tensor(lens, dtype=torch.int64, device=DEV)
    assert lens.numel() == B
    total = int(lens.sum())
    rows = pool_rows or (total + 17)
    perm = torch.randperm(rows - 1, device=DEV, generator=g)[:total] + 1  # row 0 stays unused
    max_ctx = int(lens.max()) + 3
    r2t = torch.zeros(B + 2, max_ctx, dtype=torch.int32, device=DEV)
    rpi = torch.randperm(B + 2, device=DEV, generator=g)[:B].long()
    off = 0
    loc = torch.empty(B, dtype=torch.int64, device=DEV)
    for b in range(B):
        n = int(lens[b])
        r2t[rpi[b], :n] = perm[off:off + n].int()
        loc[b] = perm[off + n - 1]
        off += n
    kb = torch.randn(rows, Hk, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(rows, Hk, D, device=DEV, generator=g).to(dtype)
    pos = (lens - 1).clone()
    cache = torch.randn(int(lens.max()) + 1, D, device=DEV, generator=g)

    def partials():
        p = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, dtype, bias)
        assert p is not None
        return p
    return dict(partials=partials, lens=lens, r2t=r2t, rpi=rpi, loc=loc, kb=kb, vb=vb, pos=pos, cache=cache)


def _two_launch(s, Hq, Hk, D, dtype, scale, cap):
    from sglang_npu_amd import ops
    kb, vb = s["kb"].clone(), s["vb"].clone()
    q = ops.rope_set_kv_from_partials(s["partials"](), s["pos"], Hq, Hk, D, s["cache"], kb, vb, s["loc"], True)
    o = torch.empty(q.shape[0], Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention_paged(q.view(-1, Hq, D), kb, vb, o, s["r2t"], s["rpi"], s["lens"], None, 1, scale, cap)
    return o, kb, vb, q.view(-1, Hq, D)


def _truth_f32(q, kb, vb, s, scale, cap):
    """Plain fp32 softmax attention over each request's page-table slice (q rotated, pools final)."""
    B, Hq, D = q.shape
    Hk = kb.shape[1]
    out = torch.empty(B, Hq, D, dtype=torch.float32, device=DEV)
    for b in range(B):
        n = int(s["lens"][b])
        idx = s["r2t"][s["rpi"][b], :n].long()
        k = kb[idx].float().repeat_interleave(Hq // Hk, dim=1)  # [n, Hq, D]
        v = vb[idx].float().repeat_interleave(Hq // Hk, dim=1)
        sc = torch.einsum("hd,nhd->hn", q[b].float(), k) * scale
        if cap > 0:
            sc = cap * torch.tanh(sc / cap)
        out[b] = torch.einsum("hn,nhd->hd", torch.softmax(sc, dim=-1), v)
    return out


CASES = [
    # B, Hq, Hk, D, dtype, bias, cap, lens builder
    (64, 32, 8, 128, torch.bfloat16, False, 0.0, "mixed"),
    (64, 32, 8, 128, torch.float16, True, 0.0, "mixed"),
    (43, 28, 7, 128, torch.bfloat16, False, 30.0, "mixed"),     # odd number of items: the last workgroup has one
    (40, 16, 8, 128, torch.bfloat16, True, 0.0, "short"),       # group 2; lengths 1..40 (new token in the first tile)
    (48, 24, 6, 64, torch.bfloat16, False, 0.0, "mixed"),       # head size 64
    (33, 128, 8, 128, torch.bfloat16, False, 0.0, "mixed"),     # group 16
    (36, 32, 8, 128, torch.bfloat16, False, 0.0, "one_long"),   # one request beyond the staged page-table window
]


@pytest.mark.parametrize("B,Hq,Hk,D,dtype,with_bias,cap,kind", CASES)
def test_fused_qkv_decode_equals_two_launches(B, Hq, Hk, D, dtype, with_bias, cap, kind):
    from sglang_npu_amd import ops
    gen = torch.Generator().manual_seed(B * 7 + Hq)
    if kind == "mixed":
        lens = torch.randint(1, 600, (B,), generator=gen)
        lens[0], lens[1], lens[2] = 1, 32, 33
    elif kind == "short":
        lens = torch.arange(1, B + 1)
    else:
        lens = torch.randint(1, 300, (B,), generator=gen)
        lens[5] = 9000
    s = _setup(B, Hq, Hk, D, dtype, lens.tolist(), with_bias, seed=B + Hq + D)
    scale = D ** -0.5
    o_ref, kb_ref, vb_ref, q_rot = _two_launch(s, Hq, Hk, D, dtype, scale, cap)
    kb, vb = s["kb"].clone(), s["vb"].clone()
    o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
    done = ops.decode_attention_qkv_partials(s["partials"](), s["pos"], s["cache"], True, s["loc"], kb, vb, o, s["r2t"],
                                             s["rpi"], s["lens"], Hq, scale, cap)
    assert done, "shape is inside the fused form: the call must launch"
    torch.cuda.synchronize()
    assert torch.equal(kb, kb_ref) and torch.equal(vb, vb_ref), "RoPE + KV write must be bit-exact"
    truth = _truth_f32(q_rot, kb_ref, vb_ref, s, scale, cap)
    # the P-rounding allowance of conftest.p_rounding_term: this attention on |V| (short requests: a handful of comparable p_j)
    term = p_rounding_term(dtype, _truth_f32(q_rot, kb_ref, vb_ref.abs(), s, scale, cap))
    assert_elem_close(o, truth, dtype, what="fused launch vs the fp32 truth", extra=term)
    assert_elem_close(o, o_ref, dtype, pair=True, what="fused launch vs the two launches", extra=2 * term)  # both sides round P


def test_fused_qkv_decode_declines_outside_its_form():
    """<= 256 items (the split kernels' territory), gpt-j pairs, an e4m3 pool: False, nothing written."""
    from sglang_npu_amd import ops
    B, Hq, Hk, D = 8, 32, 8, 128
    s = _setup(B, Hq, Hk, D, torch.bfloat16, [50] * B, False, seed=3)
    kb, vb = s["kb"].clone(), s["vb"].clone()
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    args = (s["pos"], s["cache"], True, s["loc"], kb, vb, o, s["r2t"], s["rpi"], s["lens"], Hq, D ** -0.5)
    assert ops.decode_attention_qkv_partials(s["partials"](), *args) is False
    B = 64
    s = _setup(B, Hq, Hk, D, torch.bfloat16, [50] * B, False, seed=4)
    kb, vb = s["kb"].clone(), s["vb"].clone()
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    assert ops.decode_attention_qkv_partials(s["partials"](), s["pos"], s["cache"], False, s["loc"], kb, vb, o, s["r2t"],
                                             s["rpi"], s["lens"], Hq, D ** -0.5) is False
    kb8 = kb.to(torch.float8_e4m3fn)
    assert ops.decode_attention_qkv_partials(s["partials"](), s["pos"], s["cache"], True, s["loc"], kb8, kb8.clone(), o,
                                             s["r2t"], s["rpi"], s["lens"], Hq, D ** -0.5) is False
    torch.cuda.synchronize()
    assert torch.equal(kb, s["kb"]) and torch.equal(vb, s["vb"]) and not o.any()


def test_fused_qkv_decode_rejects_mismatched_gemm():
    from sglang_npu_amd import ops
    B, Hq, Hk, D = 64, 32, 8, 128
    s = _setup(B, Hq, Hk, D, torch.bfloat16, [20] * B, False, seed=5)
    o = torch.zeros(B, Hq + 1, D, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="qkv projection"):
        ops.decode_attention_qkv_partials(s["partials"](), s["pos"], s["cache"], True, s["loc"], s["kb"], s["vb"], o,
                                          s["r2t"], s["rpi"], s["lens"], Hq + 1, D ** -0.5)


def test_model_step_with_the_fused_prologue_matches_the_two_launch_step(monkeypatch):
    """SGL_MI355_QKV_ATTN_FUSION wiring: LlamaAttention.forward_fp8 -> MI355AttnBackend.forward_decode_qkv_partials.
    Layer 0 sees identical inputs, so its pool rows are bit-identical; logits agree within 16-bit rounding noise."""
    from sglang_npu_amd import model as M, ops
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 8, 128, 1024, 2048, 3, 512, 256)
    B = 40  # 40 requests x 8 kv heads = 320 items: the pairs-of-items kernel
    outs, taken = [], []
    real = ops.decode_attention_qkv_partials

    def counted(*a, **kw):
        done = real(*a, **kw)
        taken.append(done)
        return done

    monkeypatch.setattr(ops, "decode_attention_qkv_partials", counted)
    for fuse in (False, True):
        monkeypatch.setattr(M, "FUSE_QKV_ATTN", fuse)
        net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
        net.defer_epilogues = True
        r2t = ReqToTokenPool(B, 256, DEV)
        pool = MHATokenToKVPool(B * 256 + 1, 1, torch.bfloat16, 8, 128, 3, DEV)
        g = torch.Generator(device=DEV).manual_seed(0)
        for l in range(3):
            pool.k_buffer[l].normal_(generator=g)
            pool.v_buffer[l].normal_(generator=g)
        r2t.req_to_token.copy_((torch.randperm(B * 256, device=DEV, generator=g) + 1).view(B, 256).to(torch.int32))
        runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        seq = (torch.arange(B, device=DEV) * 6 + 1).clamp(max=255)
        ids = torch.arange(B, device=DEV) + 1
        rows = torch.arange(B, device=DEV)
        fb = ForwardBatch(ForwardMode.DECODE, B, ids, rows, seq, r2t.req_to_token[rows, seq - 1].long(), int(seq.sum()),
                          seq.cpu(), seq - 1, req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        outs.append((net(ids, seq - 1, fb).float(), pool.k_buffer[0].clone(), pool.v_buffer[0].clone()))
    assert taken == [True] * 3, f"the fused kernel must have run in all three layers, got {taken}"
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), "layer-0 pool rows differ"
    assert torch.isfinite(outs[1][0]).all()
    scale = outs[0][0].abs().max().item()
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 0.02 * scale
