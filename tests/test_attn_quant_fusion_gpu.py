"""GPU: the per-token FP8 quant of the decode attention output folded into its neighbours --
sgl_mi355_decode_attention_absmax (row absmax by atomic max in the attention epilogue) and
sgl_mi355_fp8_scaled_mm_partials_a16 (the o_proj GEMM quantises while staging) -- against the sequence it replaces:
decode attention -> sgl_per_token_quant_fp8 -> fp8_scaled_mm_partials.  Everything bit-identical."""
import pytest
import torch

from sglang_npu_amd import ops

pytestmark = [pytest.mark.gpu, pytest.mark.optin_fusions]  # (skipped unless the loaded library was built with
# -DSGLM_OPTIN_FUSIONS=1: tests/conftest.py; the default library returns UNSUPPORTED from these entry points)
DEV = "cuda"


def _attn_inputs(B, Hq, Hk, D, dtype, lens, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    lens = torch.as_tensor(lens, dtype=torch.int64, device=DEV)
    total = int(lens.sum())
    rows = total + 9
    perm = torch.randperm(rows - 1, device=DEV, generator=g)[:total] + 1
    r2t = torch.zeros(B, int(lens.max()) + 2, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(B):
        n = int(lens[b])
        r2t[b, :n] = perm[off:off + n].int()
        off += n
    kb = torch.randn(rows, Hk, D, device=DEV, generator=g).to(dtype)
    vb = (torch.randn(rows, Hk, D, device=DEV, generator=g) * 3).to(dtype)
    q = torch.randn(B, Hq, D, device=DEV, generator=g).to(dtype)
    return q, kb, vb, r2t, torch.arange(B, device=DEV), lens


@pytest.mark.parametrize("B,Hq,Hk,D,dtype", [(64, 32, 8, 128, torch.bfloat16), (43, 28, 7, 128, torch.float16),
                                               (48, 24, 6, 64, torch.bfloat16), (36, 128, 8, 128, torch.bfloat16)])
def test_attention_row_absmax_is_exact(B, Hq, Hk, D, dtype):
    gen = torch.Generator().manual_seed(B)
    lens = torch.randint(1, 500, (B,), generator=gen)
    lens[0], lens[1] = 1, 33
    if B == 36:
        lens[5] = 9000  # beyond the staged page-table window
    q, kb, vb, r2t, rpi, lens = _attn_inputs(B, Hq, Hk, D, dtype, lens.tolist(), seed=B + D)
    o_ref = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, lens, None, 1, D ** -0.5, 0.0)
    o = torch.empty_like(o_ref)
    amax = torch.zeros(B, dtype=torch.float32, device=DEV)
    assert ops.decode_attention_paged_absmax(q, kb, vb, o, amax, r2t, rpi, lens, D ** -0.5, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(o, o_ref)
    assert torch.equal(amax, o_ref.float().abs().amax(dim=(1, 2)))
    # a second call on the same (not re-zeroed) buffer keeps the max: the caller owns the zeroing
    amax.fill_(1e9)
    assert ops.decode_attention_paged_absmax(q, kb, vb, o, amax, r2t, rpi, lens, D ** -0.5, 0.0)
    assert bool((amax == 1e9).all())


@pytest.mark.parametrize("B,Hq,Hk,D,dtype", [(64, 32, 8, 128, torch.bfloat16), (43, 28, 7, 128, torch.float16),
                                               (48, 24, 6, 64, torch.bfloat16), (36, 128, 8, 128, torch.bfloat16),
                                               (260, 16, 1, 64, torch.float16), (33, 8, 8, 128, torch.bfloat16)])
def test_attention_quant_in_launch_equals_attention_then_quant(B, Hq, Hk, D, dtype):
    """sgl_mi355_decode_attention_quant: the last workgroup of a request quantises its row.  Odd item counts (the last
    workgroup holds one item), pairs that straddle two requests (odd kv-head counts), one item per request (Hk = 1: both
    items of a workgroup complete a request), empty and beyond-the-window sequences; twice on the same counters."""
    gen = torch.Generator().manual_seed(B + 1)
    lens = torch.randint(1, 500, (B,), generator=gen)
    lens[0], lens[1], lens[3] = 1, 33, 0
    if B == 36:
        lens[5] = 9000  # beyond the staged page-table window
    q, kb, vb, r2t, rpi, lens = _attn_inputs(B, Hq, Hk, D, dtype, lens.tolist(), seed=B + D + 1)
    o_ref = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, lens, None, 1, D ** -0.5, 0.0)
    q_ref = torch.empty(B, Hq * D, dtype=torch.float8_e4m3fn, device=DEV)
    s_ref = torch.empty(B, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(o_ref.view(B, Hq * D), q_ref, s_ref)
    counters = torch.zeros(B + 3, dtype=torch.int32, device=DEV)
    for _ in range(2):
        o = torch.full_like(o_ref, 7.0)
        done = ops.decode_attention_paged_quant(q, kb, vb, o, r2t, rpi, lens, counters, D ** -0.5, 0.0)
        assert done is not False
        oq, os_ = done
        torch.cuda.synchronize()
        assert torch.equal(o, o_ref)
        assert torch.equal(os_, s_ref)
        assert torch.equal(oq.view(torch.uint8), q_ref.view(torch.uint8))
        assert not counters.any()  # left zero for the next launch
    assert float(s_ref[3]) == 0.0 and not oq[3].view(torch.uint8).any()  # the empty sequence: zero row, scale 0


def test_attention_quant_in_launch_declines_small_batches_and_fp8_pools():
    q, kb, vb, r2t, rpi, lens = _attn_inputs(8, 32, 8, 128, torch.bfloat16, [50] * 8, seed=1)
    o = torch.zeros(8, 32, 128, dtype=torch.bfloat16, device=DEV)
    counters = torch.zeros(64, dtype=torch.int32, device=DEV)
    assert ops.decode_attention_paged_quant(q, kb, vb, o, r2t, rpi, lens, counters, 0.1) is False
    q, kb, vb, r2t, rpi, lens = _attn_inputs(64, 32, 8, 128, torch.bfloat16, [50] * 64, seed=2)
    o = torch.zeros(64, 32, 128, dtype=torch.bfloat16, device=DEV)
    kb8 = kb.to(torch.float8_e4m3fn)
    assert ops.decode_attention_paged_quant(q, kb8, kb8.clone(), o, r2t, rpi, lens, counters, 0.1) is False
    torch.cuda.synchronize()
    assert not o.any() and not counters.any()


def test_attention_absmax_declines_small_batches_and_fp8_pools():
    q, kb, vb, r2t, rpi, lens = _attn_inputs(8, 32, 8, 128, torch.bfloat16, [50] * 8, seed=1)
    o = torch.zeros(8, 32, 128, dtype=torch.bfloat16, device=DEV)
    amax = torch.zeros(8, device=DEV)
    assert ops.decode_attention_paged_absmax(q, kb, vb, o, amax, r2t, rpi, lens, 0.1) is False
    q, kb, vb, r2t, rpi, lens = _attn_inputs(64, 32, 8, 128, torch.bfloat16, [50] * 64, seed=2)
    o = torch.zeros(64, 32, 128, dtype=torch.bfloat16, device=DEV)
    amax = torch.zeros(64, device=DEV)
    kb8 = kb.to(torch.float8_e4m3fn)
    assert ops.decode_attention_paged_absmax(q, kb8, kb8.clone(), o, amax, r2t, rpi, lens, 0.1) is False
    torch.cuda.synchronize()
    assert not o.any() and not amax.any()


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64])
@pytest.mark.parametrize("N,K,shuffled", [(4096, 4096, True), (4096, 4096, False), (4096, 1024, True), (1280, 4096, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_a16_equals_quant_then_partials(M, N, K, shuffled, dtype):
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = (torch.randn(M, K, device=DEV, generator=g) * 2.5).to(dtype)
    if M > 2:
        x[2].zero_()  # an all-zero row: scale 0, quantised zeros
    w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sb = torch.rand(N, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    wt = ops.fp8_shuffle_weight(w) if shuffled else w.t()
    xq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
    xs = torch.empty(M, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(x, xq, xs)
    ref = ops.fp8_scaled_mm_partials(xq, wt, xs, sb, dtype)
    amax = x.float().abs().amax(dim=1).contiguous()
    part = ops.fp8_scaled_mm_partials_a16(x, amax, wt, sb, dtype)
    if ref is None or ref.num_slices * 1024 < K:  # no split-K form, or slices longer than one phase: declined
        assert part is None or ref is not None
        if part is None:
            return
    assert part is not None and ref is not None and part.num_slices == ref.num_slices
    ref_out = ref.finalize()
    # the workspace is shared: recompute the a16 form after taking the reference
    part = ops.fp8_scaled_mm_partials_a16(x, amax, wt, sb, dtype)
    assert torch.equal(part.x_scale.view(-1), xs.view(-1))
    assert torch.equal(part.finalize(), ref_out)


def test_attention_then_o_proj_chain_is_bit_identical():
    """The chain as the model runs it: absmax attention -> a16 partials -> add + RMSNorm + quant from the partials."""
    B, Hq, Hk, D = 64, 32, 8, 128
    q, kb, vb, r2t, rpi, lens = _attn_inputs(B, Hq, Hk, D, torch.bfloat16, [300] * B, seed=5)
    g = torch.Generator(device=DEV).manual_seed(9)
    w = ((torch.rand(4096, Hq * D, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sb = torch.rand(4096, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    wt = ops.fp8_shuffle_weight(w)
    wn = (torch.rand(4096, device=DEV, generator=g) + 0.5).bfloat16()
    res0 = torch.randn(B, 4096, device=DEV, generator=g).bfloat16()
    # reference sequence
    o = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o, r2t, rpi, lens, None, 1, D ** -0.5, 0.0)
    a2 = o.view(B, Hq * D)
    aq = torch.empty_like(a2, dtype=torch.float8_e4m3fn)
    a_s = torch.empty(B, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(a2, aq, a_s)
    r1 = res0.clone()
    q_ref, s_ref = ops.rmsnorm_quant_fp8_from_partials(ops.fp8_scaled_mm_partials(aq, wt, a_s, sb, torch.bfloat16), r1, wn, 1e-5)
    q_ref, s_ref = q_ref.clone(), s_ref.clone()
    # fused
    o2 = torch.empty_like(o)
    amax = torch.zeros(B, device=DEV)
    assert ops.decode_attention_paged_absmax(q, kb, vb, o2, amax, r2t, rpi, lens, D ** -0.5, 0.0)
    part = ops.fp8_scaled_mm_partials_a16(o2.view(B, Hq * D), amax, wt, sb, torch.bfloat16)
    assert part is not None
    r2 = res0.clone()
    q2, s2 = ops.rmsnorm_quant_fp8_from_partials(part, r2, wn, 1e-5)
    assert torch.equal(q2.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s2, s_ref) and torch.equal(r1, r2)


def test_model_step_is_bit_identical_with_and_without_the_fusion(monkeypatch):
    """LlamaAttention.forward_fp8 -> MI355AttnBackend.forward_decode_absmax -> RowParallelLinear.forward_a16_partials:
    same logits and pool contents as with the separate quant launch, bit for bit; the fused path really ran."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 8, 128, 1024, 2048, 3, 512, 256)
    B = 40  # 40 requests x 8 kv heads = 320 items: the pairs-of-items kernel
    outs, taken = [], []
    real = ops.fp8_scaled_mm_partials_a16

    def counted(*a, **kw):
        r = real(*a, **kw)
        taken.append(r is not None)
        return r

    monkeypatch.setattr(ops, "fp8_scaled_mm_partials_a16", counted)
    monkeypatch.setattr(M, "FUSE_QKV_ATTN", False)  # (SGL_MI355_QKV_ATTN_FUSION=1 would take the attention first)
    for fuse in (False, True):
        monkeypatch.setattr(M, "FUSE_ATTN_QUANT", fuse)
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
        logits = net(ids, seq - 1, fb)
        logits2 = net(ids, seq - 1, fb)  # a second step on the same buffers: the absmax rows were re-zeroed
        outs.append((logits.clone(), logits2.clone(), [pool.k_buffer[l].clone() for l in range(3)]))
    assert taken == [True] * 6, f"the a16 GEMM must have run in all three layers of both steps, got {taken}"
    assert torch.isfinite(outs[1][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for l in range(3):
        assert torch.equal(outs[0][2][l], outs[1][2][l])


def test_model_step_is_bit_identical_with_the_quant_in_the_attention_launch(monkeypatch):
    """LlamaAttention.forward_fp8 -> MI355AttnBackend.forward_decode(fp8_out=True) with SGL_MI355_DECODE_QUANT_FUSION:
    same logits and pool contents as with the separate quant launch, bit for bit; the fused launch really ran."""
    from sglang_npu_amd import attention_backend as AB
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 8, 128, 1024, 2048, 3, 512, 256)
    B = 40  # 40 requests x 8 kv heads = 320 items: the pairs-of-items kernel
    outs, taken = [], []
    real = ops.decode_attention_paged_quant

    def counted(*a, **kw):
        r = real(*a, **kw)
        taken.append(r is not False)
        return r

    monkeypatch.setattr(ops, "decode_attention_paged_quant", counted)
    monkeypatch.setattr(M, "FUSE_QKV_ATTN", False)  # (SGL_MI355_QKV_ATTN_FUSION=1 would take the attention first)
    monkeypatch.setattr(M, "FUSE_ATTN_QUANT", False)
    for fuse in (False, True):
        monkeypatch.setattr(AB, "FUSE_DECODE_QUANT", fuse)
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
        logits = net(ids, seq - 1, fb)
        logits2 = net(ids, seq - 1, fb)  # a second step: the arrival counters were left zero
        outs.append((logits.clone(), logits2.clone(), [pool.k_buffer[l].clone() for l in range(3)]))
    assert taken == [True] * 6, f"the quantising launch must have run in all three layers of both steps, got {taken}"
    assert torch.isfinite(outs[1][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for l in range(3):
        assert torch.equal(outs[0][2][l], outs[1][2][l])
