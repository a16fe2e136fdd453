"""GPU: the drop-in classes (MI355AttnBackend behind the AttentionBackend surface, the FP8/AWQ linear
methods behind LinearMethodBase) driven the way SGLang drives them, checked against the oracle."""
import pytest
import torch

import oracle
from conftest import assert_elem_close, p_rounding_term
from sglang_npu_amd import model as M
from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                    RadixAttention, ReqToTokenPool, ServerArgs, install_attention_backend)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, Hq, Hkv, D, max_len, layers=2, flat=False):
    cfg = ModelConfig(Hq, Hkv, D, Hq * D, 4 * Hq * D, layers, 1000, max_len)
    r2t = ReqToTokenPool(B + 2, max_len, DEV)
    n_tok = (B + 2) * max_len + 1
    pool = MHATokenToKVPool(n_tok, 1, torch.bfloat16, Hkv, D, layers, DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    r2t.req_to_token.copy_((torch.randperm(n_tok - 1, device=DEV, generator=g) + 1)[: (B + 2) * max_len]
                           .view(B + 2, max_len).to(torch.int32))
    runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
    from sglang_npu_amd.attention_backend import MI355AttnBackend
    backend = MI355AttnBackend(runner, flat_kv_indices=flat)
    runner.attn_backend = backend
    return cfg, r2t, pool, runner, backend


@pytest.mark.parametrize("flat", [False, True])
def test_backend_prefill_then_decode_matches_oracle(flat):
    """EXTEND (ragged, with a cached prefix for one request) then 3 DECODE steps through the
    AttentionBackend API; K/V land in the pool through the backend; every output is checked against the
    oracle run on a CPU copy of the same pool."""
    B, Hq, Hkv, D, max_len = 3, 32, 8, 128, 400
    cfg, r2t, pool, runner, backend = _setup(B, Hq, Hkv, D, max_len, flat=flat)
    layer = RadixAttention(Hq, D, D ** -0.5, Hkv, layer_id=1)
    g = torch.Generator(device=DEV).manual_seed(1)
    rpi = torch.tensor([2, 0, 4], device=DEV)
    prefix = torch.tensor([0, 37, 0], device=DEV)
    ext = torch.tensor([150, 80, 1], device=DEV)
    seq = prefix + ext
    # pre-populate the cached prefix of request 1
    pool.k_buffer[1].normal_(generator=g)
    pool.v_buffer[1].normal_(generator=g)
    T = int(ext.sum())
    q = torch.randn(T, Hq * D, device=DEV, generator=g).bfloat16()
    k = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    v = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    start = torch.zeros(B, dtype=torch.int64, device=DEV)
    start[1:] = torch.cumsum(ext[:-1], 0)
    loc = torch.cat([r2t.req_to_token[rpi[b], prefix[b]:seq[b]] for b in range(B)]).long()
    fb = ForwardBatch(ForwardMode.EXTEND, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), None,
                      extend_num_tokens=T, extend_seq_lens=ext, extend_prefix_lens=prefix, extend_start_loc=start,
                      extend_prefix_lens_cpu=prefix.tolist(), extend_seq_lens_cpu=ext.tolist(),
                      req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    kb_cpu, vb_cpu = pool.k_buffer[1].cpu().clone(), pool.v_buffer[1].cpu().clone()
    backend.init_forward_metadata(fb)
    o = layer(q, k, v, fb)
    assert o.shape == (T, Hq * D)
    # oracle: write then attend
    kb_cpu[loc.cpu()] = k.cpu().view(T, Hkv, D)
    vb_cpu[loc.cpu()] = v.cpu().view(T, Hkv, D)
    assert torch.equal(pool.k_buffer[1].cpu().view(torch.int16), kb_cpu.view(torch.int16)), "backend must write K/V"
    o_ref = torch.zeros(T, Hq, D, dtype=torch.bfloat16)
    oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D), o_ref, kb_cpu,
                            vb_cpu, r2t.req_to_token.cpu(), rpi.cpu(), seq.cpu(), ext.cpu(), start.cpu(), int(ext.max()),
                            D ** -0.5, 0.0)
    a = torch.zeros_like(o_ref)  # the attention of |V|: the per-element P-rounding allowance (conftest.p_rounding_term)
    oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D).abs(), a, kb_cpu,
                            vb_cpu.abs(), r2t.req_to_token.cpu(), rpi.cpu(), seq.cpu(), ext.cpu(), start.cpu(), int(ext.max()),
                            D ** -0.5, 0.0, p_round=False)
    assert_elem_close(o.view(T, Hq, D), o_ref, torch.bfloat16, pair=True, what="backend extend vs oracle",
                      extra=p_rounding_term(torch.bfloat16, a, pair=True))
    # decode steps
    for step in range(3):
        seq = seq + 1
        loc = r2t.req_to_token[rpi, seq - 1].long()
        qd = torch.randn(B, Hq * D, device=DEV, generator=g).bfloat16()
        kd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
        vd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
        fb = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), seq - 1,
                          req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        od = layer(qd, kd, vd, fb)
        od_ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
        oracle.decode_attention(qd.cpu().view(B, Hq, D), kb_cpu, vb_cpu, od_ref, kd.cpu().view(B, Hkv, D),
                                vd.cpu().view(B, Hkv, D), loc.cpu(), torch.zeros(B, Hq, 2, D + 1), r2t.req_to_token.cpu(),
                                rpi.cpu(), seq.cpu(), D ** -0.5, 0.0, p_round=True)
        assert torch.equal(pool.k_buffer[1].cpu().view(torch.int16), kb_cpu.view(torch.int16))
        from test_decode_gpu import decode_p_term  # (short ragged sequences: the per-element P-rounding allowance)
        term = decode_p_term(qd.view(B, Hq, D), kb_cpu, vb_cpu, r2t.req_to_token, rpi, seq, D ** -0.5, 0.0, torch.bfloat16)
        assert_elem_close(od.view(B, Hq, D), od_ref, torch.bfloat16, pair=True, what=f"backend decode step {step} vs oracle",
                          extra=term)


def test_backend_sliding_window_layers_use_window_indices():
    """A model with sliding_window_size: layers that carry it attend the last W+1 cached tokens only
    (triton_backend.py:187-203, 301-311, 656-665, 711-713; window built by update_sliding_window_buffer :927-955),
    other layers the full context.  Checked against the oracle run on a page table holding just the window."""
    B, Hq, Hkv, D, max_len, W = 3, 8, 2, 128, 300, 31
    cfg, r2t, pool, runner, _ = _setup(B, Hq, Hkv, D, max_len)
    runner.sliding_window_size = W
    from sglang_npu_amd.attention_backend import MI355AttnBackend
    backend = MI355AttnBackend(runner)
    win_layer = RadixAttention(Hq, D, D ** -0.5, Hkv, layer_id=0, sliding_window_size=W)
    full_layer = RadixAttention(Hq, D, D ** -0.5, Hkv, layer_id=1)
    g = torch.Generator(device=DEV).manual_seed(2)
    for l in range(2):
        pool.k_buffer[l].normal_(generator=g)
        pool.v_buffer[l].normal_(generator=g)
    rpi = torch.tensor([1, 3, 0], device=DEV)
    # ---- decode: seq_lens straddle the window length
    seq = torch.tensor([200, 20, W + 1], device=DEV)
    loc = r2t.req_to_token[rpi, seq - 1].long()
    fb = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), seq - 1,
                      req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    qd = torch.randn(B, Hq * D, device=DEV, generator=g).bfloat16()
    kd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    vd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    for layer, windowed in ((win_layer, True), (full_layer, False)):
        od = layer(qd, kd, vd, fb)
        lid = layer.layer_id
        kb_cpu, vb_cpu = pool.k_buffer[lid].cpu(), pool.v_buffer[lid].cpu()
        lens = torch.clamp(seq, max=W + 1) if windowed else seq
        tab = torch.zeros(B, max_len, dtype=torch.int32)
        for b in range(B):
            n, L = int(lens[b]), int(seq[b])
            tab[b, :n] = r2t.req_to_token[rpi[b], L - n:L].cpu()
        od_ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
        oracle.decode_attention(qd.cpu().view(B, Hq, D), kb_cpu, vb_cpu, od_ref, None, None, None,
                                torch.zeros(B, Hq, 2, D + 1), tab, torch.arange(B), lens.cpu(), D ** -0.5, 0.0, p_round=True)
        assert_elem_close(od.view(B, Hq, D), od_ref, torch.bfloat16, pair=True, what="windowed decode vs oracle")
    # ---- extend with a cached prefix longer than the window: prefix part = last W+1 cached tokens
    prefix = torch.tensor([100, 0, 10], device=DEV)
    ext = torch.tensor([40, 25, 7], device=DEV)
    seq = prefix + ext
    T = int(ext.sum())
    q = torch.randn(T, Hq * D, device=DEV, generator=g).bfloat16()
    k = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    v = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    start = torch.zeros(B, dtype=torch.int64, device=DEV)
    start[1:] = torch.cumsum(ext[:-1], 0)
    loc = torch.cat([r2t.req_to_token[rpi[b], prefix[b]:seq[b]] for b in range(B)]).long()
    fb = ForwardBatch(ForwardMode.EXTEND, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), None,
                      extend_num_tokens=T, extend_seq_lens=ext, extend_prefix_lens=prefix, extend_start_loc=start,
                      extend_prefix_lens_cpu=prefix.tolist(), extend_seq_lens_cpu=ext.tolist(),
                      req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    o = win_layer(q, k, v, fb)
    kb_cpu, vb_cpu = pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu()
    wp = torch.clamp(prefix, max=W + 1).cpu()
    tab = torch.zeros(B, max_len, dtype=torch.int32)
    for b in range(B):
        n, P, L = int(wp[b]), int(prefix[b]), int(seq[b])
        tab[b, :n] = r2t.req_to_token[rpi[b], P - n:P].cpu()
        tab[b, n:n + int(ext[b])] = r2t.req_to_token[rpi[b], P:L].cpu()
    o_ref = torch.zeros(T, Hq, D, dtype=torch.bfloat16)
    oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D), o_ref, kb_cpu,
                            vb_cpu, tab, torch.arange(B), wp + ext.cpu(), ext.cpu(), start.cpu(), int(ext.max()),
                            D ** -0.5, 0.0)
    a = torch.zeros_like(o_ref)
    oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D).abs(), a, kb_cpu,
                            vb_cpu.abs(), tab, torch.arange(B), wp + ext.cpu(), ext.cpu(), start.cpu(), int(ext.max()),
                            D ** -0.5, 0.0, p_round=False)
    assert_elem_close(o.view(T, Hq, D), o_ref, torch.bfloat16, pair=True, what="windowed extend vs oracle",
                      extra=p_rounding_term(torch.bfloat16, a, pair=True))


def test_idle_mode_returns_empty_like_reference():
    cfg, r2t, pool, runner, backend = _setup(2, 8, 2, 64, 64)
    layer = RadixAttention(8, 64, 0.125, 2, 0)
    fb = ForwardBatch(ForwardMode.IDLE, 0, None, None, None, None, 0, attn_backend=backend)
    out = layer(torch.zeros(0, 8 * 64, dtype=torch.bfloat16, device=DEV), None, None, fb)
    assert out.shape == (0, 8 * 64)  # base_attn_backend.py:67-68


def test_decode_under_hip_graph_replay_with_changing_lengths():
    """Capture one decode step (backend graph hooks), then replay it with different seq_lens / page-table
    rows written into the static inputs, as cuda_graph_runner.py:773 does; compare with eager."""
    B, Hq, Hkv, D, max_len = 8, 8, 1, 128, 700  # 70B-TP8 geometry: few workgroups -> kv-splits are used
    cfg, r2t, pool, runner, backend = _setup(B, Hq, Hkv, D, max_len, layers=1)
    layer = RadixAttention(Hq, D, D ** -0.5, Hkv, 0)
    g = torch.Generator(device=DEV).manual_seed(2)
    pool.k_buffer[0].normal_(generator=g)
    pool.v_buffer[0].normal_(generator=g)
    rpi = torch.arange(B, device=DEV)
    seq = torch.full((B,), backend.get_cuda_graph_seq_len_fill_value(), dtype=torch.int64, device=DEV)
    loc = torch.zeros(B, dtype=torch.int64, device=DEV)
    q = torch.randn(B, Hq * D, device=DEV, generator=g).bfloat16()
    k = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    v = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    out = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=DEV)
    fb = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq, loc, B, None, None, req_to_token_pool=r2t,
                      token_to_kv_pool=pool, attn_backend=backend)
    backend.init_cuda_graph_state(B, B)
    backend.init_forward_metadata_capture_cuda_graph(B, B, rpi, seq, None, ForwardMode.DECODE, None)
    assert backend.forward_metadata.num_kv_splits > 1
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out.copy_(layer(q, k, v, fb))
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.copy_(layer(q, k, v, fb))
    for trial in range(3):
        new_seq = torch.randint(1, max_len, (B,), device=DEV, generator=g)
        seq.copy_(new_seq)
        loc.copy_(r2t.req_to_token[rpi, seq - 1].long())
        q.normal_(generator=g)
        backend.init_forward_metadata_replay_cuda_graph(B, rpi, seq, int(seq.sum()), None, ForwardMode.DECODE, None, None)
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        fb2 = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq.clone(), loc.clone(), int(seq.sum()), seq.cpu(), None,
                           req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb2)
        eager = layer(q, k, v, fb2)
        # the replayed graph uses the capture's kv-split count, the eager pass the one its lengths ask for: the 16-bit
        # probabilities are rounded against different running maxima (per-element allowance: conftest.p_rounding_term)
        from test_decode_gpu import decode_p_term
        term = decode_p_term(q.view(B, Hq, D), pool.k_buffer[0], pool.v_buffer[0], r2t.req_to_token, rpi, seq, D ** -0.5, 0.0,
                             torch.bfloat16)
        assert_elem_close(got.view(B, Hq, D), eager.view(B, Hq, D), torch.bfloat16, pair=True, what="graph replay vs eager",
                          extra=term)


@pytest.mark.parametrize("quant", ["w8a8_fp8", "awq"])
def test_tiny_model_decode_step_runs_and_is_deterministic(quant):
    """End-to-end plumbing of the synthetic model (linear methods + norms + rope + backend): two identical
    decode steps give bit-identical logits and finite values."""
    cfg = ModelConfig(8, 2, 64, 512, 1024, 2, 512, 128)
    dtype = torch.float16 if quant == "awq" else torch.bfloat16
    net = M.LlamaForCausalLM(cfg, quant, dtype, DEV).load_dummy_weights()
    B = 4
    r2t = ReqToTokenPool(B, 128, DEV)
    pool = MHATokenToKVPool(B * 128 + 1, 1, dtype, 2, 64, 2, DEV)
    r2t.req_to_token.copy_((torch.arange(B * 128, device=DEV) + 1).view(B, 128).to(torch.int32))
    runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
    backend = install_attention_backend(runner)
    seq = torch.tensor([5, 17, 64, 100], device=DEV)
    ids = torch.tensor([1, 2, 3, 4], device=DEV)
    fb = ForwardBatch(ForwardMode.DECODE, B, ids, torch.arange(B, device=DEV), seq,
                      r2t.req_to_token[torch.arange(B, device=DEV), seq - 1].long(), int(seq.sum()), seq.cpu(), seq - 1,
                      req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    l1 = net(ids, seq - 1, fb)
    l2 = net(ids, seq - 1, fb)
    assert l1.shape == (B, 512) and torch.isfinite(l1.float()).all()
    assert torch.equal(l1, l2)


def test_fused_fp8_layer_path_equals_unfused():
    """The fused-producer decode path (norm+quant, silu+quant, rope+KV-write) must give the same logits as the
    reference-shaped unfused sequence (bit-identical FP8 activations -> identical GEMM inputs)."""
    cfg = ModelConfig(8, 2, 64, 512, 1024, 2, 512, 128)
    outs = []
    for fuse in (False, True):
        net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV, fuse_quant=fuse).load_dummy_weights()
        B = 4
        r2t = ReqToTokenPool(B, 128, DEV)
        pool = MHATokenToKVPool(B * 128 + 1, 1, torch.bfloat16, 2, 64, 2, DEV)
        g = torch.Generator(device=DEV).manual_seed(0)
        for l in range(2):
            pool.k_buffer[l].normal_(generator=g)
            pool.v_buffer[l].normal_(generator=g)
        r2t.req_to_token.copy_((torch.arange(B * 128, device=DEV) + 1).view(B, 128).to(torch.int32))
        runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        seq = torch.tensor([5, 17, 64, 100], device=DEV)
        ids = torch.tensor([1, 2, 3, 4], device=DEV)
        fb = ForwardBatch(ForwardMode.DECODE, B, ids, torch.arange(B, device=DEV), seq,
                          r2t.req_to_token[torch.arange(B, device=DEV), seq - 1].long(), int(seq.sum()), seq.cpu(), seq - 1,
                          req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        outs.append((net(ids, seq - 1, fb), pool.k_buffer[1].clone(), pool.v_buffer[0].clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), "KV pool contents must match"
    assert torch.equal(outs[0][0], outs[1][0]), "fused and unfused paths must agree bit for bit"


def test_fused_fp8_path_equals_unfused_in_prefill():
    """The same fused producers run in EXTEND mode (ragged prefill with a cached prefix): logits and the written K/V
    must equal the unfused sequence bit for bit."""
    cfg = ModelConfig(8, 2, 64, 512, 1024, 2, 512, 256)
    outs = []
    for fuse in (False, True):
        net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV, fuse_quant=fuse).load_dummy_weights()
        B = 3
        r2t = ReqToTokenPool(B, 256, DEV)
        pool = MHATokenToKVPool(B * 256 + 1, 1, torch.bfloat16, 2, 64, 2, DEV)
        g = torch.Generator(device=DEV).manual_seed(0)
        for l in range(2):
            pool.k_buffer[l].normal_(generator=g)
            pool.v_buffer[l].normal_(generator=g)
        r2t.req_to_token.copy_((torch.randperm(B * 256, device=DEV, generator=g) + 1).view(B, 256).to(torch.int32))
        runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        prefix = torch.tensor([0, 20, 3], device=DEV)
        ext = torch.tensor([70, 33, 1], device=DEV)
        seq = prefix + ext
        T = int(ext.sum())
        rows = torch.arange(B, device=DEV)
        ids = torch.randint(0, 500, (T,), device=DEV, generator=g)
        positions = torch.cat([torch.arange(int(prefix[b]), int(seq[b]), device=DEV) for b in range(B)])
        loc = torch.cat([r2t.req_to_token[b, prefix[b]:seq[b]] for b in range(B)]).long()
        start = torch.zeros(B, dtype=torch.int64, device=DEV)
        start[1:] = torch.cumsum(ext[:-1], 0)
        fb = ForwardBatch(ForwardMode.EXTEND, B, ids, rows, seq, loc, int(seq.sum()), seq.cpu(), positions,
                          extend_num_tokens=T, extend_seq_lens=ext, extend_prefix_lens=prefix, extend_start_loc=start,
                          extend_prefix_lens_cpu=prefix.tolist(), extend_seq_lens_cpu=ext.tolist(),
                          req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        outs.append((net(ids, positions, fb), pool.k_buffer[1].clone(), pool.v_buffer[0].clone()))
    assert outs[0][0].shape == (3, 512)  # one row of logits per request (LogitsProcessor pruning)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_deferred_gemm_epilogues_are_bit_identical(monkeypatch):
    """defer=True leaves the qkv / o_proj / down_proj epilogues to the consumer kernels (RoPE + KV write, add + RMSNorm +
    quant) on split-K partials.  Same slice order, same roundings -> logits and pool contents identical to the
    fused-producer path without deferral, bit for bit."""
    cfg = ModelConfig(8, 2, 128, 1024, 2048, 3, 512, 256)  # K = 1024 / 2048: on the split-K weight-streaming path
    outs = []
    monkeypatch.setattr(M, "DEFER_MIN_ROWS", 0)  # the model only defers above 32 rows (a speed policy); test it at 5
    for defer in (False, True):
        net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
        net.defer_epilogues = defer
        B = 5
        r2t = ReqToTokenPool(B, 256, DEV)
        pool = MHATokenToKVPool(B * 256 + 1, 1, torch.bfloat16, 2, 128, 3, DEV)
        g = torch.Generator(device=DEV).manual_seed(0)
        for l in range(3):
            pool.k_buffer[l].normal_(generator=g)
            pool.v_buffer[l].normal_(generator=g)
        r2t.req_to_token.copy_((torch.randperm(B * 256, device=DEV, generator=g) + 1).view(B, 256).to(torch.int32))
        runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        seq = torch.tensor([5, 17, 64, 100, 255], device=DEV)
        ids = torch.tensor([1, 2, 3, 4, 5], device=DEV)
        rows = torch.arange(B, device=DEV)
        fb = ForwardBatch(ForwardMode.DECODE, B, ids, rows, seq, r2t.req_to_token[rows, seq - 1].long(), int(seq.sum()),
                          seq.cpu(), seq - 1, req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        outs.append((net(ids, seq - 1, fb), [pool.k_buffer[l].clone() for l in range(3)],
                     [pool.v_buffer[l].clone() for l in range(3)]))
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0]), "logits differ"
    for l in range(3):
        assert torch.equal(outs[0][1][l], outs[1][1][l]) and torch.equal(outs[0][2][l], outs[1][2][l]), f"pool layer {l}"


@pytest.mark.parametrize("M", [1, 16, 33, 64, 100, 128])  # (65..128: the streamer's 128-row phases)
@pytest.mark.parametrize("with_bias", [False, True])
def test_from_partials_ops_equal_unfused_sequence(M, with_bias):
    """sgl_mi355_fp8_scaled_mm_partials + {finalize, rmsnorm_quant_fp8_from_partials, silu_and_mul_quant_fp8_from_partials,
    rope_set_kv_from_partials}
    against fp8_scaled_mm followed by the unfused op: bit-identical."""
    from sglang_npu_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + int(with_bias))
    K, H = 2048, 1024
    a = ((torch.rand(M, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    w = ((torch.rand(H, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    sb = torch.rand(H, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    bias = torch.randn(H, device=DEV, generator=g).bfloat16() if with_bias else None
    full = ops.fp8_scaled_mm(a, w.t(), sa, sb, torch.bfloat16, bias)
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    assert part is not None and part.num_slices >= 1
    assert torch.equal(part.finalize(), full)
    # add + rmsnorm + quant
    wn = (torch.rand(H, device=DEV, generator=g) + 0.5).bfloat16()
    res0 = torch.randn(M, H, device=DEV, generator=g).bfloat16()
    r1, r2 = res0.clone(), res0.clone()
    q_ref, s_ref, _ = ops.rmsnorm_quant_fp8(full.clone(), wn, 1e-5, residual=r1)
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    q, s = ops.rmsnorm_quant_fp8_from_partials(part, r2, wn, 1e-5)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s, s_ref) and torch.equal(r1, r2)
    # SiLU * mul + quant: treat the H = 1024 outputs as [gate | up] with d = 512
    q_ref, s_ref = ops.silu_and_mul_quant_fp8(full)
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    q, s = ops.silu_and_mul_quant_fp8_from_partials(part)
    assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s, s_ref)
    # RoPE + KV write: treat the H = 1024 outputs as qkv of Hq = 4, Hk = 2, D = 128
    Hq, Hk, D = 4, 2, 128
    pos = torch.randint(0, 500, (M,), device=DEV, generator=g)
    cache = torch.randn(512, D, device=DEV, generator=g)
    loc = (torch.randperm(200, device=DEV, generator=g)[:M] + 1).long()
    kb1, vb1 = torch.zeros(201, Hk, D, dtype=torch.bfloat16, device=DEV), torch.zeros(201, Hk, D, dtype=torch.bfloat16, device=DEV)
    kb2, vb2 = torch.zeros_like(kb1), torch.zeros_like(vb1)
    qkv = full.clone()
    q1, k1, v1 = qkv.split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.apply_rope_and_set_kv_buffer(pos, q1, k1, v1, D, cache, kb1, vb1, loc, True)
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    q2 = ops.rope_set_kv_from_partials(part, pos, Hq, Hk, D, cache, kb2, vb2, loc, True)
    assert torch.equal(q2, q1) and torch.equal(kb1, kb2) and torch.equal(vb1, vb2)
    # the same two fused writes into an FP8 (e4m3) pool == 16-bit result cast by set_kv_buffer_fp8: bit-identical
    kb8_ref, vb8_ref = torch.zeros(201, Hk, D, dtype=torch.uint8, device=DEV), torch.zeros(201, Hk, D, dtype=torch.uint8, device=DEV)
    ops.set_kv_buffer_fp8(kb8_ref, vb8_ref, loc, k1.reshape(M, Hk, D), v1.reshape(M, Hk, D))  # k1 is rotated by now
    kb8, vb8 = torch.zeros_like(kb8_ref), torch.zeros_like(vb8_ref)
    qkv = full.clone()
    q3, k3, v3 = qkv.split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.apply_rope_and_set_kv_buffer(pos, q3, k3, v3, D, cache, kb8, vb8, loc, True)
    assert torch.equal(q3, q1) and torch.equal(kb8, kb8_ref) and torch.equal(vb8, vb8_ref)
    kb8.zero_(), vb8.zero_()
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    q4 = ops.rope_set_kv_from_partials(part, pos, Hq, Hk, D, cache, kb8.view(torch.float8_e4m3fn), vb8.view(torch.float8_e4m3fn), loc, True)
    assert torch.equal(q4, q1) and torch.equal(kb8, kb8_ref) and torch.equal(vb8, vb8_ref)


def test_prefill_graph_runner_matches_the_eager_prefill():
    """harness.PrefillGraphRunner (round 4): a short single-request prefill replayed from a HIP graph captured at a bucketed
    token count.  n == bucket: logits and the KV-pool rows bit-identical to the eager pass; n < bucket (padded with token 0,
    padded rows' K/V into slot 0, the attention reading n from the device-side qo_indptr): the same greedy token, logits within
    the 16-bit tolerance (the GEMMs tile 128 rows differently from n rows), the request's pool rows bit-identical, nothing
    written outside them but the padding slot; a second prompt through the same graph."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        PrefillGraphRunner, ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 2, 128, 1024, 2048, 2, 512, 1024)
    net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
    n_tok = 601
    r2t_pool = ReqToTokenPool(2, 300, DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    r2t_pool.req_to_token.copy_((torch.randperm(n_tok - 1, device=DEV, generator=g) + 1)[:600].view(2, 300).to(torch.int32))

    def fresh_pool():
        return MHATokenToKVPool(n_tok, 1, torch.bfloat16, 2, 128, 2, DEV)

    def eager(ids, pool):
        n = ids.numel()
        runner = ModelRunnerLike(cfg, r2t_pool, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        seq = torch.full((1,), n, dtype=torch.int64, device=DEV)
        zero = torch.zeros(1, dtype=torch.int64, device=DEV)
        fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, zero.clone(), seq, r2t_pool.req_to_token[0, :n].to(torch.int64), n,
                          seq.cpu(), torch.arange(n, device=DEV), extend_num_tokens=n, extend_seq_lens=seq.clone(),
                          extend_prefix_lens=zero, extend_start_loc=zero.clone(), extend_prefix_lens_cpu=[0],
                          extend_seq_lens_cpu=[n], req_to_token_pool=r2t_pool, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        return net(ids, torch.arange(n, device=DEV), fb).clone()

    pool_g = fresh_pool()
    runner = ModelRunnerLike(cfg, r2t_pool, pool_g, DEV, 0, 1, ServerArgs())
    backend = install_attention_backend(runner)
    pg = PrefillGraphRunner(net, runner, backend, DEV, buckets=(128, 256))
    for n in (128, 77, 128, 200):
        ids = torch.randint(0, 512, (n,), device=DEV, generator=g)
        pool_e = fresh_pool()
        ref = eager(ids, pool_e)
        for b in pool_g.k_buffer + pool_g.v_buffer:
            b.zero_()
        slots = r2t_pool.req_to_token[0, :n].to(torch.int64)
        logits, tok = pg.run(ids, slots)
        torch.cuda.synchronize()
        assert int(tok) == int(ref.float().argmax())
        if n in pg.buckets:
            assert torch.equal(logits, ref)
        else:
            torch.testing.assert_close(logits.float(), ref.float(), rtol=2.0 ** -6, atol=1e-3 * float(ref.float().abs().max()) + 1e-3)
        for l in range(2):
            ke, kg = pool_e.get_key_buffer(l), pool_g.get_key_buffer(l)
            ve, vg = pool_e.get_value_buffer(l), pool_g.get_value_buffer(l)
            if l == 0 or n in pg.buckets:  # (layer 1's K/V come from layer 0's output: equal only where the GEMMs tile alike)
                assert torch.equal(kg[slots], ke[slots]) and torch.equal(vg[slots], ve[slots])
            untouched = torch.ones(kg.shape[0], dtype=torch.bool, device=DEV)
            untouched[slots] = False
            untouched[0] = False
            assert not bool(kg[untouched].any()) and not bool(vg[untouched].any())


@pytest.mark.parametrize("n", [33, 100, 128])
def test_short_prefill_with_deferred_epilogues_is_bit_identical(n, monkeypatch):
    """Round 4: an extend pass of 33 ... 128 new tokens leaves the o_proj / down_proj GEMM epilogues to the next norm
    (`*_from_partials` consumers, as the decode step does): the same logits and pool rows, bit for bit, as with the
    finalize launches (SGL_MI355_DEFER_EXTEND_MAX_ROWS=0)."""
    from sglang_npu_amd import model as M, ops
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 2, 128, 1024, 2048, 2, 512, 1024)
    net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
    g = torch.Generator(device=DEV).manual_seed(n)
    r2t_pool = ReqToTokenPool(1, 300, DEV)
    r2t_pool.req_to_token.copy_((torch.randperm(400, device=DEV, generator=g) + 1)[:300].view(1, 300).to(torch.int32))
    ids = torch.randint(0, 512, (n,), device=DEV, generator=g)
    outs = []
    calls = []
    real = ops.rmsnorm_quant_fp8_from_partials
    monkeypatch.setattr(ops, "rmsnorm_quant_fp8_from_partials", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for max_rows in (0, 128):
        monkeypatch.setattr(M, "DEFER_EXTEND_MAX_ROWS", max_rows)
        pool = MHATokenToKVPool(401, 1, torch.bfloat16, 2, 128, 2, DEV)
        runner = ModelRunnerLike(cfg, r2t_pool, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        seq = torch.full((1,), n, dtype=torch.int64, device=DEV)
        zero = torch.zeros(1, dtype=torch.int64, device=DEV)
        fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, zero.clone(), seq, r2t_pool.req_to_token[0, :n].to(torch.int64), n,
                          seq.cpu(), torch.arange(n, device=DEV), extend_num_tokens=n, extend_seq_lens=seq.clone(),
                          extend_prefix_lens=zero, extend_start_loc=zero.clone(), extend_prefix_lens_cpu=[0],
                          extend_seq_lens_cpu=[n], req_to_token_pool=r2t_pool, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        before = len(calls)
        logits = net(ids, torch.arange(n, device=DEV), fb).clone()
        outs.append((logits, [b.clone() for b in pool.k_buffer + pool.v_buffer], len(calls) - before))
    assert outs[0][2] == 0 and outs[1][2] > 0  # the second pass really went through the from-partials norms
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)

def test_prefill_graph_runner_behind_a_cached_prefix():
    """harness.PrefillGraphRunner with prefix_buckets (round 4): a radix hit -- n new tokens behind p cached ones -- replayed
    from the graph captured for (token bucket, prefix bucket).  p == the prefix bucket and n == the token bucket: bit-identical
    to the eager pass (same launches, the extend kernel's KV-range parts planned from the same bound); a shorter prefix
    through the same graph: the same greedy token, logits within the 16-bit tolerance (the parts are planned from the bucket's
    bound, the eager pass from the true length); the new tokens' pool rows equal."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        PrefillGraphRunner, ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 2, 128, 1024, 2048, 2, 512, 4096)
    net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
    n_tok, width = 2601, 2600
    r2t_pool = ReqToTokenPool(1, width, DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    r2t_pool.req_to_token.copy_((torch.randperm(n_tok - 1, device=DEV, generator=g) + 1)[:width].view(1, width).to(torch.int32))
    rows = MHATokenToKVPool(n_tok, 1, torch.bfloat16, 2, 128, 2, DEV).get_key_buffer(0).shape[0]
    kv0 = [torch.randn(rows, 2, 128, device=DEV, generator=g).to(torch.bfloat16) for _ in range(4)]  # the cached tokens' K / V

    def fresh_pool():
        pool = MHATokenToKVPool(n_tok, 1, torch.bfloat16, 2, 128, 2, DEV)
        for l in range(2):
            pool.get_key_buffer(l).copy_(kv0[2 * l])
            pool.get_value_buffer(l).copy_(kv0[2 * l + 1])
        return pool

    def eager(ids, p, pool, backend=None):
        n = ids.numel()
        if backend is None:
            runner = ModelRunnerLike(cfg, r2t_pool, pool, DEV, 0, 1, ServerArgs())
            backend = install_attention_backend(runner)
        seq = torch.full((1,), p + n, dtype=torch.int64, device=DEV)
        zero = torch.zeros(1, dtype=torch.int64, device=DEV)
        pos = torch.arange(p, p + n, device=DEV)
        fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, zero.clone(), seq, r2t_pool.req_to_token[0, p:p + n].to(torch.int64), n,
                          seq.cpu(), pos, extend_num_tokens=n, extend_seq_lens=torch.full((1,), n, dtype=torch.int64, device=DEV),
                          extend_prefix_lens=zero + p, extend_start_loc=zero.clone(), extend_prefix_lens_cpu=[p],
                          extend_seq_lens_cpu=[n], req_to_token_pool=r2t_pool, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        return net(ids, pos, fb).clone()

    pool_g = fresh_pool()
    runner = ModelRunnerLike(cfg, r2t_pool, pool_g, DEV, 0, 1, ServerArgs())
    backend = install_attention_backend(runner)
    pg = PrefillGraphRunner(net, runner, backend, DEV, buckets=(128,), prefix_buckets=(0, 2048))
    # the graphs' qo_indptr / kv_indptr are the backend's shared buffers.  The order ends with the prefix-bucket-0
    # graph replayed AFTER prefix passes left kv_indptr[1] != 0 there -- a replay of the other bucket, and an eager prefix
    # prefill on the runner's own backend (`dirty`) -- which used to pair the stale length with a 1-entry kv_indices.
    for n, p, dirty in ((128, 2048, False), (128, 1500, False), (100, 2048, False), (128, 0, False), (128, 2048, False),
                        (128, 0, False), (128, 0, True)):
        ids = torch.randint(0, 512, (n,), device=DEV, generator=g)
        pool_e = fresh_pool()
        ref = eager(ids, p, pool_e)
        if dirty:
            eager(torch.randint(0, 512, (64,), device=DEV, generator=g), 1800, pool_g, backend=backend)
        for l in range(2):
            pool_g.get_key_buffer(l).copy_(kv0[2 * l])
            pool_g.get_value_buffer(l).copy_(kv0[2 * l + 1])
        slots = r2t_pool.req_to_token[0, p:p + n].to(torch.int64)
        pre = r2t_pool.req_to_token[0, :p].to(torch.int64) if p else None
        logits, tok = pg.run(ids, slots, pre)
        torch.cuda.synchronize()
        assert int(tok) == int(ref.float().argmax())
        if n == 128 and p in (0, 2048):
            assert torch.equal(logits, ref)
        else:
            torch.testing.assert_close(logits.float(), ref.float(), rtol=2.0 ** -6, atol=1e-3 * float(ref.float().abs().max()) + 1e-3)
        ke, kg = pool_e.get_key_buffer(0), pool_g.get_key_buffer(0)
        assert torch.equal(kg[slots], ke[slots])
