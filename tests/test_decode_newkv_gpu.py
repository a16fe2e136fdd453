"""GPU: paged decode with the step's KV write inside the attention launch (sgl_mi355_decode_attention_newkv, round 5) against
the two-launch sequence it replaces (set_kv_buffer, then decode attention) and against the oracle: both pools bit-identical,
outputs within the per-element decode tolerance (the new token enters the softmax as one more partial state instead of
through the last streamed tile, so the roundings differ).  Measured no faster (profiles/r05_decode_kv_write_fusion.txt): the
kernel form is compiled only into the variant build (-DSGLM_OPTIN_FUSIONS=1), where these tests run; on the default library
the entry point declines and the callers make the two calls (last test but one)."""
import pytest
import torch

import oracle
from conftest import assert_elem_close
from sglang_npu_amd import ops
from test_decode_gpu import decode_p_term

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, Hq, Hkv, D, dtype, lens, seed):
    g = torch.Generator().manual_seed(seed)
    seq = torch.tensor(lens)
    S = int(seq.max())
    n_tok = int(seq.sum()) + 8
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    # key / value: rows of a wider [B, (Hq + 2 Hkv) D] tensor, as q, k, v = qkv.split(...) hands them over
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, generator=g).to(dtype)
    key, val = qkv[:, Hq * D:(Hq + Hkv) * D].view(B, Hkv, D), qkv[:, (Hq + Hkv) * D:].view(B, Hkv, D)
    perm = torch.randperm(n_tok - 1, generator=g) + 1
    r2t = torch.zeros(B, S, dtype=torch.int32)
    off = 0
    for b in range(B):
        L = int(seq[b])
        r2t[b, :L] = perm[off:off + L].int()
        off += L
    loc = torch.stack([r2t[b, seq[b] - 1] for b in range(B)]).long()
    return q, kb, vb, qkv, key, val, r2t, seq, loc


@pytest.mark.optin_fusions
@pytest.mark.parametrize("B,Hq,Hkv,D,dtype,loc64", [(64, 32, 8, 128, torch.bfloat16, True), (70, 8, 8, 64, torch.float16, False),
                                                     (140, 40, 2, 128, torch.bfloat16, True), (300, 8, 1, 128, torch.bfloat16, True)])
def test_newkv_matches_the_two_launches_and_the_oracle(B, Hq, Hkv, D, dtype, loc64):
    base = [1, 2, 33, 64, 65, 700, 4096, 4097, 5000, 2048]
    g = torch.Generator().manual_seed(B)
    lens = [base[b % len(base)] if b < 20 else int(torch.randint(1, 400, (1,), generator=g)) for b in range(B)]
    q, kb, vb, qkv, key, val, r2t, seq, loc = _case(B, Hq, Hkv, D, dtype, lens, seed=B + Hq)
    rpi = torch.arange(B)
    # oracle: write, then attend
    kb_ref, vb_ref = kb.clone(), vb.clone()
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q, kb_ref, vb_ref, o_ref, key.contiguous(), val.contiguous(), loc, torch.zeros(B, Hq, 1, D + 1), r2t, rpi,
                            seq, D ** -0.5, 0.0, p_round=True)
    d = lambda t: t.to(DEV)  # noqa: E731
    qkv_d = d(qkv)
    key_d, val_d = qkv_d[:, Hq * D:(Hq + Hkv) * D].view(B, Hkv, D), qkv_d[:, (Hq + Hkv) * D:].view(B, Hkv, D)
    loc_d = d(loc) if loc64 else d(loc).to(torch.int32)
    # two launches
    kb2, vb2 = d(kb), d(vb)
    o2 = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
    ops.set_kv_buffer(kb2, vb2, d(loc), key_d, val_d)
    ops.decode_attention_paged(d(q), kb2, vb2, o2, d(r2t), d(rpi), d(seq), None, 1, D ** -0.5, 0.0)
    # one launch
    kb1, vb1 = d(kb), d(vb)
    o1 = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
    assert ops.decode_attention_paged_newkv(d(q), kb1, vb1, o1, key_d, val_d, loc_d, d(r2t), d(rpi), d(seq), D ** -0.5, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(kb1, kb2) and torch.equal(vb1, vb2), "the pools must be bit-identical to set_kv_buffer's"
    assert torch.equal(kb1.cpu().view(torch.int16), kb_ref.view(torch.int16)) and torch.equal(vb1.cpu().view(torch.int16), vb_ref.view(torch.int16))
    term = decode_p_term(q, kb_ref, vb_ref, r2t, rpi, seq, D ** -0.5, 0.0, dtype)
    assert_elem_close(o1, o_ref, dtype, pair=True, what="one launch vs oracle", extra=term)
    assert_elem_close(o1, o2, dtype, pair=True, what="one launch vs two launches", extra=term)


@pytest.mark.optin_fusions
def test_newkv_with_logit_cap_and_deterministic():
    B, Hq, Hkv, D, dtype = 64, 32, 8, 128, torch.bfloat16
    lens = [int(x) for x in torch.randint(1, 600, (B,), generator=torch.Generator().manual_seed(3))]
    q, kb, vb, qkv, key, val, r2t, seq, loc = _case(B, Hq, Hkv, D, dtype, lens, seed=9)
    rpi = torch.arange(B)
    kb_ref, vb_ref = kb.clone(), vb.clone()
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q, kb_ref, vb_ref, o_ref, key.contiguous(), val.contiguous(), loc, torch.zeros(B, Hq, 1, D + 1), r2t, rpi,
                            seq, 0.2, 20.0, p_round=True)
    d = lambda t: t.to(DEV)  # noqa: E731
    outs = []
    for _ in range(2):
        kb1, vb1 = d(kb), d(vb)
        o1 = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
        assert ops.decode_attention_paged_newkv(d(q), kb1, vb1, o1, d(key.contiguous()), d(val.contiguous()), d(loc), d(r2t), d(rpi),
                                                d(seq), 0.2, 20.0)
        outs.append(o1)
    assert torch.equal(outs[0], outs[1])
    assert_elem_close(outs[0], o_ref, dtype, pair=True, what="logit cap",
                      extra=decode_p_term(q, kb_ref, vb_ref, r2t, rpi, seq, 0.2, 20.0, dtype))


def test_newkv_declines_outside_its_form_and_writes_nothing():
    B, Hq, Hkv, D, dtype = 8, 32, 8, 128, torch.bfloat16   # 64 items: the split kernels' territory
    q, kb, vb, qkv, key, val, r2t, seq, loc = _case(B, Hq, Hkv, D, dtype, [100] * B, seed=1)
    d = lambda t: t.to(DEV)  # noqa: E731
    kb1, vb1 = d(kb), d(vb)
    o = torch.full((B, Hq, D), 7.0, dtype=dtype, device=DEV)
    assert ops.decode_attention_paged_newkv(d(q), kb1, vb1, o, d(key.contiguous()), d(val.contiguous()), d(loc), d(r2t),
                                            torch.arange(B, device=DEV), d(seq), D ** -0.5, 0.0) is False
    torch.cuda.synchronize()
    assert torch.equal(kb1.cpu(), kb) and torch.equal(vb1.cpu(), vb) and bool((o == 7.0).all())
    # an FP8 pool: declined on the host
    kb8 = torch.zeros(kb.shape, dtype=torch.uint8, device=DEV)
    assert ops.decode_attention_paged_newkv(d(q), kb8, kb8.clone(), o, d(key.contiguous()), d(val.contiguous()), d(loc), d(r2t),
                                            torch.arange(B, device=DEV), d(seq), D ** -0.5, 0.0) is False


@pytest.mark.optin_fusions
def test_backend_decode_with_and_without_the_fused_kv_write(monkeypatch):
    """MI355AttnBackend.forward_decode(save_kv_cache=True) at bs = 64: the one-launch form against set_kv_buffer + attention."""
    from sglang_npu_amd import attention_backend as AB
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike, RadixAttention,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    B, Hq, Hkv, D, max_len = 64, 32, 8, 128, 300
    cfg = ModelConfig(Hq, Hkv, D, Hq * D, 1024, 1, 512, 2048)
    outs, pools = [], []
    for fuse in (True, False):
        monkeypatch.setattr(AB, "FUSE_DECODE_KV_WRITE", fuse)
        g = torch.Generator(device=DEV).manual_seed(4)
        r2t = ReqToTokenPool(B, max_len, DEV)
        pool = MHATokenToKVPool(B * max_len + 1, 1, torch.bfloat16, Hkv, D, 1, DEV)
        r2t.req_to_token.copy_((torch.randperm(B * max_len, device=DEV, generator=g) + 1).view(B, max_len).to(torch.int32))
        pool.k_buffer[0].normal_(generator=g)
        pool.v_buffer[0].normal_(generator=g)
        runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
        backend = install_attention_backend(runner)
        layer = RadixAttention(Hq, D, D ** -0.5, Hkv, 0)
        rpi = torch.arange(B, device=DEV)
        seq = torch.randint(1, max_len, (B,), device=DEV, generator=g)
        loc = r2t.req_to_token[rpi, seq - 1].long()
        qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=DEV, generator=g).bfloat16()
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        fb = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), seq - 1, req_to_token_pool=r2t,
                          token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        calls = []
        real = ops.decode_attention_paged_newkv
        monkeypatch.setattr(ops, "decode_attention_paged_newkv", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
        outs.append(layer(q, k, v, fb).clone())
        monkeypatch.setattr(ops, "decode_attention_paged_newkv", real)
        assert bool(calls) == fuse
        pools.append((pool.k_buffer[0].clone(), pool.v_buffer[0].clone()))
    assert torch.equal(pools[0][0], pools[1][0]) and torch.equal(pools[0][1], pools[1][1])
    kpool, vpool = pools[1]
    term = decode_p_term(q.reshape(B, Hq, D), kpool, vpool, r2t.req_to_token, rpi, seq, D ** -0.5, 0.0, torch.bfloat16)
    assert_elem_close(outs[0].view(B, Hq, D), outs[1].view(B, Hq, D), torch.bfloat16, pair=True, what="backend: one launch vs two",
                      extra=term)
