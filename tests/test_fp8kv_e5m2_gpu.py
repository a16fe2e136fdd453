"""GPU: float8_e5m2 KV cache (`--kv-cache-dtype fp8_e5m2`, server_args.py:829-833) -- the pool write (bit-exact against
torch's own cast and the oracle), paged decode, the extend prefix stage, the fused RoPE + pool write (the backend on an
e5m2 MHATokenToKVPool: tests/test_fp8kv_gpu.py, parametrised over the pool dtype).  Bars as for the e4m3 pool (tests/test_fp8kv_gpu.py): P (and Q in the extend prefix stage) is
rounded to the pool format before the products, so the kernel's error against the exact-P truth must stay within 1.5x of
the oracle's own rounded-P error; e5m2 has 2 mantissa bits, so that noise is about twice the e4m3 one."""
import numpy as np
import pytest
import torch

import oracle
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E5 = torch.float8_e5m2


def test_e5m2_cast_every_16_bit_value_matches_torch():
    """common.h cvt_e5m2_torch through set_kv_buffer_fp8: every fp16 and every bf16 bit pattern."""
    bits = torch.from_numpy(np.arange(65536, dtype=np.uint16).view(np.int16))
    for dtype in (torch.float16, torch.bfloat16):
        x = bits.view(dtype)
        src = x.view(512, 1, 128)
        kb = torch.zeros(513, 1, 128, dtype=torch.uint8, device=DEV)
        vb = torch.zeros_like(kb)
        loc = torch.arange(1, 513, device=DEV)
        ops.set_kv_buffer_fp8(kb, vb, loc, src.to(DEV), src.to(DEV), fp8_dtype=E5)
        got = kb[1:].cpu().view(-1)
        ref = x.to(E5).view(torch.uint8)
        nan = torch.isnan(x.float())
        assert torch.equal(got[~nan], ref[~nan])
        assert bool(((got[nan] & 0x7f) == 0x7f).all())  # NaN stays NaN (0x7f | sign)
        assert torch.equal(vb[1:].cpu().view(-1), got)
        assert torch.equal(oracle.cvt_f32_to_e5m2(x.float())[~nan], ref[~nan])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [None, (0.5, 2.0)])
def test_set_kv_buffer_e5m2_bit_exact(dtype, scales):
    g = torch.Generator().manual_seed(1)
    T, Hkv, D = 37, 4, 128
    k = (torch.randn(T, Hkv, D, generator=g) * 4).to(dtype)
    v = (torch.randn(T, Hkv, D, generator=g) * 4).to(dtype)
    special = torch.tensor([57344.0, -60000.0, 61440.0, 1e-6, float("nan"), float("inf"), -float("inf"), 2.0 ** -17, 3e-5])
    k[0, 0, :special.numel()] = special.to(dtype)
    v[1, 1, :special.numel()] = special.to(dtype)
    loc = (torch.randperm(99, generator=g)[:T] + 1)
    kb, vb = torch.zeros(100, Hkv, D, dtype=torch.uint8), torch.zeros(100, Hkv, D, dtype=torch.uint8)
    ks, vs = scales if scales else (None, None)
    oracle.set_kv_buffer_fp8(kb, vb, k, v, loc, ks, vs, kv_dtype=E5)
    kb_d, vb_d = torch.zeros_like(kb, device=DEV), torch.zeros_like(vb, device=DEV)
    ops.set_kv_buffer_fp8(kb_d, vb_d, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs, fp8_dtype=E5)
    assert torch.equal(kb_d.cpu(), kb) and torch.equal(vb_d.cpu(), vb)
    for pool, src, sc in ((kb, k, ks), (vb, v, vs)):
        ref = (src / sc if sc else src).to(E5).view(torch.uint8)
        got = pool[loc]
        nan = (ref & 0x7f) > 0x7c
        assert torch.equal(got[~nan], ref[~nan]) and bool(((got[nan] & 0x7f) > 0x7c).all())
    # the typed pool (no explicit fp8_dtype) takes the same path
    kb_t = torch.zeros(100, Hkv, D, dtype=torch.uint8, device=DEV).view(E5)
    vb_t = torch.zeros(100, Hkv, D, dtype=torch.uint8, device=DEV).view(E5)
    ops.set_kv_buffer_fp8(kb_t, vb_t, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs)
    assert torch.equal(kb_t.view(torch.uint8).cpu(), kb)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (32, 32, 128)])
@pytest.mark.parametrize("splits", [1, 4])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_e5m2_kv_vs_oracle(Hq, Hkv, D, splits, dtype):
    g = torch.Generator().manual_seed(Hq + D + splits)
    B, max_len = 5, 700
    seq = torch.tensor([700, 1, 33, 256, 417])
    n_tok = B * max_len + 1
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(E5)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(E5)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).view(B, max_len).int()
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    rpi = torch.arange(B)
    truth, ref = torch.zeros(B, Hq, D, dtype=dtype), torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention_fp8kv(q, kb, vb, truth, torch.zeros(B, Hq, splits, D + 1), r2t, rpi, seq, D ** -0.5, p_fp8=False)
    oracle.decode_attention_fp8kv(q, kb, vb, ref, torch.zeros(B, Hq, splits, D + 1), r2t, rpi, seq, D ** -0.5, p_fp8=True)
    o = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV) if splits > 1 else None
    ops.decode_attention_paged(q.to(DEV), kb.to(DEV), vb.to(DEV), o, r2t.to(DEV), rpi.to(DEV), seq.to(DEV), logits, splits,
                               D ** -0.5, 0.0)
    err_hip = (o.float().cpu() - truth.float()).abs()
    err_ref = (ref.float() - truth.float()).abs()
    scale = float(truth.float().abs().max())
    assert float(err_ref.max()) > 0  # the rounded-P reference really differs from the truth (the format is in effect)
    assert float(err_hip.pow(2).mean().sqrt()) <= 1.5 * float(err_ref.pow(2).mean().sqrt()) + 2.0 ** -9 * scale
    assert float(err_hip.max()) <= 1.5 * float(err_ref.max()) + 2.0 ** -8 * scale
    assert float(err_hip[1].max()) <= 2.0 ** -8 * scale  # a single-token request has p = 1 exactly
    # decoding the SAME bytes as e4m3 must give something else (the dispatch really follows the dtype)
    o_wrong = torch.zeros_like(o)
    ops.decode_attention_paged(q.to(DEV), kb.view(torch.uint8).to(DEV), vb.view(torch.uint8).to(DEV), o_wrong, r2t.to(DEV),
                               rpi.to(DEV), seq.to(DEV), logits, splits, D ** -0.5, 0.0)
    assert not torch.allclose(o_wrong.float(), o.float(), atol=1e-2)
    # the flattened (Triton) form agrees with the page-table form
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    kv_indptr[1:] = torch.cumsum(seq.to(DEV), 0)
    kv_indices = torch.cat([r2t[b, :seq[b]] for b in range(B)]).to(DEV)
    o2 = torch.zeros_like(o)
    al = torch.zeros(B, Hq, max(splits, 1), D, device=DEV)
    ls = torch.zeros(B, Hq, max(splits, 1), device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o2, kv_indptr, kv_indices, al if splits > 1 else None,
                             ls if splits > 1 else None, None, splits, D ** -0.5, 0.0)
    assert float((o2.float() - o.float()).abs().max()) <= 1.5 * float(err_ref.max()) + 2.0 ** -8 * scale


def test_decode_e5m2_full_batch_shape():
    """bs=64 x 8 kv heads, ctx 2048 (BASELINE's decode shape) on an e5m2 pool against an fp32 evaluation with exact P on
    the same bytes: the error stays in the fp8-P noise band (2^-3 relative per probability, averaged over ~2k keys)."""
    g = torch.Generator(device=DEV).manual_seed(0)
    B, Hq, Hkv, D, S = 64, 32, 8, 128, 2048
    n_tok = B * S + 1
    kb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(E5)
    vb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(E5)
    r2t = (torch.randperm(n_tok - 1, device=DEV, generator=g) + 1).view(B, S).int()
    q = torch.randn(B, Hq, D, device=DEV, generator=g).bfloat16()
    seq = torch.full((B,), S, dtype=torch.int64, device=DEV)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o, r2t, torch.arange(B, device=DEV), seq, None, 1, D ** -0.5, 0.0)
    for b in (0, 31, 63):
        idx = r2t[b].long()  # (gather on the uint8 view: indexing is not implemented for every float8 dtype)
        k = kb.view(torch.uint8)[idx].view(E5).float().repeat_interleave(Hq // Hkv, dim=1)
        v = vb.view(torch.uint8)[idx].view(E5).float().repeat_interleave(Hq // Hkv, dim=1)
        p = torch.softmax(torch.einsum("hd,nhd->hn", q[b].float(), k) * D ** -0.5, dim=-1)
        truth = torch.einsum("hn,nhd->hd", p, v)
        assert float((o[b].float() - truth).abs().max()) <= 0.02 + 2.0 ** -8 * float(truth.abs().max())


def _extend_case(g, B, Hq, Hkv, D, prefix, ext, dtype):
    max_len = int(max(p + e for p, e in zip(prefix, ext)))
    n_tok = B * max_len + 1
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(E5)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(E5)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).view(B, max_len).int()
    T = int(sum(ext))
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    k = torch.randn(T, Hkv, D, generator=g).to(dtype)
    v = torch.randn(T, Hkv, D, generator=g).to(dtype)
    ext_t, pre_t = torch.tensor(ext), torch.tensor(prefix)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(ext_t, 0)
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(pre_t, 0)
    kv_indices = torch.cat([r2t[b, :prefix[b]] for b in range(B)] + [torch.zeros(0, dtype=torch.int32)])
    return dict(kb=kb, vb=vb, r2t=r2t, q=q, k=k, v=v, ext=ext_t, seq=ext_t + pre_t, start=torch.cumsum(ext_t, 0) - ext_t,
                qo_indptr=qo_indptr, kv_indptr=kv_indptr, kv_indices=kv_indices, rpi=torch.arange(B))


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (14, 2, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_extend_e5m2_prefix_vs_oracle(Hq, Hkv, D, dtype):
    g = torch.Generator().manual_seed(Hq * 3 + D)
    c = _extend_case(g, 4, Hq, Hkv, D, prefix=[300, 0, 64, 517], ext=[70, 33, 1, 129], dtype=dtype)
    o = torch.zeros(c["q"].shape, dtype=dtype, device=DEV)
    d = lambda t: t.to(DEV)
    ops.extend_attention_fwd(d(c["q"]), d(c["k"]), d(c["v"]), o, d(c["kb"]), d(c["vb"]), d(c["qo_indptr"]), d(c["kv_indptr"]),
                             d(c["kv_indices"]), None, True, None, int(c["ext"].max()), D ** -0.5)
    o = o.float().cpu()

    def orc(**kw):
        out = torch.zeros(c["q"].shape, dtype=dtype)
        oracle.extend_attention_fp8kv(c["q"], c["k"], c["v"], out, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                                      c["start"], D ** -0.5, **kw)
        return out.float()

    truth, ref = orc(p_fp8=False), orc(p_fp8=True)
    err_hip, err_ref = (o - truth).abs(), (ref - truth).abs()
    scale = float(truth.abs().max())
    assert float(err_hip.pow(2).mean().sqrt()) <= 1.5 * float(err_ref.pow(2).mean().sqrt()) + 2.0 ** -9 * scale
    assert float(err_hip.max()) <= 1.5 * float(err_ref.max()) + 2.0 ** -8 * scale
    assert float(err_hip[70:103].max()) <= 2.0 ** -7 * scale  # the request without a prefix never touches the pool


def test_rope_and_kv_write_into_e5m2_pool_bit_exact():
    """apply_rope_and_set_kv_buffer / rope_set_kv_from_partials on an e5m2 pool == the 16-bit results cast by
    set_kv_buffer_fp8(fp8_dtype=e5m2)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    M, Hq, Hk, D = 33, 4, 2, 128
    pos = torch.randint(0, 500, (M,), device=DEV, generator=g)
    cache = torch.randn(512, D, device=DEV, generator=g)
    loc = (torch.randperm(200, device=DEV, generator=g)[:M] + 1).long()
    qkv = torch.randn(M, (Hq + 2 * Hk) * D, device=DEV, generator=g).bfloat16()
    # 16-bit pool as the intermediate truth
    kb16, vb16 = torch.zeros(201, Hk, D, dtype=torch.bfloat16, device=DEV), torch.zeros(201, Hk, D, dtype=torch.bfloat16, device=DEV)
    q1, k1, v1 = qkv.clone().split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.apply_rope_and_set_kv_buffer(pos, q1, k1, v1, D, cache, kb16, vb16, loc, True)
    kb_ref, vb_ref = torch.zeros(201, Hk, D, dtype=torch.uint8, device=DEV), torch.zeros(201, Hk, D, dtype=torch.uint8, device=DEV)
    ops.set_kv_buffer_fp8(kb_ref, vb_ref, loc, k1.reshape(M, Hk, D), v1.reshape(M, Hk, D), fp8_dtype=E5)
    kb8, vb8 = torch.zeros_like(kb_ref).view(E5), torch.zeros_like(vb_ref).view(E5)
    q2, k2, v2 = qkv.clone().split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.apply_rope_and_set_kv_buffer(pos, q2, k2, v2, D, cache, kb8, vb8, loc, True)
    assert torch.equal(q2, q1)
    assert torch.equal(kb8.view(torch.uint8), kb_ref) and torch.equal(vb8.view(torch.uint8), vb_ref)
    assert torch.equal(kb8.view(torch.uint8)[loc], k1.reshape(M, Hk, D).to(E5).view(torch.uint8))  # and it is torch's cast
