"""GPU: FP8 (e4m3fn) KV cache -- pool write and paged decode over byte rows.
P is rounded to FP8 before P.V (decode_attention.py:373), so results carry fp8-P noise (relative 2^-4 per probability);
which probabilities round up or down depends on the running max they are scaled by, i.e. on the tile order, and the
reference itself changes with num_kv_splits.  The bar therefore is statistical and stated next to each assert: the HIP
kernel's error against the exact-P truth must not exceed the oracle's own fp8-P error by more than 50 %, and the pool
write is bit-exact."""
import pytest
import torch

import oracle
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [None, (0.5, 2.0)])
def test_set_kv_buffer_fp8_bit_exact(dtype, scales):
    g = torch.Generator().manual_seed(1)
    T, Hkv, D = 37, 4, 128
    k = (torch.randn(T, Hkv, D, generator=g) * 4).to(dtype)
    v = (torch.randn(T, Hkv, D, generator=g) * 4).to(dtype)
    k[0, 0, :4] = torch.tensor([500.0, -1000.0, 448.0, 1e-4]).to(dtype)  # saturation and a subnormal
    loc = (torch.randperm(99, generator=g)[:T] + 1)
    kb, vb = torch.zeros(100, Hkv, D, dtype=torch.uint8), torch.zeros(100, Hkv, D, dtype=torch.uint8)
    ks, vs = scales if scales else (None, None)
    oracle.set_kv_buffer_fp8(kb, vb, k, v, loc, ks, vs)
    kb_d, vb_d = torch.zeros_like(kb, device=DEV), torch.zeros_like(vb, device=DEV)
    ops.set_kv_buffer_fp8(kb_d, vb_d, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs)
    assert torch.equal(kb_d.cpu(), kb) and torch.equal(vb_d.cpu(), vb)
    # in range it is torch's own cast (memory_pool.py:389-391)
    ref = (v / vs if vs else v).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(vb[loc], ref)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (32, 32, 128)])
@pytest.mark.parametrize("splits", [1, 4])
def test_decode_fp8_kv_vs_oracle(Hq, Hkv, D, splits):
    g = torch.Generator().manual_seed(Hq + D + splits)
    B, max_len = 5, 700
    seq = torch.tensor([700, 1, 33, 256, 417])
    n_tok = B * max_len + 1
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).view(B, max_len).int()
    q = torch.randn(B, Hq, D, generator=g).bfloat16()
    rpi = torch.arange(B)
    truth, ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16), torch.zeros(B, Hq, D, dtype=torch.bfloat16)
    kb8, vb8 = kb.view(torch.uint8), vb.view(torch.uint8)
    oracle.decode_attention_fp8kv(q, kb8, vb8, truth, torch.zeros(B, Hq, splits, D + 1), r2t, rpi, seq, D ** -0.5, p_fp8=False)
    oracle.decode_attention_fp8kv(q, kb8, vb8, ref, torch.zeros(B, Hq, splits, D + 1), r2t, rpi, seq, D ** -0.5, p_fp8=True)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV) if splits > 1 else None
    ops.decode_attention_paged(q.to(DEV), kb.to(DEV), vb.to(DEV), o, r2t.to(DEV), rpi.to(DEV), seq.to(DEV), logits, splits,
                               D ** -0.5, 0.0)
    err_hip = (o.float().cpu() - truth.float()).abs()
    err_ref = (ref.float() - truth.float()).abs()
    scale = float(truth.float().abs().max())
    # fp8-P noise: RMS and max of the HIP kernel within 1.5x of the oracle's (same rounding, other tile order)
    assert float(err_hip.pow(2).mean().sqrt()) <= 1.5 * float(err_ref.pow(2).mean().sqrt()) + 2.0 ** -9 * scale
    assert float(err_hip.max()) <= 1.5 * float(err_ref.max()) + 2.0 ** -8 * scale
    # a single-token request has p = 1 exactly: no fp8 noise at all -> the usual 16-bit bound
    assert float(err_hip[1].max()) <= 2.0 ** -8 * scale
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


def test_fp8_pool_unsupported_shapes_raise():
    q = torch.zeros(1, 2, 80, dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros(9, 2, 80, dtype=torch.float8_e4m3fn, device=DEV)
    with pytest.raises(NotImplementedError):
        ops.decode_attention_paged(q, kb, kb, torch.zeros_like(q), torch.zeros(1, 4, dtype=torch.int32, device=DEV),
                                   torch.zeros(1, dtype=torch.int64, device=DEV), torch.ones(1, dtype=torch.int64, device=DEV),
                                   None, 1, 1.0, 0.0)
    with pytest.raises(NotImplementedError):
        ops.decode_attention_paged(torch.zeros(1, 2, 128, dtype=torch.bfloat16, device=DEV),
                                   torch.zeros(9, 2, 128, dtype=torch.float8_e5m2, device=DEV),
                                   torch.zeros(9, 2, 128, dtype=torch.float8_e5m2, device=DEV),
                                   torch.zeros(1, 2, 128, dtype=torch.bfloat16, device=DEV),
                                   torch.zeros(1, 4, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
                                   torch.ones(1, dtype=torch.int64, device=DEV), None, 1, 1.0, 0.0)
