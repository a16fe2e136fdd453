"""GPU: paged decode attention (HIP, through the C ABI) against the golden vectors, the oracle
and size-independent properties at BASELINE.json's full size."""
import pytest
import torch

import oracle
from conftest import assert_elem_close, golden_names, load_golden, p_rounding_term
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to(d, keys):
    return {k: d[k].to(DEV) for k in keys}


@pytest.mark.parametrize("name", golden_names("decode_"))
def test_decode_op_form_vs_golden(name):
    g = load_golden(name)
    B, Hq, Hkv, D, Dv, S, splits = [int(x) for x in g["meta"]]
    t = _to(g, ["q", "k_buffer", "v_buffer", "key", "value", "loc", "req_to_token", "req_pool_indices", "seq_lens"])
    o = torch.zeros(B, Hq, Dv, dtype=t["q"].dtype, device=DEV)
    logits = torch.zeros(B, Hq, splits, Dv + 1, device=DEV)
    ops.decode_attention(t["q"], t["k_buffer"], t["v_buffer"], o, t["key"], t["value"], t["loc"], logits,
                         t["req_to_token"], t["req_pool_indices"], t["seq_lens"], g["sm_scale"], g["logit_cap"])
    torch.cuda.synchronize()
    exp_k, exp_v = g["k_buffer"].clone(), g["v_buffer"].clone()
    exp_k[g["loc"]] = g["key"]
    exp_v[g["loc"]] = g["value"]
    assert torch.equal(t["k_buffer"].cpu().view(torch.int16), exp_k.view(torch.int16)), "KV write must be bit-exact"
    assert torch.equal(t["v_buffer"].cpu().view(torch.int16), exp_v.view(torch.int16)), "KV write must be bit-exact"
    assert_elem_close(o, g["o_f32"], g["dtype"], what=f"{name}: hip vs the fp32 truth")  # per element: 1e-3 + ulp |ref_i|
    if g["ref_valid"]:  # the compiled reference kernel's own 16-bit output: two rounded results
        assert_elem_close(o, g["o_ref"], g["dtype"], pair=True, what=f"{name}: hip vs the reference kernel")


def decode_p_term(q, kb, vb, r2t, rpi, seq, scale, cap, dtype, pair=True):
    """conftest.p_rounding_term for a decode call: the oracle's attention of |V| with unrounded probabilities.  The HIP kernels
    hand the softmax numerators to the PV MFMA in the 16-bit type (the oracle's p_round=True models that; the reference's
    decode.cpp keeps them in fp32): for SHORT sequences -- a handful of keys with comparable weights -- the two roundings
    (taken relative to different running maxima) differ by up to half an ulp of each p_j times |v_j|, per element."""
    B, Hq, D = q.shape
    a = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q.cpu(), kb.cpu(), vb.cpu().abs(), a, None, None, None, torch.zeros(B, Hq, 1, D + 1), r2t.cpu(),
                            rpi.cpu(), seq.cpu(), scale, cap, p_round=False)
    return p_rounding_term(dtype, a, pair=pair)


def _random_case(B, Hq, Hkv, D, S, dtype, seed, ragged=True, idx_dtype=torch.int32):
    g = torch.Generator().manual_seed(seed)
    n_tok = B * S + 8
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    seq = torch.randint(max(1, S // 2), S + 1, (B,), generator=g) if ragged else torch.full((B,), S)
    perm = torch.randperm(n_tok - 1, generator=g) + 1
    r2t = torch.zeros(B, S, dtype=idx_dtype)
    off = 0
    for b in range(B):
        L = int(seq[b])
        r2t[b, :L] = perm[off:off + L].to(idx_dtype)
        off += L
    return q, kb, vb, r2t, torch.arange(B), seq


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (32, 32, 128), (14, 2, 64), (64, 2, 128)])
@pytest.mark.parametrize("splits", [1, 3, 8])
def test_decode_backend_form_vs_oracle(Hq, Hkv, D, splits):
    B, S = 5, 700
    dtype = torch.bfloat16
    q, kb, vb, r2t, rpi, seq = _random_case(B, Hq, Hkv, D, S, dtype, seed=Hq * 131 + splits)
    # oracle (CPU restatement of decode.cpp / decode_attention.py), p rounded like the Triton kernel
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q, kb.clone(), vb.clone(), o_ref, None, None, None, torch.zeros(B, Hq, 8, D + 1), r2t, rpi,
                            seq, 1.0 / D ** 0.5, 0.0, p_round=True)
    # HIP: flatten the page table, then the backend form with per-request split counts
    d = dict(q=q.to(DEV), kb=kb.to(DEV), vb=vb.to(DEV), r2t=r2t.to(DEV), rpi=rpi.to(DEV), seq=seq.to(DEV))
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    kv_indptr[1:] = torch.cumsum(d["seq"], 0)
    kv_indices = torch.empty(int(seq.sum()), dtype=torch.int32, device=DEV)
    ops.create_kv_indices(d["r2t"], d["rpi"], d["seq"], kv_indptr, None, kv_indices)
    exp_idx = torch.cat([r2t[b, :int(seq[b])] for b in range(B)])
    assert torch.equal(kv_indices.cpu(), exp_idx), "kv_indices must be bit-exact"
    o = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
    if splits == 1:
        ops.decode_attention_fwd(d["q"], d["kb"], d["vb"], o, kv_indptr, kv_indices, None, None, None, 1,
                                 1.0 / D ** 0.5)
    else:
        logits = torch.full((B, Hq, splits, D), float("nan"), device=DEV)
        lse = torch.full((B, Hq, splits), float("nan"), device=DEV)
        nks = torch.tensor([1 + (b % splits) for b in range(B)], dtype=torch.int32, device=DEV)
        ops.decode_attention_fwd(d["q"], d["kb"], d["vb"], o, kv_indptr, kv_indices, logits, lse, nks, splits,
                                 1.0 / D ** 0.5)
    torch.cuda.synchronize()
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle")


def test_kv_indices_vs_reference_fixture():
    """The page-table flatten against tests/golden/kv_indices.npz -- outputs of the reference's own gather
    (test/srt/test_create_kvindices.py:41-49), written by tests/golden/make_golden.py.  Integer work: bit-exact."""
    g = load_golden("kv_indices")
    for i in range(int(g["n"])):
        max_batch, max_ctx = [int(x) for x in g[f"shape{i}"]]
        r2t = torch.arange(max_batch * max_ctx, dtype=torch.int32, device=DEV).reshape(max_batch, max_ctx)
        indptr = g[f"indptr{i}"].to(DEV)
        out = torch.full((int(indptr[-1]),), -1, dtype=torch.int32, device=DEV)
        ops.create_kv_indices(r2t, g[f"rpi{i}"].to(DEV), g[f"lens{i}"].to(DEV), indptr, None, out)
        assert torch.equal(out.cpu(), g[f"kv_indices{i}"]), f"case {i}"


def test_decode_fp16_int64_table_and_logit_cap():
    B, Hq, Hkv, D, S = 3, 32, 8, 128, 300
    q, kb, vb, r2t, rpi, seq = _random_case(B, Hq, Hkv, D, S, torch.float16, seed=5, idx_dtype=torch.int64)
    o_ref = torch.zeros(B, Hq, D, dtype=torch.float16)
    logits = torch.zeros(B, Hq, 4, D + 1)
    oracle.decode_attention(q, kb.clone(), vb.clone(), o_ref, None, None, None, logits, r2t, rpi, seq, 0.2, 20.0,
                            p_round=True)
    o = torch.zeros(B, Hq, D, dtype=torch.float16, device=DEV)
    ops.decode_attention(q.to(DEV), kb.to(DEV), vb.to(DEV), o, None, None, None, logits.to(DEV), r2t.to(DEV),
                         rpi.to(DEV), seq.to(DEV), 0.2, 20.0)
    assert_elem_close(o, o_ref, torch.float16, pair=True, what="hip vs oracle (fp16, logit cap)")


def test_decode_online_softmax_rescale_is_exercised():
    """Force the running max to jump late (a spiked key near the end of the sequence and another
    in the middle), so every rescale branch of the online softmax is taken with a large factor."""
    B, Hq, Hkv, D, S = 2, 32, 8, 128, 1500
    dtype = torch.bfloat16
    q, kb, vb, r2t, rpi, seq = _random_case(B, Hq, Hkv, D, S, dtype, seed=77, ragged=False)
    for b in range(B):
        for pos, gain in ((S // 2 + 3, 3.0), (S - 37, 6.0)):
            tok = int(r2t[b, pos])
            kb[tok] = (q[b].view(Hkv, Hq // Hkv, D)[:, 0] * gain).to(dtype)  # aligned with head 0 of each group
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q, kb.clone(), vb.clone(), o_ref, None, None, None, torch.zeros(B, Hq, 1, D + 1), r2t, rpi,
                            seq, 1.0 / D ** 0.5, 0.0, p_round=True)
    for splits in (1, 4):
        o = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
        ops.decode_attention(q.to(DEV), kb.to(DEV), vb.to(DEV), o, None, None, None,
                             torch.zeros(B, Hq, splits, D + 1, device=DEV), r2t.to(DEV), rpi.to(DEV), seq.to(DEV),
                             1.0 / D ** 0.5, 0.0)
        assert_elem_close(o, o_ref, dtype, pair=True, what=f"hip vs oracle, {splits} kv splits")


def test_decode_full_size_properties():
    """BASELINE config 2 (Llama-3-8B, bs=64, S=2048) on the GPU only -- too big for the oracle in
    seconds, so check size-independent properties:
      (1) split count does not change the result (1 vs 8 splits);
      (2) relabelling pool slots (permuting the pool + page table together) does not change it;
      (3) constant V rows come back unchanged (softmax weights sum to one);
      (4) a bounded sample of rows agrees with the oracle."""
    B, Hq, Hkv, D, S = 64, 32, 8, 128, 2048
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(1)
    n_tok = B * S + 1
    q = torch.randn(B, Hq, D, device=DEV, generator=g).to(dtype)
    kb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    perm = torch.randperm(n_tok - 1, device=DEV, generator=g) + 1
    r2t = perm.view(B, S).to(torch.int32)
    rpi = torch.arange(B, device=DEV)
    seq = torch.randint(S // 2, S + 1, (B,), device=DEV, generator=g)
    seq[0] = S
    scale = 1.0 / D ** 0.5

    def run(kb_, vb_, r2t_, splits):
        o = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
        ops.decode_attention(q, kb_, vb_, o, None, None, None, torch.zeros(B, Hq, splits, D + 1, device=DEV), r2t_,
                             rpi, seq, scale, 0.0)
        return o

    o1 = run(kb, vb, r2t, 1)
    o8 = run(kb, vb, r2t, 8)
    assert_elem_close(o8, o1, dtype, pair=True, what="8 kv splits vs 1")  # (1)
    relabel = torch.randperm(n_tok, device=DEV, generator=g)
    inv = torch.empty_like(relabel)
    inv[relabel] = torch.arange(n_tok, device=DEV)
    o_p = run(kb[relabel], vb[relabel], inv[r2t.long()].to(torch.int32), 1)  # slot s moved to inv[s]
    assert torch.equal(o_p.view(torch.int16), o1.view(torch.int16))  # (2) same arithmetic, same order: bit-exact
    c = torch.randn(1, Hkv, D, device=DEV, generator=g).to(dtype)
    o_c = run(kb, c.expand(n_tok, Hkv, D).contiguous(), r2t, 2)
    exp = c[0].repeat_interleave(Hq // Hkv, 0).float()
    # (3) the probabilities are rounded to 16 bits before PV, so their sum is 1 only to ~2^-9 relative: one more ulp
    assert_elem_close(o_c, exp[None].expand_as(o_c), dtype, pair=True, what="constant V")
    # (4) oracle on 2 requests
    sel = [0, 37]
    o_ref = torch.zeros(len(sel), Hq, D, dtype=dtype)
    oracle.decode_attention(q[sel].cpu(), kb.cpu(), vb.cpu(), o_ref, None, None, None,
                            torch.zeros(len(sel), Hq, 1, D + 1), r2t.cpu(), torch.tensor(sel), seq[sel].cpu(), scale,
                            0.0, p_round=True)
    assert_elem_close(o1[sel], o_ref, dtype, pair=True, what="full size vs oracle on 2 requests")


def test_decode_empty_batch_and_zero_length():
    dtype = torch.bfloat16
    q = torch.randn(2, 32, 128, device=DEV).to(dtype)
    kb = torch.randn(64, 8, 128, device=DEV).to(dtype)
    o = torch.full((2, 32, 128), 7.0, dtype=dtype, device=DEV)
    r2t = torch.arange(64, dtype=torch.int32, device=DEV).view(2, 32)
    seq = torch.tensor([0, 5], device=DEV)
    ops.decode_attention(q, kb, kb, o, None, None, None, torch.zeros(2, 32, 2, 129, device=DEV), r2t,
                         torch.arange(2, device=DEV), seq, 0.1, 0.0)
    assert (o[0] == 0).all()  # empty sequence -> zero row
    ops.decode_attention(q[:0], kb, kb, o[:0], None, None, None, torch.zeros(0, 32, 2, 129, device=DEV), r2t,
                         torch.arange(0, device=DEV), seq[:0], 0.1, 0.0)  # empty batch: no-op


@pytest.mark.parametrize("Hq,Hkv,D", [(8, 1, 128), (32, 8, 128), (14, 2, 64)])
@pytest.mark.parametrize("splits", [2, 4])
@pytest.mark.parametrize("fp8_pool", [False, True])
def test_merge_fused_with_fp8_quant_is_bit_identical(Hq, Hkv, D, splits, fp8_pool):
    """sgl_mi355_decode_attention with output = NULL (stage 1 only) + sgl_mi355_decode_merge_quant_fp8 against the full
    decode followed by sgl_per_token_quant_fp8 of the [B, Hq*D] result: same bytes, same scales, same 16-bit rows."""
    g = torch.Generator(device=DEV).manual_seed(Hq + D + splits)
    B, max_len = 6, 600
    seq = torch.tensor([600, 1, 33, 256, 417, 0], device=DEV)  # one empty request: zero row, scale 0
    n_tok = B * max_len + 1
    kv_dt = torch.float8_e4m3fn if fp8_pool else torch.bfloat16
    kb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(kv_dt)
    vb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(kv_dt)
    r2t = (torch.randperm(n_tok - 1, device=DEV, generator=g) + 1).view(B, max_len).int()
    q = torch.randn(B, Hq, D, device=DEV, generator=g).bfloat16()
    rpi = torch.arange(B, device=DEV)
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV)
    o_ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, seq, logits, splits, D ** -0.5, 0.0)
    q_ref = torch.empty(B, Hq * D, dtype=torch.float8_e4m3fn, device=DEV)
    s_ref = torch.empty(B, 1, device=DEV)
    ops.sgl_per_token_quant_fp8(o_ref.view(B, Hq * D), q_ref, s_ref)
    logits2 = torch.full_like(logits, float("nan"))
    ops.decode_attention_paged(q, kb, vb, None, r2t, rpi, seq, logits2, splits, D ** -0.5, 0.0)
    o2 = torch.zeros_like(o_ref)
    qq, ss = ops.decode_merge_quant_fp8(logits2, splits, torch.bfloat16, o2)
    assert torch.equal(o2, o_ref)
    assert torch.equal(ss, s_ref) and torch.equal(qq.view(torch.uint8), q_ref.view(torch.uint8))
    qq2, ss2 = ops.decode_merge_quant_fp8(logits2, splits, torch.bfloat16)  # without the 16-bit copy
    assert torch.equal(ss2, s_ref) and torch.equal(qq2.view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("Hq,Hkv,D,dtype", [(32, 8, 128, torch.bfloat16), (8, 8, 64, torch.float16), (40, 2, 128, torch.bfloat16)])
def test_decode_pairs_of_items_per_workgroup(Hq, Hkv, D, dtype):
    """More (request, kv-head block) items than CUs with one split: decode_mfma_pair_kernel streams two items per
    workgroup.  Ragged lengths incl. empty, one token, exactly / just over the staged window (4096 entries: the fused
    pair needs both items inside it, longer ones go one item at a time), an odd item count -- against the oracle."""
    nhb = (Hq // Hkv + 15) // 16
    B = 260 // (Hkv * nhb) + 1  # > 256 items, so the pair kernel is chosen
    if (B * Hkv * nhb) % 2 == 0 and Hkv * nhb % 2 == 1:
        B += 1
    g = torch.Generator().manual_seed(Hq + D)
    lens = [0, 1, 33, 4096, 4097, 5000, 700, 64]
    seq = torch.tensor([lens[b % len(lens)] if b < 16 else int(torch.randint(1, 300, (1,), generator=g)) for b in range(B)])
    S = int(seq.max())
    n_tok = int(seq.sum()) + 8
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    perm = torch.randperm(n_tok - 1, generator=g) + 1
    r2t = torch.zeros(B, S, dtype=torch.int32)
    off = 0
    for b in range(B):
        L = int(seq[b])
        r2t[b, :L] = perm[off:off + L].int()
        off += L
    rpi = torch.arange(B)
    o_ref = torch.zeros(B, Hq, D, dtype=dtype)
    oracle.decode_attention(q, kb.clone(), vb.clone(), o_ref, None, None, None, torch.zeros(B, Hq, 1, D + 1), r2t, rpi,
                            seq, 1.0 / D ** 0.5, 0.0, p_round=True)
    o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
    ops.decode_attention_paged(q.to(DEV), kb.to(DEV), vb.to(DEV), o, r2t.to(DEV), rpi.to(DEV), seq.to(DEV), None, 1,
                               1.0 / D ** 0.5, 0.0)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle (ragged lengths)",
                      extra=decode_p_term(q, kb, vb, r2t, rpi, seq, 1.0 / D ** 0.5, 0.0, dtype))
    assert float(o[seq == 0].float().abs().max()) == 0.0
