"""GPU: FP8 (e4m3fn) KV cache -- pool write, paged decode and the extend prefix stage over byte rows.
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
    # torch's cast does not saturate: NaN, infinities and everything that rounds past 448 (|x| > 464) are stored as NaN
    special = torch.tensor([500.0, -1000.0, 448.0, 1e-4, float("nan"), float("inf"), -float("inf"), 449.0, 464.0, -464.0,
                            466.0, -470.0, 479.0, 480.0])
    k[0, 0, :special.numel()] = special.to(dtype)
    v[1, 1, :special.numel()] = special.to(dtype)
    loc = (torch.randperm(99, generator=g)[:T] + 1)
    kb, vb = torch.zeros(100, Hkv, D, dtype=torch.uint8), torch.zeros(100, Hkv, D, dtype=torch.uint8)
    ks, vs = scales if scales else (None, None)
    oracle.set_kv_buffer_fp8(kb, vb, k, v, loc, ks, vs)
    kb_d, vb_d = torch.zeros_like(kb, device=DEV), torch.zeros_like(vb, device=DEV)
    ops.set_kv_buffer_fp8(kb_d, vb_d, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs)
    assert torch.equal(kb_d.cpu(), kb) and torch.equal(vb_d.cpu(), vb)
    # it is torch's own cast (memory_pool.py:389-391), out-of-range values and NaN included (NaN's sign bit aside)
    for pool, src, sc in ((kb, k, ks), (vb, v, vs)):
        ref = (src / sc if sc else src).to(torch.float8_e4m3fn).view(torch.uint8)
        got = pool[loc]
        nan = (ref & 0x7f) == 0x7f
        assert torch.equal(got[~nan], ref[~nan]) and bool(((got[nan] & 0x7f) == 0x7f).all()) and int(nan.sum()) >= 3


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


@pytest.mark.parametrize("kv_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1), (32, 32)])
@pytest.mark.parametrize("splits", [1, 4])
def test_decode_fp8_kv_exact_when_every_p_is_a_power_of_two(kv_dtype, Hq, Hkv, splits):
    """A DETERMINISTIC case for the byte-MFMA path: q is one-hot and the keys carry small integers
    on that dimension, sm_scale = ln 2, so every score is an integer n and every probability 2^(n - max) -- exact in e4m3
    and in e5m2 whatever running max it is scaled by (tile order, split count).  The rounding of P to the pool format is
    then the identity, fp8 x fp8 products are exact, and the kernel must agree with the oracle AND with the closed form
    sum(2^n v) / sum(2^n) evaluated in float64 to one output ulp -- no statistical slack."""
    import math
    g = torch.Generator().manual_seed(Hq * 7 + splits)
    B, D, max_len = 4, 128, 600
    seq = torch.tensor([600, 1, 257, 96])
    group = Hq // Hkv
    n_tok = B * max_len + 1
    kb = torch.randn(n_tok, Hkv, D, generator=g)
    # integer scores in [-8, 0] on the first `group` dimensions (head h of a group looks at dimension h % group);
    # -8 keeps 2^(n - max) >= 2^-8: representable in e4m3 (subnormal) and far from its 2^-10 rounding-to-zero tie
    n = torch.randint(-8, 1, (n_tok, Hkv, group), generator=g)
    kb[:, :, :group] = n.float()
    vb = torch.randn(n_tok, Hkv, D, generator=g) * 2
    kb8, vb8 = kb.to(kv_dtype), vb.to(kv_dtype)
    assert torch.equal(kb8[:, :, :group].float(), n.float())  # small integers are exact in both formats
    q = torch.zeros(B, Hq, D)
    for h in range(Hq):
        q[:, h, h % group] = 1.0
    q = q.bfloat16()
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).view(B, max_len).int()
    rpi = torch.arange(B)
    sm = math.log(2.0)
    ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
    oracle.decode_attention_fp8kv(q, kb8, vb8, ref, torch.zeros(B, Hq, splits, D + 1), r2t, rpi, seq, sm, p_fp8=True)
    # closed form in float64
    truth = torch.zeros(B, Hq, D, dtype=torch.float64)
    mag = torch.zeros(B, Hq, D, dtype=torch.float64)  # sum p |v| / sum p: what the fp32 summation noise scales with
    for b in range(B):
        rows = r2t[b, :int(seq[b])].long()
        for h in range(Hq):
            kvh = h // group
            e = kb8[rows, kvh, h % group].double()
            w = torch.exp2(e - e.max())
            truth[b, h] = (w[:, None] * vb8[rows, kvh].double()).sum(0) / w.sum()
            mag[b, h] = (w[:, None] * vb8[rows, kvh].double().abs()).sum(0) / w.sum()
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV) if splits > 1 else None
    ops.decode_attention_paged(q.to(DEV), kb8.to(DEV), vb8.to(DEV), o, r2t.to(DEV), rpi.to(DEV), seq.to(DEV), logits, splits,
                               sm, 0.0)
    got = o.float().cpu()
    t16 = truth.float().bfloat16().float()  # the correctly rounded answer
    ulp = 2.0 ** -7  # one bf16 ulp relative to the element (8 significant bits)
    # one output ulp, plus the fp32 noise of a sum whose terms (random signs) may cancel: 2^-16 of sum p|v| / sum p (measured
    # on the GPU: up to 2^-17.4 -- the online-softmax rescales go through v_exp_f32, ~1 ulp each, and compound over the
    # tiles) -- still three to four orders of magnitude below what a wrongly rounded P would cause (2^-4 per probability)
    slack = (2.0 ** -16 * mag).float()
    for name, want in (("closed form", t16), ("oracle", ref.float())):
        diff = (got - want).abs()
        assert bool((diff <= ulp * want.abs() + slack).all()), (name, float(((diff - ulp * want.abs()) / mag.float()).max()))
        # ... and nearly all elements are bit-equal (a 1-ulp step needs the fp32 value to sit on a rounding boundary)
        assert float((diff == 0).float().mean()) > 0.98, (name, float((diff == 0).float().mean()))


def test_fp8_pool_unsupported_shapes_raise():
    q = torch.zeros(1, 2, 80, dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros(9, 2, 80, dtype=torch.float8_e4m3fn, device=DEV)
    with pytest.raises(NotImplementedError):
        ops.decode_attention_paged(q, kb, kb, torch.zeros_like(q), torch.zeros(1, 4, dtype=torch.int32, device=DEV),
                                   torch.zeros(1, dtype=torch.int64, device=DEV), torch.ones(1, dtype=torch.int64, device=DEV),
                                   None, 1, 1.0, 0.0)
    with pytest.raises(NotImplementedError):  # mixed pool formats
        ops.decode_attention_paged(torch.zeros(1, 2, 128, dtype=torch.bfloat16, device=DEV),
                                   torch.zeros(9, 2, 128, dtype=torch.float8_e5m2, device=DEV),
                                   torch.zeros(9, 2, 128, dtype=torch.float8_e4m3fn, device=DEV),
                                   torch.zeros(1, 2, 128, dtype=torch.bfloat16, device=DEV),
                                   torch.zeros(1, 4, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
                                   torch.ones(1, dtype=torch.int64, device=DEV), None, 1, 1.0, 0.0)
    with pytest.raises(NotImplementedError):  # head size 80 on an e5m2 pool
        kb5 = torch.zeros(9, 2, 80, dtype=torch.float8_e5m2, device=DEV)
        ops.decode_attention_paged(q, kb5, kb5, torch.zeros_like(q), torch.zeros(1, 4, dtype=torch.int32, device=DEV),
                                   torch.zeros(1, dtype=torch.int64, device=DEV), torch.ones(1, dtype=torch.int64, device=DEV),
                                   None, 1, 1.0, 0.0)


@pytest.mark.parametrize("kv_dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
def test_backend_prefill_then_decode_with_fp8_pool(kv_dtype):
    """MI355AttnBackend on an FP8 pool (e4m3fn, e5m2): EXTEND without a cached prefix (16-bit kernel on the new tokens, K/V cast into
    the pool), DECODE steps reading the byte rows, then an EXTEND over a cached prefix in the pool."""
    from sglang_npu_amd.attention_backend import MI355AttnBackend
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        RadixAttention, ReqToTokenPool, ServerArgs)
    B, Hq, Hkv, D, max_len = 2, 8, 2, 128, 200
    cfg = ModelConfig(Hq, Hkv, D, Hq * D, 4 * Hq * D, 1, 1000, max_len)
    r2t = ReqToTokenPool(B, max_len, DEV)
    n_tok = B * max_len + 1
    pool = MHATokenToKVPool(n_tok, 1, kv_dtype, Hkv, D, 1, DEV)
    assert pool.k_buffer[0].dtype == torch.uint8 and pool.get_key_buffer(0).dtype == kv_dtype
    g = torch.Generator(device=DEV).manual_seed(0)
    r2t.req_to_token.copy_((torch.randperm(n_tok - 1, device=DEV, generator=g) + 1)[: B * max_len].view(B, max_len).int())
    backend = MI355AttnBackend(ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs()))
    layer = RadixAttention(Hq, D, D ** -0.5, Hkv, layer_id=0)
    ext = torch.tensor([90, 41], device=DEV)
    seq, prefix = ext.clone(), torch.zeros_like(ext)
    T = int(ext.sum())
    q = torch.randn(T, Hq * D, device=DEV, generator=g).bfloat16()
    k = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    v = torch.randn(T, Hkv * D, device=DEV, generator=g).bfloat16()
    rpi = torch.arange(B, device=DEV)
    start = torch.tensor([0, 90], device=DEV)
    loc = torch.cat([r2t.req_to_token[b, :seq[b]] for b in range(B)]).long()
    fb = ForwardBatch(ForwardMode.EXTEND, B, None, rpi, seq, loc, int(seq.sum()), seq.cpu(), None, extend_num_tokens=T,
                      extend_seq_lens=ext, extend_prefix_lens=prefix, extend_start_loc=start,
                      extend_prefix_lens_cpu=[0, 0], extend_seq_lens_cpu=ext.tolist(), req_to_token_pool=r2t,
                      token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    o = layer(q, k, v, fb)
    assert torch.isfinite(o.float()).all()
    kb_ref = torch.zeros(n_tok + 1, Hkv, D, dtype=torch.uint8)
    vb_ref = torch.zeros_like(kb_ref)
    oracle.set_kv_buffer_fp8(kb_ref, vb_ref, k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D), loc.cpu(), kv_dtype=kv_dtype)
    assert torch.equal(pool.k_buffer[0].cpu(), kb_ref) and torch.equal(pool.v_buffer[0].cpu(), vb_ref)
    # decode
    seq = seq + 1
    locd = r2t.req_to_token[rpi, seq - 1].long()
    qd = torch.randn(B, Hq * D, device=DEV, generator=g).bfloat16()
    kd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    vd = torch.randn(B, Hkv * D, device=DEV, generator=g).bfloat16()
    fb = ForwardBatch(ForwardMode.DECODE, B, None, rpi, seq, locd, int(seq.sum()), seq.cpu(), seq - 1,
                      req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    od = layer(qd, kd, vd, fb)
    oracle.set_kv_buffer_fp8(kb_ref, vb_ref, kd.cpu().view(B, Hkv, D), vd.cpu().view(B, Hkv, D), locd.cpu(), kv_dtype=kv_dtype)
    assert torch.equal(pool.k_buffer[0].cpu(), kb_ref)
    truth, ref = torch.zeros(B, Hq, D, dtype=torch.bfloat16), torch.zeros(B, Hq, D, dtype=torch.bfloat16)
    args = (qd.cpu().view(B, Hq, D), kb_ref, vb_ref)
    oracle.decode_attention_fp8kv(*args, truth, torch.zeros(B, Hq, 1, D + 1), r2t.req_to_token.cpu(), rpi.cpu(), seq.cpu(),
                                  D ** -0.5, p_fp8=False, kv_dtype=kv_dtype)
    oracle.decode_attention_fp8kv(*args, ref, torch.zeros(B, Hq, 1, D + 1), r2t.req_to_token.cpu(), rpi.cpu(), seq.cpu(),
                                  D ** -0.5, p_fp8=True, kv_dtype=kv_dtype)
    err = float((od.float().cpu().view(B, Hq, D) - truth.float()).abs().max())
    assert err <= 1.5 * float((ref.float() - truth.float()).abs().max()) + 2.0 ** -8 * float(truth.float().abs().max())
    # a cached prefix in the FP8 pool: the prefix stage reads the byte rows (q and p rounded to FP8)
    loc4 = r2t.req_to_token[0, 91:95].long()
    fb = ForwardBatch(ForwardMode.EXTEND, 1, None, rpi[:1], torch.tensor([95], device=DEV), loc4, 95, torch.tensor([95]),
                      None, extend_num_tokens=4, extend_seq_lens=torch.tensor([4], device=DEV),
                      extend_prefix_lens=torch.tensor([91], device=DEV), extend_start_loc=torch.tensor([0], device=DEV),
                      extend_prefix_lens_cpu=[91], extend_seq_lens_cpu=[4], req_to_token_pool=r2t, token_to_kv_pool=pool,
                      attn_backend=backend)
    backend.init_forward_metadata(fb)
    oe = layer(q[:4], k[:4], v[:4], fb)
    oracle.set_kv_buffer_fp8(kb_ref, vb_ref, k[:4].cpu().view(4, Hkv, D), v[:4].cpu().view(4, Hkv, D), loc4.cpu(), kv_dtype=kv_dtype)
    assert torch.equal(pool.k_buffer[0].cpu(), kb_ref)
    outs = []
    for p_fp8 in (False, True):
        o_ref = torch.zeros(4, Hq, D, dtype=torch.bfloat16)
        oracle.extend_attention_fp8kv(q[:4].cpu().view(4, Hq, D), k[:4].cpu().view(4, Hkv, D), v[:4].cpu().view(4, Hkv, D),
                                      o_ref, kb_ref, vb_ref, r2t.req_to_token.cpu(), torch.tensor([0]), torch.tensor([95]),
                                      torch.tensor([4]), torch.tensor([0]), D ** -0.5, p_fp8=p_fp8, kv_dtype=kv_dtype)
        outs.append(o_ref.float())
    err = float((oe.float().cpu().view(4, Hq, D) - outs[0]).abs().max())
    assert err <= 1.5 * float((outs[1] - outs[0]).abs().max()) + 2.0 ** -8 * float(outs[0].abs().max())


def _extend_case(g, B, Hq, Hkv, D, prefix, ext, dtype=torch.bfloat16):
    max_len = int(max(p + e for p, e in zip(prefix, ext)))
    n_tok = B * max_len + 1
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).view(B, max_len).int()
    T = int(sum(ext))
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    k = torch.randn(T, Hkv, D, generator=g).to(dtype)
    v = torch.randn(T, Hkv, D, generator=g).to(dtype)
    ext_t, pre_t = torch.tensor(ext), torch.tensor(prefix)
    start = torch.cumsum(ext_t, 0) - ext_t
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(ext_t, 0)
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(pre_t, 0)
    kv_indices = torch.cat([r2t[b, :prefix[b]] for b in range(B)] + [torch.zeros(0, dtype=torch.int32)])
    return dict(kb=kb, vb=vb, r2t=r2t, q=q, k=k, v=v, ext=ext_t, seq=ext_t + pre_t, start=start, qo_indptr=qo_indptr,
                kv_indptr=kv_indptr, kv_indices=kv_indices, rpi=torch.arange(B))


def _run_extend(c, D, **kw):
    o = torch.zeros(c["q"].shape, dtype=c["q"].dtype, device=DEV)
    d = lambda t: t.to(DEV)
    mask = kw.pop("custom_mask", None)
    mi = kw.pop("mask_indptr", None)
    ops.extend_attention_fwd(d(c["q"]), d(c["k"]), d(c["v"]), o, d(c["kb"]), d(c["vb"]), d(c["qo_indptr"]), d(c["kv_indptr"]),
                             d(c["kv_indices"]), d(mask) if mask is not None else None, True,
                             d(mi) if mi is not None else None, int(c["ext"].max()), D ** -0.5, **kw)
    return o.float().cpu()


def _oracle_extend(c, D, **kw):
    o = torch.zeros(c["q"].shape, dtype=c["q"].dtype)
    oracle.extend_attention_fp8kv(c["q"], c["k"], c["v"], o, c["kb"].view(torch.uint8), c["vb"].view(torch.uint8), c["r2t"],
                                  c["rpi"], c["seq"], c["ext"], c["start"], D ** -0.5, **kw)
    return o.float()


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (14, 2, 64), (6, 6, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_extend_fp8_prefix_vs_oracle(Hq, Hkv, D, dtype):
    g = torch.Generator().manual_seed(Hq * 3 + D)
    c = _extend_case(g, 4, Hq, Hkv, D, prefix=[300, 0, 64, 517], ext=[70, 33, 1, 129], dtype=dtype)
    o = _run_extend(c, D)
    truth = _oracle_extend(c, D, p_fp8=False)
    ref = _oracle_extend(c, D, p_fp8=True)
    err_hip, err_ref = (o - truth).abs(), (ref - truth).abs()
    scale = float(truth.abs().max())
    # same rounding points as the oracle (Q -> fp8, P -> fp8 per 64-key block); exp2 vs expf decide single roundings
    assert float(err_hip.pow(2).mean().sqrt()) <= 1.5 * float(err_ref.pow(2).mean().sqrt()) + 2.0 ** -9 * scale
    assert float(err_hip.max()) <= 1.5 * float(err_ref.max()) + 2.0 ** -8 * scale
    # the request without a prefix never touches the pool: plain 16-bit accuracy
    rows = slice(70, 103)
    assert float(err_hip[rows].max()) <= 2.0 ** -7 * scale


def test_extend_fp8_rounds_q_like_the_reference():
    """One cached key (p = 1 exactly, so no fp8-P noise): the result depends on the prefix logit q8 . k alone and must
    follow the oracle WITH Q rounded to FP8 (extend_attention.py:149), not the unrounded one."""
    g = torch.Generator().manual_seed(5)
    Hq, Hkv, D = 8, 2, 128
    c = _extend_case(g, 3, Hq, Hkv, D, prefix=[1, 1, 1], ext=[1, 2, 40])
    c["q"] = (c["q"].float() * 1.7).bfloat16()
    o = _run_extend(c, D)
    with_q8 = _oracle_extend(c, D, q_fp8=True, p_fp8=True)
    without = _oracle_extend(c, D, q_fp8=False, p_fp8=True)
    e_with, e_without = float((o - with_q8).abs().max()), float((o - without).abs().max())
    assert e_with <= 2.0 ** -7 * float(with_q8.abs().max())  # 16-bit output rounding only
    assert e_without > 4 * e_with


def test_extend_fp8_long_prefix_window_and_mask():
    """Prefix longer than one page-table pass (4096 entries), sliding window on the prefix, custom mask on both parts."""
    g = torch.Generator().manual_seed(9)
    Hq, Hkv, D = 4, 1, 128
    c = _extend_case(g, 2, Hq, Hkv, D, prefix=[4300, 130], ext=[40, 70])
    for kw in (dict(), dict(sliding_window_size=100)):
        o = _run_extend(c, D, **kw)
        truth, ref = _oracle_extend(c, D, p_fp8=False, **kw), _oracle_extend(c, D, p_fp8=True, **kw)
        scale = float(truth.abs().max())
        assert float((o - truth).abs().max()) <= 1.5 * float((ref - truth).abs().max()) + 2.0 ** -8 * scale
    # custom mask over [ext][prefix + ext], a random subset of the causal mask that keeps the diagonal
    masks, indptr = [], [0]
    for b in range(2):
        pre, ext = int(c["seq"][b] - c["ext"][b]), int(c["ext"][b])
        m = torch.rand(ext, pre + ext, generator=g) < 0.6
        m[:, pre:] &= torch.tril(torch.ones(ext, ext, dtype=torch.bool))
        m[torch.arange(ext), pre + torch.arange(ext)] = True
        masks.append(m.reshape(-1))
        indptr.append(indptr[-1] + m.numel())
    mask, mi = torch.cat(masks), torch.tensor(indptr, dtype=torch.int64)
    kw = dict(custom_mask=mask, mask_indptr=mi, skip_prefix_custom_mask=False)
    o = _run_extend(c, D, **kw)
    truth, ref = _oracle_extend(c, D, p_fp8=False, **kw), _oracle_extend(c, D, p_fp8=True, **kw)
    scale = float(truth.abs().max())
    assert float((o - truth).abs().max()) <= 1.5 * float((ref - truth).abs().max()) + 2.0 ** -8 * scale
