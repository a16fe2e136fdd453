"""GPU: ragged prefix + extend attention (HIP, through the C ABI) against the golden vectors, the
oracle and size-independent properties."""
import pytest
import torch

import oracle
from conftest import assert_elem_close, count_beyond, golden_names, load_golden, p_rounding_term
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", golden_names("extend_"))
def test_extend_op_form_vs_golden(name):
    g = load_golden(name)
    B, Hq, Hkv, D, Dv, max_len_extend = [int(x) for x in g["meta"]]
    keys = ["q_extend", "k_extend", "v_extend", "k_buffer", "v_buffer", "req_to_token", "req_pool_indices", "seq_lens",
            "extend_seq_lens", "extend_start_loc"]
    t = {k: g[k].to(DEV) for k in keys}
    o = torch.zeros(g["q_extend"].size(0), Hq, Dv, dtype=t["q_extend"].dtype, device=DEV)
    ops.extend_attention(t["q_extend"], t["k_extend"], t["v_extend"], o, t["k_buffer"], t["v_buffer"],
                         t["req_to_token"], t["req_pool_indices"], t["seq_lens"],
                         t["extend_seq_lens"].to(torch.int32), t["extend_start_loc"].to(torch.int32), max_len_extend,
                         g["sm_scale"], 0.0)
    torch.cuda.synchronize()
    # per element: 1e-3 + ulp |ref_i| + the P-rounding term (conftest.p_rounding_term: the probabilities enter the PV MFMA in
    # the 16-bit type, here as in the reference's extend.cpp, whose own output is 0.1-0.3 % of elements beyond the strict
    # bound on these fixtures -- tests/test_oracle_golden.py keeps that statement checked)
    a = torch.zeros(g["q_extend"].size(0), Hq, Dv, dtype=g["q_extend"].dtype)
    oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"].abs(), a, g["k_buffer"], g["v_buffer"].abs(),
                            g["req_to_token"], g["req_pool_indices"], g["seq_lens"], g["extend_seq_lens"],
                            g["extend_start_loc"], max_len_extend, g["sm_scale"], 0.0, p_round=False)
    assert_elem_close(o, g["o_f32"], g["dtype"], what=f"{name}: hip vs the fp32 truth", extra=p_rounding_term(g["dtype"], a))
    if g["ref_valid"]:  # the compiled reference kernel's own 16-bit output: two rounded results
        assert_elem_close(o, g["o_ref"], g["dtype"], pair=True, what=f"{name}: hip vs the reference kernel",
                          extra=p_rounding_term(g["dtype"], a, pair=True))
        # ... and under the STRICT bound the HIP kernel leaves no more elements outside than the reference kernel does
        n_ref, n_hip = count_beyond(g["o_ref"], g["o_f32"], g["dtype"]), count_beyond(o, g["o_f32"], g["dtype"])
        assert n_hip <= n_ref + 8, f"{name}: {n_hip} hip elements beyond 1e-3 + ulp |f32_i|, the reference kernel has {n_ref}"


def _case(B, Hq, Hkv, D, max_prefix, max_ext, dtype, seed, zero_prefix=False, pin_first_prefix=False):
    g = torch.Generator().manual_seed(seed)
    prefix = torch.randint(0, max_prefix + 1, (B,), generator=g)
    if zero_prefix:
        prefix.zero_()
    if pin_first_prefix:  # the first request has the longest prefix the case allows
        prefix[0] = max_prefix
    ext = torch.randint(1, max_ext + 1, (B,), generator=g)
    ext[0] = max_ext
    seq = prefix + ext
    n_tok = int(seq.sum()) + 4
    perm = torch.randperm(n_tok - 1, generator=g) + 1
    r2t = torch.zeros(B, int(seq.max()), dtype=torch.int32)
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    T = int(ext.sum())
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    ke, ve = torch.empty(T, Hkv, D, dtype=dtype), torch.empty(T, Hkv, D, dtype=dtype)
    start = torch.zeros(B, dtype=torch.int64)
    start[1:] = torch.cumsum(ext[:-1], 0)
    off = 0
    for b in range(B):
        L, p, e, s0 = int(seq[b]), int(prefix[b]), int(ext[b]), int(start[b])
        toks = perm[off:off + L]
        off += L
        r2t[b, :L] = toks.to(torch.int32)
        ke[s0:s0 + e], ve[s0:s0 + e] = kb[toks[p:]], vb[toks[p:]]
    return dict(q=q, ke=ke, ve=ve, kb=kb, vb=vb, r2t=r2t, rpi=torch.arange(B), seq=seq, ext=ext, start=start, prefix=prefix)


def _p_term(c, dtype, scale, cap=0.0, pair=True, **kw):
    """conftest.p_rounding_term for a _case: the oracle's attention of |V| with unrounded probabilities."""
    T, Hq, D = c["q"].shape
    a = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"].abs(), a, c["kb"], c["vb"].abs(), c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), scale, cap, p_round=False, **kw)
    return p_rounding_term(dtype, a, pair=pair)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (32, 32, 128), (14, 2, 64), (6, 2, 128), (3, 1, 64)])
@pytest.mark.parametrize("zero_prefix", [False, True])
def test_extend_backend_form_vs_oracle(Hq, Hkv, D, zero_prefix):
    B = 4
    dtype = torch.bfloat16
    c = _case(B, Hq, Hkv, D, 300, 260, dtype, seed=Hq * 7 + D + int(zero_prefix), zero_prefix=zero_prefix)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0)
    d = {k: v.to(DEV) for k, v in c.items()}
    # Triton-form metadata exactly as triton_backend.py:284-312 builds it
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    kv_indptr[1:] = torch.cumsum(d["prefix"], 0)
    kv_indices = torch.empty(max(int(c["prefix"].sum()), 1), dtype=torch.int32, device=DEV)
    ops.create_kv_indices(d["r2t"], d["rpi"], d["prefix"], kv_indptr, None, kv_indices)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    qo_indptr[1:] = torch.cumsum(d["ext"], 0)
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None,
                             True, None, int(c["ext"].max()), D ** -0.5, 0.0)
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle", extra=_p_term(c, dtype, D ** -0.5))


def test_extend_long_prefix_multi_pass_and_fp16_cap():
    # prefix > 4096 page-table entries exercises the multi-pass staging; fp16 + logit cap + int64 table
    B, Hq, Hkv, D = 2, 8, 2, 128
    dtype = torch.float16
    c = _case(B, Hq, Hkv, D, 4500, 70, dtype, seed=3)
    c["prefix"][0] = 4500
    c = _case(B, Hq, Hkv, D, 4500, 70, dtype, seed=4)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), 0.2, 25.0)
    d = {k: v.to(DEV) for k, v in c.items()}
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], d["r2t"].long(), d["rpi"], d["seq"], d["ext"],
                         d["start"], int(c["ext"].max()), 0.2, 25.0)
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle", extra=_p_term(c, dtype, 0.2, 25.0))


def test_extend_full_size_properties():
    """Prefill-sized case (Llama-3-8B geometry, 4 x 2048 new tokens + cached prefixes) on the GPU only:
      (1) chunked prefill is consistent: extending [0:L) in one go equals extending the second half with the
          first half as cached prefix (same keys, different kernel stage) to within one ulp;
      (2) the last row of each request equals single-token decode over the same keys;
      (3) a bounded sample of rows agrees with the oracle."""
    B, Hq, Hkv, D, L = 4, 32, 8, 128, 2048
    dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(5)
    n_tok = B * L + 1
    kb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    r2t = (torch.randperm(n_tok - 1, device=DEV, generator=g) + 1).view(B, L).to(torch.int32)
    q = torch.randn(B * L, Hq, D, device=DEV, generator=g).to(dtype)
    ke = kb[r2t.long().view(-1)].contiguous()
    ve = vb[r2t.long().view(-1)].contiguous()
    rpi = torch.arange(B, device=DEV)
    full = torch.full((B,), L, device=DEV)
    start = torch.arange(B, device=DEV) * L
    o_full = torch.zeros(B * L, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention(q, ke, ve, o_full, kb, vb, r2t, rpi, full, full, start, L, D ** -0.5, 0.0)
    # the P-rounding allowance at this size: the attention of |V| from the kernel itself (the oracle would take minutes; a
    # smooth positive quantity, its own 16-bit rounding is covered by the factor 1.01)
    a_full = torch.zeros_like(o_full)
    ops.extend_attention(q, ke, ve.abs(), a_full, kb, vb.abs(), r2t, rpi, full, full, start, L, D ** -0.5, 0.0)
    a_full = a_full.float() * 1.01
    # (1) second half with the first half cached
    h = L // 2
    sel = (torch.arange(B, device=DEV)[:, None] * L + h + torch.arange(h, device=DEV)[None]).view(-1)
    o_half = torch.zeros(B * h, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention(q[sel].contiguous(), ke[sel].contiguous(), ve[sel].contiguous(), o_half, kb, vb, r2t, rpi,
                         full, torch.full((B,), h, device=DEV), torch.arange(B, device=DEV) * h, h, D ** -0.5, 0.0)
    assert_elem_close(o_half, o_full[sel], dtype, pair=True, what="second half behind a cached first half vs the full pass",
                      extra=p_rounding_term(dtype, a_full[sel], pair=True))
    # (2) last row == decode
    last = start + L - 1
    o_dec = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention(q[last].contiguous(), kb, vb, o_dec, None, None, None, torch.zeros(B, Hq, 1, D + 1, device=DEV),
                         r2t, rpi, full, D ** -0.5, 0.0)
    assert_elem_close(o_dec, o_full[last], dtype, pair=True, what="decode of the last row vs the full pass",
                      extra=p_rounding_term(dtype, a_full[last], pair=True))
    # (3) oracle on one short slice: request 1, rows [0, 96)
    rows = 96
    o_ref = torch.zeros(rows, Hq, D, dtype=dtype)
    oracle.extend_attention(q[L:L + rows].cpu(), ke[L:L + rows].cpu(), ve[L:L + rows].cpu(), o_ref, kb.cpu(), vb.cpu(),
                            r2t.cpu(), torch.tensor([1]), torch.tensor([rows]), torch.tensor([rows]), torch.tensor([0]),
                            rows, D ** -0.5, 0.0)
    assert_elem_close(o_full[L:L + rows], o_ref, dtype, pair=True, what="full size vs oracle on one slice",
                      extra=p_rounding_term(dtype, a_full[L:L + rows], pair=True))


def _triton_meta(d, B):
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    kv_indptr[1:] = torch.cumsum(d["prefix"], 0)
    kv_indices = torch.empty(max(int(d["prefix"].sum()), 1), dtype=torch.int32, device=DEV)
    ops.create_kv_indices(d["r2t"], d["rpi"], d["prefix"], kv_indptr, None, kv_indices)
    qo_indptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    qo_indptr[1:] = torch.cumsum(d["ext"], 0)
    return qo_indptr, kv_indptr, kv_indices


def _masks(c, kind, g):
    """custom_mask as the reference's own test builds it (test_triton_attention_kernels.py:125-141): per request
    [ext][prefix+ext] flattened; `causal` = all-ones prefix + lower triangle, `tree` = a random subset of it that keeps
    the diagonal (every row sees itself, like a token-tree ancestor mask)."""
    B = c["seq"].numel()
    lens = c["ext"] * c["seq"]
    indptr = torch.zeros(B + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(lens, 0)
    mask = torch.zeros(int(lens.sum()), dtype=torch.bool)
    for b in range(B):
        e, p = int(c["ext"][b]), int(c["prefix"][b])
        tri = torch.tril(torch.ones(e, e, dtype=torch.bool))
        pre = torch.ones(e, p, dtype=torch.bool)
        if kind == "tree":
            tri = tri & (torch.rand(e, e, generator=g) < 0.6) | torch.eye(e, dtype=torch.bool)
            pre = pre & (torch.rand(e, p, generator=g) < 0.7)
        mask[indptr[b]:indptr[b + 1]] = torch.cat([pre, tri], 1).flatten()
    return mask, indptr


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (14, 2, 64), (4, 4, 80)])  # MFMA x2, generic kernel
@pytest.mark.parametrize("kind,skip_prefix", [("causal", True), ("causal", False), ("tree", True), ("tree", False)])
def test_extend_custom_mask(Hq, Hkv, D, kind, skip_prefix):
    """custom_mask (extend_attention.py:171-183, 246-259).  A mask equal to the causal mask must reproduce the unmasked
    result (the reference's own check, test_triton_attention_kernels.py:153-172); random tree masks are checked
    against the oracle's restatement."""
    B, dtype = 3, torch.bfloat16
    c = _case(B, Hq, Hkv, D, 150, 90, dtype, seed=Hq + D + len(kind) + int(skip_prefix))
    g = torch.Generator().manual_seed(5)
    mask, mask_indptr = _masks(c, kind, g)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0, custom_mask=mask, mask_indptr=mask_indptr,
                            skip_prefix_custom_mask=skip_prefix)
    d = {k: v.to(DEV) for k, v in c.items()}
    qo_indptr, kv_indptr, kv_indices = _triton_meta(d, B)
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices,
                             mask.to(DEV), True, mask_indptr.to(DEV), int(c["ext"].max()), D ** -0.5, 0.0, skip_prefix)
    term = _p_term(c, dtype, D ** -0.5, custom_mask=mask, mask_indptr=mask_indptr, skip_prefix_custom_mask=skip_prefix)
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle (custom mask)", extra=term)
    if kind == "causal":
        o_plain = torch.zeros_like(o)
        ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o_plain, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices,
                                 None, True, None, int(c["ext"].max()), D ** -0.5, 0.0)
        # same math; the unmasked launch may take the key-split variant (other summation order), so not bit-equal
        assert_elem_close(o, o_plain, dtype, pair=True, what="all-visible mask vs no mask", extra=term)


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (6, 2, 128), (4, 4, 80)])
@pytest.mark.parametrize("window", [1, 37, 100000])
def test_extend_sliding_window(Hq, Hkv, D, window):
    """SLIDING_WINDOW_SIZE (extend_attention.py:184-189): prefix key n is visible to extend row q iff q <= n + W.
    A window larger than every extend length changes nothing."""
    B, dtype = 3, torch.float16
    c = _case(B, Hq, Hkv, D, 200, 120, dtype, seed=window % 97 + D)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0, sliding_window_size=window)
    d = {k: v.to(DEV) for k, v in c.items()}
    qo_indptr, kv_indptr, kv_indices = _triton_meta(d, B)
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None,
                             True, None, int(c["ext"].max()), D ** -0.5, 0.0, True, window)
    term = _p_term(c, dtype, D ** -0.5, sliding_window_size=window)
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle", extra=term)
    if window >= 100000:
        o_plain = torch.zeros_like(o)
        ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o_plain, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices,
                                 None, True, None, int(c["ext"].max()), D ** -0.5, 0.0)
        assert_elem_close(o, o_plain, dtype, pair=True, what="window larger than every length vs no window", extra=term)


def test_extend_non_causal_vs_oracle():
    """is_causal=False (ENCODER_ONLY, triton_backend.py:651-653): every extend key is visible."""
    B, Hq, Hkv, D, dtype = 2, 8, 2, 128, torch.bfloat16
    c = _case(B, Hq, Hkv, D, 60, 100, dtype, seed=11)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0, causal=False)
    d = {k: v.to(DEV) for k, v in c.items()}
    qo_indptr, kv_indptr, kv_indices = _triton_meta(d, B)
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None,
                             False, None, int(c["ext"].max()), D ** -0.5, 0.0)
    assert_elem_close(o, o_ref, dtype, pair=True, what="hip vs oracle", extra=_p_term(c, dtype, D ** -0.5, causal=False))


def test_extend_mask_argument_checks():
    x = torch.zeros(1, 1, 64, dtype=torch.bfloat16, device=DEV)
    i = torch.zeros(2, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="mask_indptr"):
        ops.extend_attention_fwd(x, x, x, x, x, x, i, i, i, torch.zeros(1, dtype=torch.bool, device=DEV), True, None, 1)
