"""CPU: the oracle (oracle/sgl_oracle.c) against every committed golden vector.

The fixtures were produced in the build container by tests/golden/make_golden.py from the
reference's own compiled CPU kernels and the torch references inside the reference's tests.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import assert_elem_close, count_beyond, golden_names, load_golden, p_rounding_term


@pytest.mark.parametrize("name", golden_names("decode_"))
def test_decode_oracle_vs_reference(name):
    g = load_golden(name)
    B, Hq, Hkv, D, Dv, S, splits = [int(x) for x in g["meta"]]
    kb, vb = g["k_buffer"].clone(), g["v_buffer"].clone()
    o = torch.zeros(B, Hq, Dv, dtype=g["q"].dtype)
    logits = torch.zeros(B, Hq, splits, Dv + 1)
    oracle.decode_attention(g["q"], kb, vb, o, g["key"], g["value"], g["loc"], logits, g["req_to_token"],
                            g["req_pool_indices"], g["seq_lens"], g["sm_scale"], g["logit_cap"])
    # KV write is a byte copy: exact
    exp_k, exp_v = g["k_buffer"].clone(), g["v_buffer"].clone()
    exp_k[g["loc"]] = g["key"]
    exp_v[g["loc"]] = g["value"]
    assert torch.equal(kb.view(torch.int16), exp_k.view(torch.int16))
    assert torch.equal(vb.view(torch.int16), exp_v.view(torch.int16))
    assert_elem_close(o, g["o_f32"], g["dtype"], what=f"{name}: oracle vs the fp32 truth")
    if g["ref_valid"]:
        assert_elem_close(o, g["o_ref"], g["dtype"], pair=True, what=f"{name}: oracle vs the compiled reference kernel")


@pytest.mark.parametrize("name", golden_names("decode_"))
def test_decode_blocked_port_vs_reference_and_token_loop(name):
    """The BLOCKED form of the decode oracle (what bench.py times as cpu_baseline, structure of decode.cpp:942-985)
    against the same reference fixtures, and against the token-at-a-time oracle (same arithmetic up to the fp32
    summation order: a 16-bit output element may sit one rounding apart on a few elements, never more)."""
    g = load_golden(name)
    B, Hq, Hkv, D, Dv, S, splits = [int(x) for x in g["meta"]]
    outs = []
    for blocked in (False, True):
        kb, vb = g["k_buffer"].clone(), g["v_buffer"].clone()
        o = torch.zeros(B, Hq, Dv, dtype=g["q"].dtype)
        oracle.decode_attention(g["q"], kb, vb, o, g["key"], g["value"], g["loc"], torch.zeros(B, Hq, splits, Dv + 1),
                                g["req_to_token"], g["req_pool_indices"], g["seq_lens"], g["sm_scale"], g["logit_cap"],
                                blocked=blocked)
        outs.append(o.float())
    assert_elem_close(outs[1], g["o_f32"], g["dtype"], what=f"{name}: blocked oracle vs the fp32 truth")
    if g["ref_valid"]:
        assert_elem_close(outs[1], g["o_ref"], g["dtype"], pair=True, what=f"{name}: blocked oracle vs the compiled reference")
    ulp = 2.0 ** -7 if g["dtype"] == "bf16" else 2.0 ** -10
    d = (outs[1] - outs[0]).abs()
    assert bool((d <= ulp * outs[0].abs() + 1e-6).all()), float(d.max())
    assert float((d == 0).float().mean()) > 0.97


@pytest.mark.parametrize("name", golden_names("extend_"))
def test_extend_oracle_vs_reference(name):
    g = load_golden(name)
    B, Hq, Hkv, D, Dv, max_len_extend = [int(x) for x in g["meta"]]
    o = torch.zeros(g["q_extend"].size(0), Hq, Dv, dtype=g["q_extend"].dtype)
    oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"], o, g["k_buffer"], g["v_buffer"],
                            g["req_to_token"], g["req_pool_indices"], g["seq_lens"], g["extend_seq_lens"],
                            g["extend_start_loc"], max_len_extend, g["sm_scale"], 0.0)
    # The reference's extend kernel rounds the softmax numerators to the 16-bit type for its PV GEMM; its own output
    # (o_ref) is 0.1-0.3 % of elements beyond the strict per-element bound of the fp32 truth on these fixtures (checked
    # below, so the statement stays true).  The per-element allowance for that rounding is derived, not fitted:
    # half an ulp of every p_j times |v_j|, i.e. the same attention evaluated on |V| (conftest.p_rounding_term).
    a = torch.zeros_like(o)
    oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"].abs(), a, g["k_buffer"], g["v_buffer"].abs(),
                            g["req_to_token"], g["req_pool_indices"], g["seq_lens"], g["extend_seq_lens"],
                            g["extend_start_loc"], max_len_extend, g["sm_scale"], 0.0, p_round=False)
    assert_elem_close(o, g["o_f32"], g["dtype"], what=f"{name}: oracle vs the fp32 truth",
                      extra=p_rounding_term(g["dtype"], a))
    if g["ref_valid"]:
        assert_elem_close(g["o_ref"], g["o_f32"], g["dtype"], what=f"{name}: compiled reference vs the fp32 truth",
                          extra=p_rounding_term(g["dtype"], a))
        assert_elem_close(o, g["o_ref"], g["dtype"], pair=True, what=f"{name}: oracle vs the compiled reference kernel",
                          extra=p_rounding_term(g["dtype"], a, pair=True))
        # the restatement is no further from the truth than the kernel it restates (strict bound, element counts)
        n_ref, n_orc = count_beyond(g["o_ref"], g["o_f32"], g["dtype"]), count_beyond(o, g["o_f32"], g["dtype"])
        assert n_orc <= n_ref + 8, f"{name}: {n_orc} oracle elements beyond the strict bound, the reference kernel has {n_ref}"


def test_kv_indices_exact():
    z = np.load("tests/golden/kv_indices.npz") if False else None
    g = load_golden("kv_indices")
    for i in range(int(g["n"])):
        max_batch, max_ctx = [int(x) for x in g[f"shape{i}"]]
        r2t = torch.arange(max_batch * max_ctx, dtype=torch.int32).reshape(max_batch, max_ctx)
        out = torch.empty(int(g[f"indptr{i}"][-1]), dtype=torch.int32)
        oracle.create_kv_indices(r2t, g[f"rpi{i}"], g[f"lens{i}"], g[f"indptr{i}"], None, out)
        assert torch.equal(out, g[f"kv_indices{i}"])


def _h(arr, dtype):
    return arr.view(torch.int16).view(torch.bfloat16 if dtype == "bf16" else torch.float16)


def test_per_token_quant_fp8_bit_exact():
    z = np.load("tests/golden/per_token_quant_fp8.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        x = _h(torch.from_numpy(z[f"x{i}"].view(np.int16).copy()), dtype)
        q = torch.empty(x.shape, dtype=torch.uint8)
        s = torch.empty(x.size(0), dtype=torch.float32)
        oracle.per_token_quant_fp8(x, q, s)
        assert torch.equal(s, torch.from_numpy(z[f"s{i}"]))
        assert torch.equal(q, torch.from_numpy(z[f"q{i}"]))


def test_fp8_scaled_mm_vs_reference_torch():
    z = np.load("tests/golden/fp8_scaled_mm.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        dt = torch.bfloat16 if dtype == "bf16" else torch.float16
        a = torch.from_numpy(z[f"a{i}"]).view(torch.float8_e4m3fn)
        b = torch.from_numpy(z[f"b{i}"]).view(torch.float8_e4m3fn)
        bias = _h(torch.from_numpy(z[f"bias{i}"].view(np.int16).copy()), dtype) if f"bias{i}" in z.files else None
        o = oracle.fp8_scaled_mm(a, b.t(), torch.from_numpy(z[f"sa{i}"]), torch.from_numpy(z[f"sb{i}"]), dt, bias,
                                 bias_after_round=True)
        ref = _h(torch.from_numpy(z[f"o{i}"].view(np.int16).copy()), dtype)
        # reference tolerance: rtol 0.02 / atol 1 (sgl-kernel/tests/test_fp8_gemm.py:33-35); ours: 1 ulp
        torch.testing.assert_close(o.float(), ref.float(), rtol=2.0 ** -7, atol=2e-2)


def test_awq_dequant_exact_and_gemm():
    z = np.load("tests/golden/awq.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        qw, qz = torch.from_numpy(z[f"qweight{i}"]), torch.from_numpy(z[f"qzeros{i}"])
        sc = _h(torch.from_numpy(z[f"scales{i}"].view(np.int16).copy()), dtype)
        w = oracle.awq_dequantize(qw, sc, qz)
        assert torch.equal(w.view(torch.int16), torch.from_numpy(z[f"w{i}"].view(np.int16).copy()))
        x = _h(torch.from_numpy(z[f"x{i}"].view(np.int16).copy()), dtype)
        y = oracle.awq_gemm(x, qw, sc, qz)
        torch.testing.assert_close(y.float(), torch.from_numpy(z[f"y_f32_{i}"]),
                                   rtol=2.0 ** (-7 if dtype == "bf16" else -10), atol=1e-2)


def test_scalar_converters_match_torch():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * 100, torch.randn(4096, generator=g) * 1e-2,
                   torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 240.0, 464.0])])
    xc = x.clamp(-448, 448)
    lib = oracle.load()
    import ctypes
    y = torch.empty(x.numel(), dtype=torch.uint8)
    lib.orc_cvt_f32_to_e4m3(ctypes.c_void_p(xc.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_int64(x.numel()))
    assert torch.equal(y, xc.to(torch.float8_e4m3fn).view(torch.uint8))
    # the KV-pool cast is torch's UNclamped one (memory_pool.py:385-394): NaN, infinities and |x| > 464 give NaN
    xe = torch.cat([x, torch.tensor([float("nan"), -float("nan"), float("inf"), -float("inf"), 449.0, 463.9, 464.0,
                                     -464.0, 464.1, -465.0, 479.9, 480.0, 1e6, -1e30])])
    ye = torch.empty(xe.numel(), dtype=torch.uint8)
    lib.orc_cvt_f32_to_e4m3_torch(ctypes.c_void_p(xe.data_ptr()), ctypes.c_void_p(ye.data_ptr()), ctypes.c_int64(xe.numel()))
    ref_e = xe.to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(ye & 0x7f, ref_e & 0x7f)                       # magnitudes incl. the NaN code 0x7f
    fin = ~torch.isnan(xe)
    assert torch.equal(ye[fin], ref_e[fin])                           # the sign too wherever the input is a number
    back = torch.empty(256, dtype=torch.float32)
    allb = torch.arange(256, dtype=torch.uint8)
    lib.orc_cvt_e4m3_to_f32(ctypes.c_void_p(allb.data_ptr()), ctypes.c_void_p(back.data_ptr()), ctypes.c_int64(256))
    ref = allb.view(torch.float8_e4m3fn).float()
    assert torch.equal(torch.nan_to_num(back, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))
    for code, dt in ((0, torch.bfloat16), (1, torch.float16)):
        h = torch.empty(x.numel(), dtype=torch.int16)
        xs = x * (1e-3 if code else 1.0)
        lib.orc_cvt_f32_to_h(ctypes.c_void_p(xs.data_ptr()), ctypes.c_void_p(h.data_ptr()), ctypes.c_int64(x.numel()),
                             ctypes.c_int(code))
        assert torch.equal(h, xs.to(dt).view(torch.int16))


@pytest.mark.parametrize("mode", ["mask_skip_prefix", "mask_full", "window", "non_causal"])
def test_extend_oracle_masks_vs_dense_torch(mode):
    """The optional masks of the Triton extend kernel (extend_attention.py:171-189, 246-259) as restated by the oracle,
    against a dense fp32 softmax with an explicit boolean visibility matrix built straight from those rules."""
    g = torch.Generator().manual_seed(len(mode))
    Hq, Hkv, D, P, E = 4, 2, 32, 23, 17
    dtype = torch.bfloat16
    n_tok = P + E + 3
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(dtype)
    q = torch.randn(E, Hq, D, generator=g).to(dtype)
    toks = torch.randperm(n_tok - 1, generator=g)[:P + E] + 1
    r2t = toks.to(torch.int32).view(1, -1)
    ke, ve = kb[toks[P:]].clone(), vb[toks[P:]].clone()
    vis = torch.cat([torch.ones(E, P, dtype=torch.bool), torch.tril(torch.ones(E, E, dtype=torch.bool))], 1)
    kw = {}
    if mode.startswith("mask"):
        m = vis & (torch.rand(E, P + E, generator=g) < 0.6)
        m[:, P:] |= torch.eye(E, dtype=torch.bool)
        kw = dict(custom_mask=m.flatten(), mask_indptr=torch.tensor([0, E * (P + E)]),
                  skip_prefix_custom_mask=(mode == "mask_skip_prefix"))
        vis = m.clone()
        if mode == "mask_skip_prefix":
            vis[:, :P] = True
    elif mode == "window":
        W = 5
        kw = dict(sliding_window_size=W)
        qi, ni = torch.arange(E)[:, None], torch.arange(P)[None, :]
        vis[:, :P] = qi <= ni + W
    else:
        kw = dict(causal=False)
        vis[:, P:] = True
    o = torch.zeros(E, Hq, D, dtype=dtype)
    oracle.extend_attention(q, ke, ve, o, kb, vb, r2t, torch.tensor([0]), torch.tensor([P + E]), torch.tensor([E]),
                            torch.tensor([0]), E, D ** -0.5, 0.0, **kw)
    K = kb[toks].float().repeat_interleave(Hq // Hkv, 1)  # [P+E, Hq, D]
    V = vb[toks].float().repeat_interleave(Hq // Hkv, 1)
    s = torch.einsum("ehd,nhd->hen", q.float(), K) * D ** -0.5
    s = s.masked_fill(~vis[None], float("-inf"))
    ref = torch.einsum("hen,nhd->ehd", torch.softmax(s, -1), V)
    absv = torch.einsum("hen,nhd->ehd", torch.softmax(s, -1), V.abs())  # the P-rounding allowance, per element
    assert_elem_close(o, ref, "bf16", what="oracle vs an fp32 evaluation", extra=p_rounding_term("bf16", absv))


def test_extend_fp8kv_oracle_reduces_to_the_16bit_oracle():
    """With the roundings switched off, the FP8-pool restatement (blocks of 64 prefix keys) is the 16-bit extend oracle
    on the upcast pool (e4m3 -> bf16 is exact); with them on it differs by fp8 noise only, and Q rounding matters."""
    g = torch.Generator().manual_seed(3)
    B, Hq, Hkv, D = 2, 4, 2, 64
    prefix, ext = torch.tensor([150, 7]), torch.tensor([9, 20])
    seq = prefix + ext
    n_tok = 400
    kb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    vb = torch.randn(n_tok, Hkv, D, generator=g).to(torch.float8_e4m3fn)
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1)[: B * 170].view(B, 170).int()
    T = int(ext.sum())
    q = torch.randn(T, Hq, D, generator=g).bfloat16()
    k = torch.randn(T, Hkv, D, generator=g).bfloat16()
    v = torch.randn(T, Hkv, D, generator=g).bfloat16()
    rpi, start = torch.arange(B), torch.tensor([0, 9])
    o16 = torch.zeros_like(q)
    oracle.extend_attention(q, k, v, o16, kb.bfloat16(), vb.bfloat16(), r2t, rpi, seq, ext, start, int(ext.max()), D ** -0.5, 0.0)
    outs = {}
    for q_fp8, p_fp8 in ((False, False), (True, False), (True, True)):
        o = torch.zeros_like(q)
        oracle.extend_attention_fp8kv(q, k, v, o, kb.view(torch.uint8), vb.view(torch.uint8), r2t, rpi, seq, ext, start,
                                      D ** -0.5, q_fp8=q_fp8, p_fp8=p_fp8)
        outs[(q_fp8, p_fp8)] = o.float()
    scale = float(o16.float().abs().max())
    assert float((outs[(False, False)] - o16.float()).abs().max()) <= 2.0 ** -7 * scale
    d_q = float((outs[(True, False)] - outs[(False, False)]).abs().max())
    d_p = float((outs[(True, True)] - outs[(True, False)]).abs().max())
    assert 0 < d_q < 0.25 * scale and 0 < d_p < 0.1 * scale


# ----------------------------------------------------------------------------- "next" rows: pinned oracles
def _z16(z, key, dtype):
    a = z[key]
    if dtype == "f32":
        return torch.from_numpy(a.copy())
    return _h(torch.from_numpy(a.view(np.int16).copy()), dtype)


def _ulp16(a, b):
    return (a.contiguous().view(torch.int16).int() - b.contiguous().view(torch.int16).int()).abs()


def test_rmsnorm_oracle_vs_reference_fixture():
    """Expected = llama_rms_norm / fused_add_rms_norm of the reference's sgl-kernel/tests/test_norm.py (and the compiled
    rmsnorm_cpu / fused_add_rmsnorm_cpu): at most one 16-bit ulp apart on < 0.5 % of elements (fp32 summation order);
    the residual update is exact."""
    z = np.load("tests/golden/rmsnorm.npz")
    eps = float(z["eps"])
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        x, w, r = (_z16(z, f"{k}{i}", dtype) for k in "xwr")
        for exp_key, res in ((f"y{i}", None), (f"y_cpu{i}", None), (f"y_add{i}", r.clone()), (f"y_add_cpu{i}", r.clone())):
            y = oracle.rmsnorm(x, w, eps, residual=res)
            d = _ulp16(y, _z16(z, exp_key, dtype))
            assert int(d.max()) <= 1 and (d > 0).float().mean().item() < 5e-3, exp_key
            if res is not None:
                assert torch.equal(res.view(torch.int16), _z16(z, f"r_out{i}", dtype).view(torch.int16))


def test_silu_and_mul_oracle_bit_exact_vs_reference_fixture():
    z = np.load("tests/golden/silu_and_mul.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        y = oracle.silu_and_mul(_z16(z, f"x{i}", dtype))
        assert torch.equal(y.view(torch.int16), _z16(z, f"y{i}", dtype).view(torch.int16))
        assert int(_ulp16(y, _z16(z, f"y_cpu{i}", dtype)).max()) <= 2  # the compiled CPU op rounds once, torch twice


def test_rope_neox_oracle_bit_exact_vs_reference_fixture():
    z = np.load("tests/golden/rope_neox.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        hs, rd, T, Hq, Hkv = [int(v) for v in z[f"meta{i}"]]
        pos, cache = torch.from_numpy(z[f"pos{i}"]), torch.from_numpy(z[f"cache{i}"])
        for name, H in (("q", Hq), ("k", Hkv)):
            x = _z16(z, f"{name}{i}", dtype).clone().view(T, H, hs)
            oracle.rope_neox(x, pos, cache, rot_dim=rd)
            assert torch.equal(x.reshape(T, -1).view(torch.int16), _z16(z, f"{name}_out{i}", dtype).view(torch.int16))


def test_merge_state_oracle_vs_reference_fixture():
    z = np.load("tests/golden/merge_state.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        po, so = _z16(z, f"po{i}", dtype), _z16(z, f"so{i}", dtype)
        out, lse = oracle.merge_state(po, torch.from_numpy(z[f"pl{i}"].copy()), so, torch.from_numpy(z[f"sl{i}"].copy()))
        torch.testing.assert_close(lse, torch.from_numpy(z[f"lse{i}"]), rtol=1e-6, atol=1e-6)
        if dtype == "f32":
            torch.testing.assert_close(out, torch.from_numpy(z[f"o_f32_{i}"]), rtol=1e-6, atol=1e-6)
        else:
            d = _ulp16(out, _z16(z, f"o{i}", dtype))
            assert int(d.max()) <= 1 and (d > 0).float().mean().item() < 5e-3


def _gq_input(z, i):
    dtype = z[f"gdtype{i}"].item().decode()
    if dtype == "f32":
        return torch.from_numpy(z[f"gx{i}"].copy()), dtype
    return _h(torch.from_numpy(z[f"gx{i}"].view(np.int16).copy()), dtype), dtype


def test_group_and_tensor_quant_oracle_bit_exact_vs_reference_fixture():
    """tests/golden/group_tensor_quant_fp8.npz: native_per_token_group_quant_fp8 (python/sglang/test/test_block_fp8.py)
    and torch_scaled_fp8_quant (sgl-kernel/tests/test_per_tensor_quant_fp8.py) of the reference."""
    z = np.load("tests/golden/group_tensor_quant_fp8.npz")
    for i in range(int(z["gn"])):
        T, K, G = [int(v) for v in z[f"gmeta{i}"]]
        x, _ = _gq_input(z, i)
        q, s = torch.empty(T, K, dtype=torch.uint8), torch.empty(T, K // G)
        oracle.per_token_group_quant_fp8(x, q, s, G, 1e-10, -448.0, 448.0)
        assert torch.equal(s, torch.from_numpy(z[f"gs{i}"])) and torch.equal(q, torch.from_numpy(z[f"gq{i}"]))
    for i in range(int(z["tn"])):
        dtype = z[f"tdtype{i}"].item().decode()
        x = _h(torch.from_numpy(z[f"tx{i}"].view(np.int16).copy()), dtype)
        q, s = torch.empty(x.shape, dtype=torch.uint8), torch.zeros(1)
        oracle.per_tensor_quant_fp8(x, q, s, False)
        assert torch.equal(s, torch.from_numpy(z[f"ts{i}"])) and torch.equal(q, torch.from_numpy(z[f"tq{i}"]))
        oracle.per_tensor_quant_fp8(x, q, torch.from_numpy(z[f"ts_static{i}"].copy()), True)
        assert torch.equal(q, torch.from_numpy(z[f"tq_static{i}"]))


def test_vocab_parallel_embedding_oracle_vs_reference_fixture():
    """oracle.vocab_parallel_embedding against the fixture made with the reference's own get_masked_input_and_mask and shard
    ranges (vocab_parallel_embedding.py:126-150, 284-330): exact, every rank of every case."""
    z = np.load("tests/golden/vocab_parallel_embedding.npz")
    for i in range(int(z["n"])):
        dt = torch.bfloat16 if z[f"dtype{i}"].item().decode() == "bf16" else torch.float16
        table = torch.from_numpy(z[f"table{i}"].view(np.int16).copy()).view(dt)
        ids = torch.from_numpy(z[f"ids{i}"]).long()
        tp = int(z[f"tp{i}"])
        per = table.shape[0] // tp
        total = torch.zeros(*ids.shape, table.shape[1])
        for r in range(tp):
            o = oracle.vocab_parallel_embedding(ids, table[r * per:(r + 1) * per].contiguous(), int(z[f"start{i}_{r}"]),
                                                int(z[f"end{i}_{r}"]))
            assert torch.equal(o.view(torch.int16), torch.from_numpy(z[f"o{i}_{r}"].view(np.int16).copy()))
            total += o.float()
        assert torch.equal(total, table[ids].float())
