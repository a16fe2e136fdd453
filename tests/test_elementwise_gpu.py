"""GPU: RMSNorm / fused-add RMSNorm / SiLU*mul / RoPE and their FP8-fused forms vs the oracle."""
import pytest
import torch

import oracle
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, dtype, extra=0.0):
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(a.float().cpu(), b.float(), rtol=ulp, atol=1e-3 * float(b.float().abs().max()) + extra)


@pytest.mark.parametrize("T,H", [(64, 4096), (1, 896), (7, 8192), (33, 1368), (3, 16384)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rmsnorm_and_fused_add(T, H, dtype):
    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, H, generator=g).to(dtype)
    res = torch.randn(T, H, generator=g).to(dtype)
    w = (torch.rand(H, generator=g) + 0.5).to(dtype)
    ref = oracle.rmsnorm(x, w, 1e-5)
    _close(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5), ref, dtype)
    res_ref = res.clone()
    ref2 = oracle.rmsnorm(x, w, 1e-5, residual=res_ref)
    xd, rd = x.to(DEV).clone(), res.to(DEV).clone()
    ops.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-5)
    assert torch.equal(rd.cpu().view(torch.int16), res_ref.view(torch.int16)), "residual = round(x + residual) exactly"
    _close(xd, ref2, dtype)


@pytest.mark.parametrize("T,d", [(64, 14336), (5, 4864), (1, 11008), (16, 3584)])
def test_silu_and_mul(T, d):
    g = torch.Generator().manual_seed(d)
    x = (torch.randn(T, 2 * d, generator=g) * 2).bfloat16()
    _close(ops.silu_and_mul(x.to(DEV)), oracle.silu_and_mul(x), torch.bfloat16)


@pytest.mark.parametrize("Hq,Hk,D", [(32, 8, 128), (14, 2, 64), (8, 1, 128)])
def test_rope_neox(Hq, Hk, D):
    from sglang_npu_amd.layers import RotaryEmbedding
    g = torch.Generator().manual_seed(D + Hq)
    T = 37
    rot = RotaryEmbedding(D, D, 4096, 500000.0, True)
    qkv = torch.randn(T, (Hq + 2 * Hk) * D, generator=g).bfloat16()
    pos = torch.randint(0, 4096, (T,), generator=g)
    q_ref = qkv[:, :Hq * D].clone().view(T, Hq, D)
    k_ref = qkv[:, Hq * D:(Hq + Hk) * D].clone().view(T, Hk, D)
    oracle.rope_neox(q_ref, pos, rot.cos_sin_cache)
    oracle.rope_neox(k_ref, pos, rot.cos_sin_cache)
    d = qkv.to(DEV)
    q, k, v = d.split([Hq * D, Hk * D, Hk * D], dim=-1)  # strided views, like llama.py:188
    ops.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), q, k, D, rot.cos_sin_cache.to(DEV), True)
    _close(q.reshape(T, Hq, D), q_ref, torch.bfloat16)
    _close(k.reshape(T, Hk, D), k_ref, torch.bfloat16)
    assert torch.equal(v.cpu(), qkv[:, (Hq + Hk) * D:]), "v must be untouched"


def test_fused_norm_quant_equals_unfused_pair():
    g = torch.Generator().manual_seed(11)
    T, H = 64, 4096
    x = torch.randn(T, H, generator=g).bfloat16().to(DEV)
    res = torch.randn(T, H, generator=g).bfloat16().to(DEV)
    w = (torch.rand(H, generator=g) + 0.5).bfloat16().to(DEV)
    x2, r2 = x.clone(), res.clone()
    ops.fused_add_rmsnorm(x2, r2, w, 1e-5)
    q_ref = torch.empty(T, H, dtype=torch.float8_e4m3fn, device=DEV)
    s_ref = torch.empty(T, 1, device=DEV)
    ops.sgl_per_token_quant_fp8(x2, q_ref, s_ref)
    r3 = res.clone()
    q, s, out = ops.rmsnorm_quant_fp8(x, w, 1e-5, residual=r3, want_out=True)
    assert torch.equal(r3, r2) and torch.equal(out, x2)
    assert torch.equal(s, s_ref) and torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)), "fusion must be bit-exact"
    # silu + quant
    y = (torch.randn(T, 2 * 14336, generator=g) * 2).bfloat16().to(DEV)
    a = ops.silu_and_mul(y)
    qa_ref = torch.empty(a.shape, dtype=torch.float8_e4m3fn, device=DEV)
    sa_ref = torch.empty(T, 1, device=DEV)
    ops.sgl_per_token_quant_fp8(a, qa_ref, sa_ref)
    qa, sa = ops.silu_and_mul_quant_fp8(y)
    assert torch.equal(sa, sa_ref) and torch.equal(qa.view(torch.uint8), qa_ref.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(64, 128256), (1, 32000), (7, 151936), (3, 5), (130, 4099)])
def test_argmax_equals_torch_argmax(dtype, rows, cols):
    """Greedy sampling (sampler.py:72-75): the same index as torch.argmax -- first maximal value, NaN counts as the
    maximum -- on rows full of ties (coarse values), with -0/+0, infinities and NaNs."""
    from sglang_npu_amd import ops
    g = torch.Generator(device=DEV).manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device=DEV, generator=g) * 3).round().to(dtype)  # few distinct values: ties everywhere
    assert torch.equal(ops.argmax(x), torch.argmax(x, dim=-1))
    y = torch.randn(rows, cols, device=DEV, generator=g).to(dtype)
    assert torch.equal(ops.argmax(y), torch.argmax(y, dim=-1))
    z = torch.zeros(rows, cols, device=DEV, dtype=dtype)
    z[:, cols // 2:] = -0.0
    z[0, 0] = -0.0
    assert torch.equal(ops.argmax(z), torch.argmax(z, dim=-1))
    y[0, cols - 1] = float("inf")
    y[rows - 1, cols // 3] = float("nan")
    y[rows - 1, cols - 1] = float("nan")
    y[rows // 2, 0] = float("-inf")
    assert torch.equal(ops.argmax(y), torch.argmax(y, dim=-1))
    # a strided view (rows of a wider matrix) and repeated calls on the self-resetting workspace
    wide = torch.randn(rows, cols + 24, device=DEV, generator=g).to(dtype)
    for _ in range(3):
        assert torch.equal(ops.argmax(wide[:, 8:8 + cols]), torch.argmax(wide[:, 8:8 + cols], dim=-1))


# ----------------------------------------------------------------------------- reference-generated fixtures
# tests/golden/{rmsnorm,silu_and_mul,rope_neox}.npz hold inputs and the outputs of the reference's own torch
# references (sgl-kernel/tests/test_norm.py, test_rotary_embedding.py; test/srt/cpu/utils.py) and compiled CPU ops
# (norm.cpp / activation.cpp / rope.cpp), written by tests/golden/make_golden.py in the build container.
import numpy as np  # noqa: E402


def _z16(z, key, dtype):
    t = torch.from_numpy(z[key].view(np.int16).copy())
    return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16)


def _ulp16(a, b):
    return (a.cpu().contiguous().view(torch.int16).int() - b.contiguous().view(torch.int16).int()).abs()


def _one_ulp_rare(got, exp, what, frac=5e-3, max_ulp=1):
    d = _ulp16(got, exp)
    assert int(d.max()) <= max_ulp, f"{what}: {int(d.max())} ulp"
    assert (d > 0).float().mean().item() < frac, f"{what}: {(d > 0).float().mean().item():.2e} of elements differ"


def test_rmsnorm_vs_reference_fixture():
    """At most one 16-bit ulp from the reference's torch result on < 0.5 % of the elements (fp32 summation order); the
    residual update round(x + residual) is exact."""
    z = np.load("tests/golden/rmsnorm.npz")
    eps = float(z["eps"])
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        x, w, r = (_z16(z, f"{k}{i}", dtype).to(DEV) for k in "xwr")
        _one_ulp_rare(ops.rmsnorm(x, w, eps), _z16(z, f"y{i}", dtype), f"rmsnorm case {i}")
        xd, rd = x.clone(), r.clone()
        ops.fused_add_rmsnorm(xd, rd, w, eps)
        assert torch.equal(rd.cpu().view(torch.int16), _z16(z, f"r_out{i}", dtype).view(torch.int16))
        _one_ulp_rare(xd, _z16(z, f"y_add{i}", dtype), f"fused_add_rmsnorm case {i}")
        _one_ulp_rare(xd, _z16(z, f"y_add_cpu{i}", dtype), f"fused_add_rmsnorm vs compiled CPU op, case {i}")


def test_silu_and_mul_vs_reference_fixture():
    """SiluAndMul.forward_native (and the reference's CUDA kernel, activation.cu:56-60) round silu(x) to the 16-bit dtype
    before the product; the kernel does the same, so only the last bit of expf can move silu(x) across a rounding
    boundary -- rarely (< 0.5 % of elements), and a one-ulp step of silu(x) is at most two ulps of the product."""
    z = np.load("tests/golden/silu_and_mul.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        y = ops.silu_and_mul(_z16(z, f"x{i}", dtype).to(DEV))
        _one_ulp_rare(y, _z16(z, f"y{i}", dtype), f"silu_and_mul case {i}", max_ulp=2)
        assert int(_ulp16(y, _z16(z, f"y_cpu{i}", dtype)).max()) <= 2


def test_rope_neox_bit_exact_vs_reference_fixture():
    z = np.load("tests/golden/rope_neox.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        hs, rd, T, Hq, Hkv = [int(v) for v in z[f"meta{i}"]]
        q, k = _z16(z, f"q{i}", dtype).to(DEV), _z16(z, f"k{i}", dtype).to(DEV)
        pos, cache = torch.from_numpy(z[f"pos{i}"]).to(DEV), torch.from_numpy(z[f"cache{i}"]).to(DEV)
        ops.apply_rope_with_cos_sin_cache_inplace(pos, q, k, hs, cache, True)
        assert torch.equal(q.cpu().view(torch.int16), _z16(z, f"q_out{i}", dtype).view(torch.int16)), f"q case {i}"
        assert torch.equal(k.cpu().view(torch.int16), _z16(z, f"k_out{i}", dtype).view(torch.int16)), f"k case {i}"


def test_vocab_parallel_embedding_bit_exact_vs_reference_fixture():
    """ops.vocab_parallel_embedding against outputs of the REFERENCE's own get_masked_input_and_mask / shard ranges
    (vocab_parallel_embedding.py:126-150, 284-330, 462-482; tests/golden/make_golden.py run_vocab_embedding_cases): every
    rank's masked lookup, bf16 / fp16 tables, int32 / int64 ids, shard-edge ids, TP 2 / 4 / 8."""
    from sglang_npu_amd.layers import pad_vocab_size, vocab_shard_range
    z = np.load("tests/golden/vocab_parallel_embedding.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        table = _z16(z, f"table{i}", dtype)
        ids = torch.from_numpy(z[f"ids{i}"])
        vocab, tp = int(z[f"vocab{i}"]), int(z[f"tp{i}"])
        per = table.shape[0] // tp
        assert pad_vocab_size(vocab, 64) == table.shape[0]
        for r in range(tp):
            start, end = int(z[f"start{i}_{r}"]), int(z[f"end{i}_{r}"])
            assert vocab_shard_range(vocab, table.shape[0], r, tp) == (start, end, per)  # the host-side shard arithmetic too
            got = ops.vocab_parallel_embedding(ids.to(DEV), table[r * per:(r + 1) * per].contiguous().to(DEV), start, end)
            assert torch.equal(got.cpu().view(torch.int16), _z16(z, f"o{i}_{r}", dtype).view(torch.int16)), f"case {i} rank {r}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_vocab_parallel_embedding_vs_oracle(dtype, idt):
    """ops.vocab_parallel_embedding (one rank's masked lookup of VocabParallelEmbedding.forward,
    vocab_parallel_embedding.py:126-150, 462-482) against the oracle's restatement: a byte copy -- exact; the shards of
    all ranks summed give the plain lookup (the all-reduce of :483)."""
    import oracle
    from sglang_npu_amd.layers import pad_vocab_size, vocab_shard_range
    g = torch.Generator().manual_seed(3)
    vocab, H, world = 1000, 896, 4
    padded = pad_vocab_size(vocab, 64 * world)
    table = torch.randn(vocab, H, generator=g).to(dtype)
    ids = torch.randint(0, vocab, (2, 37), generator=g).to(idt)
    ids[0, 0], ids[0, 1], ids[1, 0] = 0, vocab - 1, 250
    total = torch.zeros(2, 37, H, dtype=torch.float64)
    for rank in range(world):
        start, end, per = vocab_shard_range(vocab, padded, rank, world)
        shard = torch.zeros(per, H, dtype=dtype)
        shard[:end - start] = table[start:end]
        ref = oracle.vocab_parallel_embedding(ids.long(), shard, start, end)
        got = ops.vocab_parallel_embedding(ids.to(DEV), shard.to(DEV), start, end)
        assert got.shape == (2, 37, H) and torch.equal(got.cpu(), ref)
        total += got.double().cpu()
    assert torch.equal(total.to(dtype), table[ids.long()])
    assert ops.vocab_parallel_embedding(ids[:0].to(DEV), table.to(DEV), 0, vocab).shape == (0, 37, H)
    with pytest.raises(RuntimeError, match="does not fit"):
        ops.vocab_parallel_embedding(ids.to(DEV), table[:10].contiguous().to(DEV), 0, vocab)
