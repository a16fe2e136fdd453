"""GPU: AWQ INT4 dequant (bit-exact) and the fused dequant-GEMM vs golden vectors and the oracle."""
import numpy as np
import pytest
import torch

import oracle
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _h(arr, dtype):
    t = torch.from_numpy(arr.view(np.int16).copy())
    return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16)


def test_awq_golden_dequant_exact_and_gemm():
    z = np.load("tests/golden/awq.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        qw, qz = torch.from_numpy(z[f"qweight{i}"]).to(DEV), torch.from_numpy(z[f"qzeros{i}"]).to(DEV)
        sc = _h(z[f"scales{i}"], dtype).to(DEV)
        w = ops.awq_dequantize(qw, sc, qz)
        assert torch.equal(w.cpu().view(torch.int16), torch.from_numpy(z[f"w{i}"].view(np.int16).copy())), "dequant must be exact"
        x = _h(z[f"x{i}"], dtype).to(DEV)
        y = ops.awq_gemm(x, qw, sc, qz)
        torch.testing.assert_close(y.float().cpu(), torch.from_numpy(z[f"y_f32_{i}"]),
                                   rtol=2.0 ** (-7 if dtype == "bf16" else -10), atol=1e-2)


# (K, N/8) from sgl-kernel/tests/test_awq_dequant.py:70-71 (subset) + Llama-2-7B layer shapes
@pytest.mark.parametrize("K,Nc,G", [(128, 16, 128), (3584, 448, 3584), (1536, 72, 128), (4096, 512, 128), (11008, 512, 128),
                                    (512, 4736, 512)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_awq_dequantize_vs_oracle_exact(K, Nc, G, dtype):
    g = torch.Generator().manual_seed(K + Nc)
    imax = torch.iinfo(torch.int32).max
    qw = torch.randint(0, imax, (K, Nc), dtype=torch.int32, generator=g)
    qz = torch.randint(0, imax, (K // G, Nc), dtype=torch.int32, generator=g)
    sc = torch.rand(K // G, Nc * 8, generator=g).to(dtype)
    ref = oracle.awq_dequantize(qw, sc, qz)
    out = ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("M", [1, 5, 16, 33, 64, 100])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 12288), (11008, 4096), (256, 128), (4096, 22016)])
def test_awq_gemm_vs_oracle(M, K, N):
    dtype = torch.float16
    g = torch.Generator().manual_seed(M + K + N)
    imax = torch.iinfo(torch.int32).max
    G = 128
    qw = torch.randint(0, imax, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(0, imax, (K // G, N // 8), dtype=torch.int32, generator=g)
    sc = (torch.rand(K // G, N, generator=g) * 2e-2).to(dtype)
    x = torch.randn(M, K, generator=g).to(dtype)
    bias = torch.randn(N, generator=g).to(dtype) if M % 2 else None
    ref = _awq_reference(x, qw, sc, qz, bias)
    out = ops.awq_gemm(x.to(DEV), qw.to(DEV), sc.to(DEV), qz.to(DEV), bias.to(DEV) if bias is not None else None)
    # fp32 accumulation of identical fp16 operands: summation order only -> 1 output ulp (+ bias rounding)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=2.0 ** -10, atol=2e-3 * float(ref.float().abs().max()))


def _awq_reference(x, qw, sc, qz, bias):
    """The oracle where it finishes in seconds; at the full Llama-2-7B widths with more than 16 rows an fp64 product
    of the SAME fp16 operands instead: `ops.awq_dequantize` is pinned bit-exact to the reference (golden + oracle tests
    above), and fp64 accumulation of fp16 x fp16 products is exact to ~1e-13, so this reference is order-free where the
    oracle's fp32 sum is one particular order -- the tolerance (summation order of an fp32 accumulator) is unchanged."""
    M, K = x.shape
    if M <= 16 or K * qw.size(1) * 8 <= 5e7:
        return oracle.awq_gemm(x, qw, sc, qz, bias)
    w = ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    ref = x.to(DEV).double() @ w.double()
    if bias is not None:
        ref = ref + bias.to(DEV).double()
    return ref.float().cpu()


def _awq_case(K, N, G, g):
    imax = torch.iinfo(torch.int32).max
    qw = torch.randint(0, imax, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(0, imax, (K // G, N // 8), dtype=torch.int32, generator=g)
    sc = ((torch.rand(K // G, N, generator=g) - 0.3) * 2e-2).half()  # negative scales too
    return qw, qz, sc


@pytest.mark.parametrize("K,N,G", [(512, 16, 128), (1024, 72, 256), (4096, 4096, 128), (11008, 1000, 128), (640, 64, 128)])
def test_awq_packed_dequant_is_bit_exact(K, N, G):
    """The k-packed decode path must produce the SAME fp16 weights as awq_dequantize (integer unpack exact, one fp16
    rounding in `(w - z) * s`): with one-hot activation rows x[m] = e_{k_m} the GEMM output row m is W[k_m, :] exactly
    (a single product by 1.0, fp32 accumulate, exact back-conversion)."""
    g = torch.Generator().manual_seed(K + N)
    qw, qz, sc = _awq_case(K, N, G, g)
    w_ref = oracle.awq_dequantize(qw, sc, qz)  # [K, N] fp16, pinned bit-exact to the reference by the golden test
    wp, sz = ops.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    rows = torch.randperm(K, generator=g)[:192]
    for c in range(0, 192, 64):
        ks = rows[c:c + 64]
        x = torch.zeros(64, K, dtype=torch.float16)
        x[torch.arange(64), ks] = 1.0
        out = ops.awq_gemm_packed(x.to(DEV), wp, sz, G)
        # a sum cannot return -0.0 (nib == zero with a negative scale): compare modulo the sign of zero
        assert torch.equal((out.cpu() + 0.0).view(torch.int16), (w_ref[ks] + 0.0).view(torch.int16))


@pytest.mark.parametrize("M", [1, 7, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 12288), (11008, 4096), (512, 128), (4096, 22016), (2176, 1000)])
def test_awq_gemm_packed_vs_oracle(M, K, N):
    """Every M bucket (16/32/64 rows), wide and narrow N (direct and split-K slabs), ragged N, bias on odd M."""
    g = torch.Generator().manual_seed(M * 3 + K + N)
    G = 128
    qw, qz, sc = _awq_case(K, N, G, g)
    x = torch.randn(M, K, generator=g).half()
    bias = torch.randn(N, generator=g).half() if M % 2 else None
    ref = _awq_reference(x, qw, sc, qz, bias)
    wp, sz = ops.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    out = ops.awq_gemm_packed(x.to(DEV), wp, sz, G, bias.to(DEV) if bias is not None else None)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=2.0 ** -10, atol=2e-3 * float(ref.float().abs().max()))
    # and the same numbers as the checkpoint-layout kernel up to summation order
    out2 = ops.awq_gemm(x.to(DEV), qw.to(DEV), sc.to(DEV), qz.to(DEV), bias.to(DEV) if bias is not None else None)
    torch.testing.assert_close(out.float(), out2.float(), rtol=2.0 ** -9, atol=2e-3 * float(ref.float().abs().max()))


def test_awq_linear_method_repacks_and_matches_unpacked(monkeypatch):
    from sglang_npu_amd.quantization import AWQConfig
    from sglang_npu_amd.linear import ColumnParallelLinear
    g = torch.Generator().manual_seed(9)
    K, N = 1024, 512
    outs = []
    for no_repack in ("", "1"):
        if no_repack:
            monkeypatch.setenv("SGL_MI355_AWQ_NO_REPACK", "1")
        layer = ColumnParallelLinear(K, [N], bias=False, params_dtype=torch.float16,
                                     quant_config=AWQConfig(4, 128, True)).to(DEV)
        gg = torch.Generator().manual_seed(10)
        qw, qz, sc = _awq_case(K, N, 128, gg)
        layer.qweight.data.copy_(qw), layer.qzeros.data.copy_(qz), layer.scales.data.copy_(sc)
        layer.quant_method.process_weights_after_loading(layer)
        assert (layer.awq_packed is None) == bool(no_repack)
        x = torch.randn(20, K, generator=torch.Generator().manual_seed(9)).half().to(DEV)
        y = layer(x)
        outs.append(y[0] if isinstance(y, tuple) else y)
    torch.testing.assert_close(outs[0].float(), outs[1].float(), rtol=2.0 ** -9, atol=1e-2)


@pytest.mark.parametrize("M", [65, 128, 200, 512, 1000])
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 12288), (11008, 4096), (512, 128), (2176, 1000), (4096, 22016)])
def test_awq_gemm_packed_tiled_vs_reference_computation(M, K, N):
    """Prefill (M > 64): 128 x 128 x 64 tiles with the INT4 weights unpacked in registers.  Reference computation =
    x @ awq_dequantize(...) (awq.py:413-417) with the bit-exact-pinned dequant and an fp64 product (order-free), or the C
    oracle where it finishes in seconds.  Ragged M and N tiles, K padded to 512 inside the packed copy, bias on odd M."""
    g = torch.Generator().manual_seed(M * 5 + K + N)
    G = 128
    qw, qz, sc = _awq_case(K, N, G, g)
    x = torch.randn(M, K, generator=g).half()
    bias = torch.randn(N, generator=g).half() if M % 2 else None
    ref = _awq_reference(x, qw, sc, qz, bias)
    wp, sz = ops.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    out = ops.awq_gemm_packed_tiled(x.to(DEV), wp, sz, G, bias.to(DEV) if bias is not None else None)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=2.0 ** -10, atol=2e-3 * float(ref.float().abs().max()))


@pytest.mark.parametrize("K,N,G", [(512, 128, 128), (4096, 1000, 128), (11008, 256, 256)])
def test_awq_tiled_dequant_is_bit_exact(K, N, G):
    """One-hot activation rows: output row m is W[k_m, :] exactly -- the in-register unpacking of the tiled kernel is the
    reference's dequantisation bit for bit (same argument as test_awq_packed_dequant_is_bit_exact)."""
    g = torch.Generator().manual_seed(K + N + 1)
    qw, qz, sc = _awq_case(K, N, G, g)
    w_ref = oracle.awq_dequantize(qw, sc, qz)
    wp, sz = ops.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    ks = torch.randperm(K, generator=g)[:160]
    x = torch.zeros(160, K, dtype=torch.float16)
    x[torch.arange(160), ks] = 1.0
    out = ops.awq_gemm_packed_tiled(x.to(DEV), wp, sz, G)
    assert torch.equal((out.cpu() + 0.0).view(torch.int16), (w_ref[ks] + 0.0).view(torch.int16))


def test_awq_linear_method_prefill_keeps_int4_only():
    """M > 64 through AWQLinearMethod.apply: the fused tiled kernel, no dequantised copy of the weight on the layer, and
    a reload (process_weights_after_loading again) is picked up by both the decode and the prefill path."""
    from sglang_npu_amd.quantization import AWQConfig
    from sglang_npu_amd.linear import ColumnParallelLinear
    K, N, M = 1024, 512, 200
    layer = ColumnParallelLinear(K, [N], bias=False, params_dtype=torch.float16, quant_config=AWQConfig(4, 128, True)).to(DEV)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).half().to(DEV)
    for seed in (10, 11):  # second round = a weight reload
        qw, qz, sc = _awq_case(K, N, 128, torch.Generator().manual_seed(seed))
        layer.qweight.data.copy_(qw), layer.qzeros.data.copy_(qz), layer.scales.data.copy_(sc)
        layer.quant_method.process_weights_after_loading(layer)
        y = layer(x)[0]
        ref = (x.double() @ ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV)).double())
        torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -10, atol=2e-3 * float(ref.abs().max()))
        y_dec = layer(x[:7])[0]
        torch.testing.assert_close(y_dec.double(), ref[:7], rtol=2.0 ** -10, atol=2e-3 * float(ref.abs().max()))
    assert not hasattr(layer, "awq_dequant_cache")
    assert not any(t.dtype == torch.float16 and t.numel() >= K * N for t in vars(layer).values() if isinstance(t, torch.Tensor))


@pytest.mark.parametrize("M", [65, 100, 128])
def test_awq_linear_method_65_to_128_rows_is_two_passes_of_the_decode_kernel(M):
    """AWQLinearMethod.apply for a decode batch of 65..128 rows: rows are independent, so the result must be the 64-row
    streamer's on each chunk (bit for bit), with bias, and match x @ dequantize(W)."""
    from sglang_npu_amd.quantization import AWQConfig
    from sglang_npu_amd.linear import ColumnParallelLinear
    K, N = 2048, 1024
    layer = ColumnParallelLinear(K, [N], bias=True, params_dtype=torch.float16, quant_config=AWQConfig(4, 128, True)).to(DEV)
    qw, qz, sc = _awq_case(K, N, 128, torch.Generator().manual_seed(21))
    layer.qweight.data.copy_(qw), layer.qzeros.data.copy_(qz), layer.scales.data.copy_(sc)
    layer.bias.data.copy_(torch.randn(N, generator=torch.Generator().manual_seed(3)).half())
    layer.quant_method.process_weights_after_loading(layer)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).half().to(DEV)
    y = layer(x)[0]
    assert y.shape == (M, N)
    assert torch.equal(y[:64], layer(x[:64])[0]) and torch.equal(y[64:], layer(x[64:])[0])
    ref = x.double() @ ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV)).double() + layer.bias.double()
    torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -10, atol=2e-3 * float(ref.abs().max()))
