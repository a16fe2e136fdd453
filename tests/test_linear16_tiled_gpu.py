"""The 16-bit tiled GEMM for more than 128 rows (csrc/gemm_bf16.hip gemm16_tiled_kernel, round 4): prefill of an unquantised
model and LM heads over many rows -- `F.linear` in the reference (layers/quantization/unquant.py:111-123,
logits_processor.py:430-505).  Reference here: an fp64 product of the same 16-bit values, rounded once."""
import pytest
import torch

from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, bias):
    ref = x.double() @ w.double().t()
    return ref if bias is None else ref + bias.double()


def _close(out, ref, dtype):
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(out.double(), ref, rtol=ulp, atol=ulp * float(ref.abs().max()) * 0.05)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [129, 200, 256, 1000, 1024, 1537])
@pytest.mark.parametrize("N,K", [(4096, 4096), (12288, 4096), (4096, 11008), (22016, 4096), (512, 256), (48, 512),
                                 (1008, 768), (6144, 4096)])
def test_linear16_tiled_vs_fp64(M, N, K, dtype):
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, device=DEV, generator=g).to(dtype) if M % 2 == 0 else None
    wsh = ops.linear16_shuffle_weight(w)
    out = ops.linear16(x, wsh, bias)
    assert out.shape == (M, N) and out.dtype == dtype
    _close(out, _ref(x, w, bias), dtype)
    # rows of a wider activation matrix (strided view) give the same bits
    wide = torch.zeros(M, K + 64, device=DEV, dtype=dtype)
    wide[:, :K] = x
    assert torch.equal(ops.linear16(wide[:, :K], wsh, bias), out)


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (4096, 22016, 4096), (2048, 4096, 11008), (8192, 12288, 4096)])
def test_linear16_tiled_prefill_sizes(M, N, K):
    """Llama-2-7B's prefill shapes (BASELINE configs[2]) at 2048..8192 rows: every tile form of the dispatcher."""
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    out = ops.linear16(x, ops.linear16_shuffle_weight(w))
    rows = torch.randperm(M, device=DEV, generator=g)[:96]  # (the fp64 product of everything would take seconds per case)
    _close(out[rows], _ref(x[rows], w, None), torch.bfloat16)
    # every row block and column block was written: no element kept the fill value of a fresh buffer
    assert bool(torch.isfinite(out.float()).all())
    cols = torch.randperm(N, device=DEV, generator=g)[:64]
    _close(out[:, cols], x.double() @ w[cols].double().t(), torch.bfloat16)


def test_linear16_tiled_reads_nothing_past_its_operands():
    """Operands at the very end of their allocations, NaN in front: an over-read past a ragged edge would show up as a value."""
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 300, 208, 512
    big_x = torch.full((4096 + M * K,), float("nan"), device=DEV, dtype=torch.bfloat16)
    x = big_x[-M * K:].view(M, K)
    x.copy_(torch.randn(M, K, device=DEV, generator=g))
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    wsh = ops.linear16_shuffle_weight(w)
    big_w = torch.full((8192 + wsh.data.numel(),), 0xFF, device=DEV, dtype=torch.uint8)  # 0xFFFF = a bf16 NaN
    tail = big_w[-wsh.data.numel():].view(wsh.data.shape)
    tail.copy_(wsh.data)
    out = ops.linear16(x, ops.ShuffledWeight16(tail, N, K, torch.bfloat16))
    _close(out, _ref(x, w, None), torch.bfloat16)


def test_row_major_weight_above_128_rows_is_refused():
    with pytest.raises(RuntimeError, match="fragment-major"):
        ops.linear16(torch.zeros(129, 512, device=DEV, dtype=torch.bfloat16), torch.zeros(64, 512, device=DEV, dtype=torch.bfloat16))
