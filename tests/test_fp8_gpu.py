"""GPU: per-token FP8 quant (bit-exact) and fp8_scaled_mm against the golden vectors
(reference torch oracles) and the C oracle."""
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


def test_per_token_quant_golden_bit_exact():
    z = np.load("tests/golden/per_token_quant_fp8.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        x = _h(z[f"x{i}"], dtype).to(DEV)
        q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=DEV)
        s = torch.empty(x.size(0), 1, dtype=torch.float32, device=DEV)
        ops.sgl_per_token_quant_fp8(x, q, s)
        assert torch.equal(s.cpu().view(-1), torch.from_numpy(z[f"s{i}"])), "scale must be exact"
        assert torch.equal(q.cpu().view(torch.uint8), torch.from_numpy(z[f"q{i}"])), "q must be bit-exact"


@pytest.mark.parametrize("T,K", [(1, 4096), (64, 4096), (64, 14336), (128, 1368), (512, 896), (3, 8)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_per_token_quant_vs_oracle_bit_exact(T, K, dtype):
    g = torch.Generator().manual_seed(T * 7 + K)
    x = (torch.randn(T, K, generator=g) * 2.5).to(dtype)
    x[0, :3] = torch.tensor([0.0, -0.0, 1e-30]).to(dtype)
    if T > 2:
        x[2].zero_()
    q_ref = torch.empty(T, K, dtype=torch.uint8)
    s_ref = torch.empty(T)
    oracle.per_token_quant_fp8(x, q_ref, s_ref)
    q = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.empty(T, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(x.to(DEV), q, s)
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref)


def test_quant_rejects_bad_hidden_dim():
    x = torch.zeros(2, 12, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="divisible by 8"):
        ops.sgl_per_token_quant_fp8(x, torch.empty(2, 12, dtype=torch.float8_e4m3fn, device=DEV),
                                    torch.empty(2, device=DEV))


def test_fp8_scaled_mm_golden():
    z = np.load("tests/golden/fp8_scaled_mm.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        dt = torch.bfloat16 if dtype == "bf16" else torch.float16
        a = torch.from_numpy(z[f"a{i}"]).view(torch.float8_e4m3fn).to(DEV)
        b = torch.from_numpy(z[f"b{i}"]).view(torch.float8_e4m3fn).to(DEV)
        bias = _h(z[f"bias{i}"], dtype).to(DEV) if f"bias{i}" in z.files else None
        o = ops.fp8_scaled_mm(a, b.t(), torch.from_numpy(z[f"sa{i}"]).to(DEV), torch.from_numpy(z[f"sb{i}"]).to(DEV),
                              dt, bias)
        ref = _h(z[f"o{i}"], dtype)
        # the reference's own bound is rtol 0.02 / atol 1 (sgl-kernel/tests/test_fp8_gemm.py:33-35);
        # ours: one output ulp (+ the bias rounding the reference oracle adds)
        torch.testing.assert_close(o.float().cpu(), ref.float(), rtol=2.0 ** -7, atol=2e-2)


def _rand_fp8(shape, g):
    return ((torch.rand(shape, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 65, 128, 129, 200, 256, 512])
@pytest.mark.parametrize("N,K", [(128, 512), (6144, 4096), (4096, 1024), (1280, 8192), (72, 144), (4096, 14336)])
def test_fp8_scaled_mm_vs_oracle(M, N, K):
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    dt = torch.bfloat16 if (M + N) % 2 == 0 else torch.float16
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, generator=g) * 1e-3 + 1e-4
    sb = torch.rand(N, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=g).to(dt) if M % 2 else None
    if M > 128 and N * K > 8e6:
        # the C oracle needs minutes here: an fp64 product of the same fp8 values (products exact, sum to ~1e-13)
        # with the oracle's epilogue order (x w_scale, x x_scale, + bias, one rounding) stands in for it
        ref = (a.to(DEV).double() @ w.to(DEV).double().t()) * sb.to(DEV).double() * sa.to(DEV).double()[:, None]
        if bias is not None:
            ref = ref + bias.to(DEV).double()
        ref = ref.to(dt).cpu()
    else:
        ref = oracle.fp8_scaled_mm(a, w.t(), sa, sb, dt, bias)
    out = ops.fp8_scaled_mm(a.to(DEV), w.to(DEV).t(), sa.to(DEV), sb.to(DEV), dt, bias.to(DEV) if bias is not None else None)
    # fp32 accumulation of exact products: only the summation order differs -> <= 1 output ulp
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=ulp, atol=1e-3 * float(ref.float().abs().max()))


def test_fp8_scaled_mm_full_size_properties():
    """BASELINE config 3 shapes at full size: (a) rows are independent -- the M=64 (skinny kernel)
    result equals the matching rows of an M=256 (tiled kernel) call to within one ulp; (b) scaling
    scales_a by 2 doubles the output exactly (power-of-two)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for (K, N) in [(4096, 6144), (4096, 28672), (14336, 4096)]:
        a = ((torch.rand(256, K, device=DEV, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
        w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
        sa = torch.rand(256, device=DEV, generator=g) * 1e-2 + 1e-3
        sb = torch.rand(N, device=DEV, generator=g) * 1e-2 + 1e-3
        big = ops.fp8_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
        small = ops.fp8_scaled_mm(a[:64], w.t(), sa[:64], sb, torch.bfloat16)
        torch.testing.assert_close(small.float(), big[:64].float(), rtol=2.0 ** -7, atol=1e-6)
        dbl = ops.fp8_scaled_mm(a[:64], w.t(), sa[:64] * 2, sb, torch.bfloat16)
        assert torch.equal(dbl, small * 2)
        # spot-check 4 rows x 64 columns against fp64 on the host
        rows, cols = [0, 17, 40, 63], slice(1000, 1064)
        ref = (a[rows].float().double() @ w[cols].float().double().t()) * sb[cols].double() * sa[rows].double()[:, None]
        torch.testing.assert_close(small[rows][:, cols].double(), ref, rtol=2.0 ** -7, atol=1e-6)


@pytest.mark.parametrize("M", [1, 5, 16, 17, 32, 48, 64])
@pytest.mark.parametrize("N,K", [(28672, 4096), (20480, 1024), (20488, 2048), (24576, 4224), (28672, 512)])
def test_fp8_scaled_mm_wide_n_decode_kernels(M, N, K):
    """Wide-N decode shapes take the weight-streaming kernels (LDS-DMA double-buffered A / phase-filled A,
    depending on K): every M bucket (16/32/64 rows), ragged N, K with and without a 128 tail, bias on odd M.
    Checked against an fp64 matmul of the same fp8 values (order-free up to fp32 accumulation -> 1 ulp)."""
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N + K)
    dt = torch.bfloat16 if M % 2 == 0 else torch.float16
    a = ((torch.rand(M, K, device=DEV, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=DEV, generator=g) * 1e-2 + 1e-3
    sb = torch.rand(N, device=DEV, generator=g) * 1e-2 + 1e-3
    bias = torch.randn(N, device=DEV, generator=g).to(dt) if M % 2 else None
    out = ops.fp8_scaled_mm(a, w.t(), sa, sb, dt, bias)
    ref = (a.double() @ w.double().t()) * sb.double() * sa.double()[:, None]
    if bias is not None:
        ref = ref + bias.double()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(out.double(), ref, rtol=ulp, atol=ulp * float(ref.abs().max()) * 0.05)


def test_fp8_scaled_mm_checks():
    a = torch.zeros(4, 24, dtype=torch.float8_e4m3fn, device=DEV)
    w = torch.zeros(16, 24, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.ones(16, device=DEV)
    with pytest.raises(RuntimeError, match="16 bytes"):
        ops.fp8_scaled_mm(a, w.t(), s[:4], s, torch.bfloat16)
    with pytest.raises(RuntimeError, match="column major"):
        ops.fp8_scaled_mm(a, torch.zeros(24, 16, dtype=torch.float8_e4m3fn, device=DEV), s[:4], s, torch.bfloat16)


@pytest.mark.parametrize("M", [1, 16, 33, 64])
@pytest.mark.parametrize("K,N", [(3584, 8192), (1536, 4096), (3584, 28672)])
def test_k_steps_not_multiple_of_eight(M, K, N):
    """K / 128 = 28 or 12 (Llama-3-70B down_proj per TP-8 rank: 3584): the weight streamer runs phases of four k-steps."""
    g = torch.Generator(device=DEV).manual_seed(M + K)
    a = ((torch.rand(M, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, device=DEV, generator=g) * 1e-2 + 1e-3
    bias = torch.randn(N, device=DEV, generator=g).bfloat16()
    out = ops.fp8_scaled_mm(a, w.t(), sa, sb, torch.bfloat16, bias)
    ref = (a.float() @ w.float().t()) * sb.view(1, -1) * sa + bias.float()
    assert torch.allclose(out.float(), ref, rtol=2 ** -7, atol=2 ** -7 * float(ref.abs().max()))
    part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, bias)
    if part is not None:
        assert torch.allclose(part.finalize().float(), ref, rtol=2 ** -7, atol=2 ** -7 * float(ref.abs().max()))


def test_split_k_workspace_is_per_stream_and_never_replaced_under_a_graph():
    """The split-K scratch is keyed by (device, stream); a buffer that must grow is replaced while the old one stays
    alive (captured graphs hold its pointer); growing one during capture raises."""
    dev = torch.device(DEV)
    pool = ops._ScratchPool(1024)
    a = pool.get(dev, 100)
    assert pool.get(dev, 1000) is a
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        b = pool.get(dev, 100)
    assert b.data_ptr() != a.data_ptr()
    c = pool.get(dev, 5000)
    assert c.numel() >= 5000 and c is not a and any(t is a for t in pool._retired)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cap):
        pool.get(dev, 10)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cap):
            torch.zeros(8, device=dev)
            pool.get(dev, 10)               # fits: fine
            with pytest.raises(RuntimeError, match="during graph capture"):
                pool.get(dev, 1 << 20)
    # a GEMM captured on one stream and eager GEMMs on another do not share partial sums
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = ((torch.rand(64, 4096, device=DEV, generator=gen) - 0.5) * 8).to(torch.float8_e4m3fn)
    w = ((torch.rand(4096, 4096, device=DEV, generator=gen) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa, sb = torch.rand(64, device=DEV, generator=gen) + 0.5, torch.rand(4096, device=DEV, generator=gen) + 0.5
    ref = ops.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        other = ops.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(ref, other)


def test_group_and_tensor_quant_bit_exact_vs_reference_fixture():
    """sgl_per_token_group_quant_fp8 / sgl_per_tensor_quant_fp8 against the reference's own torch references
    (tests/golden/group_tensor_quant_fp8.npz): scales and FP8 bytes must be identical."""
    z = np.load("tests/golden/group_tensor_quant_fp8.npz")
    for i in range(int(z["gn"])):
        T, K, G = [int(v) for v in z[f"gmeta{i}"]]
        dtype = z[f"gdtype{i}"].item().decode()
        x = (torch.from_numpy(z[f"gx{i}"].copy()) if dtype == "f32" else _h(z[f"gx{i}"], dtype)).to(DEV)
        q = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=DEV)
        s = torch.empty(T, K // G, device=DEV)
        ops.sgl_per_token_group_quant_fp8(x, q, s, G, 1e-10, -448.0, 448.0)
        assert torch.equal(s.cpu(), torch.from_numpy(z[f"gs{i}"])), f"group scales, case {i}"
        assert torch.equal(q.cpu().view(torch.uint8), torch.from_numpy(z[f"gq{i}"])), f"group q, case {i}"
        # the reference's column-major scale layout (create_per_token_group_quant_fp8_output_scale, fp8_kernel.py)
        s_cm = torch.empty(K // G, T, device=DEV).t()
        ops.sgl_per_token_group_quant_fp8(x, q, s_cm, G, 1e-10, -448.0, 448.0)
        assert torch.equal(s_cm.contiguous().cpu(), torch.from_numpy(z[f"gs{i}"]))
    for i in range(int(z["tn"])):
        dtype = z[f"tdtype{i}"].item().decode()
        x = _h(z[f"tx{i}"], dtype).to(DEV)
        q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=DEV)
        s = torch.zeros(1, device=DEV)
        ops.sgl_per_tensor_quant_fp8(x, q, s, False)
        assert torch.equal(s.cpu(), torch.from_numpy(z[f"ts{i}"])), f"tensor scale, case {i}"
        assert torch.equal(q.cpu().view(torch.uint8), torch.from_numpy(z[f"tq{i}"])), f"tensor q, case {i}"
        ops.sgl_per_tensor_quant_fp8(x, q, torch.from_numpy(z[f"ts_static{i}"].copy()).to(DEV), True)
        assert torch.equal(q.cpu().view(torch.uint8), torch.from_numpy(z[f"tq_static{i}"]))


@pytest.mark.parametrize("T,K,G", [(64, 4096, 128), (7, 14336, 128), (2048, 512, 64), (1, 7168, 256)])
def test_group_quant_vs_oracle_and_per_token_limit(T, K, G):
    """Against the C oracle on other shapes; and group_size = K is the per-token HIP path of apply_fp8_linear
    (fp8_utils.py:676-678): same scale as sgl_per_token_quant_fp8 whenever the row is not all zero."""
    g = torch.Generator().manual_seed(T + K + G)
    x = (torch.randn(T, K, generator=g) * 2).bfloat16()
    q_ref, s_ref = torch.empty(T, K, dtype=torch.uint8), torch.empty(T, K // G)
    oracle.per_token_group_quant_fp8(x, q_ref, s_ref, G, 1e-10, -448.0, 448.0)
    q = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.empty(T, K // G, device=DEV)
    ops.sgl_per_token_group_quant_fp8(x.to(DEV), q, s, G, 1e-10, -448.0, 448.0)
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu().view(torch.uint8), q_ref)
    if K <= 1024:
        s1, s2 = torch.empty(T, 1, device=DEV), torch.empty(T, 1, device=DEV)
        ops.sgl_per_token_group_quant_fp8(x.to(DEV), q, s1, K, 1e-10, -448.0, 448.0)
        ops.sgl_per_token_quant_fp8(x.to(DEV), torch.empty_like(q), s2)
        assert torch.equal(s1, s2)
    with pytest.raises(RuntimeError, match="scale_ue8m0"):  # the UE8M0 form takes the packed int32 tensor, not fp32 scales
        ops.sgl_per_token_group_quant_fp8(x.to(DEV), q, s, G, 1e-10, -448.0, 448.0, True)


def _ue8m0_scale_tensor(T, K, G, device):
    """create_per_token_group_quant_fp8_output_scale(..., scale_ue8m0=True), fp8_kernel.py:308-319."""
    align = lambda v, a: (v + a - 1) // a * a  # noqa: E731
    return torch.zeros((align(K // G, 4) // 4, align(T, 4)), device=device, dtype=torch.int32).transpose(0, 1)[:T, :]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("T,K,G", [(1, 512, 128), (7, 7168, 128), (64, 4096, 128), (130, 1536, 64), (33, 384, 128)])
def test_group_quant_ue8m0_packed_scales(T, K, G, dtype):
    """sgl_per_token_group_quant_fp8(..., scale_ue8m0=True) (per_token_group_quant_8bit.cu:24-137): power-of-two scales as
    exponent bytes packed four to an int32, column-major -- against the C oracle (the reference's exp2 / ceil / log2
    formula) byte for byte, and against the formula written in torch."""
    g = torch.Generator().manual_seed(T + K + G)
    x = (torch.randn(T, K, generator=g) * torch.rand(T, 1, generator=g) * 30).to(dtype)
    x[0, :G] = 0                       # an all-zero group: the 1e-10 floor
    if T > 1:
        x[1, :G] = 448.0 * 2.0 ** -3   # absmax / 448 an exact power of two: no rounding up
    q_ref, s_ref = torch.empty(T, K, dtype=torch.uint8), _ue8m0_scale_tensor(T, K, G, "cpu")
    oracle.per_token_group_quant_fp8_ue8m0(x, q_ref, s_ref, G, 1e-10, -448.0, 448.0)
    q = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=DEV)
    s = _ue8m0_scale_tensor(T, K, G, DEV)
    ops.sgl_per_token_group_quant_fp8(x.to(DEV), q, s, G, 1e-10, -448.0, 448.0, True)
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu().view(torch.uint8), q_ref)
    # the formula itself, in torch: exponent bytes and the quantised values
    gpr = K // G
    amax = x.float().view(T, gpr, G).abs().amax(-1).clamp_min(1e-10)
    e = torch.ceil(torch.log2((amax / 448.0).clamp_min(1e-10)))
    packed = s.cpu().to(torch.int64) & 0xFFFFFFFF  # [T, ceil(gpr / 4)]: byte c % 4 of int32 c / 4 is group c's exponent
    bytes_ = torch.stack([(packed[:, c // 4] >> (8 * (c % 4))) & 0xFF for c in range(gpr)], dim=1)
    assert torch.equal(bytes_, (e + 127).to(torch.int64))
    q_t = (x.float().view(T, gpr, G) / torch.exp2(e)[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    assert torch.equal(q.cpu().view(torch.uint8).view(T, gpr, G), q_t.view(torch.uint8))
    if T > 1:
        assert int(bytes_[1, 0]) == 127 - 3 and int(bytes_[0, 0]) == int(torch.ceil(torch.log2(torch.tensor(1e-10)))) + 127


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 65, 128])
@pytest.mark.parametrize("N,K", [(128256, 4096), (32000, 4096), (1000, 512), (4096, 1024), (151936, 896 + 128)])
def test_linear16_lm_head_vs_fp64(M, N, K, dtype):
    """LM head / unquantised decode linear (logits_processor.py:430-505): x @ W^T in 16 bit with fp32 accumulation.
    Reference: an fp64 product of the same 16-bit values (order-free) rounded once -- one output ulp."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, device=DEV, generator=g).to(dtype) if M % 2 == 0 else None
    out = ops.linear16(x, w, bias)
    ref = x.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(out.double(), ref, rtol=ulp, atol=ulp * float(ref.abs().max()) * 0.05)
    # rows of a wider activation matrix (strided view) give the same result
    wide = torch.zeros(M, K + 64, device=DEV, dtype=dtype)
    wide[:, :K] = x
    assert torch.equal(ops.linear16(wide[:, :K], w, bias), out)
    with pytest.raises(RuntimeError, match="M <= 128"):  # (more rows: the tiled kernel, on the fragment-major weight only)
        ops.linear16(torch.zeros(129, K, device=DEV, dtype=dtype), w)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 16, 17, 33, 64, 100, 128])
@pytest.mark.parametrize("N,K", [(128256, 4096), (32000, 4096), (1008, 512), (4096, 1024), (151936, 1024), (4096, 14336),
                                 (6144, 4096), (4096, 4096), (1280, 8192)])
def test_linear16_on_fragment_major_weight_is_bit_identical(M, N, K, dtype):
    """The LM head on a weight re-laid once for contiguous 1-KiB loads (ops.linear16_shuffle_weight, round 3): the same
    products in the same order -- only the addresses of the weight loads differ -- so the SAME BITS as the row-major call
    (which test_linear16_lm_head_vs_fp64 holds to one output ulp of an fp64 product); the layout round-trips; the storage
    has a shape of its own, so that it cannot be mistaken for the [N, K] matrix."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, device=DEV, generator=g).to(dtype) if M % 2 == 0 else None
    wsh = ops.linear16_shuffle_weight(w)
    assert wsh.data.dtype == torch.uint8 and wsh.data.shape == (N // 16, 32 * K)
    assert torch.equal(wsh.unshuffle().view(torch.int16), w.view(torch.int16))
    # the byte layout is the FP8 one applied to the [N][2K] byte image (include/sgl_mi355.h)
    b = w.view(torch.uint8).view(N // 16, 16, (2 * K) // 128, 2, 4, 16).permute(0, 2, 3, 4, 1, 5).contiguous()
    assert torch.equal(wsh.data.view(-1), b.view(-1))
    out = ops.linear16(x, wsh, bias)
    if N >= 16 * 8 * 200:  # wide: the unsplit kernel on both layouts
        assert torch.equal(out, ops.linear16(x, w, bias))
    else:              # narrow: the fragment-major weight runs the split-K form (another fp32 summation order)
        ref = x.double() @ w.double().t()
        if bias is not None:
            ref = ref + bias.double()
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        torch.testing.assert_close(out.double(), ref, rtol=ulp, atol=ulp * float(ref.abs().max()) * 0.05)
    # a strided source (rows of a wider buffer) is re-laid the same
    wide = torch.zeros(N, K + 256, device=DEV, dtype=dtype)
    wide[:, :K] = w
    assert torch.equal(ops.linear16_shuffle_weight(wide[:, :K]).data, wsh.data)


def test_linear16_shuffle_rejects_other_shapes():
    with pytest.raises(RuntimeError, match="N % 16 == 0"):
        ops.linear16_shuffle_weight(torch.zeros(1000, 512, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="K % 256 == 0"):
        ops.linear16_shuffle_weight(torch.zeros(1024, 896, device=DEV, dtype=torch.bfloat16))


@pytest.mark.parametrize("M,N,K,shuffled,partials,family", [
    (64, 28672, 4096, True, False, "wstream"),        # gate_up at decode: one column block per wave over all of K
    (64, 4096, 14336, True, True, "wstream_slab"),    # down_proj at decode: split-K slabs
    (128, 4096, 14336, True, False, "wstream_slab"),  # ... and for a batch of 65..128 rows: 128-row A phases (MB = 8) + finalize
    (100, 28672, 4096, True, False, "wstream"),       # gate_up at 65..128 rows
    (256, 4096, 14336, True, False, "wstream_slab"),  # ... and 129..256 rows: 256-row A phases of two k-steps (MB = 16)
    (128, 4096, 14336, False, False, "wstream_slab"), # (row-major weight as well)
    (64, 4096, 1024, False, False, "oneshot"),        # K <= 1024 at any width: every load issued up front
    (16, 28672, 4096 + 64, False, False, "astat_direct"),  # wide N with a K tail, row-major weight
    (16, 1024, 65536 + 64, False, False, "astat"),    # narrow N, very long K with a tail
    (7, 48, 208, False, False, "skinny"),             # ragged little shapes (K % 16 == 0 is the op's own precondition)
    (4096, 28672, 4096, True, False, "tiled3"),       # prefill on a pre-shuffled weight, >= 192 tiles of 128 x 256
    (1024, 6144, 4096, True, False, "tiled3_ks2"),    # ... exactly one 128-row tile per CU: eight waves, two k groups
    (1024, 4096, 4096, True, False, "tiled2"),        # prefill with fewer tiles
    (300, 512, 1008, False, False, "tiled"),          # K tail at M > 64
])
def test_every_gemm_kernel_family_is_reachable(M, N, K, shuffled, partials, family):
    """run_gemm's dispatch table (gemm_fp8.hip) by name: each kernel generation that is still compiled in has a shape that
    reaches it -- and gives the right product there (fp64 reference, one output ulp)."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = ((torch.rand(M, K, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    wt = ops.fp8_shuffle_weight(w) if shuffled else w.t()
    if partials:
        part = ops.fp8_scaled_mm_partials(a, wt, sa, sb, torch.bfloat16)
        assert part is not None
        name = ops.fp8_last_kernel()
        out = part.finalize()
    else:
        out = ops.fp8_scaled_mm(a, wt, sa, sb, torch.bfloat16)
        name = ops.fp8_last_kernel()
    assert name == family, (name, family)
    rows = torch.tensor(sorted({0, M // 2, M - 1}), device=DEV)
    ref = (a[rows].double() @ w.double().t()) * sb.double().view(1, -1) * sa[rows].double()
    torch.testing.assert_close(out[rows].double(), ref, rtol=2.0 ** -7, atol=1e-3 * float(ref.abs().max()))


def test_unquantized_wide_linear_uses_the_streamer_at_decode_sizes():
    """UnquantizedLinearMethod (unquant.py): a wide column-parallel bf16 layer gets a fragment-major copy after loading and
    runs ops.linear16 for up to 64 rows (narrow layers through its split-K form); larger batches and shapes without the layout
    stay on F.linear.  One output ulp of fp64."""
    from sglang_npu_amd.linear import MergedColumnParallelLinear, RowParallelLinear
    g = torch.Generator(device=DEV).manual_seed(5)
    K, I = 512, 8192
    wide = MergedColumnParallelLinear(K, [I, I], params_dtype=torch.bfloat16).to(DEV)
    wide.weight.data.copy_(torch.randn(2 * I, K, device=DEV, generator=g) * 0.05)
    wide.quant_method.process_weights_after_loading(wide)
    fm_of = lambda lin: lin.quant_method.weight_fm(lin)
    assert fm_of(wide) is not None and fm_of(wide).data.shape == (2 * I // 16, 32 * K)
    narrow = RowParallelLinear(K, 4096, params_dtype=torch.bfloat16).to(DEV)  # narrow N: the split-K form of the streamer
    narrow.weight.data.copy_(torch.randn(4096, K, device=DEV, generator=g) * 0.05)
    narrow.quant_method.process_weights_after_loading(narrow)
    assert fm_of(narrow) is not None
    odd = RowParallelLinear(K, 1000, params_dtype=torch.bfloat16).to(DEV)     # N % 16 != 0: no fragment-major layout, F.linear
    odd.quant_method.process_weights_after_loading(odd)
    assert fm_of(odd) is None
    for M in (1, 33, 64, 100):
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        y, _ = narrow(x)
        ref = x.double() @ narrow.weight.data.double().t()
        torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()) * 0.05)
    for M in (1, 7, 64, 65, 200):
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        y, _ = wide(x)
        ref = x.double() @ wide.weight.data.double().t()
        torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()) * 0.05)
        if M <= 64:  # the streamer's fp32 summation order, not the library's: the same bits as a direct ops.linear16 call
            assert torch.equal(y, ops.linear16(x, wide.weight.data))
    y3, _ = wide(torch.randn(2, 3, K, device=DEV, generator=g).bfloat16())
    assert y3.shape == (2, 3, 2 * I)


def test_fragment_major_copy_follows_in_place_weight_updates():
    """SGLang updates weights IN PLACE without re-running process_weights_after_loading (model_runner.py:831-900
    update_weights_from_tensor / _from_distributed, :1777 _model_load_weights_direct -> `param.data.copy_`).  The 16-bit
    fragment-major copy next to the parameter must follow every such write -- and into the SAME storage, which captured graphs
    hold -- and must never be multiplied with while stale."""
    from sglang_npu_amd.linear import RowParallelLinear
    from sglang_npu_amd.parameter import raw_data
    g = torch.Generator(device=DEV).manual_seed(9)
    K, N = 512, 4096
    lin = RowParallelLinear(K, N, params_dtype=torch.bfloat16).to(DEV)
    lin.weight.data.copy_(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    lin.quant_method.process_weights_after_loading(lin)
    fm = lin.quant_method.weight_fm(lin)
    assert fm is not None
    ptr = fm.data.data_ptr()
    x = torch.randn(5, K, device=DEV, generator=g).bfloat16()

    def check():
        y, _ = lin(x)
        w = raw_data(lin.weight)
        ref = x.double() @ w.double().t()
        torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()) * 0.05)
        # the streamer ran (not F.linear), on the CURRENT weights: the bits of a fresh fragment-major copy of them
        assert torch.equal(y, ops.linear16(x, ops.linear16_shuffle_weight(w)))
        cur = lin.quant_method.weight_fm(lin)
        assert cur is not None and cur.data.data_ptr() == ptr  # same storage: graphs captured over it stay valid
        return y

    y0 = check()
    tracked_copy = lin._weight_fm
    n0 = tracked_copy.rebuilds
    check()
    assert tracked_copy.rebuilds == n0  # no write in between: no re-shuffle (the hot path costs a tuple comparison)
    new = lambda: (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    lin.weight.data.copy_(new())                      # default_weight_loader's form
    assert not torch.equal(check(), y0)
    with torch.no_grad():
        lin.weight.copy_(new())                       # load_state_dict's form (bumps the parameter's own version)
    check()
    lin.weight.data = new()                           # storage replaced
    check()
    lin.weight.weight_loader(lin.weight, new())       # the layer's checkpoint loader
    check()
    sd = {k: v.clone() for k, v in lin.state_dict().items()}
    lin.weight.data.zero_()
    lin.load_state_dict(sd)
    check()
    assert tracked_copy.rebuilds > n0
    # what the epoch cannot see -- a write through an alias taken EARLIER (its own version counter) -- is what the
    # weight-update path's explicit hook is for: TrackedCopy16.refresh_all() re-shuffles every live copy into the same storage
    alias = raw_data(lin.weight)
    check()                                           # (the copy is up to date with the alias's current bytes)
    alias.copy_(new())                                # invisible to the epoch
    n1 = tracked_copy.rebuilds
    assert ops.TrackedCopy16.refresh_all() >= 1 and tracked_copy.rebuilds == n1 + 1
    check()
    # ... and nothing is re-shuffled inside a stream capture (the launch would be baked into the graph): get() hands out the
    # copy as it is, refresh_all() refuses
    lin.weight.data.copy_(new())
    n2 = tracked_copy.rebuilds
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s_):
        assert tracked_copy.get() is not None and tracked_copy.rebuilds == n2
        with pytest.raises(RuntimeError, match="stream capture"):
            ops.TrackedCopy16.refresh_all()
    check()                                           # the next eager use brings it up to date
    assert tracked_copy.rebuilds == n2 + 1
    # a parameter that no longer is this matrix: the copy is dropped, F.linear serves
    lin.weight.data = new()[:, :256].contiguous()
    assert lin.quant_method.weight_fm(lin) is None
    y, _ = lin(x[:, :256].contiguous())
    ref = x[:, :256].double() @ raw_data(lin.weight).double().t()
    torch.testing.assert_close(y.double(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * float(ref.abs().max()) * 0.05)


def test_lm_head_copy_follows_in_place_updates():
    from sglang_npu_amd import model as Mo
    from sglang_npu_amd.harness import ModelConfig
    cfg = ModelConfig(8, 8, 64, 512, 1024, 1, 4096, 256)
    net = Mo.LlamaForCausalLM(cfg, None, torch.bfloat16, DEV).load_dummy_weights()
    fm = net.lm_head_shuffled
    if fm is None:
        pytest.skip("LM head kept row-major (SGL_MI355_NO_LM_HEAD_SHUFFLE)")
    ptr = fm.data.data_ptr()
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(3, 512, device=DEV, generator=g).bfloat16()
    y0 = ops.linear16(x, net.lm_head_shuffled)
    net.lm_head.data.copy_((torch.randn(4096, 512, device=DEV, generator=g) * 0.02).bfloat16())
    fm2 = net.lm_head_shuffled
    assert fm2 is not None and fm2.data.data_ptr() == ptr
    y1 = ops.linear16(x, fm2)
    from sglang_npu_amd.parameter import raw_data
    assert torch.equal(y1, ops.linear16(x, ops.linear16_shuffle_weight(raw_data(net.lm_head)))) and not torch.equal(y1, y0)


@pytest.mark.parametrize("K", [16, 48, 112])
@pytest.mark.parametrize("M", [1, 64, 129])
@pytest.mark.parametrize("strided", [False, True])
def test_k_shorter_than_one_k_step_reads_nothing_past_its_rows(M, K, strided):
    """Regression for commit 899639d (VERDICT r3 weak #3b): the clamped K-tail load `ld16` read up to 112 bytes past
    a row shorter than one 128-byte k-step -- past the allocation on the last row of A or W, a GPU memory fault when the
    tensor ended a mapped region.  A and W are the TAIL slices of larger buffers whose other bytes are NaN (0x7F), so an
    over-read that lands in the same allocation shows as a wrong value, and the values must match the C oracle on every
    kernel family these shapes reach, with and without bias."""
    g = torch.Generator().manual_seed(M * 131 + K + (7 if strided else 0))
    NAN8 = 0x7F  # e4m3fn NaN
    for N in (16, 72, 256):
        a_cpu, w_cpu = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
        lda = K + 16 if strided else K
        # A's last byte A[M-1, K-1] is the last byte of its allocation (compact and strided alike); in the strided form the
        # bytes between rows are NaN too
        if strided:
            abuf = torch.full((M * lda + 64,), NAN8, dtype=torch.uint8, device=DEV)
            a = abuf[-((M - 1) * lda + K):].as_strided((M, K), (lda, 1))
        else:
            abuf = torch.full((M + 3, K), NAN8, dtype=torch.uint8, device=DEV)
            a = abuf.view(-1)[-(M * K):].view(M, K)
        a.copy_(a_cpu.view(torch.uint8).to(DEV))
        wbuf = torch.full((N + 5, K), NAN8, dtype=torch.uint8, device=DEV)
        w = wbuf.view(-1)[-(N * K):].view(N, K)
        w.copy_(w_cpu.view(torch.uint8).to(DEV))
        a, w = a.view(torch.float8_e4m3fn), w.view(torch.float8_e4m3fn)
        sa = torch.rand(M, generator=g) * 1e-3 + 1e-4
        sb = torch.rand(N, generator=g) * 1e-3 + 1e-4
        for dt, with_bias in ((torch.bfloat16, False), (torch.float16, True)):
            bias = torch.randn(N, generator=g).to(dt) if with_bias else None
            ref = oracle.fp8_scaled_mm(a_cpu, w_cpu.t(), sa, sb, dt, bias)
            out = ops.fp8_scaled_mm(a, w.t(), sa.to(DEV), sb.to(DEV), dt, bias.to(DEV) if with_bias else None)
            assert bool(torch.isfinite(out.float()).all()), f"NaN padding leaked into the product ({ops.fp8_last_kernel()})"
            ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
            torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=ulp, atol=1e-3 * float(ref.float().abs().max()),
                                       msg=lambda m: f"{m} (kernel {ops.fp8_last_kernel()}, N={N})")
            part = ops.fp8_scaled_mm_partials(a, w.t(), sa.to(DEV), sb.to(DEV), dt, bias.to(DEV) if with_bias else None)
            if part is not None:
                torch.testing.assert_close(part.finalize().float().cpu(), ref.float(), rtol=ulp,
                                           atol=1e-3 * float(ref.float().abs().max()))


@pytest.mark.parametrize("M", [7, 64, 100, 128, 200, 256, 300])
@pytest.mark.parametrize("N,K,shuffled", [(4096, 14336, True), (6144, 4096, True), (28672, 1024, True), (1280, 3584, False)])
def test_fp8_scaled_mm_takes_rows_of_a_wider_activation_buffer(M, N, K, shuffled):
    """mat_a as a strided view (rows of a wider buffer, stride(0) > K), the form the model hands over when the activation is
    a slice: every kernel family on the way (weight streamer with 64- / 128- / 256-row phases, tiled) must read the rows it
    is given -- same bits as on a compact copy."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    wide = ((torch.rand(M, K + 256, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    a = wide[:, :K]
    assert a.stride(0) == K + 256
    w = ((torch.rand(N, K, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    wt = ops.fp8_shuffle_weight(w) if shuffled else w.t()
    out = ops.fp8_scaled_mm(a, wt, sa, sb, torch.bfloat16)
    name = ops.fp8_last_kernel()
    ref = ops.fp8_scaled_mm(a.contiguous(), wt, sa, sb, torch.bfloat16)
    assert ops.fp8_last_kernel() == name
    assert torch.equal(out, ref)
    rows = torch.tensor(sorted({0, M // 2, M - 1}), device=DEV)
    truth = (a[rows].double() @ w.double().t()) * sb.double().view(1, -1) * sa[rows].double()
    torch.testing.assert_close(out[rows].double(), truth, rtol=2.0 ** -7, atol=1e-3 * float(truth.abs().max()))
