"""GPU: the pre-shuffled (fragment-major) FP8 weight layout of the decode GEMMs.
* the re-layout kernel against the index formula in include/sgl_mi355.h, and its inverse;
* fp8_scaled_mm / fp8_scaled_mm_partials on a shuffled weight against the same call on the row-major weight:
  bit-identical wherever both run the same kernel (same products, same fp32 summation order -- only the addresses of the
  loads differ), within one output ulp where the shuffled weight forces another kernel, and against an fp64 product;
* W8A8Fp8LinearMethod.process_weights_after_loading shuffles eligible weights, leaves others row-major, and a
  model step with shuffled weights equals the step with SGL_MI355_NO_WSHUFFLE semantics;
* the layout is STRUCTURAL (round 4): a uint8 [N / 16, 16 K] tensor -- it survives .data / detach / deepcopy / state_dict,
  and what is not that tensor is never read as shuffled bytes."""
import copy


import pytest
import torch

from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand_fp8(shape, g):
    return ((torch.rand(shape, generator=g, device=DEV) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)


def _shuffle_by_formula(w):
    """piece = ((n/16) * (K/128) + k/128) * 2 + (k%128)/64 ; inside = ((k%64)/16)*256 + (n%16)*16 + k%16."""
    N, K = w.shape
    b = w.view(torch.uint8).view(N // 16, 16, K // 128, 2, 4, 16)  # [nb, r16, ks, half, g, byte]
    return b.permute(0, 2, 3, 4, 1, 5).contiguous().view(N // 16, 16 * K)  # [nb, ks, half, g, r16, byte]: one row per 16-column block


@pytest.mark.parametrize("N,K", [(16, 512), (48, 1024), (4096, 4096), (1280, 3584), (28672, 4096)])
def test_shuffle_weight_layout_and_inverse(N, K):
    g = torch.Generator(device=DEV).manual_seed(N + K)
    w = torch.randint(0, 256, (N, K), dtype=torch.uint8, device=DEV, generator=g).view(torch.float8_e4m3fn)
    sh = ops.fp8_shuffle_weight(w)
    assert ops.is_wshuffled(sh) and sh.dtype == torch.uint8 and sh.shape == (N // 16, 16 * K)
    assert ops.fp8_weight_kn(sh) == (K, N) and ops.fp8_weight_kn(w.t()) == (K, N)
    assert torch.equal(sh, _shuffle_by_formula(w))
    back = ops.fp8_shuffle_weight(sh, inverse=True)
    assert not ops.is_wshuffled(back) and back.dtype == torch.float8_e4m3fn and back.shape == (N, K)
    assert torch.equal(back.view(torch.uint8), w.view(torch.uint8))
    # into existing storage (what a reload under captured graphs needs)
    again = torch.zeros_like(sh)
    assert ops.fp8_shuffle_weight(w, out=again) is again and torch.equal(again, sh)
    # a strided source (rows of a wider buffer) is re-laid the same
    wide = torch.zeros(N, K + 256, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    wide[:, :K] = w
    assert torch.equal(ops.fp8_shuffle_weight(wide[:, :K]).view(torch.uint8), sh.view(torch.uint8))


def test_shuffle_weight_rejects_other_shapes():
    w = torch.zeros(24, 512, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    with pytest.raises(RuntimeError, match="N % 16 == 0 and K % 512 == 0"):
        ops.fp8_shuffle_weight(w)
    w = torch.zeros(32, 640, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    with pytest.raises(RuntimeError, match="N % 16 == 0 and K % 512 == 0"):
        ops.fp8_shuffle_weight(w)
    assert not ops.fp8_shuffle_supported(24, 512) and ops.fp8_shuffle_supported(1280, 3584)


SHAPES = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (1024, 512), (48, 1024), (1280, 3584), (20480, 1024),
          (512, 7168)]


@pytest.mark.parametrize("M", [1, 7, 16, 17, 32, 33, 64, 65, 100, 128, 129, 200, 256, 1000])  # 65..128 / 129..256: the streamer's 128- / 256-row forms
@pytest.mark.parametrize("N,K", SHAPES)
def test_fp8_scaled_mm_on_shuffled_weight(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M * 1000 + N + K)
    dt = torch.bfloat16 if (M + N // 16) % 2 == 0 else torch.float16
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if M % 2 else None
    plain = ops.fp8_scaled_mm(a, w.t(), sa, sb, dt, bias)
    wsh = ops.fp8_shuffle_weight(w)
    out = ops.fp8_scaled_mm(a, wsh, sa, sb, dt, bias)
    ref = (a.double() @ w.double().t()) * sb.double().view(1, -1) * sa.double()
    if bias is not None:
        ref = ref + bias.double()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(out.float(), ref.float(), rtol=ulp, atol=1e-3 * float(ref.abs().max()))
    torch.testing.assert_close(out.float(), plain.float(), rtol=2 * ulp, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(4000, 6160, 1024), (1537, 4112, 512), (3000, 4096, 14336), (2048, 4096, 4096),
                                   (8192, 1280, 1536), (1025, 6144, 4096), (130, 28672, 512),
                                   # the headline prefill shapes bench.py's roofline_gemm times (Llama-3-8B gate_up / down at M = 4096)
                                   (4096, 28672, 4096), (4096, 4096, 14336), (4096, 6144, 4096),
                                   # and the TTFT pass (M = 1024): every shape the 1024-token prefill runs
                                   (1024, 28672, 4096), (1024, 4096, 14336),
                                   # exactly one 128-row tile per CU: the eight-wave form with two k groups (round 4)
                                   (1024, 6144, 4096), (1000, 6144, 1024), (1024, 8192, 8192),
                                   # few row tiles x a very wide output: the 256 x 256 eight-wave form (round 5 dispatch rule), ragged both ways
                                   (1000, 28000, 1024), (769, 32768, 512)])
def test_prefill_shapes_on_shuffled_weight(M, N, K):
    """>= 192 tiles of 128 x 256 on a pre-shuffled weight: fp8_gemm_tiled3_kernel (weights global -> VGPR).  Same products
    and the same fp32 summation order over k as the row-major tiled kernel -> bit-identical; ragged M, N % 256 != 0."""
    assert ((M + 127) // 128) * ((N + 255) // 256) >= 128
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dt = torch.bfloat16 if M % 2 == 0 else torch.float16
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if N % 32 else None
    plain = ops.fp8_scaled_mm(a, w.t(), sa, sb, dt, bias)
    out = ops.fp8_scaled_mm(a, ops.fp8_shuffle_weight(w), sa, sb, dt, bias)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    if ops.fp8_last_kernel() == "tiled3_ks2":  # two k groups: even and odd k-steps summed apart, then added
        assert ((M + 127) // 128) * min((N + 255) // 256, (N + 191) // 192) <= 256
        torch.testing.assert_close(out.float(), plain.float(), rtol=2 * ulp, atol=1e-3 * float(plain.float().abs().max()))
    else:
        assert torch.equal(out, plain)
    rows = torch.tensor([0, 1, 127, 128, M // 2, M - 2, M - 1], device=DEV)  # fp64 truth on a few rows (incl. the ragged edge)
    ref = (a[rows].double() @ w.double().t()) * sb.double().view(1, -1) * sa[rows].double()
    if bias is not None:
        ref = ref + bias.double()
    torch.testing.assert_close(out[rows].float(), ref.float(), rtol=ulp, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("M", [16, 64])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)])
def test_headline_decode_shapes_are_bit_identical_to_row_major(M, N, K):
    """Both layouts run fp8_gemm_wstream_kernel with the same (phase, consumer, slice) choice here: same bits."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    wsh = ops.fp8_shuffle_weight(w)
    assert torch.equal(ops.fp8_scaled_mm(a, wsh, sa, sb, torch.bfloat16), ops.fp8_scaled_mm(a, w.t(), sa, sb, torch.bfloat16))
    p0 = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16)
    if p0 is not None:
        f0, n0 = p0.finalize(), p0.num_slices
        p1 = ops.fp8_scaled_mm_partials(a, wsh, sa, sb, torch.bfloat16)
        assert p1 is not None and p1.num_slices == n0
        assert torch.equal(p1.finalize(), f0)


def test_linear_method_shuffles_eligible_weights_only(monkeypatch):
    from sglang_npu_amd import quantization as Q
    from sglang_npu_amd.linear import RowParallelLinear

    def make(n, k, shuffle):
        monkeypatch.setattr(Q, "PRESHUFFLE_FP8_WEIGHTS", shuffle)
        lin = RowParallelLinear(k, n, bias=False, quant_config=Q.W8A8Fp8Config(is_checkpoint_fp8_serialized=False),
                                params_dtype=torch.bfloat16).to(DEV)
        g = torch.Generator(device=DEV).manual_seed(n + k)
        lin.weight.data.copy_(torch.randn(n, k, device=DEV, generator=g) * 0.05)
        lin.quant_method.process_weights_after_loading(lin)
        return lin

    x = torch.randn(40, 1024, device=DEV, dtype=torch.bfloat16)
    a, b = make(512, 1024, True), make(512, 1024, False)
    assert ops.is_wshuffled(a.weight) and not ops.is_wshuffled(b.weight)
    assert a.weight.shape == (512 // 16, 16 * 1024) and a.weight.dtype == torch.uint8 and b.weight.shape == (1024, 512)
    assert torch.equal(ops.fp8_shuffle_weight(a.weight.data, inverse=True).view(torch.uint8), b.weight.t().contiguous().view(torch.uint8))
    ya, yb = a(x)[0], b(x)[0]
    torch.testing.assert_close(ya.float(), yb.float(), rtol=2.0 ** -6, atol=1e-3 * float(yb.float().abs().max()))
    # the hook is idempotent: re-entered on its own result (the reference re-runs it after a reload, loader.py:456) it returns
    # at once -- same bytes, same output, and the SAME STORAGE (captured graphs hold its raw pointer), whatever the
    # re-layout switch says by then
    before, ptr_a, ptr_b = a.weight.data.clone(), a.weight.data_ptr(), b.weight.data_ptr()
    for flag in (True, False):
        monkeypatch.setattr(Q, "PRESHUFFLE_FP8_WEIGHTS", flag)
        a.quant_method.process_weights_after_loading(a)
        assert ops.is_wshuffled(a.weight) and a.weight.data_ptr() == ptr_a and torch.equal(a.weight.data, before)
        assert torch.equal(a(x)[0], ya)
        b.quant_method.process_weights_after_loading(b)
        assert b.weight.shape == (1024, 512) and b.weight.data_ptr() == ptr_b and torch.equal(b(x)[0], yb)
    c = make(520, 1024, True)  # N % 16 != 0: stays row-major
    assert not ops.is_wshuffled(c.weight)
    d = make(512, 640, True)   # K % 512 != 0
    assert not ops.is_wshuffled(d.weight)


@pytest.mark.parametrize("M,N,K", [(1024, 28672, 4096), (4096, 28672, 4096), (1000, 14400, 1024), (1537, 6176, 512),
                                   (200, 2 * 16384, 512)])
def test_gate_up_gemm_with_silu_mul_in_the_epilogue(M, N, K):
    """sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled (fp8_gemm_tiled3_kernel, SILU): the gate_up GEMM that writes
    silu(gate) * up.  Bit-identical to fp8_scaled_mm on the same weight followed by silu_and_mul -- the Llama-3-8B prefill
    shapes (M = 1024 / 4096), ragged M, I % 128 != 0, bias, fp16."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dt = torch.bfloat16 if M % 2 == 0 else torch.float16
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if (N // 2) % 128 else None
    wsh = ops.fp8_shuffle_weight(w)
    y = ops.fp8_scaled_mm(a, wsh, sa, sb, dt, bias)
    ref = torch.empty(M, N // 2, dtype=dt, device=DEV)
    ops.silu_and_mul(y, ref)
    out = ops.fp8_scaled_mm_silu_mul(a, wsh, sa, sb, dt, bias)
    assert out is not None and ops.fp8_last_kernel() == "tiled3_silu"
    assert torch.equal(out, ref)
    assert float(ref.float().abs().max()) > 0  # (not a comparison of zeros)
    # ... and the row quant on it is what the fused silu * mul + quant kernel gives on y (the model's two prefill paths)
    q_ref, s_ref = ops.silu_and_mul_quant_fp8(y)
    q = torch.empty_like(out, dtype=torch.float8_e4m3fn)
    s = torch.empty(M, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(out, q, s)
    assert torch.equal(s, s_ref) and torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8))


def test_gate_up_gemm_with_silu_mul_declines_other_shapes():
    g = torch.Generator(device=DEV).manual_seed(3)
    a, w = _rand_fp8((64, 512), g), _rand_fp8((4096, 512), g)
    sa, sb = torch.ones(64, 1, device=DEV), torch.ones(4096, 1, device=DEV)
    wsh = ops.fp8_shuffle_weight(w)
    assert ops.fp8_scaled_mm_silu_mul(a, wsh, sa, sb, torch.bfloat16) is None          # decode rows
    a = _rand_fp8((512, 512), g)
    sa = torch.ones(512, 1, device=DEV)
    assert ops.fp8_scaled_mm_silu_mul(a, wsh, sa, sb, torch.bfloat16) is None          # 4 x 16 = 64 tiles: too few
    assert ops.fp8_scaled_mm_silu_mul(a, w.t(), sa, sb, torch.bfloat16) is None        # row-major weight


def test_prefill_mlp_is_bit_identical_with_and_without_the_silu_gemm_fusion(monkeypatch):
    """LlamaMLP.forward_fp8 at a prefill size: the fused gate_up launch + row quant against gate_up GEMM + silu * mul + quant."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import ModelConfig
    cfg = ModelConfig(8, 8, 128, 1024, 3072, 1, 512, 2048)
    net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
    mlp = net.layers[0].mlp
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(1024, 1024, device=DEV, generator=g).to(torch.bfloat16)
    xq = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    xs = torch.empty(1024, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(x, xq, xs)
    taken = []
    real = ops.fp8_scaled_mm_silu_mul

    def counted(*a, **kw):
        r = real(*a, **kw)
        taken.append(r is not None)
        return r

    monkeypatch.setattr(ops, "fp8_scaled_mm_silu_mul", counted)
    outs = []
    for fuse in (False, True):
        monkeypatch.setattr(M, "FUSE_SILU_GEMM", fuse)
        outs.append(mlp.forward_fp8(xq, xs, torch.bfloat16).clone())
    assert taken == [ops.is_wshuffled(mlp.gate_up_proj.weight)]  # (row-major weights under SGL_MI355_NO_WSHUFFLE: two launches)
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,dt", [(256, 2 * 12288, 512, torch.bfloat16), (1024, 2 * 3072, 512, torch.float16)])
def test_gate_up_gemm_with_silu_mul_against_the_oracle_chain(M, N, K, dt):
    """The fused launch against the CPU oracle's own chain: orc_fp8_scaled_mm (fp8_gemm_kernel.cu:1071-1146 restated) then
    orc_silu_and_mul (activation.cu:56-60) -- not only against this library's two launches."""
    import oracle
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a, w = _rand_fp8((M, K), g), _rand_fp8((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-3 + 1e-4  # full-range e4m3 operands: y of order 1, nothing overflows fp16
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-3 + 1e-4
    bias = (torch.randn(N, generator=g, device=DEV) * 0.1).to(dt)
    out = ops.fp8_scaled_mm_silu_mul(a, ops.fp8_shuffle_weight(w), sa, sb, dt, bias)
    assert out is not None and ops.fp8_last_kernel() == "tiled3_silu"
    y = oracle.fp8_scaled_mm(a.cpu(), w.cpu().t(), sa.cpu(), sb.cpu(), dt, bias.cpu())
    ref = oracle.silu_and_mul(y).float()
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0.1
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    # The GEMM itself is held to (one output ulp) + 1e-3 max|y| against the oracle (test_fp8_scaled_mm_vs_oracle: fp32 sums of
    # exact products in another order; the absolute part matters where a row's products cancel).  Carried through
    # out = silu(g) u with |silu'| <= 1.1 and |silu(g)| <= |g|, plus the roundings of silu(g) and of the product:
    yf = y.float()
    gate, up = yf[:, : N // 2].abs(), yf[:, N // 2:].abs()
    ymax = float(yf.abs().max())
    dg, du = ulp * gate + 1e-3 * ymax, ulp * up + 1e-3 * ymax
    tol = 1.1 * up * dg + gate * du + 3.0 * ulp * gate * up + 1e-6
    err = (out.float().cpu() - ref).abs()
    assert bool((err <= tol).all()), f"max excess {float((err - tol).max())}"
    # ... and on the GEMM result this library itself produces, the fused launch is the oracle's silu_and_mul bit for bit but
    # for the exp (expf there, the hardware exp2 path here: an ulp of silu(g) in a few elements per million)
    y_gpu = ops.fp8_scaled_mm(a, ops.fp8_shuffle_weight(w), sa, sb, dt, bias)
    ref2 = oracle.silu_and_mul(y_gpu.cpu()).float()
    diff = out.float().cpu() != ref2
    assert float(diff.float().mean()) < 1e-3
    assert bool(((out.float().cpu() - ref2).abs() <= 2.0 * ulp * ref2.abs() + 1e-6).all())


def test_shuffled_layout_is_structural_not_a_tag():
    """whatever drops Python attributes (.data, .detach(), deepcopy, a state_dict round trip) keeps the
    layout, because the layout IS the dtype and shape; and nothing else is taken for it."""
    from sglang_npu_amd import quantization as Q
    from sglang_npu_amd.linear import RowParallelLinear
    N, K = 1024, 2048
    lin = RowParallelLinear(K, N, bias=False, quant_config=Q.W8A8Fp8Config(is_checkpoint_fp8_serialized=False),
                            params_dtype=torch.bfloat16).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    w16 = (torch.randn(N, K, device=DEV, generator=g) * 0.05).bfloat16()
    lin.weight.data.copy_(w16)
    lin.quant_method.process_weights_after_loading(lin)
    assert ops.is_wshuffled(lin.weight)
    x = torch.randn(33, K, device=DEV, generator=g).bfloat16()
    y = lin(x)[0]
    # the truth: the same per-channel quantised weight, row-major
    wq, ws = Q.per_channel_quant_fp8_weight(w16)
    xq = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    xs = torch.empty(33, 1, dtype=torch.float32, device=DEV)
    ops.sgl_per_token_quant_fp8(x, xq, xs)
    ref = ops.fp8_scaled_mm(xq, wq.t(), xs, ws, torch.bfloat16)
    torch.testing.assert_close(y.float(), ref.float(), rtol=2.0 ** -6, atol=1e-3 * float(ref.float().abs().max()))
    # every attribute-dropping copy is still the shuffled weight and multiplies the same
    for w in (lin.weight.data, lin.weight.detach(), copy.deepcopy(lin.weight), lin.weight.data.clone()):
        assert ops.is_wshuffled(w)
        assert torch.equal(ops.fp8_scaled_mm(xq, w, xs, lin.weight_scale, torch.bfloat16), y)
    twin = copy.deepcopy(lin)
    assert ops.is_wshuffled(twin.weight) and torch.equal(twin(x)[0], y)
    sd = {k: v.clone() for k, v in lin.state_dict().items()}
    lin.weight.data.zero_()
    lin.load_state_dict(sd)
    assert ops.is_wshuffled(lin.weight) and torch.equal(lin(x)[0], y)
    # the [N, K] / [K, N] matrix does not fit into the parameter: an in-place reload fails on the shape (loudly) ...
    with pytest.raises(RuntimeError):
        lin.weight.data.copy_(wq)
    with pytest.raises(RuntimeError):
        lin.weight.data.copy_(wq.t())
    # ... the same bytes seen as FP8 or under another shape are refused by the GEMM, never multiplied as row-major
    with pytest.raises(RuntimeError, match="column major|cannot be multiplied"):
        ops.fp8_scaled_mm(xq, lin.weight.data.view(torch.float8_e4m3fn), xs, lin.weight_scale, torch.bfloat16)
    with pytest.raises(RuntimeError, match="fragment-major"):
        ops.fp8_scaled_mm(xq, lin.weight.data.view(N, K), xs, lin.weight_scale, torch.bfloat16)  # uint8 but [N, K], K % 8192 != 0
    with pytest.raises(RuntimeError, match="fragment-major"):
        ops.fp8_scaled_mm(xq, lin.weight.data.t(), xs, lin.weight_scale, torch.bfloat16)         # not contiguous
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        ops.fp8_scaled_mm(xq[:, :1024].contiguous(), lin.weight.data, xs, lin.weight_scale, torch.bfloat16)
