"""torch.ops.sgl_kernel.* registration (sglang_npu_amd/torch_ops.py): CPU test of the schemas, GPU test of the dispatch."""
import pytest
import torch

from sglang_npu_amd import torch_ops


def test_schemas_match_the_reference_registration():
    rep = torch_ops.register()
    assert set(rep) == set(torch_ops._SCHEMAS) and all(v in ("defined+impl", "impl", "kept existing") for v in rep.values())
    assert torch_ops.register() is rep  # idempotent
    # the dispatcher holds exactly the reference's schema strings (common_extension.cc:98-130, torch_extension_cpu.cpp:263-275)
    s = str(torch.ops.sgl_kernel.fp8_scaled_mm.default._schema)
    assert s == ("sgl_kernel::fp8_scaled_mm(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, "
                 "ScalarType out_dtype, Tensor? bias) -> Tensor")
    s = str(torch.ops.sgl_kernel.decode_attention_cpu.default._schema)
    assert s.startswith("sgl_kernel::decode_attention_cpu(Tensor query, Tensor k_cache, Tensor v_cahce, Tensor output, Tensor key")
    assert "float sm_scale, float logit_cap) -> ()" in s
    s = str(torch.ops.sgl_kernel.sgl_per_token_group_quant_fp8.default._schema)
    assert "int group_size, float eps, float fp8_min, float fp8_max, bool scale_ue8m0) -> ()" in s
    s = str(torch.ops.sgl_kernel.fused_add_rmsnorm.default._schema)
    assert "! -> ) input" in s and "! -> ) residual" in s and "Tensor weight, float eps, bool enable_pdl) -> ()" in s  # Tensor! input
    # there is no CPU kernel behind these names: the product has no CPU path
    x = torch.zeros(2, 8, dtype=torch.bfloat16)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.sgl_kernel.sgl_per_token_quant_fp8(x, torch.empty(2, 8, dtype=torch.float8_e4m3fn), torch.empty(2))


@pytest.mark.gpu
def test_dispatcher_calls_reach_the_hip_library():
    from sglang_npu_amd import ops
    torch_ops.register()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(5, 512, device=dev, generator=g).bfloat16()
    q, s = torch.empty(5, 512, dtype=torch.float8_e4m3fn, device=dev), torch.empty(5, 1, device=dev)
    torch.ops.sgl_kernel.sgl_per_token_quant_fp8(x, q, s)
    q2, s2 = torch.empty_like(q), torch.empty_like(s)
    ops.sgl_per_token_quant_fp8(x, q2, s2)
    assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(s, s2)
    w = ((torch.rand(256, 512, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sb = torch.rand(256, device=dev, generator=g) * 1e-2
    y = torch.ops.sgl_kernel.fp8_scaled_mm(q, w.t(), s, sb, torch.bfloat16, None)
    assert torch.equal(y, ops.fp8_scaled_mm(q, w.t(), s, sb, torch.bfloat16))
    # elementwise ops with the reference's output-first signatures
    wt = (torch.rand(512, device=dev, generator=g) + 0.5).bfloat16()
    out = torch.empty_like(x)
    torch.ops.sgl_kernel.rmsnorm(out, x, wt, 1e-6, False)
    assert torch.equal(out, ops.rmsnorm(x, wt, 1e-6))
    y2 = torch.randn(5, 1024, device=dev, generator=g).bfloat16()
    o2 = torch.empty(5, 512, dtype=torch.bfloat16, device=dev)
    torch.ops.sgl_kernel.silu_and_mul(o2, y2)
    assert torch.equal(o2, ops.silu_and_mul(y2))
    # awq_dequantize through the dispatcher
    qw = torch.randint(0, 2 ** 31 - 1, (128, 16), dtype=torch.int32, device=dev, generator=g)
    qz = torch.randint(0, 2 ** 31 - 1, (1, 16), dtype=torch.int32, device=dev, generator=g)
    sc = torch.rand(1, 128, device=dev, generator=g).half()
    assert torch.equal(torch.ops.sgl_kernel.awq_dequantize(qw, sc, qz), ops.awq_dequantize(qw, sc, qz))
    # RoPE in the flashinfer-style op form the sgl_kernel wrapper uses (elementwise.py apply_rope_with_cos_sin_cache_inplace)
    T, H, D = 7, 4, 64
    qq = torch.randn(T, H * D, device=dev, generator=g).bfloat16()
    kk = torch.randn(T, 2 * D, device=dev, generator=g).bfloat16()
    cache = torch.rand(128, D, device=dev, generator=g)
    pos = torch.randint(0, 128, (T,), device=dev, generator=g)
    q_ref, k_ref = qq.clone(), kk.clone()
    ops.apply_rope_with_cos_sin_cache_inplace(pos, q_ref, k_ref, D, cache, True)
    q3, k3 = qq.view(T, H, D), kk.view(T, 2, D)
    torch.ops.sgl_kernel.apply_rope_pos_ids_cos_sin_cache(q3, k3, q3, k3, cache, pos, False, 0)
    assert torch.equal(qq, q_ref) and torch.equal(kk, k_ref)
