"""GPU: FP8 companions -- the drop-in RMSNorm / SiluAndMul quantise their output per token in the same pass once an FP8 linear
has asked for it, and W8A8Fp8LinearMethod.apply then skips its own quant launch.  Reference call order, untouched model code
(RMSNorm -> LinearMethodBase.apply, layernorm.py:59-172 -> fp8_utils.py:653-704); results must not change by one bit."""
import pytest
import torch

from sglang_npu_amd import ops
from sglang_npu_amd.layers import RMSNorm, SiluAndMul
from sglang_npu_amd.linear import MergedColumnParallelLinear, RowParallelLinear
from sglang_npu_amd.quantization import W8A8Fp8Config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fp8_linear(cls, k, n, g, dtype):
    lin = cls(k, [n] if cls is MergedColumnParallelLinear else n, params_dtype=dtype,
              quant_config=W8A8Fp8Config(is_checkpoint_fp8_serialized=False)).to(DEV)
    w = (torch.rand(n, k, generator=g, device=DEV) * 2e-2 - 1e-2).to(dtype)
    lin.weight.weight_loader(lin.weight, w) if cls is RowParallelLinear else lin.weight.data.copy_(w)
    lin.quant_method.process_weights_after_loading(lin)
    return lin


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H,I", [(64, 4096, 14336), (1, 1024, 512), (300, 2048, 1536)])
def test_norm_and_activation_companions_are_bit_identical_and_save_the_quant_launch(dtype, T, H, I):
    g = torch.Generator(device=DEV).manual_seed(T + H)
    norm = RMSNorm(H, 1e-5, dtype).to(DEV)
    norm.weight.data = (torch.rand(H, generator=g, device=DEV) + 0.5).to(dtype)
    act = SiluAndMul()
    up = _fp8_linear(MergedColumnParallelLinear, H, 2 * I, g, dtype)
    down = _fp8_linear(RowParallelLinear, I, H, g, dtype)
    x0 = torch.randn(T, H, generator=g, device=DEV).to(dtype)
    r0 = torch.randn(T, H, generator=g, device=DEV).to(dtype)
    quant_calls = []
    real = ops.sgl_per_token_quant_fp8
    ops.sgl_per_token_quant_fp8 = lambda *a, **k: (quant_calls.append(1), real(*a, **k))[1]
    try:
        outs = []
        for it in range(3):
            x, r = x0.clone(), r0.clone()
            n_before = len(quant_calls)
            h, r = norm(x, r)               # models/llama.py:245-268 call order, nothing of it changed
            y, _ = up(h)
            a = act(y)
            z, _ = down(a)
            torch.cuda.synchronize()
            outs.append((h.clone(), r.clone(), y.clone(), a.clone(), z.clone(), len(quant_calls) - n_before))
        # pass 0: both linears quantise themselves (and tell their producers); passes 1, 2: no quant launch at all
        assert [o[5] for o in outs] == [2, 0, 0]
        assert norm.emit_fp8_companion and act.emit_fp8_companion
        for o in outs[1:]:
            for a_, b_ in zip(o[:5], outs[0][:5]):
                assert torch.equal(a_, b_), "a companion changed a result"
        # the companion is exactly the standalone quantiser's output
        h, _ = norm(x0.clone(), r0.clone())
        q_ref = torch.empty(T, H, dtype=torch.float8_e4m3fn, device=DEV)
        s_ref = torch.empty(T, 1, device=DEV)
        real(h.contiguous(), q_ref, s_ref)
        q, s = ops.take_fp8_companion(h)
        assert torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s, s_ref)
        # ... and is dropped the moment somebody writes to the tensor: through torch (version counter) ...
        h.add_(1)
        assert ops.take_fp8_companion(h) is None
        y2, _ = up(h)
        torch.cuda.synchronize()
        assert len(quant_calls) == sum(o[5] for o in outs) + 1, "a stale companion must not be used"
        # ... or through one of this library's in-place ops (raw pointers: the op drops the tag itself)
        h2, _ = norm(x0.clone(), r0.clone())
        assert ops.take_fp8_companion(h2) is not None
        ops.fused_add_rmsnorm(h2, r0.clone(), norm.weight.data, 1e-5)
        assert ops.take_fp8_companion(h2) is None
        # a view is another object: it has no companion of its own
        h3, _ = norm(x0.clone(), r0.clone())
        assert ops.take_fp8_companion(h3[: max(1, T // 2)]) is None
    finally:
        ops.sgl_per_token_quant_fp8 = real


def test_companions_can_be_switched_off(monkeypatch):
    monkeypatch.setattr(ops, "FP8_COMPANIONS", False)
    norm = RMSNorm(512, 1e-5, torch.bfloat16).to(DEV)
    norm.emit_fp8_companion = True
    x = torch.randn(8, 512, device=DEV).bfloat16()
    out = norm(x)
    assert ops.take_fp8_companion(out) is None and not hasattr(out, "_sgl_mi355_producer")


def test_decode_attention_with_kv_splits_emits_the_companion_for_o_proj():
    """Fewer (request, kv head) items than CUs: the kv-splits are merged inside the attention launch, and once the FP8 o_proj
    that receives the output has asked, the merging workgroup quantises the row as well (attn -> o_proj, models/llama.py:189-190,
    untouched): no quant launch from the second pass on, not one bit different."""
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike, RadixAttention,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    B, Hq, Hk, D, dtype = 64, 8, 1, 128, torch.bfloat16   # one rank of Llama-3-70B at TP 8
    g = torch.Generator(device=DEV).manual_seed(5)
    cfg = ModelConfig(Hq, Hk, D, Hq * D, 1024, 1, 512, 2048)
    r2t = ReqToTokenPool(B, 2048, DEV)
    pool = MHATokenToKVPool(B * 2048 + 1, 1, dtype, Hk, D, 1, DEV)
    r2t.req_to_token.copy_((torch.randperm(B * 2048, device=DEV, generator=g) + 1).view(B, 2048).to(torch.int32))
    pool.k_buffer[0].normal_(generator=g)
    pool.v_buffer[0].normal_(generator=g)
    backend = install_attention_backend(ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs()))
    attn = RadixAttention(Hq, D, D ** -0.5, Hk, 0)
    o_proj = _fp8_linear(RowParallelLinear, Hq * D, 4096, g, dtype)
    rows = torch.arange(B, device=DEV)
    seq = torch.randint(900, 2000, (B,), device=DEV, generator=g)
    fb = ForwardBatch(ForwardMode.DECODE, B, None, rows, seq, r2t.req_to_token[rows, seq - 1].long(), int(seq.sum()), seq.cpu(),
                      seq - 1, req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    assert backend.forward_metadata.num_kv_splits > 1 and backend._fuse_split_merge(backend.forward_metadata.num_kv_splits, B)
    qkv = torch.randn(B, (Hq + 2 * Hk) * D, generator=g, device=DEV).to(dtype)
    q, k, v = qkv.split([Hq * D, Hk * D, Hk * D], dim=-1)
    quant_calls = []
    real = ops.sgl_per_token_quant_fp8
    ops.sgl_per_token_quant_fp8 = lambda *a, **kw: (quant_calls.append(1), real(*a, **kw))[1]
    try:
        outs = []
        for it in range(3):
            n0 = len(quant_calls)
            o = attn(q, k, v, fb)
            z, _ = o_proj(o)
            torch.cuda.synchronize()
            outs.append((o.clone(), (z + 0).clone(), len(quant_calls) - n0))
        assert [x[2] for x in outs] == [1, 0, 0] and attn.emit_fp8_companion
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[1][1], outs[2][1])
    finally:
        ops.sgl_per_token_quant_fp8 = real
