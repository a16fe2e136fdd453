"""GPU: host-side guards of the model harness (what must raise instead of reaching a kernel)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_sequences_past_the_context_length_are_refused_on_the_host():
    """Positions index the rotary table unchecked on the device (as upstream's rotary_embedding op): a batch whose host-side
    lengths exceed the model's context length must raise before any launch -- bench.py --ctx 8192 on the 8192-row table of
    the Llama-3-8B shape once read past it (a GPU memory fault)."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelConfig, ModelRunnerLike,
                                        ReqToTokenPool, ServerArgs, install_attention_backend)
    cfg = ModelConfig(8, 8, 128, 1024, 2048, 1, 512, 64)  # context_len = 64
    net = M.LlamaForCausalLM(cfg, "w8a8_fp8", torch.bfloat16, DEV).load_dummy_weights()
    B = 2
    r2t = ReqToTokenPool(B, 128, DEV)
    pool = MHATokenToKVPool(B * 128 + 1, 1, torch.bfloat16, 8, 128, 1, DEV)
    r2t.req_to_token.copy_((torch.arange(B * 128, device=DEV) + 1).view(B, 128).to(torch.int32))
    runner = ModelRunnerLike(cfg, r2t, pool, DEV, 0, 1, ServerArgs())
    backend = install_attention_backend(runner)
    ids = torch.tensor([1, 2], device=DEV)
    rows = torch.arange(B, device=DEV)
    for lens, ok in (([64, 10], True), ([65, 10], False)):
        seq = torch.tensor(lens, device=DEV)
        fb = ForwardBatch(ForwardMode.DECODE, B, ids, rows, seq, r2t.req_to_token[rows, seq - 1].long(), int(seq.sum()),
                          seq.cpu(), seq - 1, req_to_token_pool=r2t, token_to_kv_pool=pool, attn_backend=backend)
        backend.init_forward_metadata(fb)
        if ok:
            assert torch.isfinite(net(ids, seq - 1, fb).float()).all()
        else:
            with pytest.raises(RuntimeError, match="context length"):
                net(ids, seq - 1, fb)
