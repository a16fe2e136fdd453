"""GPU: kv-split decode whose merge (and per-token FP8 quant) runs inside the attention launch
(sgl_mi355_decode_attention_merged: the workgroup that publishes a request's last partial merges it) against the separate
launches it replaces -- sgl_mi355_decode_attention (+ stage 2) and sgl_mi355_decode_merge_quant_fp8: same bits, for every
pool format, ragged lengths (empty splits, empty sequences), repeated calls on one counter buffer, and under a HIP graph."""
import pytest
import torch

from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, Hq, Hkv, D, lens, dtype, kv_dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    S = max(1, int(max(lens)))
    n_tok = B * S + 1
    kb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(n_tok, Hkv, D, device=DEV, generator=g).to(dtype)
    if kv_dtype is not None:
        kb, vb = kb.to(kv_dtype), vb.to(kv_dtype)
    q = torch.randn(B, Hq, D, device=DEV, generator=g).to(dtype)
    r2t = (torch.randperm(n_tok - 1, device=DEV, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
    rpi = torch.arange(B, device=DEV)
    seq = torch.tensor(lens, device=DEV, dtype=torch.int64)
    return q, kb, vb, r2t, rpi, seq


CASES = [  # B, Hq, Hkv, D, splits, lens
    (64, 8, 1, 128, 4, [2048] * 64),                                  # one rank of Llama-3-70B TP=8
    (64, 4, 1, 128, 4, [1500 + 7 * i for i in range(64)]),            # one rank of Llama-3-8B TP=8, ragged
    (16, 32, 8, 128, 2, [300 + 50 * i for i in range(16)]),           # Llama-3-8B, small batch
    (5, 14, 2, 64, 8, [1, 3, 0, 700, 9]),                              # Qwen2-0.5B: empty splits and an empty sequence
    (3, 40, 8, 128, 3, [513, 2, 1025]),                                # group 5
    (2, 32, 1, 64, 5, [4000, 77]),                                     # two head blocks per kv head (group 32)
]


@pytest.mark.parametrize("kv_dtype", [None, torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}-H{c[1]}.{c[2]}-D{c[3]}-S{c[4]}" for c in CASES])
def test_merged_launch_is_bit_identical(case, kv_dtype):
    B, Hq, Hkv, D, splits, lens = case
    dtype = torch.bfloat16 if (B + Hq) % 2 == 0 else torch.float16
    q, kb, vb, r2t, rpi, seq = _setup(B, Hq, Hkv, D, lens, dtype, kv_dtype, seed=B * 100 + Hq)
    scale = D ** -0.5
    # reference: stage 1 + stage 2, and stage 1 + merge_quant
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV)
    o_ref = torch.full((B, Hq, D), 7.0, dtype=dtype, device=DEV)
    ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, seq, logits, splits, scale, 0.0)
    logits2 = torch.zeros_like(logits)
    ops.decode_attention_paged(q, kb, vb, None, r2t, rpi, seq, logits2, splits, scale, 0.0)
    q_ref, s_ref = ops.decode_merge_quant_fp8(logits2, splits, dtype)

    counters = torch.zeros(B + 3, dtype=torch.int32, device=DEV)
    for rep in range(3):  # the counters come back to zero: the same buffer serves every call
        lg = torch.zeros_like(logits)
        o = torch.full((B, Hq, D), -3.0, dtype=dtype, device=DEV)
        assert ops.decode_attention_paged_merged(q, kb, vb, o, r2t, rpi, seq, lg, splits, counters, scale, 0.0) is True
        assert torch.equal(o, o_ref), f"rep {rep}"
        assert int(counters.abs().sum()) == 0
        lg = torch.zeros_like(logits)
        got = ops.decode_attention_paged_merged(q, kb, vb, None, r2t, rpi, seq, lg, splits, counters, scale, 0.0, fp8_out=True)
        assert got is not False
        assert torch.equal(got[0].view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(got[1], s_ref)
        assert int(counters.abs().sum()) == 0
        # both outputs at once
        o2 = torch.empty_like(o)
        got = ops.decode_attention_paged_merged(q, kb, vb, o2, r2t, rpi, seq, lg, splits, counters, scale, 0.0, fp8_out=True)
        assert torch.equal(o2, o_ref) and torch.equal(got[0].view(torch.uint8), q_ref.view(torch.uint8))


def test_merged_launch_under_a_graph():
    B, Hq, Hkv, D, splits = 64, 8, 1, 128, 4
    q, kb, vb, r2t, rpi, seq = _setup(B, Hq, Hkv, D, [1024] * B, torch.bfloat16, None, seed=5)
    logits = torch.zeros(B, Hq, splits, D + 1, device=DEV)
    counters = torch.zeros(B, dtype=torch.int32, device=DEV)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    o_ref = torch.zeros_like(o)
    ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, seq, torch.zeros_like(logits), splits, D ** -0.5, 0.0)
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.decode_attention_paged_merged(q, kb, vb, o, r2t, rpi, seq, logits, splits, counters, D ** -0.5, 0.0)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        for _ in range(4):  # four launches back to back on the same counters, as the layers of a step
            ops.decode_attention_paged_merged(q, kb, vb, o, r2t, rpi, seq, logits, splits, counters, D ** -0.5, 0.0)
    for rep in range(4):  # new queries per replay: a partial left in some cache by an earlier launch would show
        if rep:
            q.copy_(torch.randn(B, Hq, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(100 + rep)).to(q.dtype))
            ops.decode_attention_paged(q, kb, vb, o_ref, r2t, rpi, seq, torch.zeros_like(logits), splits, D ** -0.5, 0.0)
        o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o, o_ref) and int(counters.abs().sum()) == 0


def test_merged_launch_declines_other_shapes():
    B, Hq, Hkv, D = 4, 8, 2, 80  # head size 80: the generic kernel has no fused merge
    q, kb, vb, r2t, rpi, seq = _setup(B, Hq, Hkv, D, [100] * B, torch.bfloat16, None, seed=1)
    logits = torch.zeros(B, Hq, 2, D + 1, device=DEV)
    counters = torch.zeros(B, dtype=torch.int32, device=DEV)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    assert ops.decode_attention_paged_merged(q, kb, vb, o, r2t, rpi, seq, logits, 2, counters, D ** -0.5, 0.0) is False
    with pytest.raises(RuntimeError, match="merge_counters"):
        ops.decode_attention_paged_merged(q, kb, vb, o, r2t, rpi, seq, logits, 2, counters[:2], D ** -0.5, 0.0)
