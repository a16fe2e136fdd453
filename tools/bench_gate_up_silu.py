#!/usr/bin/env python3
"""Prefill MLP front half on a pre-shuffled FP8 gate_up weight (Llama-3-8B: 4096 -> 2 x 14336), HIP-graph timed:
  two launches : fp8_scaled_mm [M, 2I] -> silu * mul + per-token quant
  fused        : fp8_scaled_mm_silu_mul [M, I] (activation in the GEMM epilogue) -> per-token quant
Both give the same bits (tests/test_fp8_wshuffled_gpu.py)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
K, I = 4096, 14336
NW = 4
ws = [ops.fp8_shuffle_weight(((torch.rand(2 * I, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
      for _ in range(NW)]
sb = torch.rand(2 * I, device=dev, generator=g) * 1e-2


def graph_us(fn, reps=5):
    fn(0)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(NW * 4):
            fn(i)
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) * 1e3 / (NW * 4))
    return sorted(ts)[len(ts) // 2]


for M in [int(x) for x in os.environ.get("MS", "1024,2048,4096").split(",")]:
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=dev, generator=g) * 1e-2

    def two(i):
        return ops.silu_and_mul_quant_fp8(ops.fp8_scaled_mm(a, ws[i % NW], sa, sb, torch.bfloat16))

    def gemm_only(i):
        return ops.fp8_scaled_mm(a, ws[i % NW], sa, sb, torch.bfloat16)

    def fused_gemm_only(i):
        return ops.fp8_scaled_mm_silu_mul(a, ws[i % NW], sa, sb, torch.bfloat16)

    def fused(i):
        act = ops.fp8_scaled_mm_silu_mul(a, ws[i % NW], sa, sb, torch.bfloat16)
        q = torch.empty_like(act, dtype=torch.float8_e4m3fn)
        s = torch.empty(M, 1, dtype=torch.float32, device=dev)
        ops.sgl_per_token_quant_fp8(act, q, s)
        return q, s

    print(json.dumps(dict(M=M, gemm_us=round(graph_us(gemm_only), 1), gemm_silu_quant_us=round(graph_us(two), 1),
                          fused_gemm_us=round(graph_us(fused_gemm_only), 1), fused_gemm_quant_us=round(graph_us(fused), 1))), flush=True)
