#!/usr/bin/env python3
"""Cost of one dependent kernel boundary inside a captured HIP graph (the decode step is ~330 such launches):
chains of N identical small kernels, per-kernel time = graph replay time / N."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops

dev = "cuda:0"
x = torch.randn(64, 4096, device=dev).bfloat16()
q = torch.empty(64, 4096, dtype=torch.float8_e4m3fn, device=dev)
s = torch.empty(64, 1, device=dev)
w = torch.ones(4096, device=dev).bfloat16()
tiny = torch.zeros(64, device=dev)

def chain(fn, n):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(ts)[2]

N = 400
print("per_token_quant 64x4096 (64 WG x 512):", round(chain(lambda: ops.sgl_per_token_quant_fp8(x, q, s), N), 2), "us/kernel")
print("rmsnorm 64x4096:", round(chain(lambda: ops.rmsnorm(x, w, 1e-5, out=x), N), 2), "us/kernel")
print("torch add_ on 64 floats:", round(chain(lambda: tiny.add_(1.0), N), 2), "us/kernel")
big = torch.zeros(1 << 20, device=dev)
print("torch add_ on 1M floats:", round(chain(lambda: big.add_(1.0), N), 2), "us/kernel")
