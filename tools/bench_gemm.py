#!/usr/bin/env python3
"""Micro-benchmark of fp8_scaled_mm (+ per-token quant) on BASELINE config 3/5 shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops  # noqa: E402

dev = "cuda:0"


def bench(fn, iters=30):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(iters):
        fn(i)
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


if __name__ == "__main__":
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096), (8192, 1280), (1024, 8192), (8192, 7168),
              (3584, 8192)]
    for (K, N) in shapes:
        nw = max(2, int(600e6 // (K * N)))  # rotate weights so they do not sit in the Infinity Cache
        ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
        sb = torch.rand(N, device=dev, generator=g) * 1e-2
        for M in (1, 16, 64, 512, 4096):
            a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
            sa = torch.rand(M, device=dev, generator=g) * 1e-2
            ms = bench(lambda i: ops.fp8_scaled_mm(a, ws[i % nw].t(), sa, sb, torch.bfloat16))
            flops = 2.0 * M * N * K
            nbytes = M * K + K * N + 2 * M * N + 4 * (M + N)
            print(json.dumps(dict(K=K, N=N, M=M, ms=round(ms, 4), TFLOPs=round(flops / ms / 1e9, 1),
                                  GBps=round(nbytes / ms / 1e6, 1))), flush=True)
    for (T, K) in [(64, 4096), (64, 14336), (4096, 4096)]:
        x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
        q = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=dev)
        s = torch.empty(T, device=dev)
        ms = bench(lambda i: ops.sgl_per_token_quant_fp8(x, q, s))
        print(json.dumps(dict(op="per_token_quant", T=T, K=K, ms=round(ms, 4), GBps=round(3 * T * K / ms / 1e6, 1))))
