#!/usr/bin/env python3
"""LM head at decode sizes: the 16-bit weight streamer (ops.linear16) vs torch.matmul (hipBLASLt), Llama-3-8B vocabulary.
Two weight copies rotate so that no call finds its weight in the 256 MiB Infinity Cache."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
N, K = 128256, 4096
ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(2)]
wsh = [ops.linear16_shuffle_weight(w) for w in ws]  # fragment-major copies (round 3)
for M in (1, 16, 64, 128):
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()

    def t(fn, n=20):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    us_k = t(lambda i: ops.linear16(x, ws[i & 1]))
    us_l = t(lambda i: torch.matmul(x, ws[i & 1].t()))
    us_s = t(lambda i: ops.linear16(x, wsh[i & 1]))
    assert torch.equal(ops.linear16(x, wsh[0]), ops.linear16(x, ws[0]))
    nbytes = N * K * 2 + M * K * 2 + M * N * 2
    print(json.dumps(dict(M=M, linear16_shuffled_us=round(us_s, 1), linear16_shuffled_GBps=round(nbytes / us_s / 1e3, 1),
                          pb=os.environ.get("SGL_MI355_GEMM16_PB", "2"), linear16_us=round(us_k, 1), linear16_GBps=round(nbytes / us_k / 1e3, 1),
                          hipblaslt_us=round(us_l, 1), hipblaslt_GBps=round(nbytes / us_l / 1e3, 1),
                          nt=os.environ.get("SGL_MI355_GEMM16_NT", "0"))), flush=True)
