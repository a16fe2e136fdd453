#!/usr/bin/env python3
"""Unquantised (bf16) decode linears of Llama-3-8B (BASELINE config 2): the 16-bit weight streamer on a fragment-major copy
(ops.linear16 + ops.linear16_shuffle_weight) against torch.matmul (hipBLASLt), graph-timed over enough weight copies to
defeat the 256 MiB Infinity Cache."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
for name, K, N in shapes:
    nl = max(3, int(700e6 // (2 * K * N)))
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16() for _ in range(nl)]
    wsh = [ops.linear16_shuffle_weight(w) for w in ws]
    for M in (1, 16, 64, 128):
        x = torch.randn(M, K, device=dev, generator=g).bfloat16()

        def timed(fn):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(2 * nl):
                    fn(i)
            gr.replay(); torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(3):
                gr.replay()
            en.record(); torch.cuda.synchronize()
            return st.elapsed_time(en) * 1e3 / (6 * nl)

        us_s = timed(lambda i: ops.linear16(x, wsh[i % nl]))
        us_l = timed(lambda i: torch.matmul(x, ws[i % nl].t()))
        nbytes = 2 * N * K + 2 * M * K + 2 * M * N
        print(json.dumps(dict(shape=name, K=K, N=N, M=M, linear16_shuffled_us=round(us_s, 2), hipblaslt_us=round(us_l, 2),
                              linear16_GBps=round(nbytes / us_s / 1e3), hipblaslt_GBps=round(nbytes / us_l / 1e3))), flush=True)
    del ws, wsh
