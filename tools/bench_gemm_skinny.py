#!/usr/bin/env python3
"""Kernel-time sweep of the skinny FP8 GEMM configs (run under SGL_MI355_SKINNY=NB,WK)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
    nw = max(2, int(600e6 // (K * N)))
    ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    for M in (16, 64):
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 1e-2
        for i in range(3): ops.fp8_scaled_mm(a, ws[i % nw].t(), sa, sb, torch.bfloat16)
        torch.cuda.synchronize()
        evs = []
        for i in range(20):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record(); ops.fp8_scaled_mm(a, ws[i % nw].t(), sa, sb, torch.bfloat16); en.record()
            evs.append((st, en))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        us = ts[len(ts) // 2]
        print(json.dumps(dict(cfg=os.environ.get("SGL_MI355_SKINNY", "auto"), K=K, N=N, M=M, us=round(us, 1),
                              GBps=round((K * N + M * K + 2 * M * N) / us / 1e3, 0))), flush=True)
