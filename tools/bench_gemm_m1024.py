#!/usr/bin/env python3
"""FP8 GEMM at the prefill shapes of Llama-3-8B (M = 1024 / 2048).  The tiled-kernel variant is fixed per process by
SGL_MI355_TILED_V2 (0, 2, 3, 8, 22, 48, 84; unset = the library's choice), so sweep it from the shell:
  for v in 2 3 8 22 48 84; do SGL_MI355_TILED_V2=$v python tools/bench_gemm_m1024.py; done"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def bench(fn, iters=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(iters): fn(i)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


for M in (1024, 2048):
    for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
        nw = max(2, int(600e6 // (K * N)))
        ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
        sb = torch.rand(N, device=dev, generator=g) * 1e-2
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 1e-2
        ms = bench(lambda i: ops.fp8_scaled_mm(a, ws[i % nw].t(), sa, sb, torch.bfloat16))
        print(json.dumps(dict(M=M, K=K, N=N, variant=os.environ.get("SGL_MI355_TILED_V2", "auto"), us=round(ms * 1e3, 1),
                              TFLOPs=round(2.0 * M * N * K / ms / 1e9, 1))), flush=True)
        del ws
