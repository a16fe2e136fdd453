#!/usr/bin/env python3
"""FP8 GEMM at prefill sizes (M = 1024 / 2048 / 4096 / 8192) on PRE-SHUFFLED weights (what the linear method stores), with a
check of every output against the row-major call.  The tiled variant is fixed per process:
  SGL_MI355_TILED_V3=0 (v2 kernels) | 1 (v3 256x256, 8 waves) | 2 (v3 128x256, 4 waves x 2 workgroups) | unset (library's choice)"""
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


Ms = [int(x) for x in os.environ.get("MS", "1024,4096").split(",")]
for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
    nw = max(2, int(600e6 // (K * N)))
    ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
    wsh = [ops.fp8_shuffle_weight(w) for w in ws]
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    for M in Ms:
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 1e-2
        ref = ops.fp8_scaled_mm(a, ws[0].t(), sa, sb, torch.bfloat16).float()
        out = ops.fp8_scaled_mm(a, wsh[0], sa, sb, torch.bfloat16).float()
        err = float((out - ref).abs().max() / ref.abs().max())
        ms = bench(lambda i: ops.fp8_scaled_mm(a, wsh[i % nw], sa, sb, torch.bfloat16))
        print(json.dumps(dict(M=M, K=K, N=N, v3=os.environ.get("SGL_MI355_TILED_V3", "auto"), us=round(ms * 1e3, 1),
                              TFLOPs=round(2.0 * M * N * K / ms / 1e9, 1), rel_err_vs_rowmajor=err)), flush=True)
    del ws, wsh
