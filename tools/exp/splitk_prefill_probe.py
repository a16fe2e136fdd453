#!/usr/bin/env python3
"""What a cross-workgroup split-K of the prefill down_proj could buy at best: the GEMM BODY of `S` K-slices on 256 x 256 tiles is
emulated by one GEMM with K / S and N x S (the same number of tiles, the same operand bytes per CU; the fp32 partial traffic and
the reduction are NOT in it).  M = 1024.  SGL_MI355_TILED_V3 fixes the tile form per process (unset: the library's choice)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


M = 1024
for name, K, N in [("down as it is", 14336, 4096), ("down, 2 slices emulated", 7168, 8192), ("down, 4 slices emulated", 3584, 16384),
                   ("o as it is", 4096, 4096), ("o, 2 slices emulated", 2048, 8192), ("qkv as it is", 4096, 6144), ("qkv, 2 slices emulated", 2048, 12288)]:
    nw = max(2, int(600e6 // (K * N)))
    wsh = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)) for _ in range(nw)]
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    ms = bench(lambda i: ops.fp8_scaled_mm(a, wsh[i % nw], sa, sb, torch.bfloat16))
    print(json.dumps(dict(case=name, M=M, K=K, N=N, v3=os.environ.get("SGL_MI355_TILED_V3", "auto"), us=round(ms * 1e3, 1),
                          TFLOPs=round(2.0 * M * N * K / ms / 1e9, 1), family=ops.last_gemm_kernel() if hasattr(ops, "last_gemm_kernel") else None)), flush=True)
    del wsh
