#!/usr/bin/env python3
"""FP8 prefill GEMM time against K at fixed M x N (pre-shuffled weights, HIP-graph timed over rotating weights): the slope is
the cost of a k-step, the intercept what a launch pays before and after its K loop."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def graph_us(fn, n=8, reps=5):
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for i in range(n): fn(i)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


M = int(os.environ.get("M", "1024"))
for N in [int(x) for x in os.environ.get("NS", "6144,4096").split(",")]:
    row = dict(M=M, N=N, ks=os.environ.get("SGL_MI355_T3_KS", "auto"))
    for K in (1024, 2048, 4096, 8192, 16384):
        nw = 4
        ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)) for _ in range(nw)]
        sb = torch.rand(N, device=dev, generator=g) * 1e-2
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 1e-2
        row[f"K{K}_us"] = round(graph_us(lambda i: ops.fp8_scaled_mm(a, ws[i % nw], sa, sb, torch.bfloat16)), 1)
        row[f"K{K}_kernel"] = ops.fp8_last_kernel()
        del ws
    print(json.dumps(row), flush=True)
