#!/usr/bin/env python3
"""Graph-timed FP8 decode GEMMs (M = 64) at the per-rank shapes of Llama-3-8B (MODEL=llama3-70b: the 70B) under TP = 1, 2, 4, 8 -- what the
driver's scaling run launches.  Weights rotate through > 256 MB so the Infinity Cache does not flatter the numbers."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
ops.reserve_gemm_workspace(dev, 64, 4 * 28672)  # the split-K scratch may not grow under graph capture
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("M", "64"))
H, I, Hq, Hkv, D = 4096, 14336, 32, 8, 128
if os.environ.get("MODEL") == "llama3-70b":
    H, I, Hq, Hkv, D = 8192, 28672, 64, 8, 128
for tp in [int(x) for x in os.environ.get("TP_LIST", "1,2,4,8").split(",")]:
    shapes = {"qkv": (H, (Hq + 2 * max(Hkv, tp)) * D // tp), "o": (Hq * D // tp, H), "gate_up": (H, 2 * I // tp), "down": (I // tp, H)}
    for name, (K, N) in shapes.items():
        nw = max(2, int(600e6 // (K * N)))
        nw = min(nw, 64)
        ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
        if os.environ.get("WSHUF") and ops.fp8_shuffle_supported(N, K):  # the pre-shuffled layout (what the linear method stores)
            ws = [ops.fp8_shuffle_weight(w) for w in ws]
            tag = lambda w: w
        else:
            tag = lambda w: w.t()
        wt = [tag(w) for w in ws]
        sb = torch.rand(N, 1, device=dev, generator=g) * 1e-2
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, 1, device=dev, generator=g) * 1e-2
        if os.environ.get("PARTIALS"):  # the split-K form the model's deferred-epilogue path calls, completed by finalize
            def run(i):
                part = ops.fp8_scaled_mm_partials(a, wt[i % nw], sa, sb, torch.bfloat16)
                return part.finalize() if part is not None else ops.fp8_scaled_mm(a, wt[i % nw], sa, sb, torch.bfloat16)
        else:
            run = lambda i: ops.fp8_scaled_mm(a, wt[i % nw], sa, sb, torch.bfloat16)
        cs = torch.cuda.Stream(device=dev)  # warm up ON the capture stream: the split-K scratch is per stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            for i in range(3): run(i)
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        reps = 2 * nw
        with torch.cuda.graph(gr, stream=cs):
            for i in range(reps): out = run(i)
        gr.replay(); torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(3): gr.replay()
        en.record(); torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1e3 / (3 * reps)
        nbytes = K * N + M * K + 2 * M * N
        print(json.dumps(dict(tp=tp, op=name, K=K, N=N, M=M, us=round(us, 2), GBps=round(nbytes / us / 1e3),
                              floor_us=round(nbytes / 6.0e6, 2))), flush=True)
        del ws, wt
