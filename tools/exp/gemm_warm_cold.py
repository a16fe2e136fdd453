#!/usr/bin/env python3
"""Decode-size FP8 GEMMs (M = 64, Llama-3-8B shapes, pre-shuffled weights) against where their weights are when the launch
starts: the same weight every call (L2 / Infinity Cache warm), a rotation of NW weights that fits the Infinity Cache but
not the L2s, and a rotation that fits neither.  Bounds what a weight prefetch beside the preceding elementwise launch could
buy.  HIP-graph timed, 32 calls per replay."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("M", "64"))


def graph_us(fn, n=32, reps=7):
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


for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
    mb = K * N / 1e6
    n_cold = max(2, int(700 // mb) + 1)           # > 256 MB Infinity Cache in rotation
    n_mall = max(2, min(n_cold, int(200 // mb)))  # > the 32 MB of L2, inside the Infinity Cache where the shape allows
    ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
          for _ in range(n_cold)]
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    row = dict(shape=name, M=M, weight_MB=round(mb, 1))
    for label, nw in (("same_weight", 1), (f"rotate_{n_mall}_in_infinity_cache", n_mall), (f"rotate_{n_cold}_cold", n_cold)):
        row[label + "_us"] = round(graph_us(lambda i: ops.fp8_scaled_mm(a, ws[i % nw], sa, sb, torch.bfloat16)), 2)
    row["hbm_time_at_6TBps_us"] = round(mb / 6.0, 2)
    print(json.dumps(row), flush=True)
    del ws
