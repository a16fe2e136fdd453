#!/usr/bin/env python3
"""16-bit GEMM above 128 rows: the tiled kernel on the fragment-major weight (ops.linear16) against the vendor library
(F.linear -> hipBLASLt), Llama-2-7B / Llama-3-8B shapes, HIP-graph timed over rotating weights.
SGL_MI355_G16T_TILE=1|2|3 forces a tile form (256x256 / 128x256 / 128x128)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
dt = torch.bfloat16


def graph_us(fn, n, reps=5):
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


SHAPES = [("qkv_7b", 4096, 12288), ("o", 4096, 4096), ("gate_up_7b", 4096, 22016), ("down_7b", 11008, 4096),
          ("gate_up_8b", 4096, 28672), ("down_8b", 14336, 4096)]
Ms = [int(x) for x in os.environ.get("MS", "256,512,1024,2048,4096").split(",")]
for name, K, N in SHAPES:
    nw = 3
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.05).to(dt) for _ in range(nw)]
    fm = [ops.linear16_shuffle_weight(w) for w in ws]
    for M in Ms:
        x = torch.randn(M, K, device=dev, generator=g).to(dt)
        t_tiled = graph_us(lambda i: ops.linear16(x, fm[i % nw]), 6)
        t_lib = graph_us(lambda i: torch.nn.functional.linear(x, ws[i % nw]), 6)
        fl = 2.0 * M * N * K
        print(json.dumps(dict(shape=name, M=M, K=K, N=N, tile=os.environ.get("SGL_MI355_G16T_TILE", "auto"),
                              tiled_us=round(t_tiled, 1), tiled_TF=round(fl / t_tiled / 1e6, 1),
                              library_us=round(t_lib, 1), library_TF=round(fl / t_lib / 1e6, 1))), flush=True)
    del ws, fm
