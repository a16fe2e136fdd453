#!/usr/bin/env python3
"""A split-K decode GEMM TOGETHER with the kernel that consumes its partial sums, graph-timed over 32 layers' weights:
down_proj (14336 -> 4096) + add-RMSNorm-quant from partials, o_proj (4096 -> 4096) + the same, qkv (4096 -> 6144) + RoPE/KV
write from partials.  For tuning the slice count (SGL_MI355_WSTREAM_SLAB_FORCE="PH,nc,SK"): fewer slices = less slab
traffic for the consumer, fewer workgroups streaming weights for the producer."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M, L = 64, 32
which = os.environ.get("SHAPES", "down,o").split(",")
shapes = {"down": (14336, 4096), "o": (4096, 4096)}
for name in which:
    K, N = shapes[name]
    ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
          for _ in range(L)]
    sb = torch.rand(N, 1, device=dev, generator=g) * 1e-2
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, device=dev, generator=g) * 1e-2
    res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16)

    def layer(i):
        part = ops.fp8_scaled_mm_partials(a, ws[i], sa, sb, torch.bfloat16)
        return ops.rmsnorm_quant_fp8_from_partials(part, res, nw, 1e-5), part.num_slices

    for i in range(3):
        _, sk = layer(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(L):
            layer(i)
    gr.replay(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) * 1e3 / L)
    ts.sort()
    print(json.dumps(dict(shape=name, K=K, N=N, M=M, slices=sk, us_gemm_plus_norm=round(ts[2], 2),
                          force=os.environ.get("SGL_MI355_WSTREAM_SLAB_FORCE", ""))), flush=True)
