#!/usr/bin/env python3
"""down_proj (14336 -> 4096, FP8) + the RMSNorm that consumes it at prefill row counts, graph-replayed on rotating weights:
the plain pair (GEMM with its own epilogue, fused_add_rmsnorm [+ FP8 companion]) against the raw split-K form + norm from
partials (deferred.py).  SGL_MI355_NO_PREFILL_SPLITK=1 switches the form off in the library: run both ways."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import deferred, ops
from sglang_npu_amd.layers import RMSNorm
from sglang_npu_amd.linear import RowParallelLinear
from sglang_npu_amd.quantization import W8A8Fp8Config
DEV = "cuda:0"
K, H, NL, dtype = 14336, 4096, 8, torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
lins = []
for _ in range(NL):
    lin = RowParallelLinear(K, H, params_dtype=dtype, quant_config=W8A8Fp8Config(is_checkpoint_fp8_serialized=False)).to(DEV)
    lin.weight.weight_loader(lin.weight, (torch.rand(H, K, generator=g, device=DEV) * 2e-2 - 1e-2).to(dtype))
    lin.quant_method.process_weights_after_loading(lin)
    lins.append(lin)
norm = RMSNorm(H, 1e-5, dtype).to(DEV)
norm.emit_fp8_companion = True
deferred.hint_decode = False
for T in (256, 300, 384, 512, 768, 1024, 1536, 2048):
    x = torch.randn(T, K, generator=g, device=DEV).to(dtype)
    r = torch.randn(T, H, generator=g, device=DEV).to(dtype)
    side = torch.cuda.Stream()  # (the split-K workspace is per stream: warm up on the stream the graph is captured on)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for lin in lins:
                norm(lin(x)[0], r)
        kinds = type(lins[0](x)[0]).__name__
        norm(lins[0](x)[0], r)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for lin in lins:
                norm(lin(x)[0], r)
    torch.cuda.current_stream().wait_stream(side)
    ts = []
    for _ in range(7):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); graph.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / NL * 1e3)
    print(f"T={T:5d}  quant + down_proj + norm: {sorted(ts)[3]:7.1f} us   ({kinds})", flush=True)
