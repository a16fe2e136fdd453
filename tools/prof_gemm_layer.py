#!/usr/bin/env python3
"""Profiling target: the four FP8 GEMMs of a Llama-3-8B layer (qkv 4096 -> 6144, o 4096 -> 4096, gate_up 4096 -> 28672, down
14336 -> 4096) at M rows (env M, default 1024) on pre-shuffled weights, 4 launches each over rotating weights, in that order.
Counter passes: tools/exp/prof_gemm_layer_pmc.sh."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("M", 1024))
for K, N in ((4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)):
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
          for _ in range(2)]
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    for i in range(4):
        ops.fp8_scaled_mm(a, ws[i % 2], sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    del ws
