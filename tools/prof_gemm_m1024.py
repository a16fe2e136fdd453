#!/usr/bin/env python3
"""Profiling target: the M = 1024 prefill FP8 GEMMs that have exactly one 128 x 128 tile per CU (o_proj 4096 -> 4096, down_proj
14336 -> 4096) on pre-shuffled weights, 8 launches each over rotating weights.  Counter passes: tools/exp/prof_gemm_m1024_pmc.sh"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = int(os.environ.get("M", 1024))
for K, N in ((14336, 4096), (4096, 4096)):
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
          for _ in range(4)]
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    for i in range(8):
        ops.fp8_scaled_mm(a, ws[i % 4], sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
