#!/usr/bin/env python3
"""Profiling target for the prefill FP8 GEMM on pre-shuffled weights: gate_up (K = 4096, N = 28672) at M = 8192, 12 launches
over rotating weights.  See tools/exp/prof_gemm_prefill_pmc.sh for the counter passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M, K, N = int(os.environ.get("M", 8192)), 4096, 28672
a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
      for _ in range(4)]
sa = torch.rand(M, device=dev, generator=g) * 1e-2
sb = torch.rand(N, device=dev, generator=g) * 1e-2
for i in range(12):
    ops.fp8_scaled_mm(a, ws[i % 4], sa, sb, torch.bfloat16)
torch.cuda.synchronize()
