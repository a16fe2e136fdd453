#!/usr/bin/env python3
"""Profiling target: a few launches of one decode-shaped fp8_scaled_mm (env: PROF_K, PROF_N, PROF_M)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
K, N, M = int(os.environ.get("PROF_K", 4096)), int(os.environ.get("PROF_N", 28672)), int(os.environ.get("PROF_M", 64))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(4)]
sb = torch.rand(N, device=dev, generator=g) * 1e-2
a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
sa = torch.rand(M, device=dev, generator=g) * 1e-2
for i in range(8):
    ops.fp8_scaled_mm(a, ws[i % 4].t(), sa, sb, torch.bfloat16)
torch.cuda.synchronize()
