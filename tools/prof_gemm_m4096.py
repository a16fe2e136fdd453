#!/usr/bin/env python3
"""Profiling target for the tiled FP8 GEMM: a few launches at M = N = K = 4096 and at the M = 1024 gate_up shape.
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python3 tools/prof_gemm_m4096.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (M, K, N) in [(4096, 4096, 4096), (1024, 4096, 28672)]:
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(3)]
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    for i in range(6):
        ops.fp8_scaled_mm(a, ws[i % 3].t(), sa, sb, torch.bfloat16)
torch.cuda.synchronize()
