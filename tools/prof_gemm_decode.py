#!/usr/bin/env python3
"""Profiling target for the decode-time FP8 GEMMs (M = 64 and 16): gate_up 4096 -> 28672 and down 14336 -> 4096, weights
rotating through > 256 MB.
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -- python3 tools/prof_gemm_decode.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (K, N) in [(4096, 28672), (14336, 4096)]:
    ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(5)]
    if not os.environ.get("PLAIN"):  # the layout the linear method stores (fragment-major)
        ws = [ops.fp8_shuffle_weight(w) for w in ws]
    sb = torch.rand(N, 1, device=dev, generator=g) * 1e-2
    for M in (64, 16):
        a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sa = torch.rand(M, 1, device=dev, generator=g) * 1e-2
        for i in range(10):
            ops.fp8_scaled_mm(a, ws[i % 5].t() if os.environ.get('PLAIN') else ws[i % 5], sa, sb, torch.bfloat16)
torch.cuda.synchronize()
