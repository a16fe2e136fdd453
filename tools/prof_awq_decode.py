#!/usr/bin/env python3
"""Profiling target for the AWQ decode GEMM (awq_gemm_packed): Llama-2-7B gate_up 4096 -> 22016 and down 11008 -> 4096 at
M = 64 and 1, packed weights rotating through > 256 MB.
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
            --kernel-trace -- python3 tools/prof_awq_decode.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
G = 128
imax = torch.iinfo(torch.int32).max
for (K, N) in [(4096, 22016), (11008, 4096)]:
    packs = []
    for _ in range(8):
        qw = torch.randint(0, imax, (K, N // 8), dtype=torch.int32, device=dev, generator=g)
        qz = torch.randint(0, imax, (K // G, N // 8), dtype=torch.int32, device=dev, generator=g)
        sc = ((torch.rand(K // G, N, device=dev, generator=g) - 0.3) * 2e-2).half()
        packs.append(ops.awq_repack(qw, sc, qz))
    for M in (64, 1):
        x = torch.randn(M, K, device=dev, generator=g).half()
        for i in range(16):
            ops.awq_gemm_packed(x, packs[i % 8][0], packs[i % 8][1], G)
torch.cuda.synchronize()
