#!/usr/bin/env python3
"""Profiling target: N launches of the paged decode kernel at BASELINE config 2
(Llama-3-8B geometry, bs=64, S=2048, random page table).  Run under rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops  # noqa: E402

B, Hq, Hkv, D = 64, 32, 8, 128
S = int(os.environ.get("PROF_S", "2048"))
SPLITS = int(os.environ.get("PROF_SPLITS", "1"))
ITERS = int(os.environ.get("PROF_ITERS", "20"))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
n_tok = B * S + 1
q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
NL = 4
kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
rpi = torch.arange(B, device=dev)
seq = torch.full((B,), S, device=dev)
o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
logits = torch.zeros(B, Hq, SPLITS, D + 1, device=dev)
for i in range(ITERS):
    ops.decode_attention(q, kbs[i % NL], vbs[i % NL], o, None, None, None, logits, r2t, rpi, seq, 1.0 / D ** 0.5, 0.0)
torch.cuda.synchronize()
nbytes = B * S * Hkv * 2 * D * 2 + 4 * B * S + 2 * B * Hq * 2 * D
print("algorithmic bytes per launch:", nbytes)
