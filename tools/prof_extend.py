#!/usr/bin/env python3
"""Launches of the extend attention kernel for a profiler: CASE = "B,L,P" (default 1,4096,0), N launches (default 10)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
B, L, P = [int(x) for x in os.environ.get("CASE", "1,4096,0").split(",")]
Hq, Hkv, D = 32, 8, 128
g = torch.Generator(device=dev).manual_seed(0)
n_tok = B * (L + P) + 1
kb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
vb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
perm = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).to(torch.int32)
q = torch.randn(B * L, Hq, D, device=dev, generator=g).bfloat16()
ke = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
ve = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
o = torch.zeros(B * L, Hq, D, dtype=torch.bfloat16, device=dev)
qo = (torch.arange(B + 1, device=dev) * L).to(torch.int32)
kvp = (torch.arange(B + 1, device=dev) * P).to(torch.int32)
idx = perm[: B * P].contiguous() if P else torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("N", "10"))):
    ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, idx, None, True, None, L, D ** -0.5, 0.0)
torch.cuda.synchronize()
