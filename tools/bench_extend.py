#!/usr/bin/env python3
"""Micro-benchmark of the extend (prefill) attention kernel: TFLOP/s on causal self-attention."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops

dev = "cuda:0"
for (B, Hq, Hkv, D, L, P) in [(1, 32, 8, 128, 1024, 0), (1, 32, 8, 128, 8192, 0), (4, 32, 8, 128, 2048, 0),
                               (4, 32, 8, 128, 2048, 2048), (8, 8, 1, 128, 2048, 0), (2, 32, 32, 128, 2048, 0)]:
    g = torch.Generator(device=dev).manual_seed(0)
    n_tok = B * (L + P) + 1
    kb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
    vb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, L + P).to(torch.int32)
    q = torch.randn(B * L, Hq, D, device=dev, generator=g).bfloat16()
    ke = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
    ve = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
    o = torch.zeros(B * L, Hq, D, dtype=torch.bfloat16, device=dev)
    rpi = torch.arange(B, device=dev)
    seq = torch.full((B,), L + P, device=dev)
    ext = torch.full((B,), L, device=dev)
    start = torch.arange(B, device=dev) * L
    f = lambda: ops.extend_attention(q, ke, ve, o, kb, vb, r2t, rpi, seq, ext, start, L, D ** -0.5, 0.0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(10): f()
    en.record(); torch.cuda.synchronize()
    ms = st.elapsed_time(en) / 10
    flops = 4.0 * B * Hq * D * (L * P + L * (L + 1) / 2)
    print(json.dumps(dict(B=B, Hq=Hq, Hkv=Hkv, D=D, L=L, prefix=P, ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1))), flush=True)
