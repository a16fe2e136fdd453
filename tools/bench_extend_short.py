#!/usr/bin/env python3
"""Extend attention of ONE short request (the TTFT case: bs=1, no prefix), Llama-3-8B heads; kernel time by events."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
B, Hq, Hkv, D = 1, 32, 8, 128
for L in [int(x) for x in os.environ.get("L_LIST", "256,512,1024,2048,4096").split(",")]:
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(L, Hq, D, device=dev, generator=g).bfloat16()
    ke = torch.randn(L, Hkv, D, device=dev, generator=g).bfloat16()
    ve = torch.randn(L, Hkv, D, device=dev, generator=g).bfloat16()
    o = torch.zeros_like(q)
    kb = torch.zeros(8, Hkv, D, device=dev, dtype=torch.bfloat16)
    qo = torch.tensor([0, L], dtype=torch.int32, device=dev)
    kvi = torch.zeros(2, dtype=torch.int32, device=dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    f = lambda: ops.extend_attention_fwd(q, ke, ve, o, kb, kb, qo, kvi, idx, None, True, None, L, D ** -0.5)
    for _ in range(3): f()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20): f()
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / 20 * 1e3
    flops = 4.0 * Hq * D * L * (L + 1) / 2
    print(json.dumps(dict(L=L, us=round(us, 1), TFLOPs=round(flops / us / 1e6, 1))), flush=True)
