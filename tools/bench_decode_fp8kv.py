#!/usr/bin/env python3
"""Paged decode attention over an FP8 (e4m3) KV pool vs the bf16 pool, bs=64, random page table
(IDENTITY=1: consecutive slots per request)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, Hq, Hkv, D = 64, 32, 8, 128
for S in [int(x) for x in os.environ.get("S_LIST", "2048,8192").split(",")]:
    n_tok = B * S + 1
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
    if os.environ.get("IDENTITY") == "1":
        r2t = (torch.arange(n_tok - 1, device=dev) + 1).view(B, S).to(torch.int32).contiguous()
    rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
    for kvdt in (torch.bfloat16, torch.float8_e4m3fn):
        esz = 2 if kvdt == torch.bfloat16 else 1
        NL = max(2, int(1.2e9 // (n_tok * Hkv * D * 2 * esz)))
        kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(kvdt) for _ in range(NL)]
        vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(kvdt) for _ in range(NL)]
        run = lambda i: ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, None, 1, D ** -0.5, 0.0)
        for i in range(3): run(i)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(20): run(i)
        en.record(); torch.cuda.synchronize()
        ms = st.elapsed_time(en) / 20
        nbytes = B * S * Hkv * 2 * D * esz + 4 * B * S + 2 * B * Hq * 2 * D
        print(json.dumps(dict(S=S, kv=str(kvdt).split(".")[-1], us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6),
                              frac_hbm=round(nbytes / ms / 1e6 / 8000, 3))), flush=True)
        del kbs, vbs
