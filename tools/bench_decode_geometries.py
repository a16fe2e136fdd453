#!/usr/bin/env python3
"""Paged decode attention across the model geometries of SURVEY 8 (bs=64, ctx=2048, random page table):
Llama-3-8B (32/8, D=128), Llama-2-7B (MHA 32/32), Llama-3-70B TP8 rank (8/1), Qwen2-0.5B (14/2, D=64)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
from sglang_npu_amd.attention_backend import MI355AttnBackend
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, S = 64, 2048
for name, Hq, Hkv, D in [("llama3-8b", 32, 8, 128), ("llama2-7b", 32, 32, 128), ("llama3-70b/tp8", 8, 1, 128), ("qwen2-0.5b", 14, 2, 64)]:
    n_tok = B * S + 1
    NL = max(2, int(1.2e9 // (n_tok * Hkv * D * 4)))
    kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
    rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
    group = Hq // Hkv
    wgs = B * Hkv * ((group + 15) // 16)
    splits = 1 if wgs >= 256 else max(1, min(8, -(-256 // wgs), S // 256))  # MI355AttnBackend.choose_num_kv_splits
    logits = torch.zeros(B, Hq, splits, D + 1, device=dev) if splits > 1 else None
    def run(i):
        ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, logits, splits, D ** -0.5, 0.0)
    for i in range(3): run(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(20): run(i)
    en.record(); torch.cuda.synchronize()
    ms = st.elapsed_time(en) / 20
    nbytes = B * S * Hkv * 2 * D * 2 + 4 * B * S + 2 * B * Hq * 2 * D
    print(json.dumps(dict(model=name, Hq=Hq, Hkv=Hkv, D=D, splits=splits, us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6),
                          frac_hbm=round(nbytes / ms / 1e6 / 8000, 3))), flush=True)
    del kbs, vbs
