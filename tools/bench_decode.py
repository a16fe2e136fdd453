#!/usr/bin/env python3
"""Micro-benchmark of the paged decode kernel (GPU box).  Prints algorithmic GB/s per config.

Algorithmic bytes per call:
  B*S*Hkv*(D+Dv)*2  (K+V, 16-bit)  + 4*B*S (int32 page table) + 2*B*(Hq)*(D+Dv) (q, o)
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops  # noqa: E402


def run(B, Hq, Hkv, D, S, splits, layout, iters=20, dtype=torch.bfloat16, nlayers=4):
    dev = "cuda:0"
    n_tok = B * S + 1
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(dtype)
    # several "layers" of pool so consecutive calls do not hit in the 256 MiB Infinity Cache
    kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(dtype) for _ in range(nlayers)]
    vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(dtype) for _ in range(nlayers)]
    if layout == "random":
        perm = torch.randperm(n_tok - 1, device=dev, generator=g) + 1
    elif layout.startswith("page"):
        # what a paged allocator with page_size P hands out (allocator.py:428-449): runs of P consecutive slots, the pages
        # themselves scattered; the page table stays token-level (req_to_token has one entry per token whatever the page size)
        P = int(layout[4:])
        npages = (n_tok - 1) // P
        order = torch.randperm(npages, device=dev, generator=g)
        perm = (order[:, None] * P + torch.arange(P, device=dev)[None, :]).reshape(-1)[: B * S] + 1
    else:
        perm = torch.arange(1, n_tok, device=dev)
    r2t = perm.view(B, S).to(torch.int32).contiguous()
    rpi = torch.arange(B, device=dev)
    seq = torch.full((B,), S, device=dev)
    o = torch.zeros(B, Hq, D, dtype=dtype, device=dev)
    logits = torch.zeros(B, Hq, splits, D + 1, device=dev)
    scale = 1.0 / D ** 0.5

    def call(i):
        ops.decode_attention(q, kbs[i % nlayers], vbs[i % nlayers], o, None, None, None, logits, r2t, rpi, seq,
                             scale, 0.0)

    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(iters):
        call(i)
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en) / iters
    nbytes = B * S * Hkv * 2 * D * 2 + 4 * B * S + 2 * B * Hq * 2 * D
    return ms, nbytes / ms / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--headline", action="store_true", help="bs=64 32/8/128, S = 512 / 2048, random page table, one split")
    ap.add_argument("--pages", action="store_true", help="bs=64 32/8/128 S=2048: token-level random vs page_size 16 / 64 / 128 vs identity")
    a = ap.parse_args()
    if a.pages:
        for layout in ("random", "page16", "page64", "page128", "identity", "random"):
            ms, gbs = run(64, 32, 8, 128, 2048, 1, layout, iters=64, nlayers=8)
            print(json.dumps(dict(layout=layout, us=round(ms * 1e3, 2), GBps=round(gbs, 1), frac=round(gbs / 8000, 4))), flush=True)
        sys.exit(0)
    if a.headline:
        for S in (512, 2048, 4096):
            ms, gbs = run(64, 32, 8, 128, S, 1, "random", iters=64, nlayers=8)
            print(json.dumps(dict(S=S, us=round(ms * 1e3, 2), GBps=round(gbs, 1), frac=round(gbs / 8000, 4))), flush=True)
        sys.exit(0)
    rows = []
    cfgs = [
        # B, Hq, Hkv, D, S
        (64, 32, 8, 128, 2048),
        (64, 32, 8, 128, 512),
        (64, 32, 8, 128, 8192),
        (64, 8, 1, 128, 2048),
        (64, 32, 32, 128, 2048),
        (1, 32, 8, 128, 32768),
        (64, 14, 2, 64, 2048),
    ]
    if a.quick:
        cfgs = cfgs[:1]
    for (B, Hq, Hkv, D, S) in cfgs:
        for layout in ("random", "identity"):
            for splits in (1, 2, 4, 8, 16):
                if B * Hkv * splits < 64 and splits < 16:
                    continue
                ms, gbs = run(B, Hq, Hkv, D, S, splits, layout)
                rows.append(dict(B=B, Hq=Hq, Hkv=Hkv, D=D, S=S, layout=layout, splits=splits, ms=round(ms, 4),
                                 GBps=round(gbs, 1)))
                print(json.dumps(rows[-1]), flush=True)
