#!/usr/bin/env python3
"""Extend (prefill) attention over a set of cases, graph-timed: us and TFLOP/s per case.  A/B two builds with SGL_MI355_LIB
(python -m sglang_npu_amd.build_ext --variant NAME ...)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops

dev = "cuda:0"
CASES = [(1, 32, 8, 256, 0), (1, 32, 8, 512, 0), (1, 32, 8, 1024, 0), (1, 32, 8, 2048, 0), (1, 32, 8, 4096, 0),
         (1, 32, 8, 8192, 0), (4, 32, 8, 512, 2048), (4, 32, 8, 2048, 0), (8, 8, 1, 1024, 0), (1, 32, 32, 1024, 0),
         (16, 32, 8, 128, 1024)]
if os.environ.get("CASES"):  # "B,Hq,Hkv,L,P;..." replaces the list
    CASES = [tuple(int(x) for x in c.split(",")) for c in os.environ["CASES"].split(";")]
VARIANTS = [("kernel", None), ("parts", "parts")]  # parts: with the host's prefix bound + scratch (KV-range parts where they apply)
scratch = ops.ExtendPartsScratch(dev)
for (B, Hq, Hkv, L, P) in CASES:
    D = 128
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
    row = dict(B=B, Hq=Hq, Hkv=Hkv, L=L, prefix=P)
    flops = 4.0 * B * Hq * D * (L * P + L * (L + 1) / 2)
    for name, var in VARIANTS:
        kw = dict(max_prefix_len=P, parts_scratch=scratch) if var == "parts" else {}
        f = lambda: ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, idx, None, True, None, L, D ** -0.5, 0.0, **kw)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(8): f()
        gr.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
            ts.append(st.elapsed_time(en) * 1e3 / 8)
        ts.sort()
        row[name + "_us"] = round(ts[2], 2)
        row[name + "_TF"] = round(flops / ts[2] / 1e6, 1)
        del gr
    print(json.dumps(row), flush=True)
