#!/usr/bin/env python3
"""kv-split decode at the per-rank geometries of TP=8 (bs=64, ctx=2048): stage 1 + merge_quant launch vs the in-launch
merge (sgl_mi355_decode_attention_merged), each as a HIP graph of 32 calls over rotating pools."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, S = 64, 2048
CASES = [("70b/tp8", 8, 1, 128, 4), ("8b/tp8", 4, 1, 128, 4), ("8b/tp4", 8, 2, 128, 2), ("8b bs=16", 32, 8, 128, 2)]
if os.environ.get("SPLITS"):  # A/B aid: the 70B TP=8 rank geometry at other split counts (with SGL_MI355_DECODE_WAVES=2|4)
    CASES = [("70b/tp8", 8, 1, 128, int(x)) for x in os.environ["SPLITS"].split(",")]
for name, Hq, Hkv, D, splits in CASES:
    Bn = 16 if "bs=16" in name else B
    n_tok = Bn * S + 1
    NL = 8
    kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    q = torch.randn(Bn, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(Bn, S).to(torch.int32).contiguous()
    rpi, seq = torch.arange(Bn, device=dev), torch.full((Bn,), S, device=dev)
    logits = torch.zeros(Bn, Hq, splits, D + 1, device=dev)
    counters = torch.zeros(Bn, dtype=torch.int32, device=dev)
    def two(i):
        ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], None, r2t, rpi, seq, logits, splits, D ** -0.5, 0.0)
        return ops.decode_merge_quant_fp8(logits, splits, torch.bfloat16)
    def one(i):
        return ops.decode_attention_paged_merged(q, kbs[i % NL], vbs[i % NL], None, r2t, rpi, seq, logits, splits, counters,
                                                 D ** -0.5, 0.0, fp8_out=True)
    res = {}
    for label, fn in (("two_launches", two), ("merged", one)):
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(0)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(32): fn(i)
        graph.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record(); graph.replay(); en.record(); torch.cuda.synchronize()
            ts.append(st.elapsed_time(en) * 1e3 / 32)
        ts.sort(); res[label] = round(ts[len(ts) // 2], 2)
    print(json.dumps(dict(geometry=name, Hq=Hq, Hkv=Hkv, splits=splits, **res)), flush=True)
