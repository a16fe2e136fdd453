"""What does the decode attention kernel cost when other kernels run between its launches, as in the model step?
Difference of two graph-replayed loops (with / without the attention launch), bs=64, 32/8/128, S=2048, 8 pools."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
B, Hq, Hkv, D, S, NL = 64, 32, 8, 128, 2048, 8
n_tok = B * S + 1
q = torch.randn(B, Hq, D, device=dev, generator=g).bfloat16()
kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16() for _ in range(NL)]
vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16() for _ in range(NL)]
r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
K, N = 4096, 28672
ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn))
      for _ in range(NL)]
a8 = ((torch.rand(B, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
sa, sb = torch.rand(B, 1, device=dev, generator=g) * 1e-2, torch.rand(N, 1, device=dev, generator=g) * 1e-2
x16 = torch.randn(B, 4096, device=dev, generator=g).bfloat16()
wn = torch.ones(4096, device=dev, dtype=torch.bfloat16)


def attn(i):
    ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, None, 1, D ** -0.5, 0.0)


def gemm(i):
    return ops.fp8_scaled_mm(a8, ws[i % NL], sa, sb, torch.bfloat16)


def small(i):
    xq = torch.empty_like(x16, dtype=torch.float8_e4m3fn)
    xs = torch.empty(B, 1, dtype=torch.float32, device=dev)
    ops.sgl_per_token_quant_fp8(x16, xq, xs)


def graph_us(body, n=32, reps=7):
    for i in range(2):
        body(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n):
            body(i)
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


res = {}
res["attention alone"] = graph_us(attn)
for name, other in (("gate_up GEMM (117 MB of weights)", gemm), ("per-token quant (small)", small)):
    base = graph_us(other)
    both = graph_us(lambda i: (other(i), attn(i)))
    res[f"attention between launches of: {name}"] = both - base
    res[f"  ({name} alone)"] = base
both = graph_us(lambda i: (gemm(i), small(i), attn(i), small(i)))
base = graph_us(lambda i: (gemm(i), small(i), small(i)))
res["attention between GEMM + quant ... quant"] = both - base
for k, v in res.items():
    print(f"{k:70s} {v:8.2f} us")
