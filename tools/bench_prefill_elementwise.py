#!/usr/bin/env python3
"""The elementwise launches of a 1024-token FP8 prefill layer (Llama-3-8B shapes), each as a HIP graph of NL calls over NL
different buffer sets (nothing is served from a cache the layer would not have), us per call and GB/s of the bytes each
must move (VERDICT r3 ask 1c: <= 1.3 x bytes / 6.3 TB/s)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
T = int(os.environ.get("T", "1024"))
H, I, Hq, Hk, D, NL = 4096, 14336, 32, 8, 128, 8
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
xs_h = [rnd(T, H) for _ in range(NL)]
res = [rnd(T, H) for _ in range(NL)]
xs_i = [rnd(T, I) for _ in range(NL)]
qkv = [rnd(T, (Hq + 2 * Hk) * D) for _ in range(NL)]
w = rnd(H)
q8_h = [torch.empty(T, H, dtype=torch.float8_e4m3fn, device=dev) for _ in range(NL)]
q8_i = [torch.empty(T, I, dtype=torch.float8_e4m3fn, device=dev) for _ in range(NL)]
sc = [torch.empty(T, 1, device=dev) for _ in range(NL)]
pos = torch.arange(T, device=dev)
cache = torch.randn(8192, D, device=dev, generator=g)
n_tok = T + 8
kbs = [torch.zeros(n_tok, Hk, D, dtype=torch.bfloat16, device=dev) for _ in range(NL)]
vbs = [torch.zeros(n_tok, Hk, D, dtype=torch.bfloat16, device=dev) for _ in range(NL)]
loc = (torch.randperm(n_tok - 1, device=dev, generator=g)[:T] + 1)


def rope(i):
    q, k, v = qkv[i].split([Hq * D, Hk * D, Hk * D], dim=-1)
    ops.apply_rope_and_set_kv_buffer(pos, q, k, v, D, cache, kbs[i], vbs[i], loc, True)


CASES = [
    ("per_token_quant [T,4096] (attention output)", lambda i: ops.sgl_per_token_quant_fp8(xs_h[i], q8_h[i], sc[i]), T * H * 3 + 4 * T),
    ("per_token_quant [T,14336] (SiLU*mul output)", lambda i: ops.sgl_per_token_quant_fp8(xs_i[i], q8_i[i], sc[i]), T * I * 3 + 4 * T),
    ("fused add + RMSNorm + quant [T,4096]", lambda i: ops.rmsnorm_quant_fp8(xs_h[i], w, 1e-5, residual=res[i]), T * H * (2 + 2 + 2 + 1) + 4 * T),
    ("RoPE + KV-pool write [T, 32+8+8 heads]", rope, T * ((Hq + Hk) * D * 4 + Hk * D * 2 + 2 * Hk * D * 2)),
]
for name, fn, nbytes in CASES:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(NL): fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for i in range(NL): fn(i)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) * 1e3 / NL)
    ts.sort()
    us = ts[len(ts) // 2]
    print(json.dumps(dict(op=name, T=T, us=round(us, 2), MB=round(nbytes / 1e6, 1), GBps=round(nbytes / us / 1e3, 0),
                          x_of_6p3TBps=round(us / (nbytes / 6.3e6), 2))), flush=True)
