"""Isolated timing: RoPE/KV-write-from-partials + paged decode (two launches) against the fused form
(sgl_mi355_decode_attention_qkv_partials) at the headline decode shape.  HIP events around batches of launches."""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops

p = argparse.ArgumentParser()
p.add_argument("--bs", type=int, default=64)
p.add_argument("--ctx", type=int, default=2048)
p.add_argument("--iters", type=int, default=40)
args = p.parse_args()
DEV = "cuda"
B, Hq, Hk, D, K = args.bs, 32, 8, 128, 4096
N = (Hq + 2 * Hk) * D
g = torch.Generator(device=DEV).manual_seed(0)
a = ((torch.rand(B, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
w = ((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
sa = torch.rand(B, 1, device=DEV, generator=g) * 1e-3 + 1e-4
sb = torch.rand(N, 1, device=DEV, generator=g) * 1e-3 + 1e-4
S = args.ctx
rows = B * S + 1
perm = (torch.randperm(rows - 1, device=DEV, generator=g) + 1).int()
r2t = perm.view(B, S).contiguous()
rpi = torch.arange(B, device=DEV)
lens = torch.full((B,), S, dtype=torch.int64, device=DEV)
loc = r2t[:, S - 1].long().contiguous()
kb = torch.randn(rows, Hk, D, device=DEV, generator=g).bfloat16()
vb = torch.randn(rows, Hk, D, device=DEV, generator=g).bfloat16()
pos = lens - 1
cache = torch.randn(S + 1, D, device=DEV, generator=g)
o = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
scale = D ** -0.5
part = ops.fp8_scaled_mm_partials(a, w.t(), sa, sb, torch.bfloat16, None)
print("slices", part.num_slices)


def two():
    q = ops.rope_set_kv_from_partials(part, pos, Hq, Hk, D, cache, kb, vb, loc, True)
    ops.decode_attention_paged(q.view(B, Hq, D), kb, vb, o, r2t, rpi, lens, None, 1, scale, 0.0)


def attn_only(q=torch.randn(B, Hq, D, device=DEV).bfloat16()):
    ops.decode_attention_paged(q, kb, vb, o, r2t, rpi, lens, None, 1, scale, 0.0)


def fused():
    assert ops.decode_attention_qkv_partials(part, pos, cache, True, loc, kb, vb, o, r2t, rpi, lens, Hq, scale, 0.0)


def timeit(fn, name):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(args.iters):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        gr.replay()
        en.record()
        torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) * 1e3 / args.iters)
    print(f"{name:28s} {best:8.2f} us per call")


timeit(attn_only, "attention only")
timeit(two, "rope/kv + attention")
timeit(fused, "fused")
