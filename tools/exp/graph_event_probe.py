#!/usr/bin/env python3
"""Can HIP events recorded INSIDE a captured graph time a kernel of the replay?  (bench.py would then take the decode
attention kernel's duration from the very replays it times, instead of from an eager instrumented pass.)"""
import torch
dev = "cuda:0"
x = torch.randn(1 << 26, device=dev)
y = torch.empty_like(x)
small = torch.zeros(64, device=dev)
for external in (True, False):
    try:
        kw = {"external": True} if external else {}
        evs = [torch.cuda.Event(enable_timing=True, **kw) for _ in range(4)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            y.copy_(x); small.add_(1)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            small.add_(1)
            evs[0].record()
            y.copy_(x)          # 256 MiB read + 256 MiB write
            evs[1].record()
            small.add_(1)
            evs[2].record()
            y.copy_(x)
            evs[3].record()
        for it in range(3):
            g.replay()
            torch.cuda.synchronize()
            print("external" if external else "plain", it, round(evs[0].elapsed_time(evs[1]) * 1e3, 1), "us copy,",
                  round(evs[1].elapsed_time(evs[2]) * 1e3, 1), "us tiny,", round(evs[2].elapsed_time(evs[3]) * 1e3, 1), "us copy")
    except Exception as e:
        print("external" if external else "plain", "FAILED:", type(e).__name__, str(e)[:300])
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); y.copy_(x); b.record(); torch.cuda.synchronize()
print("eager copy", round(a.elapsed_time(b) * 1e3, 1), "us")
