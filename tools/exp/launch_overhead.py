"""Host cost of one op call through the Python wrappers + C ABI (tiny inputs, so the GPU is never the bottleneck)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
x = torch.randn(4, 256, device=DEV).bfloat16()
q = torch.empty(4, 256, dtype=torch.float8_e4m3fn, device=DEV)
s = torch.empty(4, 1, dtype=torch.float32, device=DEV)
w = torch.ones(256, device=DEV, dtype=torch.bfloat16)
a8 = torch.zeros(4, 512, device=DEV).to(torch.float8_e4m3fn)
w8 = torch.zeros(256, 512, device=DEV).to(torch.float8_e4m3fn)
sa, sb = torch.ones(4, 1, device=DEV), torch.ones(256, 1, device=DEV)


def t(fn, n=3000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


print("per_token_quant_fp8      %.1f us/call (host)" % t(lambda: ops.sgl_per_token_quant_fp8(x, q, s)))
print("rmsnorm                  %.1f us/call (host)" % t(lambda: ops.rmsnorm(x, w, 1e-5)))
print("fp8_scaled_mm            %.1f us/call (host)" % t(lambda: ops.fp8_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16)))
print("torch.empty_like         %.1f us/call (host)" % t(lambda: torch.empty_like(x)))
print("torch add (aten kernel)  %.1f us/call (host)" % t(lambda: x + x))
