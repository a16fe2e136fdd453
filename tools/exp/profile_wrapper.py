import os, sys, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
x = torch.randn(4, 256, device=DEV).bfloat16()
q = torch.empty(4, 256, dtype=torch.float8_e4m3fn, device=DEV)
s = torch.empty(4, 1, dtype=torch.float32, device=DEV)
a8 = torch.zeros(4, 512, device=DEV).to(torch.float8_e4m3fn)
w8 = torch.zeros(256, 512, device=DEV).to(torch.float8_e4m3fn)
sa, sb = torch.ones(4, 1, device=DEV), torch.ones(256, 1, device=DEV)
for _ in range(200):
    ops.sgl_per_token_quant_fp8(x, q, s); ops.fp8_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    ops.sgl_per_token_quant_fp8(x, q, s)
    ops.fp8_scaled_mm(a8, w8.t(), sa, sb, torch.bfloat16)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
