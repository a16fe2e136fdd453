import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
M = int(os.environ.get("DBG_M", "64")); K = int(os.environ.get("DBG_K", "4096")); N = 28672
g = torch.Generator(device=dev).manual_seed(3)
a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
w = ((torch.rand(N, K, device=dev, generator=g) - 0.5) * 16).to(torch.float8_e4m3fn)
sa = torch.ones(M, device=dev); sb = torch.ones(N, device=dev)
torch.cuda.synchronize(); print("inputs ready", flush=True)
out = ops.fp8_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
torch.cuda.synchronize(); print("gemm done", flush=True)
out = out.float()
ref = (a.float() @ w.float().t())
torch.cuda.synchronize(); print("ref done", flush=True)
bad = ~torch.isclose(out, ref, rtol=2**-7, atol=1e-3)
print("M", M, "K", K, "bad", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(out).sum()))
if bad.any():
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten()
    print("bad rows", rows[:70])
    print("bad cols n", len(cols), "first", cols[:20].tolist(), "mod16 hist", torch.bincount(cols % 16, minlength=16).tolist())
    print("bad col blocks mod 8 hist", torch.bincount((cols // 16) % 8, minlength=8).tolist())
    # which k-steps are wrong? recompute leaving out one 128-step at a time for one bad element
    r, c = bad.nonzero()[0].tolist()
    print("elem", r, c, "out", out[r, c].item(), "ref", ref[r, c].item())

sys.exit(1 if bad.any() else 0)
