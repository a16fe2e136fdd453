import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
def rf(shape, g): return ((torch.rand(shape, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
for (M, N, K, use_bias, dt) in [(1000, 14400, 1024, False, torch.bfloat16), (1024, 14336 * 2, 1024, True, torch.bfloat16),
                                (1000, 14400, 1024, True, torch.bfloat16), (1024, 28672, 1024, True, torch.float16)]:
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a, w = rf((M, K), g), rf((N, K), g)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if use_bias else None
    wsh = ops.mark_wshuffled(ops.fp8_shuffle_weight(w).t())
    y = ops.fp8_scaled_mm(a, wsh, sa, sb, dt, bias)
    ref = ops.silu_and_mul(y)
    out = ops.fp8_scaled_mm_silu_mul(a, wsh, sa, sb, dt, bias)
    bad = (out.view(torch.int16) != ref.view(torch.int16))
    print(M, N, K, use_bias, dt, "mismatches", int(bad.sum()), "nan", int(torch.isnan(ref.float()).sum()))
    if bad.any():
        idx = bad.nonzero()[:5]
        for r, c in idx.tolist():
            print("  at", r, c, float(out[r, c]), float(ref[r, c]), "gate", float(y[r, c]), "up", float(y[r, N // 2 + c]))
        print("  rows with mismatch:", bad.any(1).nonzero().flatten()[:10].tolist(), "cols:", bad.any(0).nonzero().flatten()[:10].tolist())
