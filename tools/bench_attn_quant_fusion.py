"""o_proj at decode (M=64, 4096 -> 4096): per-token quant + split-K partials + norm-from-partials against the a16 form
(the GEMM quantises while staging).  Graph-timed over 32 different weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
M, K, N, L = 64, 4096, 4096, 32
ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=DEV, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)) for _ in range(L)]
sb = torch.rand(N, 1, device=DEV, generator=g) * 1e-2 + 1e-3
x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
amax = x.float().abs().amax(dim=1).contiguous()
wn = torch.ones(N, device=DEV).bfloat16()
res = torch.zeros(M, N, device=DEV).bfloat16()
xq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=DEV)
xs = torch.empty(M, 1, dtype=torch.float32, device=DEV)
ops.reserve_gemm_workspace(DEV, 64, 28672)


def seq(i):
    ops.sgl_per_token_quant_fp8(x, xq, xs)
    ops.rmsnorm_quant_fp8_from_partials(ops.fp8_scaled_mm_partials(xq, ws[i], xs, sb, torch.bfloat16), res, wn, 1e-5)


def fused(i):
    ops.rmsnorm_quant_fp8_from_partials(ops.fp8_scaled_mm_partials_a16(x, amax, ws[i], sb, torch.bfloat16), res, wn, 1e-5)


def timeit(fn, name):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(0)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for i in range(L):
                fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); en.record(); torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) * 1e3 / L)
    print(f"{name:40s} {best:7.2f} us per layer")


timeit(seq, "quant + partials + norm-from-partials")
timeit(fused, "a16 partials + norm-from-partials")
