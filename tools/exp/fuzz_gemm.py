"""Random-shape check of fp8_scaled_mm (row-major and pre-shuffled weights, with / without bias, bf16 / fp16, strided rows)
against an fp64 product of the same fp8 values.  A bug hunt over the dispatch table's seams (64 / 128 / 256 rows, K tails,
narrow and ragged N); prints every failing shape with the kernel family that served it."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", "0")))
N_CASES = int(os.environ.get("N", "200"))
bad = 0
fam = {}
for it in range(N_CASES):
    M = rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 130, 191, 255, 256, 257, 300, 511, 600, 1025])
    K = rng.choice([16, 144, 512, 528, 1024, 1536, 2048, 3584, 4096, 7168, 8192, 14336])
    N = rng.choice([8, 16, 48, 72, 128, 1000, 1008, 1280, 2064, 4096, 4112, 6144, 7168, 12288, 28672])
    if M * N > 12_000_000 or N * K > 130_000_000:
        continue
    dt = rng.choice([torch.bfloat16, torch.float16])
    g = torch.Generator(device=DEV).manual_seed(it)
    pad = rng.choice([0, 0, 64])
    a_full = ((torch.rand(M, K + pad, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    a = a_full[:, :K]
    w = ((torch.rand(N, K, generator=g, device=DEV) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    sb = torch.rand(N, 1, generator=g, device=DEV) * 1e-2 + 1e-3
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if rng.random() < 0.4 else None
    shuf = rng.random() < 0.6 and ops.fp8_shuffle_supported(N, K)
    wt = ops.fp8_shuffle_weight(w) if shuf else w.t()
    if os.environ.get("TRACE"):  # one line per case BEFORE it runs: a GPU fault kills the process, the log names the shape
        print("CASE", dict(it=it, M=M, N=N, K=K, shuf=shuf, pad=pad, dt=str(dt), bias=bias is not None), flush=True)
    try:
        out = ops.fp8_scaled_mm(a, wt, sa, sb, dt, bias)
        name = ops.fp8_last_kernel()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("EXC", dict(M=M, N=N, K=K, shuf=shuf, pad=pad, dt=str(dt)), repr(e)[:200])
        bad += 1
        continue
    fam[name] = fam.get(name, 0) + 1
    ref = (a.double() @ w.double().t()) * sb.double().view(1, -1) * sa.double()
    if bias is not None:
        ref = ref + bias.double()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    err = (out.double() - ref).abs()
    tol = ulp * ref.abs() + 1e-3 * float(ref.abs().max()) + 1e-9
    if not bool((err <= tol).all()):
        print("MISMATCH", dict(M=M, N=N, K=K, shuf=shuf, pad=pad, dt=str(dt), bias=bias is not None, kernel=name),
              "max excess", float((err - tol).max()))
        bad += 1
print("cases", N_CASES, "bad", bad, "families", fam)
sys.exit(1 if bad else 0)
