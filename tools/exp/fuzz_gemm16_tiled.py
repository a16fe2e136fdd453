"""Random-shape check of ops.linear16 above 128 rows (the tiled 16-bit kernel on a fragment-major weight: every tile form, the
split-K form, ragged M / N, strided rows, bias) against an fp64 product of the same 16-bit values."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", "0")))
N_CASES = int(os.environ.get("N", "200"))
bad = ran = 0
split = 0
for it in range(N_CASES):
    M = rng.choice([129, 130, 191, 255, 256, 257, 300, 383, 384, 511, 512, 600, 1000, 1024, 1025, 1537, 2048, 2500])
    K = 256 * rng.choice([1, 2, 3, 4, 6, 8, 16, 28, 32, 43, 56])
    N = 16 * rng.choice([1, 3, 4, 7, 8, 13, 16, 63, 64, 80, 129, 256, 258, 384, 768, 1376, 1792])
    if M * N > 24_000_000 or N * K > 130_000_000:
        continue
    dt = rng.choice([torch.bfloat16, torch.float16])
    g = torch.Generator(device=DEV).manual_seed(it)
    pad = rng.choice([0, 0, 64])
    x_full = torch.randn(M, K + pad, generator=g, device=DEV).to(dt)
    x = x_full[:, :K]
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dt)
    bias = torch.randn(N, generator=g, device=DEV).to(dt) if rng.random() < 0.4 else None
    fm = ops.linear16_shuffle_weight(w)
    if os.environ.get("TRACE"):  # one line per case BEFORE it runs: a GPU fault kills the process, the log names the shape
        print("CASE", dict(it=it, M=M, N=N, K=K, pad=pad, dt=str(dt), bias=bias is not None), flush=True)
    try:
        out = ops.linear16(x, fm, bias)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("EXC", dict(M=M, N=N, K=K, pad=pad, dt=str(dt)), repr(e)[:200])
        bad += 1
        continue
    ran += 1
    split += ops._linear16_tiled_slices(M, N, K) >= 2
    ref = x.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    err = (out.double() - ref).abs()
    tol = ulp * ref.abs() + 0.05 * ulp * float(ref.abs().max()) + 1e-9
    if not bool((err <= tol).all()):
        print("MISMATCH", dict(M=M, N=N, K=K, pad=pad, dt=str(dt), bias=bias is not None), "max excess", float((err - tol).max()))
        bad += 1
print(f"configs {ran} (split-K form: {split}) bad {bad}")
sys.exit(1 if bad else 0)
