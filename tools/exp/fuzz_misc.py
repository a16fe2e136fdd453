"""Random-shape checks of the remaining hot-path ops -- AWQ fused GEMMs (checkpoint layout, k-packed decode, k-packed tiled),
the 16-bit weight streamer, per-token FP8 quant, (add +) RMSNorm (+ quant), SiLU * mul (+ quant) -- against torch references.
TRACE=1 prints each case before it runs (a GPU fault kills the process; the log names the shape)."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", "0")))
N_CASES = int(os.environ.get("N", "150"))
bad = 0


def trace(**kw):
    if os.environ.get("TRACE"):
        print("CASE", kw, flush=True)


def fail(msg, **kw):
    global bad
    bad += 1
    print(msg, kw)


def awq_case(K, N, G, g):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // G, N // 8), dtype=torch.int32, generator=g)
    sc = (torch.rand(K // G, N, generator=g) * 0.02 + 0.002).half()
    return qw.to(DEV), qz.to(DEV), sc.to(DEV)


for it in range(N_CASES):
    kind = rng.choice(["awq", "awq_packed", "awq_tiled", "linear16", "quant", "norm", "silu"])
    g = torch.Generator().manual_seed(it)
    gd = torch.Generator(device=DEV).manual_seed(it)
    if kind.startswith("awq"):
        G = 128
        K = rng.choice([128, 256, 512, 1024, 2176, 4096, 11008])
        N = rng.choice([32, 64, 128, 1000 // 8 * 8 if False else 1024, 4096, 12288, 22016])
        M = rng.choice([1, 7, 16, 33, 64]) if kind != "awq_tiled" else rng.choice([65, 100, 128, 200, 513])
        if K * N > 100_000_000:
            continue
        trace(it=it, kind=kind, M=M, K=K, N=N)
        qw, qz, sc = awq_case(K, N, G, g)
        x = (torch.randn(M, K, generator=g) * 0.5).half().to(DEV)
        bias = torch.randn(N, generator=g).half().to(DEV) if rng.random() < 0.4 else None
        W = ops.awq_dequantize(qw, sc, qz).double()
        ref = x.double() @ W + (bias.double() if bias is not None else 0)
        try:
            if kind == "awq":
                out = ops.awq_gemm(x, qw, sc, qz, bias)
            else:
                wp, sz = ops.awq_repack(qw, sc, qz)
                out = (ops.awq_gemm_packed if kind == "awq_packed" else ops.awq_gemm_packed_tiled)(x, wp, sz, G, bias)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            fail("EXC", it=it, kind=kind, M=M, K=K, N=N, e=repr(e)[:160])
            continue
        tol = 2.0 ** -10 * ref.abs() + 2e-3 * float(ref.abs().max())
        if not bool(((out.double() - ref).abs() <= tol).all()):
            fail("MISMATCH", it=it, kind=kind, M=M, K=K, N=N, excess=float(((out.double() - ref).abs() - tol).max()))
    elif kind == "linear16":
        M = rng.choice([1, 2, 16, 17, 33, 64, 65, 100, 128])
        K = rng.choice([256, 512, 1024, 3584, 4096, 14336])
        N = rng.choice([8, 16, 72, 1000, 1008, 4096, 6144, 32000, 128256])
        if N * K > 600_000_000:
            continue
        dt = rng.choice([torch.bfloat16, torch.float16])
        shuf = rng.random() < 0.5 and ops.linear16_shuffle_supported(N, K)
        trace(it=it, kind=kind, M=M, K=K, N=N, shuf=shuf, dt=str(dt))
        x = torch.randn(M, K, device=DEV, generator=gd).to(dt)
        w = (torch.randn(N, K, device=DEV, generator=gd) * 0.05).to(dt)
        bias = torch.randn(N, device=DEV, generator=gd).to(dt) if rng.random() < 0.3 else None
        try:
            out = ops.linear16(x, ops.linear16_shuffle_weight(w) if shuf else w, bias)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            fail("EXC", it=it, kind=kind, M=M, K=K, N=N, shuf=shuf, e=repr(e)[:160])
            continue
        ref = x.double() @ w.double().t() + (bias.double() if bias is not None else 0)
        ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
        tol = ulp * ref.abs() + 0.05 * ulp * float(ref.abs().max()) + 1e-6
        if not bool(((out.double() - ref).abs() <= tol).all()):
            fail("MISMATCH", it=it, kind=kind, M=M, K=K, N=N, shuf=shuf, excess=float(((out.double() - ref).abs() - tol).max()))
    else:
        T = rng.choice([1, 3, 16, 63, 64, 65, 128, 333, 1024])
        H = rng.choice([8, 64, 128, 896, 1024, 3584, 4096, 8192, 14336])
        dt = rng.choice([torch.bfloat16, torch.float16])
        trace(it=it, kind=kind, T=T, H=H, dt=str(dt))
        x = (torch.randn(T, H, device=DEV, generator=gd) * 2).to(dt)
        try:
            if kind == "quant":
                q = torch.empty(T, H, dtype=torch.float8_e4m3fn, device=DEV)
                s = torch.empty(T, 1, dtype=torch.float32, device=DEV)
                ops.sgl_per_token_quant_fp8(x, q, s)
                torch.cuda.synchronize()
                s_ref = (x.float().abs().amax(dim=1, keepdim=True).cpu() / 448.0)  # (a true division: torch's GPU `/ scalar` multiplies by 1/448)
                if not torch.equal(s.cpu(), s_ref):
                    fail("MISMATCH scale", it=it, kind=kind, T=T, H=H)
                deq = q.float() * s
                if not bool(((deq - x.float()).abs() <= s * 16 + 1e-6).all()):  # e4m3 step at 448 is 32: half of it
                    fail("MISMATCH q", it=it, kind=kind, T=T, H=H)
            elif kind == "norm":
                w = (torch.rand(H, device=DEV, generator=gd) + 0.5).to(dt)
                res = torch.randn(T, H, device=DEV, generator=gd).to(dt)
                r2 = res.clone()
                qf, sf, out = ops.rmsnorm_quant_fp8(x.clone(), w, 1e-5, residual=r2, want_out=True)
                torch.cuda.synchronize()
                h = (x.float() + res.float()).to(dt)
                if not torch.equal(r2, h):
                    fail("MISMATCH residual", it=it, kind=kind, T=T, H=H)
                hf = h.float()
                ref = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dt).float() * w.float()
                ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
                if not bool(((out.float() - ref).abs() <= 2 * ulp * ref.abs() + 1e-3).all()):
                    fail("MISMATCH norm", it=it, kind=kind, T=T, H=H, excess=float((out.float() - ref).abs().max()))
            else:
                if H % 16:
                    continue
                y = ops.silu_and_mul(x)
                qf, sf = ops.silu_and_mul_quant_fp8(x)
                torch.cuda.synchronize()
                d = H // 2
                a, b = x[:, :d].float(), x[:, d:].float()
                ref = ((a / (1 + torch.exp(-a))).to(dt).float() * b).to(dt)
                ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
                if not bool(((y.float() - ref.float()).abs() <= 2 * ulp * ref.float().abs() + 1e-6).all()):
                    fail("MISMATCH silu", it=it, kind=kind, T=T, H=H)
                if not torch.equal(sf.cpu(), y.float().abs().amax(dim=1, keepdim=True).cpu() / 448.0):
                    fail("MISMATCH silu scale", it=it, kind=kind, T=T, H=H)
        except Exception as e:  # noqa: BLE001
            fail("EXC", it=it, kind=kind, T=T, H=H, e=repr(e)[:160])
print("cases", N_CASES, "bad", bad)
sys.exit(1 if bad else 0)
