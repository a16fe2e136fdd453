#!/usr/bin/env python3
"""Kernel-level measurements for BASELINE configs 2-5: one JSON line per point with the roofline
fraction against the bound that applies (HBM 8 TB/s; dense FP8 MFMA 5 PFLOP/s -- the block-scaled f8f6f4 rate of
MI355X_MICROARCH.md; the non-scaled fp8 instructions run at the 2.5 PFLOP/s BF16 rate).
Times are HIP-event means over back-to-back launches with rotating operands (nothing stays in the Infinity Cache).
    python tools/bench_configs.py [decode] [fp8] [awq] [tp8]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops  # noqa: E402

dev = "cuda:0"
HBM, MFMA = 8000.0, 5000.0  # GB/s, TFLOP/s (dense FP8)


def bench(fn, iters=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(iters):
        fn(i)
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def out(**kw):
    print(json.dumps(kw), flush=True)


def decode_points():
    B, Hq, Hkv, D = 64, 32, 8, 128
    g = torch.Generator(device=dev).manual_seed(0)
    for S in (512, 2048, 8192):
        n_tok = B * S + 1
        NL = max(2, int(1.2e9 // (n_tok * Hkv * D * 4)))
        kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
        vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
        q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
        o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
        rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
        for table in ("random", "identity"):
            if table == "random":
                r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
            else:
                r2t = (torch.arange(n_tok - 1, device=dev) + 1).view(B, S).to(torch.int32).contiguous()
            ms = bench(lambda i: ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, None, 1,
                                                            D ** -0.5, 0.0))
            nbytes = B * S * Hkv * 2 * D * 2 + 4 * B * S + 2 * B * Hq * 2 * D
            out(config=2, op="decode_attention", B=B, S=S, table=table, us=round(ms * 1e3, 1),
                GBps=round(nbytes / ms / 1e6), frac_hbm=round(nbytes / ms / 1e6 / HBM, 3))
        del kbs, vbs


def fp8_points(shapes, tag):
    g = torch.Generator(device=dev).manual_seed(0)
    for (K, N) in shapes:
        nw = max(2, int(600e6 // (K * N)))
        ws = [((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn) for _ in range(nw)]
        sb = torch.rand(N, device=dev, generator=g) * 1e-2
        for M in (1, 64, 512, 4096):
            a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
            sa = torch.rand(M, device=dev, generator=g) * 1e-2
            ms = bench(lambda i: ops.fp8_scaled_mm(a, ws[i % nw].t(), sa, sb, torch.bfloat16))
            flops = 2.0 * M * N * K
            nbytes = M * K + K * N + 2 * M * N + 4 * (M + N)
            out(config=tag, op="fp8_scaled_mm", K=K, N=N, M=M, us=round(ms * 1e3, 1), TFLOPs=round(flops / ms / 1e9, 1),
                GBps=round(nbytes / ms / 1e6), frac_hbm=round(nbytes / ms / 1e6 / HBM, 3),
                frac_mfma=round(flops / ms / 1e9 / MFMA, 3))
        del ws
    for (T, K) in [(64, 4096), (64, 14336), (4096, 4096)]:
        x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
        qq = torch.empty(T, K, dtype=torch.float8_e4m3fn, device=dev)
        s = torch.empty(T, device=dev)
        ms = bench(lambda i: ops.sgl_per_token_quant_fp8(x, qq, s))
        out(config=tag, op="per_token_quant_fp8", T=T, K=K, us=round(ms * 1e3, 1), GBps=round((3 * T * K + 4 * T) / ms / 1e6))


def awq_points():
    g = torch.Generator(device=dev).manual_seed(0)
    G = 128
    for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
        nw = max(2, int(300e6 // (K * N // 2)))
        qws = [torch.randint(0, 2 ** 31 - 1, (K, N // 8), device=dev, generator=g, dtype=torch.int32) for _ in range(nw)]
        qz = torch.randint(0, 2 ** 31 - 1, (K // G, N // 8), device=dev, generator=g, dtype=torch.int32)
        sc = (torch.rand(K // G, N, device=dev, generator=g) * 1e-2).half()
        wbytes = K * N // 2 + (K // G) * N // 2 + (K // G) * N * 2
        for M in (1, 64, 512):
            x = torch.randn(M, K, device=dev, generator=g).half()
            nbytes = wbytes + 2 * M * K + 2 * M * N
            if M <= 64:
                if M == 1:
                    packed = [ops.awq_repack(q_, sc, qz) for q_ in qws]
                ms = bench(lambda i: ops.awq_gemm_packed(x, packed[i % nw][0], packed[i % nw][1], G))
                out(config=4, op="awq_gemm_packed", K=K, N=N, M=M, us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6),
                    frac_hbm=round(nbytes / ms / 1e6 / HBM, 3))
                ms = bench(lambda i: ops.awq_gemm(x, qws[i % nw], sc, qz))
                out(config=4, op="awq_gemm_checkpoint_layout", K=K, N=N, M=M, us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6),
                    frac_hbm=round(nbytes / ms / 1e6 / HBM, 3))
            ms2 = bench(lambda i: x @ ops.awq_dequantize(qws[i % nw], sc, qz))  # the reference's path (awq.py:407-418)
            out(config=4, op="awq_dequantize+matmul", K=K, N=N, M=M, us=round(ms2 * 1e3, 1),
                GBps_algorithmic=round(nbytes / ms2 / 1e6))
        del qws


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"decode", "fp8", "awq", "tp8"}
    if "decode" in what:
        decode_points()
    if "fp8" in what:
        fp8_points([(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)], 3)
    if "tp8" in what:  # config 5: the per-rank GEMMs of Llama-3-70B at TP=8
        fp8_points([(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)], 5)
    if "awq" in what:
        awq_points()
