#!/usr/bin/env python3
"""AWQ prefill GEMM (fused tiled kernel) vs dequantise + library GEMM, Llama-2-7B shapes, M = 128 / 512 / 1024 / 4096."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    qw = torch.randint(0, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=g)
    qz = torch.randint(0, 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=g)
    sc = (torch.rand(K // 128, N, device=dev, generator=g) * 2e-2).half()
    wp, sz = ops.awq_repack(qw, sc, qz)
    wd = ops.awq_dequantize(qw, sc, qz)
    for M in (128, 512, 1024, 4096):
        x = torch.randn(M, K, device=dev, generator=g).half()

        def t(fn, n=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3

        us_f = t(lambda: ops.awq_gemm_packed_tiled(x, wp, sz, 128))
        us_l = t(lambda: torch.matmul(x, wd))
        us_d = t(lambda: ops.awq_dequantize(qw, sc, qz))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(K=K, N=N, M=M, fused_us=round(us_f, 1), fused_TFLOPs=round(fl / us_f / 1e6, 1),
                              lib_gemm_on_fp16_copy_us=round(us_l, 1), dequant_us=round(us_d, 1))), flush=True)
