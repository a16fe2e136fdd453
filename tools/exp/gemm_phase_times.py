#!/usr/bin/env python3
"""Per-workgroup phase timestamps of fp8_gemm_tiled3_kernel (variant build -DSGLM_T3_TIMING=1: the `bias` argument is a buffer of
four 100-MHz timestamps per workgroup: entry, K loop start, K loop end, after the output stores).
  python -m sglang_npu_amd.build_ext --variant t3timing --flag=-DSGLM_T3_TIMING=1
  SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_t3timing.so python tools/exp/gemm_phase_times.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
CASES = [("qkv", 1024, 6144, 4096, False), ("qkv", 1024, 6144, 16384, False), ("qkv", 4096, 6144, 4096, False),
         ("gate_up+silu", 1024, 28672, 4096, True), ("gate_up+silu", 4096, 28672, 4096, True)]
for name, M, N, K, silu in CASES:
    nw = 3
    ws = [ops.fp8_shuffle_weight(((torch.rand(N, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)) for _ in range(nw)]
    sb = torch.rand(N, device=dev, generator=g) * 1e-2
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=dev, generator=g) * 1e-2
    tb = torch.zeros(N, dtype=torch.bfloat16, device=dev)
    fn = ops.fp8_scaled_mm_silu_mul if silu else ops.fp8_scaled_mm
    rows = []
    for it in range(6):
        tb.zero_()
        torch.cuda.synchronize()
        fn(a, ws[it % nw], sa, sb, torch.bfloat16, tb)
        torch.cuda.synchronize()
        t = tb.view(torch.int64).cpu()
        n = int((t.view(-1, 4)[:, 0] != 0).sum())
        t = t.view(-1, 4)[:n].double() * 0.01  # us
        if it >= 2:
            t0 = t[:, 0].min()
            rows.append(dict(wgs=n, entry_spread=float(t[:, 0].max() - t0), prologue=float((t[:, 1] - t[:, 0]).mean()),
                             loop=float((t[:, 2] - t[:, 1]).mean()), epilogue=float((t[:, 3] - t[:, 2]).mean()),
                             span=float(t[:, 3].max() - t0), last_entry_to_end=float(t[:, 3].max() - t[:, 0].max())))
    med = {k: round(sorted(r[k] for r in rows)[len(rows) // 2], 2) for k in rows[0]}
    print(json.dumps(dict(shape=name, M=M, N=N, K=K, kernel=ops.fp8_last_kernel(), ks=os.environ.get("SGL_MI355_T3_KS", "auto"), **med)), flush=True)
    del ws
