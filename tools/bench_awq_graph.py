#!/usr/bin/env python3
"""Graph-timed AWQ decode GEMMs (Llama-2-7B shapes, BASELINE config 4): kernel time without the Python launch cost.
Each shape rotates through enough weight copies to exceed the 256 MB Infinity Cache."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
G = 128
imax = torch.iinfo(torch.int32).max
shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
Ms = [int(x) for x in os.environ.get("M_LIST", "1,16,64").split(",")]
for K, N in shapes:
    wbytes = K * N // 2
    NL = max(2, int(600e6 // wbytes))
    packs = []
    for _ in range(NL):
        qw = torch.randint(0, imax, (K, N // 8), dtype=torch.int32, device=dev, generator=g)
        qz = torch.randint(0, imax, (K // G, N // 8), dtype=torch.int32, device=dev, generator=g)
        sc = ((torch.rand(K // G, N, device=dev, generator=g) - 0.3) * 2e-2).half()
        packs.append(ops.awq_repack(qw, sc, qz))
    for M in Ms:
        x = torch.randn(M, K, device=dev, generator=g).half()
        run = lambda i: ops.awq_gemm_packed(x, packs[i % NL][0], packs[i % NL][1], G)
        for i in range(3): run(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        reps = 2 * NL
        with torch.cuda.graph(gr):
            for i in range(reps): out = run(i)
        gr.replay(); torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(3): gr.replay()
        en.record(); torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1e3 / (3 * reps)
        nbytes = wbytes + (K // G) * N * 4 + M * K * 2 + M * N * 2
        print(json.dumps(dict(op="awq_gemm_packed", K=K, N=N, M=M, us=round(us, 2), GBps=round(nbytes / us / 1e3),
                              frac_hbm=round(nbytes / us / 1e3 / 8000, 3))), flush=True)
    del packs
