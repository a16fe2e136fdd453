#!/usr/bin/env python3
"""Time the reference's OWN compiled CPU kernels (oracle/_ref/libsgl_ref_cpu.so, built from /root/reference by
oracle/ref_build/Makefile) on the hot-path shapes of the headline config, in the BUILD CONTAINER -- the reference
cannot travel to the GPU box.  3 warm-ups, median of 10.  Writes profiles/r02_reference_cpu_container.json, which
bench.py quotes as `reference_cpu_container` next to its own `cpu_baseline` (the C port timed on the GPU box's host).

    python tools/time_reference_cpu.py
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libsgl_ref_cpu.so")
    torch.ops.load_library(so)
    ref = torch.ops.sgl_ref
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    B, Hq, Hkv, D, splits = 64, 32, 8, 128, 8
    out = {"cores": cores, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
           "torch_threads": torch.get_num_threads(), "method": "3 warm-ups, median of 10", "points": []}
    for S in (1024, 2048):
        n_tok = B * S + 1
        q = torch.randn(B, Hq, D, generator=g).bfloat16()
        kb = torch.randn(n_tok, Hkv, D, generator=g).bfloat16()
        vb = torch.randn(n_tok, Hkv, D, generator=g).bfloat16()
        key, val = torch.randn(B, Hkv, D, generator=g).bfloat16(), torch.randn(B, Hkv, D, generator=g).bfloat16()
        r2t = (torch.randperm(n_tok - 1, generator=g) + 1).to(torch.int32).view(B, S)
        loc = r2t[:, -1].long().contiguous()
        o = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
        logits = torch.zeros(B, Hq, splits, D + 1)
        rpi, seq = torch.arange(B), torch.full((B,), S)
        ms = med(lambda: ref.decode_attention_cpu(q, kb, vb, o, key, val, loc, logits, r2t, rpi, seq, D ** -0.5, 0.0))
        kv_bytes = B * S * Hkv * 2 * D * 2
        out["points"].append({"op": "decode_attention_cpu", "B": B, "S": S, "Hq": Hq, "Hkv": Hkv, "D": D, "ms": round(ms, 3),
                              "kv_GBps": round(kv_bytes / ms / 1e6, 2)})
    x = torch.randn(64, 4096, generator=g).bfloat16()
    w = torch.randn(4096, generator=g).bfloat16()
    r = torch.randn(64, 4096, generator=g).bfloat16()
    out["points"].append({"op": "fused_add_rmsnorm_cpu", "T": 64, "H": 4096,
                          "ms": round(med(lambda: ref.fused_add_rmsnorm_cpu(x.clone(), r.clone(), w, 1e-5)), 4)})
    y = torch.randn(64, 2 * 14336, generator=g).bfloat16()
    out["points"].append({"op": "silu_and_mul_cpu", "T": 64, "d": 14336, "ms": round(med(lambda: ref.silu_and_mul_cpu(y)), 4)})
    # one decode layer's attention share of a step, scaled like bench.py's cpu_baseline
    p2048 = [p for p in out["points"] if p["op"] == "decode_attention_cpu" and p["S"] == 2048][0]
    out["decode_attention_32_layers_ms"] = round(32 * p2048["ms"], 1)
    path = os.path.join(ROOT, "profiles", "r02_reference_cpu_container.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
