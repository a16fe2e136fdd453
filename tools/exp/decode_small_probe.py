"""Decode attention with FEWER (request, kv head) items than CUs -- the TP-sharded geometries (70B TP 8: 8 q heads on 1 kv head
per rank) -- over context length x kv-splits, graph-replayed on rotating pools: us per launch, the fixed cost and the slope.
  python tools/exp/decode_small_probe.py [Hq Hkv [B]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops  # noqa: E402

dev = "cuda:0"
Hq, Hkv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 1)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
D, NL = 128, 8
g = torch.Generator(device=dev).manual_seed(0)
FORMS = os.environ.get("FORMS", "merged,merged_fp8,two_launch").split(",")
print(f"# B={B} Hq={Hq} Hkv={Hkv} D={D} bf16; {NL} pools in rotation, one HIP graph of {NL} launches, median of 7 replays", flush=True)
for S in (256, 512, 1024, 2048, 4096, 8192):
    n_tok = B * S + 1
    kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(NL)]
    q = torch.randn(B, Hq, D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
    r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
    rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
    counters = torch.zeros(B, dtype=torch.int32, device=dev)
    nbytes = B * S * Hkv * 2 * D * 2
    row = {}
    for form in FORMS:
        for splits in (1, 2, 4, 8, 16):
            if S // splits < 64:
                continue
            logits = torch.zeros(B, Hq, splits, D + 1, device=dev)

            def run(i):
                if splits == 1 or form == "two_launch":
                    ops.decode_attention_paged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, logits if splits > 1 else None, splits,
                                               D ** -0.5, 0.0)
                else:
                    r = ops.decode_attention_paged_merged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, logits, splits, counters,
                                                          D ** -0.5, 0.0, fp8_out=(form == "merged_fp8"))
                    assert r is not False
            if splits == 1 and form != FORMS[0]:
                continue
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for i in range(NL):
                    run(i)
            ts = []
            for _ in range(7):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                graph.replay()
                en.record()
                torch.cuda.synchronize()
                ts.append(st.elapsed_time(en) / NL * 1e3)
            row[(form, splits)] = sorted(ts)[3]
    best = min(row.items(), key=lambda kv: kv[1])
    print(f"ctx {S:5d}: {nbytes / 1e6:7.1f} MB  floor@8TB/s {nbytes / 8e6:6.1f} us | " +
          "  ".join(f"{f[:9]}/{s}: {v:5.1f}" for (f, s), v in sorted(row.items())) +
          f" | best {best[0][0]}/{best[0][1]} {best[1]:.1f} us = {nbytes / best[1] / 8e6:.3f} of 8 TB/s", flush=True)
    del kbs, vbs
