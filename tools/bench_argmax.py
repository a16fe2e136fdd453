#!/usr/bin/env python3
"""ops.argmax vs torch.argmax on the LM-head logits shapes (graph-timed)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_npu_amd import ops
dev = "cuda:0"
for rows, cols in [(64, 128256), (64, 16032), (1, 128256), (64, 32000)]:
    x = torch.randn(rows, cols, device=dev).bfloat16()
    for name, fn in (("ops.argmax", lambda: ops.argmax(x)), ("torch.argmax", lambda: torch.argmax(x, dim=-1))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): fn()
        gr.replay(); torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); gr.replay(); gr.replay(); en.record(); torch.cuda.synchronize()
        print(json.dumps(dict(rows=rows, cols=cols, op=name, us=round(st.elapsed_time(en) * 1e3 / 40, 2))), flush=True)
