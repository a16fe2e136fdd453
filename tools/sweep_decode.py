#!/usr/bin/env python3
"""Round/ramp model fit for the decode kernel: time vs batch (rounds) and vs S."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_decode import run

for (B, S) in [(16, 2048), (32, 2048), (64, 2048), (96, 2048), (128, 2048), (32, 4096), (64, 1024), (64, 4096), (32, 8192)]:
    ms, gbs = run(B, 32, 8, 128, S, 1, "random", iters=30, nlayers=max(2, 4 * 64 * 2048 // (B * S)))
    print(json.dumps(dict(B=B, S=S, wgs=B * 8, ms=round(ms, 4), GBps=round(gbs, 1))), flush=True)
