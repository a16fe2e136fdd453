"""Decode attention at batch sizes whose (request, kv head) items do not fill whole rounds of 256 workgroups: one split (pairs of
items) against 2 / 4 kv-splits + merge.  bs x 8 kv heads, 32/8/128, random page table, 8 pools; us per call."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_decode import run
for S in [int(x) for x in os.environ.get("SS", "2048").split(",")]:
    for B in [int(x) for x in os.environ.get("BS", "64,65,72,80,96,112,128").split(",")]:
        row = {"S": S, "B": B}
        for splits in (1, 2, 4):
            ms, gbs = run(B, 32, 8, 128, S, splits, "random", iters=48, nlayers=4 if S > 4096 else 8)
            row[f"splits{splits}_us"] = round(ms * 1e3, 1)
        print(json.dumps(row), flush=True)
