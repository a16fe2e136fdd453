"""Isolated decode attention (bs=64, 32/8/128, random page table, 8 pools) at the context lengths the bench's steps visit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_decode import run
for S in (2048, 2049, 2056, 2064, 2080, 2112, 2048):
    ms, gbs = run(64, 32, 8, 128, S, 1, "random", iters=96, nlayers=8)
    print(json.dumps(dict(S=S, us=round(ms * 1e3, 2), GBps=round(gbs, 1), frac=round(gbs / 8000, 4))), flush=True)
