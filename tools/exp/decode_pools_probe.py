"""Does the number of distinct KV pools the decode kernel cycles through (TLB reach / cache state) explain the 88 -> 95 us it
takes inside the model step?  bs=64, 32/8/128, S=2048, random page table, eager back-to-back launches."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_decode import run
for nl in (1, 8, 32, 8):
    ms, gbs = run(64, 32, 8, 128, 2048, 1, "random", iters=96, nlayers=nl)
    print(json.dumps(dict(pools=nl, us=round(ms * 1e3, 2), GBps=round(gbs, 1), frac=round(gbs / 8000, 4))), flush=True)
