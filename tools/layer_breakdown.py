#!/usr/bin/env python3
"""Per-kernel spans of ONE decoder layer from a rocprofv3 --kernel-trace CSV of bench.py (graph replay: a kernel's span runs
to the next kernel's start; its own duration is End - Start).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 16 --warmup 2 \
        --no-cpu-baseline --no-other-configs
    python tools/layer_breakdown.py gpurun_out/prof/*/*kernel_trace.csv [anchor-substring [must-contain-substring]]

A layer = the kernels from one launch of the anchor kernel (default: the decode attention kernel) to the next one; the
median layer (by total span) of all complete layers in the trace is printed.  must-contain: only layers in which some kernel's
name has that substring (two pass shapes that share the anchor kernel, e.g. 1024- and 128-token prefills, differ in their GEMMs)."""
import csv, glob, statistics, sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "decode_mfma"
must = sys.argv[3] if len(sys.argv) > 3 else None
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
import os
want = int(os.environ.get("LAYER_COUNT", "0"))  # only layers of exactly this many kernels (two call orders in one trace)
layers = []
for a, b in zip(idx, idx[1:]):
    if want and b - a != want:
        continue
    if 6 <= b - a <= 16:  # a decoder layer is 7-12 launches; the steps' first / last layers carry the LM head etc.
        if must is not None and not any(must in rows[i][2] for i in range(a, b)):
            continue
        span = rows[b][0] - rows[a][0]
        layers.append((span, a, b))
if not layers:
    sys.exit("no complete layers found")
layers.sort()
span, a, b = layers[len(layers) // 2]
counts = statistics.mode([l[2] - l[1] for l in layers])
import collections
print("# kernels per layer in this trace: " + ", ".join(f"{k}: {v} layers" for k, v in sorted(collections.Counter(b - a for a, b in zip(idx, idx[1:]) if b - a <= 20).items())))
print(f"# median of {len(layers)} layers: {span / 1e3:.1f} us, {b - a} kernels (most common count {counts})")
for i in range(a, b):
    st, en, name = rows[i]
    nxt = rows[i + 1][0]
    short = name.replace("void sglm::(anonymous namespace)::", "").replace("sglm::(anonymous namespace)::", "").split("(")[0]
    print(f"   {short[:72]:72s} span {(nxt - st) / 1e3:6.1f} us   busy {(en - st) / 1e3:6.1f} us")
