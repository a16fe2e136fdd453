#!/usr/bin/env python3
"""Where a launch of the kv-split decode kernel spends its time: s_memtime stamps of wave 0 of every workgroup
(timing build: python -m sglang_npu_amd.build_ext --variant dec_timing --flag=-DSGLM_DEC_TIMING=1, loaded through SGL_MI355_LIB).
CASE="B,Hq,Hkv,ctx,splits" (default 64,8,1,2048,4: one rank of Llama-3-70B TP 8)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import _lib, ops
dev = "cuda:0"
B, Hq, Hkv, S, splits = [int(x) for x in os.environ.get("CASE", "64,8,1,2048,4").split(",")]
FP8 = bool(int(os.environ.get("FP8_OUT", "0")))
D, NL = 128, 8
g = torch.Generator(device=dev).manual_seed(0)
n_tok = B * S + 1
kbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16() for _ in range(NL)]
vbs = [torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16() for _ in range(NL)]
q = torch.randn(B, Hq, D, device=dev, generator=g).bfloat16()
o = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)
r2t = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).view(B, S).to(torch.int32).contiguous()
rpi, seq = torch.arange(B, device=dev), torch.full((B,), S, device=dev)
counters = torch.zeros(B, dtype=torch.int32, device=dev)
logits = torch.zeros(B, Hq, splits, D + 1, device=dev)


def run(i):
    r = ops.decode_attention_paged_merged(q, kbs[i % NL], vbs[i % NL], o, r2t, rpi, seq, logits, splits, counters, D ** -0.5, 0.0,
                                          fp8_out=FP8)
    assert r is not False


lib = _lib.lib()
W, NS = 1024, 16
buf = np.zeros(W * NS, dtype=np.uint64)
for i in range(NL):
    run(i)
torch.cuda.synchronize()
_lib.check(lib.sgl_mi355_decode_timing_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes)))  # clears
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
REPS = 48  # back to back: the stamps that remain are the LAST launch's, taken in the steady state of a busy queue
st.record()
for i in range(REPS):
    run(i)
en.record(); torch.cuda.synchronize()
_lib.check(lib.sgl_mi355_decode_timing_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes)))
s = buf.reshape(W, NS).astype(np.int64)
n_wg = B * Hkv * ((Hq // Hkv + 15) // 16) * splits
s = s[:min(n_wg, W)]
t0 = s[:, 0].min()
TICK = float(os.environ.get("TICK_NS", "0.944"))  # s_memtime tick of this part, calibrated in profiles/r05_extend_phase_times.txt
us = lambda x: x * TICK / 1e3
print(f"# B={B} Hq={Hq} Hkv={Hkv} ctx={S} splits={splits} fp8_out={FP8}: {st.elapsed_time(en) * 1e3 / REPS:.1f} us per launch, {REPS} eager launches back to back (instrumented "
      f"build), {len(s)} workgroups; stamps of the last launch, us from each workgroup's OWN entry (the XCDs' clocks are not aligned; tick = {TICK} ns)")
names = ["0 entry", "1 split range known (rpi / seq_lens loads)", "2 page-table slice staged in LDS", "3 first K tile landed",
         "4 stream + MFMA loop done (wave 0)", "5 all waves done (barrier)", "6 wave merge + partial stores acknowledged",
         "7 counter atomic returned", "8 last arrival: merge done"]
for i, nme in enumerate(names):
    col = s[:, i]
    m = col > 0
    if not m.any():
        continue
    x = us(col[m] - s[m, 0])
    print(f"  {nme:48s} n={int(m.sum()):4d}  min {x.min():6.2f}  median {np.median(x):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f}")
print("  per-workgroup phase lengths (median / p90, us):")
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
for a_, b_ in pairs:
    m = (s[:, a_] > 0) & (s[:, b_] > 0)
    if m.any():
        x = us(s[m, b_] - s[m, a_])
        print(f"    {a_}->{b_}: {np.median(x):6.2f} / {np.percentile(x, 90):6.2f}")
