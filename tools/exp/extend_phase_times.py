#!/usr/bin/env python3
"""Where a tile step of the extend kernel spends its time: s_memtime stamps of wave 0 of the 64 heaviest workgroups
(timing build: python -m sglang_npu_amd.build_ext --variant ext_timing --flag=-DSGLM_EXT_TIMING=1, loaded through SGL_MI355_LIB).
CASE="B,L,P" (default 1,4096,0)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import _lib, ops
dev = "cuda:0"
B, L, P = [int(x) for x in os.environ.get("CASE", "1,4096,0").split(",")]
Hq, Hkv, D = 32, 8, 128
g = torch.Generator(device=dev).manual_seed(0)
n_tok = B * (L + P) + 1
kb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
vb = torch.randn(n_tok, Hkv, D, device=dev, generator=g).bfloat16()
perm = (torch.randperm(n_tok - 1, device=dev, generator=g) + 1).to(torch.int32)
q = torch.randn(B * L, Hq, D, device=dev, generator=g).bfloat16()
ke = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
ve = torch.randn(B * L, Hkv, D, device=dev, generator=g).bfloat16()
o = torch.zeros(B * L, Hq, D, dtype=torch.bfloat16, device=dev)
qo = (torch.arange(B + 1, device=dev) * L).to(torch.int32)
kvp = (torch.arange(B + 1, device=dev) * P).to(torch.int32)
idx = perm[: B * P].contiguous() if P else torch.zeros(1, dtype=torch.int32, device=dev)
run = lambda: ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, idx, None, True, None, L, D ** -0.5, 0.0)
lib = _lib.lib()
W, T, S = 64, 80, 6
buf = np.zeros(W * T * S, dtype=np.uint64)
for _ in range(3):
    run()
torch.cuda.synchronize()
_lib.check(lib.sgl_mi355_extend_timing_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes)))  # clears
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record(); run(); en.record(); torch.cuda.synchronize()
_lib.check(lib.sgl_mi355_extend_timing_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes)))
s = buf.reshape(W, T, S).astype(np.int64)
valid = s[:, :, 5] > 0
print(f"# case B,L,P = {B},{L},{P}: launch {st.elapsed_time(en) * 1e3:.1f} us (instrumented build); {int(valid.sum())} stamped tile steps of wave 0 "
      f"in {int(valid.any(1).sum())} workgroups")
names = ["s0->s1  wait for the tile (vmcnt + barrier 1)", "s1->s2  K reads + QK^T MFMAs + scale + row max + exchange",
         "s2->s3  exp2 / row sum / pack (softmax VALU)", "s3->s4  V reads + PV MFMAs", "s4->s5  barrier 2 (everyone done reading)",
         "s5->s0' DMA issue of the next tile + loop"]
d = [s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2], s[:, :, 4] - s[:, :, 3], s[:, :, 5] - s[:, :, 4]]
nxt = np.zeros_like(s[:, :, 0]); nxt[:, :-1] = s[:, 1:, 0] - s[:, :-1, 5]
v2 = valid.copy(); v2[:, :-1] &= valid[:, 1:]; v2[:, -1] = False
tot = np.zeros_like(d[0])
for i, x in enumerate(d + [nxt]):
    m = (v2 if i == 5 else valid) & (x >= 0) & (x < 10 ** 7)
    xs = x[m]
    print(f"  {names[i]:58s} median {np.median(xs):7.0f}   mean {xs.mean():7.0f}   p90 {np.percentile(xs, 90):7.0f}  (ticks of s_memtime)")
step = (s[:, 1:, 0] - s[:, :-1, 0])[v2[:, :-1]]
print(f"  whole step (s0 -> next s0): median {np.median(step):.0f}, mean {step.mean():.0f} ticks; steps per workgroup {valid.sum(1).max()}")
# steady-state tiles only (skip the first 4 of each workgroup: running maximum still moving) -- and the first vs last quartile of steps
for lo, hi, tag in ((4, 24, "tiles 4..23"), (40, 60, "tiles 40..59")):
    sub = [x[:, lo:hi][valid[:, lo:hi]] for x in d]
    if len(sub[0]):
        print(f"  {tag}: " + " | ".join(f"{np.median(x):.0f}" for x in sub))
