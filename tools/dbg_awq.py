import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from sglang_npu_amd import ops
DEV = "cuda:0"
K, N, G = 512, 16, 128
g = torch.Generator().manual_seed(K + N)
imax = torch.iinfo(torch.int32).max
qw = torch.randint(0, imax, (K, N // 8), dtype=torch.int32, generator=g)
qz = torch.randint(0, imax, (K // G, N // 8), dtype=torch.int32, generator=g)
sc = ((torch.rand(K // G, N, generator=g) - 0.3) * 2e-2).half()
w_ref = oracle.awq_dequantize(qw, sc, qz)
wp, sz = ops.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
ks = torch.arange(64)
x = torch.zeros(64, K, dtype=torch.float16); x[torch.arange(64), ks] = 1.0
out = ops.awq_gemm_packed(x.to(DEV), wp, sz, G).cpu()
bad = (out.view(torch.int16) != w_ref[ks].view(torch.int16))
print("bad", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()[:12]
for r, c in idx.tolist():
    print(r, c, "out", float(out[r, c]), hex(out.view(torch.int16)[r, c].item() & 0xffff), "ref", float(w_ref[ks][r, c]), hex(w_ref[ks].view(torch.int16)[r, c].item() & 0xffff), "scale", float(sc[r // G, c]))
