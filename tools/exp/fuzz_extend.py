"""Random-configuration check of the extend (prefill) attention -- ragged prefix + extend lengths, GQA groups, head sizes 64 / 128,
causal -- against an fp32 torch reference.  A bug hunt; TRACE=1 prints each configuration before it runs (a GPU fault kills the
process, the log names the shape)."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", "0")))
N = int(os.environ.get("N", "120"))
bad = 0
for it in range(N):
    D = rng.choice([64, 128])
    Hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 4, 7, 8])
    Hq = Hkv * group
    B = rng.choice([1, 1, 2, 3, 5, 9])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    pre = [rng.choice([0, 0, 1, 17, 64, 100, 500, 1300]) for _ in range(B)]
    ext = [rng.choice([1, 2, 31, 32, 33, 64, 65, 200, 513, 1024]) for _ in range(B)]
    if sum(e * (p + e) for p, e in zip(pre, ext)) * Hq > 60_000_000:
        continue
    if os.environ.get("TRACE"):
        print("CASE", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, D=D, dtype=str(dtype), pre=pre, ext=ext), flush=True)
    g = torch.Generator(device=DEV).manual_seed(it)
    total_pre, total_ext = sum(pre), sum(ext)
    rows = total_pre + total_ext + 7
    kb = torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype)
    q = torch.randn(total_ext, Hq, D, device=DEV, generator=g).to(dtype)
    k_e = torch.randn(total_ext, Hkv, D, device=DEV, generator=g).to(dtype)
    v_e = torch.randn(total_ext, Hkv, D, device=DEV, generator=g).to(dtype)
    perm = (torch.randperm(rows - 1, device=DEV, generator=g) + 1).to(torch.int32)
    width = max(p + e for p, e in zip(pre, ext)) + 2
    r2t = torch.zeros(B, width, dtype=torch.int32, device=DEV)
    off = 0
    for b in range(B):
        n = pre[b] + ext[b]
        r2t[b, :n] = perm[off:off + n]
        off += n
    # the extend tokens' K/V are also what sits in the pool at their slots (the op reads the prefix from the pool only)
    start = [0]
    for e in ext:
        start.append(start[-1] + e)
    for b in range(B):
        idx = r2t[b, pre[b]:pre[b] + ext[b]].long()
        kb[idx] = k_e[start[b]:start[b + 1]]
        vb[idx] = v_e[start[b]:start[b + 1]]
    o = torch.full((total_ext, Hq, D), 7.0, dtype=dtype, device=DEV)
    rpi = torch.arange(B, device=DEV)
    seq = torch.tensor([p + e for p, e in zip(pre, ext)], dtype=torch.int64, device=DEV)
    ext_t = torch.tensor(ext, dtype=torch.int64, device=DEV)
    start_t = torch.tensor(start[:-1], dtype=torch.int64, device=DEV)
    scale = D ** -0.5
    try:
        ops.extend_attention(q, k_e, v_e, o, kb, vb, r2t, rpi, seq, ext_t, start_t, max(ext), scale, 0.0)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("EXC", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, D=D, pre=pre, ext=ext), repr(e)[:200])
        bad += 1
        continue
    ref = torch.zeros(total_ext, Hq, D, device=DEV)
    for b in range(B):
        n = pre[b] + ext[b]
        idx = r2t[b, :n].long()
        k = kb[idx].float()
        v = vb[idx].float()
        qq = q[start[b]:start[b + 1]].float().view(ext[b], Hkv, group, D)
        s = torch.einsum("thgd,nhd->thgn", qq, k) * scale
        pos_q = torch.arange(pre[b], n, device=DEV).view(-1, 1, 1, 1)
        pos_k = torch.arange(n, device=DEV).view(1, 1, 1, -1)
        s = s.masked_fill(pos_k > pos_q, float("-inf"))
        p = torch.softmax(s, dim=-1)
        ref[start[b]:start[b + 1]] = torch.einsum("thgn,nhd->thgd", p, v).reshape(ext[b], Hq, D)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    err = (o.float() - ref).abs()
    tol = 3e-3 + 4 * ulp * ref.abs() + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 3.0
    if not bool((err <= tol).all()) or not bool(torch.isfinite(o.float()).all()):
        print("MISMATCH", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, D=D, dtype=str(dtype), pre=pre, ext=ext), "max err", float(err.max()),
              "excess", float((err - tol).max()))
        bad += 1
print("configs", N, "bad", bad)
sys.exit(1 if bad else 0)
