"""Random-configuration check of the KV-range-parts form of the extend kernel (ops.extend_attention_fwd with the host's prefix
bound + scratch): few requests, long ragged prefixes, short extends, GQA groups 1-8, D = 128, against an fp32 torch reference;
every configuration twice (the merge order is fixed: same bits), counters must be zero afterwards."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd import ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("SEED", "0")))
N = int(os.environ.get("N", "120"))
bad = ran = used = 0
scratch = ops.ExtendPartsScratch(DEV)
for it in range(N):
    D = 128
    Hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 4, 8])
    Hq = Hkv * group
    B = rng.choice([1, 1, 1, 2, 3])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    pre = [rng.choice([0, 100, 700, 1300, 3000, 4500, 9000]) for _ in range(B)]
    ext = [rng.choice([1, 17, 32, 33, 64, 100, 128, 200]) for _ in range(B)]
    causal = rng.random() < 0.8
    if os.environ.get("TRACE"):
        print("CASE", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, dtype=str(dtype), pre=pre, ext=ext, causal=causal), flush=True)
    g = torch.Generator(device=DEV).manual_seed(it)
    total_pre, total_ext = sum(pre), sum(ext)
    rows = total_pre + total_ext + 7
    kb = torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype)
    vb = torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype)
    q = torch.randn(total_ext, Hq, D, device=DEV, generator=g).to(dtype)
    k_e = torch.randn(total_ext, Hkv, D, device=DEV, generator=g).to(dtype)
    v_e = torch.randn(total_ext, Hkv, D, device=DEV, generator=g).to(dtype)
    perm = (torch.randperm(rows - 1, device=DEV, generator=g) + 1).to(torch.int32)
    slots, off = [], 0
    for b in range(B):
        slots.append(perm[off:off + pre[b]])
        off += pre[b]
    kv_indices = torch.cat(slots) if total_pre else torch.zeros(1, dtype=torch.int32, device=DEV)
    kv_indptr = torch.tensor([0] + [sum(pre[:b + 1]) for b in range(B)], dtype=torch.int32, device=DEV)
    qo_indptr = torch.tensor([0] + [sum(ext[:b + 1]) for b in range(B)], dtype=torch.int32, device=DEV)
    scale = D ** -0.5
    outs = []
    scratch.workspace.fill_(float("nan"))
    hint = max(pre) + rng.choice([0, 0, 500])  # (a loose bound changes the split, so one bound per configuration)
    try:
        for rep in range(2):
            o = torch.full((total_ext, Hq, D), 7.0, dtype=dtype, device=DEV)
            ops.extend_attention_fwd(q, k_e, v_e, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, causal, None, max(ext), scale,
                                     0.0, max_prefix_len=hint, parts_scratch=scratch)
            outs.append(o)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("EXC", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, pre=pre, ext=ext), repr(e)[:200])
        bad += 1
        continue
    ran += 1
    used += bool(torch.isfinite(scratch.workspace).any())
    o = outs[0]
    ref = torch.zeros(total_ext, Hq, D, device=DEV)
    for b in range(B):
        s0, s1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
        k = torch.cat([kb[slots[b].long()], k_e[s0:s1]]).float()
        v = torch.cat([vb[slots[b].long()], v_e[s0:s1]]).float()
        n = pre[b] + ext[b]
        qq = q[s0:s1].float().view(ext[b], Hkv, group, D)
        s = torch.einsum("thgd,nhd->thgn", qq, k) * scale
        if causal:
            pos_q = torch.arange(pre[b], n, device=DEV).view(-1, 1, 1, 1)
            pos_k = torch.arange(n, device=DEV).view(1, 1, 1, -1)
            s = s.masked_fill(pos_k > pos_q, float("-inf"))
        p = torch.softmax(s, dim=-1)
        ref[s0:s1] = torch.einsum("thgn,nhd->thgd", p, v).reshape(ext[b], Hq, D)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    err = (o.float() - ref).abs()
    tol = 3e-3 + 4 * ulp * ref.abs() + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 3.0
    ok = bool((err <= tol).all()) and bool(torch.isfinite(o.float()).all())
    same = torch.equal(outs[0], outs[1])
    zero = int(scratch.counters.abs().sum()) == 0
    if not (ok and same and zero):
        print("MISMATCH", dict(it=it, B=B, Hq=Hq, Hkv=Hkv, dtype=str(dtype), pre=pre, ext=ext, causal=causal), "max err", float(err.max()),
              "same", same, "counters zero", zero)
        bad += 1
print(f"configs {ran} (parts form taken: {used}) bad {bad}")
sys.exit(1 if bad else 0)
