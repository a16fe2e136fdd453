"""GPU: bounded randomised configurations of the two attention kernels against an fp32 evaluation on the GPU, under the
per-element bound of tests/conftest.py (1e-3 + ulp |ref_i| + the derived P-rounding term, here computed by the same fp32
evaluation on |V|).  The fixed seeds make every run the same 48 cases; tools/exp/fuzz_extend.py / fuzz_decode.py are the
open-ended versions (those were not part of `pytest -m gpu`)."""
import random

import pytest
import torch

from conftest import assert_elem_close, p_rounding_term
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pool(rows, Hkv, D, dtype, g):
    return (torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype), torch.randn(rows, Hkv, D, device=DEV, generator=g).to(dtype))


def _attend(q, k, v, scale, mask):
    """q [T, Hkv, G, D] fp32, k / v [N, Hkv, D] fp32, mask [T, N] bool (True = visible) -> softmax(q k) v and the same on |v|."""
    s = torch.einsum("thgd,nhd->thgn", q, k) * scale
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("thgn,nhd->thgd", p, v), torch.einsum("thgn,nhd->thgd", p, v.abs())


@pytest.mark.parametrize("seed", range(24))
def test_extend_random_configuration(seed):
    rng = random.Random(1000 + seed)
    D = rng.choice([64, 128])
    Hkv = rng.choice([1, 2, 4, 8])
    group = rng.choice([1, 2, 4, 7, 8])
    Hq = Hkv * group
    B = rng.choice([1, 1, 2, 3, 5])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    pre = [rng.choice([0, 0, 1, 17, 64, 100, 500, 1300]) for _ in range(B)]
    ext = [rng.choice([1, 2, 31, 32, 33, 64, 65, 200, 513]) for _ in range(B)]
    g = torch.Generator(device=DEV).manual_seed(seed)
    rows = sum(pre) + sum(ext) + 7
    kb, vb = _pool(rows, Hkv, D, dtype, g)
    q = torch.randn(sum(ext), Hq, D, device=DEV, generator=g).to(dtype)
    perm = (torch.randperm(rows - 1, device=DEV, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(B, max(p + e for p, e in zip(pre, ext)) + 2, dtype=torch.int32, device=DEV)
    off, start = 0, [0]
    for b in range(B):
        n = pre[b] + ext[b]
        r2t[b, :n] = perm[off:off + n]
        off += n
        start.append(start[-1] + ext[b])
    k_e = torch.cat([kb[r2t[b, pre[b]:pre[b] + ext[b]].long()] for b in range(B)])
    v_e = torch.cat([vb[r2t[b, pre[b]:pre[b] + ext[b]].long()] for b in range(B)])
    o = torch.full((sum(ext), Hq, D), 7.0, dtype=dtype, device=DEV)
    seq = torch.tensor([p + e for p, e in zip(pre, ext)], dtype=torch.int64, device=DEV)
    ops.extend_attention(q, k_e, v_e, o, kb, vb, r2t, torch.arange(B, device=DEV), seq,
                         torch.tensor(ext, dtype=torch.int64, device=DEV), torch.tensor(start[:-1], dtype=torch.int64, device=DEV),
                         max(ext), D ** -0.5, 0.0)
    ref, absv = torch.zeros(sum(ext), Hq, D, device=DEV), torch.zeros(sum(ext), Hq, D, device=DEV)
    for b in range(B):
        n = pre[b] + ext[b]
        idx = r2t[b, :n].long()
        vis = torch.arange(n, device=DEV)[None, :] <= torch.arange(pre[b], n, device=DEV)[:, None]
        r, a = _attend(q[start[b]:start[b + 1]].float().view(ext[b], Hkv, group, D), kb[idx].float(), vb[idx].float(), D ** -0.5, vis)
        ref[start[b]:start[b + 1]], absv[start[b]:start[b + 1]] = r.reshape(ext[b], Hq, D), a.reshape(ext[b], Hq, D)
    assert_elem_close(o, ref, dtype, what=f"extend B={B} Hq={Hq} Hkv={Hkv} D={D} pre={pre} ext={ext}",
                      extra=p_rounding_term(dtype, absv))


@pytest.mark.parametrize("seed", range(24))
def test_decode_random_configuration(seed):
    rng = random.Random(2000 + seed)
    D = rng.choice([64, 128])
    Hkv = rng.choice([1, 2, 4, 8, 32])
    group = rng.choice([1, 2, 4, 5, 8, 16]) if Hkv < 32 else 1
    Hq = Hkv * group
    B = rng.choice([1, 3, 8, 33, 64, 100])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    lens = [rng.choice([0, 1, 2, 31, 32, 33, 100, 700, 2049, 4100]) for _ in range(B)]
    if sum(lens) * Hkv * D > 80_000_000:
        lens = [min(l, 700) for l in lens]
    splits = rng.choice([1, 1, 2, 4, 8])
    g = torch.Generator(device=DEV).manual_seed(seed)
    rows = sum(lens) + 9
    kb, vb = _pool(rows, Hkv, D, dtype, g)
    q = torch.randn(B, Hq, D, device=DEV, generator=g).to(dtype)
    perm = (torch.randperm(rows - 1, device=DEV, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(B, max(max(lens), 1), dtype=torch.int32, device=DEV)
    off = 0
    for b in range(B):
        r2t[b, :lens[b]] = perm[off:off + lens[b]]
        off += lens[b]
    o = torch.full((B, Hq, D), 7.0, dtype=dtype, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int64, device=DEV)
    ops.decode_attention(q, kb, vb, o, None, None, None, torch.zeros(B, Hq, splits, D + 1, device=DEV), r2t,
                         torch.arange(B, device=DEV), seq, D ** -0.5, 0.0)
    ref, absv = torch.zeros(B, Hq, D, device=DEV), torch.zeros(B, Hq, D, device=DEV)
    for b in range(B):
        if lens[b] == 0:
            continue  # (no key: the op writes zeros)
        idx = r2t[b, :lens[b]].long()
        r, a = _attend(q[b:b + 1].float().view(1, Hkv, group, D), kb[idx].float(), vb[idx].float(), D ** -0.5,
                       torch.ones(1, lens[b], dtype=torch.bool, device=DEV))
        ref[b], absv[b] = r.reshape(Hq, D), a.reshape(Hq, D)
    assert_elem_close(o, ref, dtype, what=f"decode B={B} Hq={Hq} Hkv={Hkv} D={D} splits={splits}",
                      extra=p_rounding_term(dtype, absv))
