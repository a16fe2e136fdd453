"""GPU: merge_state against the oracle and against the defining property: merging the attention over two disjoint key
sets equals the attention over their union (what cascade / chunked-prefix prefill relies on; reference test
sgl-kernel/tests/test_merge_state_v2.py compares against the same torch formula)."""
import pytest
import torch

import oracle
from sglang_npu_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("N,H,D", [(1, 1, 32), (37, 8, 128), (512, 32, 128), (9, 5, 80), (3, 2, 576)])
def test_merge_state_vs_oracle(dtype, N, H, D):
    g = torch.Generator().manual_seed(N + H + D)
    po, so = torch.randn(N, H, D, generator=g).to(dtype), torch.randn(N, H, D, generator=g).to(dtype)
    pl, sl = torch.randn(N, H, generator=g) * 3, torch.randn(N, H, generator=g) * 3
    pl[0, 0] = float("inf")      # merge_state.py:29-30: +inf is an empty partial
    if N > 2:
        sl[2, 0] = float("-inf")
    ref, ref_lse = oracle.merge_state(po, pl, so, sl)
    out, out_lse = ops.merge_state(po.to(DEV), pl.to(DEV), so.to(DEV), sl.to(DEV))
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 2.0 ** -21}[dtype]
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=ulp, atol=ulp * 4)  # expf/logf differ by an ulp
    torch.testing.assert_close(out_lse.cpu(), ref_lse, rtol=1e-5, atol=1e-5)


def test_merge_state_of_two_halves_equals_attention_over_the_union():
    g = torch.Generator(device=DEV).manual_seed(0)
    N, H, D, S = 16, 4, 64, 96
    q = torch.randn(N, H, D, device=DEV, generator=g)
    k = torch.randn(S, H, D, device=DEV, generator=g)
    v = torch.randn(S, H, D, device=DEV, generator=g)

    def attn(ks, vs):
        s = torch.einsum("nhd,shd->nhs", q, ks) * D ** -0.5
        return torch.einsum("nhs,shd->nhd", torch.softmax(s, -1), vs), torch.logsumexp(s, -1)

    oa, la = attn(k[:40], v[:40])
    ob, lb = attn(k[40:], v[40:])
    full, lfull = attn(k, v)
    out, lse = ops.merge_state(oa.contiguous(), la.contiguous(), ob.contiguous(), lb.contiguous())
    torch.testing.assert_close(out, full, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse, lfull, rtol=1e-5, atol=1e-5)
    # in-place outputs, as the reference allows
    o2, l2 = torch.empty(oa.shape, device=DEV), torch.empty(la.shape, device=DEV)
    r = ops.merge_state(oa.contiguous(), la.contiguous(), ob.contiguous(), lb.contiguous(), o2, l2)
    assert r[0] is o2 and r[1] is l2 and torch.equal(o2, out)


def test_merge_state_vs_reference_fixture():
    """tests/golden/merge_state.npz: merge_state_torch of the reference's sgl-kernel/tests/test_merge_state_v2.py."""
    import numpy as np
    z = np.load("tests/golden/merge_state.npz")
    for i in range(int(z["n"])):
        dtype = z[f"dtype{i}"].item().decode()
        dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[dtype]

        def t(key):
            a = z[key]
            return torch.from_numpy(a.copy()) if dtype == "f32" else torch.from_numpy(a.view(np.int16).copy()).view(dt)

        out, lse = ops.merge_state(t(f"po{i}").to(DEV), torch.from_numpy(z[f"pl{i}"].copy()).to(DEV),
                                   t(f"so{i}").to(DEV), torch.from_numpy(z[f"sl{i}"].copy()).to(DEV))
        torch.testing.assert_close(lse.cpu(), torch.from_numpy(z[f"lse{i}"]), rtol=1e-5, atol=1e-5)
        if dtype == "f32":
            torch.testing.assert_close(out.cpu(), torch.from_numpy(z[f"o_f32_{i}"]), rtol=1e-5, atol=1e-5)
        else:
            d = (out.cpu().view(torch.int16).int() - t(f"o{i}").view(torch.int16).int()).abs()
            assert int(d.max()) <= 1 and (d > 0).float().mean().item() < 5e-3
