"""GPU: the KV-range-parts form of the extend kernel (sgl_mi355_extend_attention_fwd_parts, round 4): launches of few, long
items -- one short request behind a long cached prefix (chunked prefill's later chunks, a radix-cache hit with a short suffix), the
heaviest query blocks of a single 1024-token prefill -- cut every item's keys into ranges over several workgroups and merge
them in range order.  Same contract as extend_attention_fwd (extend_attention.py:306-438); checked against the oracle, the
unsplit launch, for determinism, and for the state it leaves behind."""
import pytest
import torch

import oracle
from conftest import assert_elem_close
from sglang_npu_amd import ops
from test_extend_gpu import _case, _p_term, _triton_meta

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(d, B, meta, max_ext, causal, scratch, max_prefix):
    qo_indptr, kv_indptr, kv_indices = meta
    T, Hq, D = d["q"].shape
    o = torch.zeros(T, Hq, D, dtype=d["q"].dtype, device=DEV)
    ops.extend_attention_fwd(d["q"], d["ke"], d["ve"], o, d["kb"], d["vb"], qo_indptr, kv_indptr, kv_indices, None, causal,
                             None, max_ext, D ** -0.5, 0.0, max_prefix_len=max_prefix, parts_scratch=scratch)
    return o


# (B, Hq, Hkv, longest prefix, longest extend, dtype, parts expected): at most 128 items of 12 tiles or more (256 from 48 tiles)
# take the parts form
CASES = [(1, 32, 8, 4096, 128, torch.bfloat16, True),    # a short suffix behind a long cached prefix: 64 items x 4 parts
         (1, 32, 8, 1000, 100, torch.bfloat16, True),    # ragged everything
         (1, 8, 1, 3000, 200, torch.bfloat16, True),     # one kv head (a TP rank): group 8
         (1, 32, 32, 1500, 90, torch.bfloat16, True),    # MHA: one head per workgroup, two position blocks
         (1, 16, 8, 5000, 64, torch.float16, True),      # group 2; a prefix of more than one page-table pass (4096 entries)
         (2, 32, 8, 9000, 33, torch.bfloat16, True),     # two requests, one query block each, three passes
         (2, 32, 8, 4096, 256, torch.bfloat16, True),    # 256 items of 68 tiles: two parts each
         # ... and what stays unsplit: the TTFT shape (512 items), several requests, short chains
         (1, 32, 8, 0, 1024, torch.bfloat16, False), (1, 32, 8, 0, 1000, torch.float16, False),
         (3, 32, 8, 2500, 130, torch.bfloat16, False), (8, 8, 1, 700, 300, torch.bfloat16, False),
         (6, 32, 8, 1200, 128, torch.bfloat16, False), (1, 32, 8, 500, 128, torch.bfloat16, False),
         (4, 32, 8, 1024, 128, torch.bfloat16, False)]   # 256 items of 18 tiles: chains too short for two parts


@pytest.mark.parametrize("B,Hq,Hkv,max_prefix,max_ext,dtype,expect_parts", CASES)
@pytest.mark.parametrize("causal", [True, False])
def test_parts_match_the_oracle_and_the_unsplit_launch(B, Hq, Hkv, max_prefix, max_ext, dtype, expect_parts, causal):
    D = 128
    c = _case(B, Hq, Hkv, D, max_prefix, max_ext, dtype, seed=B + Hq + max_prefix + max_ext, zero_prefix=max_prefix == 0,
              pin_first_prefix=True)
    T = c["q"].size(0)
    o_ref = torch.zeros(T, Hq, D, dtype=dtype)
    oracle.extend_attention(c["q"], c["ke"], c["ve"], o_ref, c["kb"], c["vb"], c["r2t"], c["rpi"], c["seq"], c["ext"],
                            c["start"], int(c["ext"].max()), D ** -0.5, 0.0, causal=causal)
    d = {k: v.to(DEV) for k, v in c.items()}
    meta = _triton_meta(d, B)
    scratch = ops.ExtendPartsScratch(DEV)
    scratch.workspace.fill_(float("nan"))
    hint = int(c["prefix"].max())
    plain = _run(d, B, meta, int(c["ext"].max()), causal, None, None)
    parts = _run(d, B, meta, int(c["ext"].max()), causal, scratch, hint)
    term = _p_term(c, dtype, D ** -0.5, causal=causal)  # per-element P-rounding allowance (conftest.p_rounding_term)
    assert_elem_close(plain, o_ref, dtype, pair=True, what="unsplit launch vs oracle", extra=term)
    assert_elem_close(parts, o_ref, dtype, pair=True, what="KV-range parts vs oracle", extra=term)
    # the parts form really ran (the workspace was written) wherever the plan says it should
    assert bool(torch.isfinite(scratch.workspace).any()) == expect_parts
    # deterministic (the merge runs in range order whichever workgroup finishes last), and the counters are zero again
    again = _run(d, B, meta, int(c["ext"].max()), causal, scratch, hint)
    assert torch.equal(again, parts)
    assert int(scratch.counters.abs().sum()) == 0
    # a bound that overstates the prefix only changes the split, not the result beyond rounding
    loose = _run(d, B, meta, int(c["ext"].max()), causal, scratch, hint + 3000)
    assert_elem_close(loose, o_ref, dtype, pair=True, what="parts planned from a loose bound vs oracle", extra=term)


def test_parts_fall_back_where_the_form_does_not_apply():
    """FP8 pools, custom masks, a scratch that is too small, many items: the call is the plain extend_attention_fwd."""
    B, Hq, Hkv, D, dtype = 1, 32, 8, 128, torch.bfloat16
    c = _case(B, Hq, Hkv, D, 2000, 128, dtype, seed=5, pin_first_prefix=True)
    d = {k: v.to(DEV) for k, v in c.items()}
    meta = _triton_meta(d, B)
    plain = _run(d, B, meta, 128, True, None, None)
    small = ops.ExtendPartsScratch(DEV, megabytes=1)
    small.workspace.fill_(float("nan"))
    out = _run(d, B, meta, 128, True, small, int(c["prefix"].max()))
    assert torch.equal(out, plain) and not bool(torch.isfinite(small.workspace).any())
    # more than 512 items (20 requests x 8 head groups x 4 query blocks): the chip is full without parts
    c20 = _case(20, Hq, Hkv, D, 1500, 128, dtype, seed=6)
    d20 = {k: v.to(DEV) for k, v in c20.items()}
    meta20 = _triton_meta(d20, 20)
    big = ops.ExtendPartsScratch(DEV)
    big.workspace.fill_(float("nan"))
    out20 = _run(d20, 20, meta20, int(c20["ext"].max()), True, big, int(c20["prefix"].max()))
    assert torch.equal(out20, _run(d20, 20, meta20, int(c20["ext"].max()), True, None, None))
    assert not bool(torch.isfinite(big.workspace).any())
