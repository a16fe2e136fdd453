"""CPU: the ordering the P2P all-reduce depends on, checked in the ISA (deterministic counterpart of the GPU stress test
`test_staging_stores_are_ordered_before_the_flags_with_the_last_wave_delayed`).

Round 4's defect (commit 195e7bb): a workgroup's staging stores were only ordered before the flag barrier by
`__syncthreads()`, which orders execution, not the completion of other waves' stores -- a peer once read a stale 16-byte
vector.  The fix is `s_waitcnt vmcnt(0)` in every wave directly in front of the workgroup barrier of block_barrier()
(csrc/allreduce.hip).  On one GPU the window cannot be forced open (a variant built without the wait passes the stress test:
profiles/r05_allreduce_race_regression.txt), so the guard that cannot go stale is this one: compile the file for gfx950,
and require in EVERY instantiation of the three kernel families that each inlined block_barrier still has the wait glued to
its `s_barrier` -- and show that the same check flags the variant built with -DSGLM_AR_NO_STAGING_WAIT=1."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SRC = os.path.join(ROOT, "sglang_npu_amd", "csrc", "allreduce.hip")
FAMILIES = ("all_reduce_kernel", "quick_reduce_kernel", "ar_add_rmsnorm_kernel")


def _kernels(extra_flags, tmp_path, tag):
    out = os.path.join(str(tmp_path), f"allreduce_{tag}.s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-ffp-contract=fast-honor-pragmas",
           "-I", os.path.join(ROOT, "include"), SRC, "-o", out] + extra_flags
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    text = open(out).read()
    # one chunk per kernel: from its label to its .Lfunc_end
    chunks = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if any(f in name for f in FAMILIES):
            chunks[name] = body
    return chunks


def _guarded_barriers(body):
    """`s_barrier`s that have the explicit `s_waitcnt vmcnt(0)` (the inline-asm statement of block_barrier) within the four
    instructions in front of them."""
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    n = 0
    for i, l in enumerate(lines):
        if l.startswith("s_barrier"):
            if any(x.startswith("s_waitcnt vmcnt(0)") for x in lines[max(0, i - 4):i]):
                n += 1
    return n


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_every_flag_barrier_waits_for_the_staging_stores(tmp_path):
    good = _kernels([], tmp_path, "default")
    assert len(good) >= 3 + 3 * 2 + 4, sorted(good)  # plain x 3 dtypes, QuickReduce x 2 dtypes x 3 regimes, fused-norm forms
    for name, body in good.items():
        # every family calls block_barrier twice (flag barrier 0 after phase A, flag barrier 1 after phase B)
        assert _guarded_barriers(body) >= 2, f"{name}: a flag barrier without `s_waitcnt vmcnt(0)` in front of its s_barrier"
    bad = _kernels(["-DSGLM_AR_NO_STAGING_WAIT=1"], tmp_path, "nowait")
    assert set(bad) == set(good)
    assert all(_guarded_barriers(body) == 0 for body in bad.values()), "the check does not see the defect it is there for"
