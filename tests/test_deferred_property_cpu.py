"""CPU, property test: whatever untrusted code does with a lazy tensor of deferred.py, it sees exactly what it would have seen
on the finished tensor.  Random little programs of torch operations (views, slices, arithmetic, in-place writes, reductions,
splits, concatenations, dtype casts, indexing) run on (a) the plain tensor and (b) the lazy handle / its lazy column ranges, with a
rotation recorded on q / k or not; results and the final contents must be equal, and the producer's work must run exactly once."""
import torch
from hypothesis import given, settings, strategies as st

from sglang_npu_amd.deferred import DeferredCols, DeferredEpilogue

M, QS, KS = 6, 8, 4
N = QS + 2 * KS


class _Part:
    def __init__(self, value):
        self.M, self.N, self.out_dtype, self.ws = value.shape[0], value.shape[1], value.dtype, torch.zeros(1)
        self._value, self.finalized = value, 0

    def finalize(self):
        self.finalized += 1
        return self._value.clone()


class _Rot:
    """Stands in for RotaryEmbedding.forward: an in-place change of q and k that depends on positions."""
    def __init__(self):
        self.calls = 0

    def forward(self, positions, q, k):
        self.calls += 1
        q.mul_(positions.view(-1, 1).to(q.dtype) + 1)
        k.add_(positions.view(-1, 1).to(k.dtype))
        return q, k


OPS = {
    "add1": lambda t: t + 1,
    "mul_self": lambda t: t * t,
    "neg": lambda t: -t,
    "float": lambda t: t.float(),
    "view_flat": lambda t: t.reshape(-1),
    "transpose": lambda t: t.transpose(0, -1),
    "slice_rows": lambda t: t[1:4],
    "index": lambda t: t[torch.tensor([0, 2])],
    "sum": lambda t: t.sum(dim=-1),
    "clone": lambda t: t.clone(),
    "contig": lambda t: t.contiguous(),
    "cat": lambda t: torch.cat([t, t], dim=0),
    "inplace_add": lambda t: t.add_(2),
    "inplace_zero_row": lambda t: t[0].zero_() if t.dim() > 1 else t.zero_(),
    "to_list_len": lambda t: torch.tensor(len(t.tolist())),
    "unsqueeze": lambda t: t.unsqueeze(0),
    "max": lambda t: t.max(),
}
op_names = st.lists(st.sampled_from(sorted(OPS)), min_size=1, max_size=5)


def _run(program, t):
    outs = []
    for name in program:
        try:
            r = OPS[name](t)
        except (IndexError, RuntimeError) as e:  # (a program may be wrong for the shape it reached: then both sides must say so)
            outs.append(torch.tensor(hash(type(e).__name__) % 1000))
            break
        outs.append(r.clone() if isinstance(r, torch.Tensor) else r)
        if isinstance(r, torch.Tensor) and r.dim() >= 1 and r.numel() and name not in ("sum", "max", "to_list_len"):
            t = r
    return outs, t


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert type(x) is torch.Tensor and x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)


@settings(max_examples=400, deadline=None)
@given(program=op_names, kind=st.sampled_from(["partials", "local", "compute"]), seed=st.integers(0, 10 ** 6))
def test_any_program_on_the_root_sees_the_finished_gemm(program, kind, seed):
    base = torch.randn(M, N, generator=torch.Generator().manual_seed(seed)).round(decimals=2)
    part = _Part(base)
    if kind == "partials":
        lazy = DeferredEpilogue(part)
    elif kind == "local":
        lazy = DeferredEpilogue(local=base.clone())
    else:
        lazy = DeferredEpilogue(compute=part.finalize, like=((M, N), base.dtype, base.device))
    ref_outs, _ = _run(program, base.clone())
    got_outs, _ = _run(program, lazy)
    _same(got_outs, ref_outs)
    assert part.finalized == (0 if kind == "local" else 1), "the producer's work runs exactly once"


@settings(max_examples=600, deadline=None)
@given(program=op_names, who=st.sampled_from(["q", "k", "v", "k3", "v3", "root"]), rope=st.booleans(),
       kind=st.sampled_from(["partials", "local"]), seed=st.integers(0, 10 ** 6))
def test_any_program_on_a_column_range_sees_the_reference_sequence(program, who, rope, kind, seed):
    """qkv.split -> (rotary_emb) -> (RadixAttention's views) -> somebody other than the attention backend reads one of them."""
    base = torch.randn(M, N, generator=torch.Generator().manual_seed(seed)).round(decimals=2)
    positions = torch.arange(M)
    # the reference: real tensors, rotation applied in place at once
    ref = base.clone()
    rq, rk, rv = ref.split([QS, KS, KS], dim=-1)
    ref_rot = _Rot()
    if rope:
        ref_rot.forward(positions, rq, rk)
    ref_t = {"q": rq, "k": rk, "v": rv, "k3": rk.view(-1, 2, KS // 2), "v3": rv.view(-1, 2, KS // 2), "root": ref}[who]
    # the lazy chain
    part = _Part(base)
    root = DeferredEpilogue(part) if kind == "partials" else DeferredEpilogue(local=base.clone())
    q, k, v = root.split([QS, KS, KS], dim=-1)
    rot = _Rot()
    if rope:  # what layers.RotaryEmbedding.forward does with two lazy column ranges: record, return them
        root._rope = (positions, rot, (q._c0, q._c1), (k._c0, k._c1))
    lazy_t = {"q": q, "k": k, "v": v, "k3": k.view(-1, 2, KS // 2), "v3": v.view(-1, 2, KS // 2), "root": root}[who]
    assert isinstance(lazy_t, (DeferredCols, DeferredEpilogue)) and rot.calls == 0
    ref_outs, _ = _run(program, ref_t)
    got_outs, _ = _run(program, lazy_t)
    _same(got_outs, ref_outs)
    assert rot.calls == (1 if rope else 0), "the recorded rotation is applied exactly once"
    # and every other handle of the same projection now shows the same memory the reference holds (in-place writes included)
    assert torch.equal(root.materialize(), ref) and torch.equal(q + 0, rq) and torch.equal(v.view(-1, 2, KS // 2) + 0, rv.view(-1, 2, KS // 2))
