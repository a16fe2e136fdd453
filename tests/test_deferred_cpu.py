"""CPU: the mechanics of deferred.DeferredEpilogue (a row-parallel GEMM's output handed through model code with its epilogue
still to run) on a stand-in for ops.GemmPartials: metadata without storage, any foreign operation finishes it first, the
consumer's resolve(), and the workspace pool finishing a pending tensor before the buffer is handed out again."""
import gc

import torch

from sglang_npu_amd import ops
from sglang_npu_amd.deferred import DeferredEpilogue, materialize


class FakePartials:
    def __init__(self, M=4, N=8, dtype=torch.bfloat16):
        self.M, self.N, self.out_dtype, self.ws = M, N, dtype, torch.zeros(1)
        self.finalized = 0

    def finalize(self):
        self.finalized += 1
        return torch.arange(self.M * self.N, dtype=torch.float32).view(self.M, self.N).to(self.out_dtype)


def test_metadata_is_real_and_nothing_runs_for_it():
    p = FakePartials()
    d = DeferredEpilogue(p)
    assert isinstance(d, torch.Tensor) and tuple(d.shape) == (4, 8) and d.dtype == torch.bfloat16 and d.dim() == 2
    assert d.is_contiguous() and d.stride() == (8, 1) and d.device.type == "cpu" and not d.is_cuda
    assert "pending" in repr(d) and getattr(d, "_sglang_needs_allreduce_fusion", False) is False
    d._tag = 5  # (python attributes travel with it like with any tensor: the all-reduce-fusion tag, the producer tags)
    assert d._tag == 5 and p.finalized == 0 and d.pending_partials() is p


def test_any_foreign_operation_finishes_it_first_and_only_once():
    p = FakePartials()
    d = DeferredEpilogue(p)
    want = p.finalize()
    p.finalized = 0
    y = d + 1
    assert type(y) is torch.Tensor and torch.equal(y, want + 1) and p.finalized == 1
    assert torch.equal(d.view(-1), want.view(-1)) and torch.equal(torch.cat([d, d]), torch.cat([want, want]))
    a, b = d.split([3, 5], dim=-1)
    assert type(a) is torch.Tensor and torch.equal(b, want[:, 3:]) and torch.equal(d[torch.tensor([1, 3])], want[[1, 3]])
    assert p.finalized == 1 and d.pending_partials() is None and "resolved" in repr(d)
    assert materialize(d) is d.materialize() and materialize(want) is want


def test_consumer_resolve_gives_the_in_place_semantics():
    """fused_add_rmsnorm overwrites its input: after the norm consumed the partials, the handle holds the normed row."""
    p = FakePartials()
    d = DeferredEpilogue(p)
    normed = torch.ones(4, 8, dtype=torch.bfloat16)
    assert d.pending_partials() is p
    d.resolve(normed)
    assert d.pending_partials() is None and p.finalized == 0 and torch.equal(d * 2, normed * 2)


def test_ops_ptr_finishes_a_deferred_tensor_instead_of_passing_null():
    p = FakePartials()
    d = DeferredEpilogue(p)
    ptr = ops._ptr(d)
    assert p.finalized == 1 and ptr.value == d.materialize().data_ptr() and ptr.value != 0
    # ... and so does anybody else who asks for the address or the values without going through the dispatcher
    p2 = FakePartials()
    d2 = DeferredEpilogue(p2)
    assert d2.data_ptr() == d2.materialize().data_ptr() != 0 and p2.finalized == 1
    assert d2.tolist() == d2.materialize().tolist() and (d2.float().numpy() == d2.materialize().float().numpy()).all()


def test_workspace_pool_finishes_the_pending_tensor_before_the_buffer_is_reused(monkeypatch):
    monkeypatch.setattr(ops, "_stream_handle", lambda device: 0)
    pool = ops._ScratchPool(16)
    dev = torch.device("cpu")
    buf = pool.get(dev, 8)
    p = FakePartials()
    p.ws = buf
    d = DeferredEpilogue(p)
    pool.set_pending(dev, d)
    assert p.finalized == 0
    assert pool.get(dev, 8) is buf and p.finalized == 1 and d.pending_partials() is None  # finished, THEN handed out
    assert pool.get(dev, 8) is buf and p.finalized == 1
    # consumed in time: nothing left to do, and a collected tensor is simply forgotten
    p2 = FakePartials()
    d2 = DeferredEpilogue(p2)
    pool.set_pending(dev, d2)
    d2.resolve(torch.zeros(4, 8, dtype=torch.bfloat16))
    assert not pool._pending and pool.get(dev, 8) is buf and p2.finalized == 0
    p3 = FakePartials()
    d3 = DeferredEpilogue(p3)
    pool.set_pending(dev, d3)
    del d3
    gc.collect()
    assert pool.get(dev, 8) is buf and p3.finalized == 0


def test_split_and_view_stay_lazy_on_a_finished_local_tensor_too():
    """The qkv projection outside the split-K window: a finished tensor behind a handle (local=...), same lazy split / view."""
    from sglang_npu_amd.deferred import DeferredCols, qkv_root, rope_target
    local = torch.arange(4 * 16, dtype=torch.float32).view(4, 16).clone()
    d = DeferredEpilogue(local=local)
    q, k, v = d.split([8, 4, 4], dim=-1)
    assert all(isinstance(t, DeferredCols) for t in (q, k, v)) and q.stride() == (16, 1) and d.pending_local() is local
    assert rope_target(q, k) is d

    class Rot:
        def forward(self, pos, q_, k_):
            q_.mul_(-1)
            k_.add_(1000)
            return q_, k_
    d._rope = (None, Rot(), (0, 8), (8, 12))
    k3, v3 = k.view(-1, 2, 2), v.view(-1, 2, 2)
    assert isinstance(k3, DeferredCols) and k3.stride() == (16, 2, 1) and qkv_root(q, k3, v3, 8, 4) is d
    assert qkv_root(q, k3, v3, 8, 8) is None
    got = v3 + 0          # somebody looks: the recorded rotation is applied, in place, by the module that recorded it
    assert torch.equal(got, local[:, 12:].view(4, 2, 2)) and torch.equal(q + 0, -torch.arange(64.).view(4, 16)[:, :8])
    assert torch.equal((k3 + 0).reshape(4, 4), torch.arange(64.).view(4, 16)[:, 8:12] + 1000)
    assert d.materialize() is local
