"""CPU: the N>1 path (TP group, all-reduce / all-gather, row/column-parallel linears) on world_size-2 gloo."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sglang_npu_amd import distributed as D
    from sglang_npu_amd.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    try:
        tp = D.init_distributed_environment(backend="gloo")
        assert tp.world_size == world and D.get_tensor_model_parallel_rank() == rank
        # integer-valued payloads: the sum is exact whatever the algorithm (test_custom_allreduce.py:118)
        for n in (1, 128, 4097, 1 << 16):
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                x = torch.randint(0, 4, (n,), generator=torch.Generator().manual_seed(rank)).to(dt)
                ref = sum(torch.randint(0, 4, (n,), generator=torch.Generator().manual_seed(r)).to(dt) for r in range(world))
                out = D.tensor_model_parallel_all_reduce(x.clone())
                assert torch.equal(out, ref), (n, dt)
                h = tp.all_reduce_async(x.clone())
                assert torch.equal(h.wait(), ref)
        # the measurement stub turns both forms into identities; the row-parallel layer's async form on CPU tensors is
        # the plain in-stream all-reduce (there is no side stream to fork to)
        tp.stub_all_reduce = True
        xs = torch.arange(4.0)
        assert tp.all_reduce(xs) is xs and tp.all_reduce_async(xs).wait() is xs
        tp.stub_all_reduce = False
        g = D.tensor_model_parallel_all_gather(torch.full((3, 2), float(rank)), dim=-1)
        assert g.shape == (3, 2 * world) and torch.equal(g[:, 2 * rank], torch.full((3,), float(rank)))
        # row-parallel (K split + all-reduce) o column-parallel (N split) == the unsharded product
        gen = torch.Generator().manual_seed(0)
        K, N = 64, 48
        w1, w2 = torch.randn(N, K, generator=gen), torch.randn(K, N, generator=gen)
        x = torch.randn(5, K, generator=gen)
        col = MergedColumnParallelLinear(K, [N // 2, N // 2], params_dtype=torch.float32)
        # the full checkpoint matrices go through the layers' weight loaders, which take this rank's slice
        col.weight.weight_loader(col.weight, w1[:N // 2], 0)
        col.weight.weight_loader(col.weight, w1[N // 2:], 1)
        half = N // 2 // world
        assert torch.equal(col.weight.data, torch.cat([w1[rank * half:(rank + 1) * half],
                                                       w1[N // 2 + rank * half: N // 2 + (rank + 1) * half]]))
        # the row-parallel layer consumes the column-parallel output: its K index runs over [gate shard | up shard] of
        # every rank in turn, so the "checkpoint" is w2 with its columns in that order
        perm = torch.cat([torch.cat([torch.arange(r * half, (r + 1) * half), N // 2 + torch.arange(r * half, (r + 1) * half)])
                          for r in range(world)])
        row = RowParallelLinear(N, K, params_dtype=torch.float32)
        row.weight.weight_loader(row.weight, w2[:, perm].contiguous())
        y, _ = col(x)
        z, _ = row(y)
        z2, _ = row(y, async_reduce=True)
        assert torch.equal(z2, z)
        ref = (x @ w1.t()) @ w2.t()
        torch.testing.assert_close(z, ref, rtol=1e-4, atol=1e-4)
        qkv = QKVParallelLinear(64, 16, 8, 1, params_dtype=torch.float32)  # fewer KV heads than ranks -> replicated
        assert qkv.num_heads == 8 // world and qkv.num_kv_heads == 1
        assert qkv.output_partition_sizes == [8 * 16 // world, 16, 16]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_tp2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _bringup_worker(rank, world, port, q):
    """a P2P bring-up that fails on ONE rank only must end with every rank on the fallback, not in a hang."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from sglang_npu_amd import distributed as D
    try:
        tp = D.init_distributed_environment(backend="gloo")
        grp = tp.device_group
        log = []

        # --- handle exchange: (a) all fine, (b) create fails on rank 1, (c) open_peers fails on rank 1
        def mk(create_fail=False, open_fail=False):
            def create():
                if create_fail and rank == 1:
                    raise RuntimeError("hipIpcGetMemHandle: invalid argument")
                return bytes([rank]) * 64
            def open_peers(blob):
                assert len(blob) == 64 * world and blob[64 * 1] == 1
                log.append("open")
                if open_fail and rank == 1:
                    raise RuntimeError("hipIpcOpenMemHandle failed")
            return create, open_peers
        assert D._exchange_ipc_handles(grp, world, *mk()) is None
        e = D._exchange_ipc_handles(grp, world, *mk(create_fail=True))
        assert e is not None and e.startswith("rank 1: create") and log == ["open"]  # nobody opened peers after the failure
        e = D._exchange_ipc_handles(grp, world, *mk(open_fail=True))
        assert e is not None and e.startswith("rank 1: open_peers")

        # --- verification in lockstep: step 2 fails on rank 1 (wrong result) / raises on rank 0; step 3 must not run anywhere
        ran = []
        def step(name, fail_rank=None, raises=False):
            def fn():
                t = torch.ones(8)
                dist.all_reduce(t, group=grp)  # the reference collective: issued by every rank
                ran.append(name)  # "the P2P kernel was launched"
                if fail_rank == rank:
                    if raises:
                        raise RuntimeError("kernel launch failed")
                    return False
                return True
            return (name, fn)
        assert D._run_lockstep(tp, [step("a"), step("b")]) is True
        ran.clear()
        assert D._run_lockstep(tp, [step("a"), step("b", fail_rank=1), step("c")]) is False and ran == ["a", "b"]
        ran.clear()
        assert D._run_lockstep(tp, [step("a", fail_rank=0, raises=True), step("b")]) is False and ran == ["a"]

        # --- the whole bring-up with a stand-in communicator: installed only when every rank agrees, closed otherwise
        class FakeComm:
            def __init__(self, disabled=False, init_error=None):
                self.disabled, self.init_error, self.closed = disabled, init_error, False
                self._comm, self.world_size, self.rank = None, world, rank
            def close(self):
                self.closed, self.disabled = True, True
        made = []
        def make(fail_rank=None, raises=False):
            def f():
                if raises and rank == fail_rank:
                    raise OSError("libsgl_mi355.so: cannot open shared object file")
                c = FakeComm(disabled=(rank == fail_rank), init_error="rank 1: create: boom" if fail_rank is not None else None)
                made.append(c)
                return c
            return f
        steps_ok = lambda ca, qr, tp_: [step("v1"), step("v2")]
        steps_bad = lambda ca, qr, tp_: [step("v1"), step("v2", fail_rank=1)]
        assert D._bring_up_custom_ar(tp, make(), steps_of=steps_ok) is True and tp.ca_comm is made[-1] and not made[-1].closed
        tp.ca_comm = tp.qr_comm = None
        assert D._bring_up_custom_ar(tp, make(), steps_of=steps_bad) is False and tp.ca_comm is None and made[-1].closed
        assert D._bring_up_custom_ar(tp, make(fail_rank=1), steps_of=steps_ok) is False and tp.ca_comm is None and made[-1].closed
        n = len(made)
        assert D._bring_up_custom_ar(tp, make(fail_rank=0, raises=True), steps_of=steps_ok) is False and tp.ca_comm is None
        assert len(made) == n + (1 if rank != 0 else 0) and (rank == 0 or made[-1].closed)
        # --- the library fails to load on ONE rank.  The constructor stands in for the real one: it starts with
        # an object all-gather (as _exchange_ipc_handles does); a rank that skipped it would leave its peer there for
        # ever -- the pre-flight agreement makes EVERY rank skip the constructor instead
        constructed = []
        def make_collective():
            constructed.append(rank)
            got = [None] * world
            dist.all_gather_object(got, rank, group=grp)
            return FakeComm()
        def preflight(fail_rank):
            def f():
                if rank == fail_rank:
                    raise OSError("libsgl_mi355.so: cannot open shared object file")
            return f
        assert D._bring_up_custom_ar(tp, make_collective, steps_of=steps_ok, preflight=preflight(1)) is False
        assert constructed == [] and tp.ca_comm is None
        assert D._bring_up_custom_ar(tp, make_collective, steps_of=steps_ok, preflight=preflight(None)) is True
        assert constructed == [rank]
        tp.ca_comm = tp.qr_comm = None
        # the group is still usable afterwards (no collective left unpaired)
        t = torch.full((4,), float(rank + 1))
        dist.all_reduce(t, group=grp)
        assert float(t[0]) == world * (world + 1) / 2
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_p2p_bring_up_fails_on_one_rank_without_hanging():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bringup_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_single_rank_all_reduce_is_identity():
    from sglang_npu_amd import distributed as D
    D.set_tp_group(D.GroupCoordinator(None, 0, 1, None))
    x = torch.arange(8.0)
    assert D.tensor_model_parallel_all_reduce(x) is x  # parallel_state.py:478-479
