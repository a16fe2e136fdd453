"""GPU: the P2P all-reduce kernel with 2 and 4 ranks sharing the ONE GPU of the box (IPC handles between
processes, same protocol as across GPUs; RCCL itself refuses duplicate devices so gloo carries the handle
exchange).  Integer-valued payloads: the result must be exact (test_custom_allreduce.py:118-146).
The 6- and 8-rank protocol runs with all ranks inside this one process (the box admits at most six processes on its
GPU): the ranks' communicators are wired by pointer (CustomAllreduce.connect_local) and their kernels run concurrently
on one stream each.  A missing peer must fail closed: NaN output, sticky status, the next call raises."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
    try:
        import torch.distributed as dist
        from sglang_npu_amd.distributed import CustomAllreduce, GroupCoordinator
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ca = CustomAllreduce(dist.group.WORLD, dev, max_size=8 * 1024 * 1024)
        assert not ca.disabled
        tp = GroupCoordinator(dist.group.WORLD, rank, world, dev)
        tp.ca_comm = ca
        # sizes 512 B .. 8 MiB (one-shot and two-shot), three dtypes, many back-to-back calls (double buffering)
        for dt in (torch.float32, torch.bfloat16):
            for nbytes in (512, 65536, 262144 + 16, 1 << 20):
                n = nbytes // torch.tensor([], dtype=dt).element_size()
                for it in range(2):
                    g = torch.Generator().manual_seed(1000 * it + n % 997)
                    parts = [torch.randint(-3, 4, (n,), generator=g).to(dt) for _ in range(world)]
                    ref = sum(p.float() for p in parts).to(dt)
                    out = tp.all_reduce(parts[rank].to(dev))
                    assert out.shape == (n,) and torch.equal(out.cpu(), ref), (dt, nbytes, it)
        # not eligible -> returns None -> coordinator falls back to the process-group all-reduce
        assert ca.custom_all_reduce(torch.zeros(3, device=dev)) is None
        # graph capture + replay
        x = torch.ones(1 << 16, device=dev, dtype=torch.bfloat16) * (rank + 1)
        y = torch.empty_like(x)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            y.copy_(ca.custom_all_reduce(x))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y.copy_(ca.custom_all_reduce(x))
        for k in range(3):
            x.fill_(float(rank + 1 + k))
            dist.barrier()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.all(y == float(sum(r + 1 + k for r in range(world)))), k
        # side-stream form (GroupCoordinator.all_reduce_async), eager and inside a captured graph: fork at the call,
        # join at wait(); the result must be the same tensor values as the in-stream call
        x2 = (torch.arange(1 << 14, device=dev, dtype=torch.float32) % 7 + rank).to(torch.bfloat16)
        want = sum(((torch.arange(1 << 14, dtype=torch.float32) % 7 + r).to(torch.bfloat16)).float() for r in range(world))
        h = tp.all_reduce_async(x2)
        assert torch.equal(h.wait().float().cpu(), want)
        y2 = torch.empty_like(x2)
        dist.barrier()
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            hh = tp.all_reduce_async(x2)
            z = x2 * 2           # main-stream work between the fork and the join
            y2.copy_(hh.wait())
            y2.add_(z)
        for k in range(2):
            dist.barrier()
            graph2.replay()
            torch.cuda.synchronize()
            assert torch.equal(y2.float().cpu(), (want + 2 * x2.float().cpu()).to(torch.bfloat16).float()), k
        tp.stub_all_reduce = True   # measurement stub: identity, also for the async form
        assert tp.all_reduce_async(x2).wait() is x2
        tp.stub_all_reduce = False
        assert not ca.timed_out()
        dist.barrier()
        ca.close()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.timeout(300)
def test_p2p_all_reduce_shared_gpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _local_ranks_worker(world, q):
    # one hardware queue per rank's stream: streams that share a queue would run their kernels one after the other
    # and the ranks could never meet at the flag barriers (read by the HIP runtime when it initialises)
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from sglang_npu_amd.distributed import CustomAllreduce
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        comms = CustomAllreduce.connect_local(world, dev, max_size=4 * 1024 * 1024)
        streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
        for dt in (torch.bfloat16, torch.float32):
            for nbytes in (512, 65536, 262144 + 16 * world, 1 << 20, (1 << 21) + 4096):   # one-shot and two-shot
                n = nbytes // torch.tensor([], dtype=dt).element_size()
                g = torch.Generator().manual_seed(n % 991)
                parts = [torch.randint(-3, 4, (n,), generator=g).to(dt).to(dev) for _ in range(world)]
                ref = sum(p.float() for p in parts).to(dt)
                for rep in range(2):             # both halves of the double buffer
                    torch.cuda.synchronize()
                    outs = []
                    for r in range(world):       # every rank's kernel on its own stream: they meet at the flag barriers
                        with torch.cuda.stream(streams[r]):
                            outs.append(comms[r].custom_all_reduce(parts[r]))
                    torch.cuda.synchronize()
                    for r in range(world):
                        assert torch.equal(outs[r], ref), (dt, nbytes, r, rep)
        # all-reduce + residual add + RMSNorm (+ FP8 quant) in one kernel == the three separate ops, bit for bit, on
        # every rank; one-shot (small) and two-shot (column-sliced) forms, several calls in a row (double buffering)
        from sglang_npu_amd import ops
        for dt in (torch.bfloat16, torch.float16):
            for (T, H) in [(4, 1024), (64, 8192), (7, 4096), (130, 2048)]:
                if H % (8 * world):
                    continue
                g = torch.Generator().manual_seed(T * H + world)
                parts = [(torch.randn(T, H, generator=g) * 0.5).to(dt).to(dev) for _ in range(world)]
                # the fused-norm staging area is bound to ONE row length per communicator (row b <-> block b): another H
                # is refused until the binding is dropped, which is legal here -- nothing is in flight on any rank
                torch.cuda.synchronize()
                for c in comms:
                    if c._norm_h not in (None, H):
                        assert not c.should_fuse_norm(parts[0])
                    c.rebind_fused_norm()
                res0 = torch.randn(T, H, generator=g).to(dt).to(dev)
                w = (torch.rand(H, generator=g) + 0.5).to(dt).to(dev)
                for quant in (False, True):
                    torch.cuda.synchronize()
                    ars = []
                    for r in range(world):
                        with torch.cuda.stream(streams[r]):
                            ars.append(comms[r].custom_all_reduce(parts[r].view(-1)).view(T, H))
                    torch.cuda.synchronize()
                    assert all(torch.equal(ars[0], a) for a in ars)
                    res_ref = res0.clone()
                    if quant:
                        q_ref, s_ref, _ = ops.rmsnorm_quant_fp8(ars[0], w, 1e-5, residual=res_ref)
                    else:
                        x_ref = ars[0].clone()
                        ops.fused_add_rmsnorm(x_ref, res_ref, w, 1e-5)
                    torch.cuda.synchronize()
                    outs, ress = [], [res0.clone() for _ in range(world)]
                    for r in range(world):
                        assert comms[r].should_fuse_norm(parts[r])
                        with torch.cuda.stream(streams[r]):
                            outs.append(comms[r].fused_add_rmsnorm(parts[r], ress[r], w, 1e-5, quant_fp8=quant))
                    torch.cuda.synchronize()
                    for r in range(world):
                        assert torch.equal(ress[r], res_ref), ("residual", dt, T, H, quant, r)
                        if quant:
                            assert torch.equal(outs[r][1], s_ref) and torch.equal(outs[r][0].view(torch.uint8), q_ref.view(torch.uint8))
                        else:
                            assert torch.equal(outs[r], x_ref), ("norm", dt, T, H, r)
                # round 5: BOTH results from one launch (the FP8 companion of the drop-in RMSNorm under TP) = the two above
                torch.cuda.synchronize()
                res_ref = res0.clone()
                q_ref, s_ref, x_ref = ops.rmsnorm_quant_fp8(ars[0], w, 1e-5, residual=res_ref, want_out=True)
                torch.cuda.synchronize()
                outs, ress = [], [res0.clone() for _ in range(world)]
                for r in range(world):
                    with torch.cuda.stream(streams[r]):
                        outs.append(comms[r].fused_add_rmsnorm(parts[r], ress[r], w, 1e-5, with_fp8_companion=True))
                torch.cuda.synchronize()
                for r in range(world):
                    o_, q_, s_ = outs[r]
                    assert torch.equal(ress[r], res_ref) and torch.equal(o_, x_ref), ("both: norm", dt, T, H, r)
                    assert torch.equal(s_, s_ref) and torch.equal(q_.view(torch.uint8), q_ref.view(torch.uint8)), ("both: quant", dt, T, H, r)
        # the same with every rank's addend still a split-K GEMM (ops.GemmPartials): the fused kernel runs the GEMM epilogue
        # while it stages the row -- bit-identical to finalize() + fused_add_rmsnorm; bias on rank 0 only (RowParallelLinear)
        # (rows kept at <= 32: all ranks' workgroups -- one per row, 1024 threads, 89 VGPRs in this form -- must be resident
        #  on the ONE GPU of this box at the same time for their flag barriers to meet)
        for dt, cases in ((torch.bfloat16, [(32, 8192, 8), (7, 2048, 1), (16, 1024, 11), (16, 3072, 4)]),
                          (torch.float16, [(24, 4096, 5)])):
            for (T, H, SK) in cases:
                if H % (8 * world):
                    continue
                g = torch.Generator().manual_seed(T + H + SK + world)
                torch.cuda.synchronize()
                for c in comms:
                    c.rebind_fused_norm()
                gps = []
                for r in range(world):
                    ws = torch.randn(SK, T, H, generator=g).to(dev)
                    xs = (torch.rand(T, 1, generator=g) * 0.1 + 0.01).to(dev)
                    wsc = (torch.rand(H, generator=g) * 0.1 + 0.01).to(dev)
                    bias = (torch.randn(H, generator=g)).to(dt).to(dev) if r == 0 else None
                    gps.append(ops.GemmPartials(ws, SK, xs, wsc, bias, T, H, dt))
                res0 = torch.randn(T, H, generator=g).to(dt).to(dev)
                w = (torch.rand(H, generator=g) + 0.5).to(dt).to(dev)
                fins = [gp.finalize() for gp in gps]
                for quant in (False, True):
                    torch.cuda.synchronize()
                    refs, ress_ref = [], [res0.clone() for _ in range(world)]
                    for r in range(world):
                        with torch.cuda.stream(streams[r]):
                            refs.append(comms[r].fused_add_rmsnorm(fins[r], ress_ref[r], w, 1e-5, quant_fp8=quant))
                    torch.cuda.synchronize()
                    outs, ress = [], [res0.clone() for _ in range(world)]
                    for r in range(world):
                        assert comms[r].should_fuse_norm_shape(T, H, dt)
                        with torch.cuda.stream(streams[r]):
                            outs.append(comms[r].fused_add_rmsnorm_partials(gps[r], ress[r], w, 1e-5, quant_fp8=quant))
                    torch.cuda.synchronize()
                    for r in range(world):
                        assert torch.equal(ress[r], ress_ref[r]), ("partials residual", dt, T, H, SK, quant, r)
                        if quant:
                            assert torch.equal(outs[r][1], refs[r][1]) and \
                                torch.equal(outs[r][0].view(torch.uint8), refs[r][0].view(torch.uint8)), ("partials quant", dt, T, H, SK, r)
                        else:
                            assert torch.equal(outs[r], refs[r]), ("partials norm", dt, T, H, SK, r)
        assert not any(c.timed_out() for c in comms)
        for c in comms:
            c.close()
        q.put("ok")
    except Exception:
        import traceback
        q.put(traceback.format_exc())


@pytest.mark.parametrize("world", [2, 6, 8])
@pytest.mark.timeout(600)
def test_p2p_all_reduce_six_and_eight_ranks_in_one_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_local_ranks_worker, args=(world, q))
    p.start()
    msg = q.get(timeout=540)
    p.join(30)
    assert msg == "ok", msg


def test_p2p_all_reduce_missing_peer_fails_closed():
    """Rank 1 never launches: rank 0's bounded wait runs out, its output is NaN (not a sum of whatever the buffers
    held), the status word is visible to the host without a device sync, and the next call raises."""
    from sglang_npu_amd import _lib
    from sglang_npu_amd.distributed import CustomAllreduce, GroupCoordinator
    import ctypes
    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    comms = CustomAllreduce.connect_local(2, dev, max_size=1 << 20)
    _lib.check(lib.sgl_mi355_ar_set_spin_limit(ctypes.c_int64(20000)))
    try:
        x = torch.ones(4096, device=dev, dtype=torch.bfloat16)
        out = comms[0].custom_all_reduce(x)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out.float()).all()), "a timed-out all-reduce must not return a partial sum"
        assert comms[0].timed_out() and not comms[1].timed_out()
        tp = GroupCoordinator(None, 0, 2, dev)
        tp.ca_comm = comms[0]
        with pytest.raises(RuntimeError, match="did not reach the barrier"):
            tp.all_reduce(x)
        assert comms[0].disabled
    finally:
        _lib.check(lib.sgl_mi355_ar_set_spin_limit(ctypes.c_int64(1 << 27)))
        for c in comms:
            c.close()


def _quick_reduce_worker(world, q):
    """QuickReduce (sgl_mi355_ar_quick_all_reduce) with all ranks in one process: every regime against the numpy oracle
    (oracle/quick_reduce.py), bit for bit, on every rank; message sizes below and far above the staging area."""
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import numpy as np
        from oracle import quick_reduce as qro
        from sglang_npu_amd.distributed import CustomAllreduce, QuickAllReduce
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        max_size = 1 << 20
        comms = CustomAllreduce.connect_local(world, dev, max_size=max_size)
        streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
        for dt in (torch.float16, torch.bfloat16):
            rnd = (lambda x: x.astype(np.float16).astype(np.float32)) if dt == torch.float16 else \
                (lambda x: torch.from_numpy(x).to(torch.bfloat16).float().numpy())
            for regime, name in ((1, "INT8"), (2, "INT6"), (3, "INT4"), (0, "FP")):
                qrs = [QuickAllReduce(c, name, max_size_mb=64) for c in comms]
                assert not any(x.disabled for x in qrs)
                for n in (32, 32 * 1000, (1 << 20) + 32 * 7, 3 * (1 << 20) + 64):   # values; the last two: several chunks
                    g = torch.Generator().manual_seed(n % 977 + regime)
                    parts = [(torch.randn(n, generator=g) * (1 + r)).to(dt) for r in range(world)]
                    parts[0][:64] = 0                                  # all-zero blocks
                    parts_d = [p.to(dev) for p in parts]
                    if regime == 0:
                        # exact two-shot: fp32 accumulation in rank order, one rounding
                        acc = torch.zeros(n)
                        for p in parts:
                            acc = acc + p.float()
                        ref = acc.to(dt)
                    else:
                        ref = torch.from_numpy(qro.quick_all_reduce([p.float().numpy() for p in parts], regime, rnd,
                                                                    max_bytes=max_size)).to(dt)
                    for rep in range(2):
                        torch.cuda.synchronize()
                        outs = []
                        for r in range(world):
                            with torch.cuda.stream(streams[r]):
                                outs.append(qrs[r].quick_all_reduce(parts_d[r]))
                        torch.cuda.synchronize()
                        for r in range(world):
                            same = torch.equal(outs[r].cpu().view(torch.int16), ref.view(torch.int16))
                            assert same, (str(dt), name, n, r, rep, float((outs[r].cpu().float() - ref.float()).abs().max()))
                # the reference test's own payload and bound (test_quick_allreduce.py:131-165)
                n = 32 * 4096
                g = torch.Generator().manual_seed(regime)
                parts = [torch.randint(1, 24, (n,), generator=g).to(dt).to(dev) for _ in range(world)]
                exact = sum(p.float() for p in parts)
                outs = []
                for r in range(world):
                    with torch.cuda.stream(streams[r]):
                        outs.append(qrs[r].quick_all_reduce(parts[r]))
                torch.cuda.synchronize()
                torch.testing.assert_close(outs[0].float(), exact, atol=1.25 * world, rtol=0.5 * world)
                if regime == 0:
                    assert torch.equal(outs[0].float(), exact)
        # the exact P2P all-reduce still works on the same communicators afterwards (shared call counters)
        x = [torch.full((4096,), float(r + 1), dtype=torch.bfloat16, device=dev) for r in range(world)]
        outs = []
        for r in range(world):
            with torch.cuda.stream(streams[r]):
                outs.append(comms[r].custom_all_reduce(x[r]))
        torch.cuda.synchronize()
        assert all(float(o[0]) == world * (world + 1) / 2 for o in outs)
        assert not any(c.timed_out() for c in comms)
        for c in comms:
            c.close()
        q.put("ok")
    except Exception:
        import traceback
        q.put(traceback.format_exc())


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.timeout(600)
def test_quick_reduce_all_regimes_vs_oracle(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_quick_reduce_worker, args=(world, q))
    p.start()
    msg = q.get(timeout=500)
    p.join(30)
    assert msg == "ok", msg


def _staging_order_worker(world, q):
    """Regression for round 4's staging-store race (commit 195e7bb; csrc/allreduce.hip block_barrier): with the test hook on,
    the LAST wave of every workgroup issues its phase-A staging stores as late as possible -- right in front of the flag
    barrier, on every call -- which is the interleaving that once delivered a stale 16-byte vector of a peer's row in the
    8-rank in-process run of the PARTIALS form.  Integer-valued addends: every sum is exact in the 16-bit type and in fp32,
    so ANY stale or torn peer vector changes the bits of the residual.  200 calls per form, alternating the two halves of
    the double buffer and alternating payloads (a stale read of the previous call's vector in the same half is then wrong
    by construction)."""
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import ctypes
        from sglang_npu_amd import _lib, ops
        from sglang_npu_amd.distributed import CustomAllreduce
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        lib = _lib.lib()
        comms = CustomAllreduce.connect_local(world, dev, max_size=1 << 20)
        streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
        T, H, SK, dt = 32, 8192, 3, torch.bfloat16   # one row per workgroup, 16 waves of 64 lanes x 16 B per row
        g = torch.Generator().manual_seed(world)
        w = torch.ones(H, dtype=dt, device=dev)
        # four payload sets (set k used by call i with i % 4 == k: the same half of the double buffer sees sets k and k + 2
        # in turn); slabs of integer-valued fp32 partial sums with unit scales, sum of the slabs in [-6, 6]
        sets = []
        for k in range(4):
            slabs = [torch.randint(-2, 3, (SK, T, H), generator=g).float().to(dev) for _ in range(world)]
            addend = [s.sum(0) for s in slabs]
            total = torch.stack(addend).sum(0)
            sets.append((slabs, addend, total))
        ones_t, ones_h = torch.ones(T, device=dev), torch.ones(H, device=dev)
        res0 = torch.randint(-3, 4, (T, H), generator=g).to(dt).to(dev)
        _lib.check(lib.sgl_mi355_ar_set_test_delay(ctypes.c_int64(2)))   # ~7 us of idling in front of the last wave's stores
        # memory-channel noise on a stream of its own: queueing in the channels is what lets a late store land after a flag
        noise_stream = torch.cuda.Stream(device=dev)
        noise_a = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
        noise_b = torch.empty_like(noise_a)
        try:
            for form in ("partials", "plain", "all_reduce"):
                bad = []
                for i in range(200):
                    slabs, addend, total = sets[i % 4]
                    outs, ress = [], [res0.clone() for _ in range(world)]
                    with torch.cuda.stream(noise_stream):
                        noise_b.copy_(noise_a)
                    for r in range(world):
                        with torch.cuda.stream(streams[r]):
                            if form == "partials":
                                gp = ops.GemmPartials(slabs[r], SK, ones_t, ones_h, None, T, H, dt)
                                outs.append(comms[r].fused_add_rmsnorm_partials(gp, ress[r], w, 1e-5))
                            elif form == "plain":
                                outs.append(comms[r].fused_add_rmsnorm(addend[r].to(dt), ress[r], w, 1e-5))
                            else:
                                outs.append(comms[r].custom_all_reduce(addend[r].to(dt).view(-1)).view(T, H))
                    torch.cuda.synchronize()
                    expect = total + (res0.float() if form != "all_reduce" else 0.0)  # |values| <= 6 * 8 + 3: exact in bf16
                    for r in range(world):
                        got = (ress[r] if form != "all_reduce" else outs[r]).float()
                        if not torch.equal(got, expect):
                            bad.append((i, r, int((got != expect).sum())))
                assert not bad, f"{form}: stale or torn peer data in {len(bad)} (call, rank) pairs, first {bad[:4]}"
                torch.cuda.synchronize()
                for c in comms:
                    c.rebind_fused_norm()
        finally:
            _lib.check(lib.sgl_mi355_ar_set_test_delay(ctypes.c_int64(0)))
        assert not any(c.timed_out() for c in comms)
        for c in comms:
            c.close()
        q.put("ok")
    except Exception:
        import traceback
        q.put(traceback.format_exc())


@pytest.mark.timeout(600)
def test_staging_stores_are_ordered_before_the_flags_with_the_last_wave_delayed():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_staging_order_worker, args=(8, q))
    p.start()
    msg = q.get(timeout=540)
    p.join(30)
    assert msg == "ok", msg
