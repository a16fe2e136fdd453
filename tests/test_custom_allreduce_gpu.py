"""GPU: the P2P all-reduce kernel with 2 and 4 ranks sharing the ONE GPU of the box (IPC handles between
processes, same protocol as across GPUs; RCCL itself refuses duplicate devices so gloo carries the handle
exchange).  Integer-valued payloads: the result must be exact (test_custom_allreduce.py:118-146)."""
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
