"""Child-process probe: can this job's all-reduce be captured into a HIP graph?

A failed stream capture leaves a sticky HIP error in the process, so bench.py asks a throw-away child
(one per rank, rendezvousing on MASTER_PORT + 17) instead of trying in-process.  Exit code 0 = yes."""
import os
import sys

import torch
import torch.distributed as dist


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("SGL_MI355_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    backend = os.environ.get("SGL_MI355_DIST_BACKEND") or "nccl"
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    x = torch.ones(1024, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            dist.all_reduce(x)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        x.fill_(1.0)
        with torch.cuda.graph(g):
            dist.all_reduce(x)
        x.fill_(1.0)
        g.replay()
        torch.cuda.synchronize()
        ok = bool(torch.all(x == float(world)).item())
    except Exception:
        ok = False
    os._exit(0 if ok else 1)  # skip destructors: the process group may be wedged after a failed capture


if __name__ == "__main__":
    sys.exit(main())
