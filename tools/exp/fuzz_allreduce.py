"""Random message sizes through the P2P all-reduce with all ranks inside ONE process (CustomAllreduce.connect_local, one stream
per rank, as tests/test_custom_allreduce_gpu.py does): one-shot / two-shot regimes, sizes that are not multiples of the block
shapes, both buffer halves.  Integer-valued payloads, so the sum is exact."""
import os, sys, random
os.environ["GPU_MAX_HW_QUEUES"] = "16"  # one hardware queue per rank's stream (read when the HIP runtime initialises)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_npu_amd.distributed import CustomAllreduce
world = int(os.environ.get("WORLD", "8"))
rng = random.Random(int(os.environ.get("SEED", "0")))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
comms = CustomAllreduce.connect_local(world, dev, max_size=4 * 1024 * 1024)
streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
bad = skipped = 0
for it in range(int(os.environ.get("N", "60"))):
    dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    esz = torch.tensor([], dtype=dt).element_size()
    nbytes = rng.choice([16, 48, 512, 4096, 65536, 262144, 1 << 20, 1 << 21, 3 << 20, 4 << 20]) + 16 * rng.randint(0, 64) * rng.choice([0, 1, world])
    nbytes = min(nbytes, 4 << 20)
    n = nbytes // esz
    if os.environ.get("TRACE"):
        print("CASE", dict(it=it, dt=str(dt), n=n, nbytes=n * esz), flush=True)
    g = torch.Generator().manual_seed(it)
    parts = [torch.randint(-3, 4, (n,), generator=g).to(dt).to(dev) for _ in range(world)]
    ref = sum(p.float() for p in parts).to(dt)
    if not comms[0].should_custom_ar(parts[0]):
        skipped += 1
        continue
    for rep in range(2):
        torch.cuda.synchronize()
        outs = []
        for r in range(world):
            with torch.cuda.stream(streams[r]):
                outs.append(comms[r].custom_all_reduce(parts[r]))
        torch.cuda.synchronize()
        for r in range(world):
            if outs[r] is None or not torch.equal(outs[r], ref):
                print("MISMATCH", dict(it=it, dt=str(dt), n=n, rank=r, rep=rep))
                bad += 1
                break
if any(c.timed_out() for c in comms):
    print("TIMED OUT flag set")
    bad += 1
for c in comms:
    c.close()
print("cases done; skipped (not eligible)", skipped, "bad", bad)
sys.exit(1 if bad else 0)
