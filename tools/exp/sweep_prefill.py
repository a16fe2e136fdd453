"""Prefill (bs = 1, empty prefix, eager) time of the headline model over input lengths: ms and tokens/s per length.
Looks for cliffs between the GEMM tile classes (M = 128 / 256 / 1024 / 4096 ...) the way sweep_bench.sh does for decode."""
import json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
lens = [int(x) for x in os.environ.get("LENS", "64,128,129,256,257,512,768,1024,1025,1536,2048,3000,4096,8192").split(",")]
args = types.SimpleNamespace(batch=1, ctx=max(lens) + 8, steps=4, warmup=2, model=os.environ.get("MODEL", "llama3-8b"),
                             quant=os.environ.get("QUANT", "w8a8_fp8"), layers=None, kv_dtype="auto")
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
from sglang_npu_amd.distributed import init_distributed_environment  # noqa: E402
tp = init_distributed_environment(device=device)
net, cfg, runner, backend, max_len = bench.build(args, device, 1)
for L in lens:
    ms, used = bench.time_ttft(net, runner, backend, device, input_len=L, reps=5)
    print(json.dumps(dict(input_len=used, ms=round(ms, 3), tokens_per_s=round(used / ms * 1e3, 0), us_per_layer=round(ms * 1e3 / len(net.layers), 1))), flush=True)
