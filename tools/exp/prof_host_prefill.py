import argparse, cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from sglang_npu_amd.harness import ForwardBatch, ForwardMode
from sglang_npu_amd.layers import greedy_sample
args = argparse.Namespace(batch=16, ctx=2048, model="llama3-8b", quant="w8a8_fp8", layers=None, kv_dtype="auto", emulate_tp=0,
                          call_order=os.environ.get("ORDER", "reference"), no_graph=True, gpus=1, steps=2, warmup=1)
device = torch.device("cuda", 0); torch.cuda.set_device(device)
from sglang_npu_amd.distributed import init_distributed_environment
init_distributed_environment(device=device)
net, cfg, runner, backend, max_len = bench.build(args, device, 1)
net.fuse_quant = os.environ.get("ORDER", "reference") == "fused"  # reference: models/llama.py's call order through the drop-in classes
r2t = runner.req_to_token_pool.req_to_token
n=128
ids = torch.randint(0, 10000, (n,), device=device); pos = torch.arange(n, device=device)
rpi = torch.zeros(1, dtype=torch.int64, device=device); seq = torch.full((1,), n, dtype=torch.int64, device=device)
loc = r2t[0, :n].to(torch.int64); zero = torch.zeros(1, dtype=torch.int64, device=device)
fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, rpi, seq, loc, n, seq.cpu(), pos, extend_num_tokens=n, extend_seq_lens=seq.clone(),
                  extend_prefix_lens=zero, extend_start_loc=zero.clone(), extend_prefix_lens_cpu=[0], extend_seq_lens_cpu=[n],
                  req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend)
def run():
    backend.init_forward_metadata(fb); logits = net(ids, pos, fb); tok = greedy_sample(logits[-1:]); tok.item()
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
print("wall per pass, eager: %.2f ms" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
