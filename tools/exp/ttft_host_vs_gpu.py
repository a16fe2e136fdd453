#!/usr/bin/env python3
"""Where a short eager prefill spends its wall time: host enqueue time (until the last launch returns) against the device
time of the same pass (events), bs=1, 128 new tokens, with and without a cached prefix, with and without KV-range parts."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sglang_npu_amd.harness import ForwardBatch, ForwardMode  # noqa: E402
from sglang_npu_amd.layers import greedy_sample  # noqa: E402

args = argparse.Namespace(batch=16, ctx=6144, model="llama3-8b", quant="w8a8_fp8", layers=None, kv_dtype="auto", emulate_tp=0,
                          call_order="fused", no_graph=True, gpus=1, steps=2, warmup=1)
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
from sglang_npu_amd.distributed import init_distributed_environment  # noqa: E402
init_distributed_environment(device=device)
net, cfg, runner, backend, max_len = bench.build(args, device, 1)
r2t = runner.req_to_token_pool.req_to_token


def one(input_len, prefix_len, parts):
    saved = backend._extend_parts
    if not parts:
        backend._extend_parts = None
    ids = torch.randint(0, 10000, (input_len,), device=device)
    pos = torch.arange(prefix_len, prefix_len + input_len, device=device)
    rpi = torch.zeros(1, dtype=torch.int64, device=device)
    seq = torch.full((1,), prefix_len + input_len, dtype=torch.int64, device=device)
    loc = r2t[0, prefix_len:prefix_len + input_len].to(torch.int64)
    zero = torch.zeros(1, dtype=torch.int64, device=device)
    ext = torch.full((1,), input_len, dtype=torch.int64, device=device)
    fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, rpi, seq, loc, input_len, seq.cpu(), pos, extend_num_tokens=input_len,
                      extend_seq_lens=ext, extend_prefix_lens=zero + prefix_len, extend_start_loc=zero.clone(),
                      extend_prefix_lens_cpu=[prefix_len], extend_seq_lens_cpu=[input_len],
                      req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend)
    rows = []
    for i in range(9):
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        st.record()
        backend.init_forward_metadata(fb)
        logits = net(ids, pos, fb)
        tok = greedy_sample(logits[-1:])
        en.record()
        t1 = time.perf_counter()
        tok.item()
        t2 = time.perf_counter()
        if i >= 2:
            rows.append(((t1 - t0) * 1e3, st.elapsed_time(en), (t2 - t0) * 1e3))
    backend._extend_parts = saved
    rows.sort(key=lambda r: r[2])
    h, g, w = rows[len(rows) // 2]
    return dict(new=input_len, prefix=prefix_len, parts=parts, host_enqueue_ms=round(h, 3), device_ms=round(g, 3), wall_ms=round(w, 3))


for (n, p) in ((128, 0), (128, 1920), (128, 4096)):
    for parts in (True, False):
        print(json.dumps(one(n, p, parts)), flush=True)
