"""GPU: the tensor-parallel decoder stack with TWO ranks sharing the one GPU of the box (gloo carries the IPC handle
exchange; the P2P all-reduce kernel carries the data, as it would over xGMI).  Checks, on a 2-layer Llama-3-8B-shaped FP8
stack at bs = 64:
  * the three ways a row-parallel layer's collective can run give the SAME bits: in-stream all-reduce + norm, side-stream
    all-reduce (AllReduceHandle) + norm, and the fused all-reduce + residual add + RMSNorm + FP8 quant kernel;
  * both ranks hold identical hidden states afterwards;
  * the TP = 2 result agrees with the TP = 1 stack to bf16 accumulation noise (different summation split).
"""
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


def _run_stack(tp_rank, tp_world, device, fuse, async_ar, with_lm_head=False, reference_order=False, passes=1, quant="w8a8_fp8"):
    """One prefill-free decode step of a 2-layer stack (with_lm_head=False: the output is the final hidden state;
    True: the vocab-parallel LM head's logits after the all-gather over the ranks, logits_processor.py:430-505)."""
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (ForwardBatch, ForwardMode, MHATokenToKVPool, ModelRunnerLike, ReqToTokenPool,
                                        ServerArgs, install_attention_backend)
    M.FUSE_AR_NORM, M.ASYNC_AR = fuse, async_ar
    cfg = M.LLAMA3_8B
    B, ctx = 64, 96
    net = M.LlamaForCausalLM(cfg, quant, torch.bfloat16, str(device), num_layers=2,
                             with_lm_head=with_lm_head).load_dummy_weights()
    max_len = ctx + 4
    n_tok = B * max_len + 1
    r2t_pool = ReqToTokenPool(B, max_len, str(device))
    kv_pool = MHATokenToKVPool(n_tok, 1, torch.bfloat16, cfg.get_num_kv_heads(tp_world), cfg.head_dim, 2, str(device))
    g = torch.Generator(device=device).manual_seed(1234 + 7 * tp_rank * 0)
    # every rank fills ITS kv heads from the same full-head tensor so that TP = 1 and TP = 2 see the same cache
    hkv_full = cfg.num_key_value_heads
    per = hkv_full // tp_world
    for l in range(2):
        rows = kv_pool.k_buffer[l].shape[0]  # size + page_size (memory_pool.py:222-241)
        kf = torch.randn(rows, hkv_full, cfg.head_dim, device=device, generator=g, dtype=torch.float32).bfloat16()
        vf = torch.randn(rows, hkv_full, cfg.head_dim, device=device, generator=g, dtype=torch.float32).bfloat16()
        kv_pool.k_buffer[l].copy_(kf[:, tp_rank * per:(tp_rank + 1) * per])
        kv_pool.v_buffer[l].copy_(vf[:, tp_rank * per:(tp_rank + 1) * per])
    perm = (torch.randperm(n_tok - 1, device=device, generator=g) + 1).to(torch.int32)
    r2t_pool.req_to_token.copy_(perm[: B * max_len].view(B, max_len))
    runner = ModelRunnerLike(cfg, r2t_pool, kv_pool, str(device), 0, tp_world, ServerArgs())
    backend = install_attention_backend(runner)
    ids = torch.randint(0, 10000, (B,), device=device, generator=g)
    seq = torch.full((B,), ctx, dtype=torch.int64, device=device)
    pos = seq - 1
    rpi = torch.arange(B, dtype=torch.int64, device=device)
    loc = r2t_pool.req_to_token[rpi, pos].to(torch.int64)
    fb = ForwardBatch(ForwardMode.DECODE, B, ids, rpi, seq, loc, B * ctx, None, pos, req_to_token_pool=r2t_pool,
                      token_to_kv_pool=kv_pool, attn_backend=backend)
    backend.init_forward_metadata(fb)
    if reference_order:  # models/llama.py's operator order through the drop-in classes (RMSNorm -> apply() -> RoPE -> attn ...)
        net.fuse_quant = False
    outs = []
    for _ in range(passes):  # (the KV write of a decode step is idempotent: same slot, same values)
        outs.append(net(ids, pos, fb).float().cpu())
        torch.cuda.synchronize()
    return outs[0] if passes == 1 else outs


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import torch.distributed as dist
        from sglang_npu_amd import distributed as D
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tp = D.GroupCoordinator(dist.group.WORLD, rank, world, dev)
        tp.ca_comm = D.CustomAllreduce(dist.group.WORLD, dev, max_size=8 * 1024 * 1024)
        D.set_tp_group(tp)
        outs = {}
        for name, fuse, asy in (("sync", False, False), ("async", False, True), ("fused", True, False)):
            dist.barrier()
            outs[name] = _run_stack(rank, world, dev, fuse, asy)
        # the reference call order, three passes: the first teaches the norms that FP8 linears follow them, from the second on
        # the fused all-reduce + norm kernel also emits the FP8 companion (round 5) -- not one bit may move
        # ... and (deferred.py) from the second pass on o_proj / down_proj -- called WITHOUT flags, as models/llama.py does --
        # hand their unreduced output to the norm as a lazy tensor: all-reduce + add + norm in one kernel, 4 per pass here
        dist.barrier()
        fused_count = lambda: D.DISPATCH_COUNTS["p2p+norm"] + D.DISPATCH_COUNTS["p2p+norm(partials)"]  # noqa: E731
        fused_before, partials_before = fused_count(), D.DISPATCH_COUNTS["p2p+norm(partials)"]
        ref_order = _run_stack(rank, world, dev, True, False, reference_order=True, passes=3)
        assert torch.equal(ref_order[0], ref_order[1]) and torch.equal(ref_order[0], ref_order[2]), \
            "FP8 companions / the lazy all-reduce changed the reference-order result under TP"
        assert fused_count() - fused_before == 8, \
            f"untouched call order: {fused_count() - fused_before} fused all-reduce + norm launches, expected 8"
        # (down_proj, 28 Mi weights per rank, comes as split-K partials; o_proj, 8 Mi, as its finished local sum)
        assert D.DISPATCH_COUNTS["p2p+norm(partials)"] - partials_before == 4
        outs["reference_order"] = ref_order[0]
        # the same through 16-bit weights (config 2 under TP): the streamer's split-K partials go to the fused all-reduce + norm on
        # unit scales where the rule takes them (down_proj), the finished local sum elsewhere -- three passes, not one bit apart
        dist.barrier()
        ref16 = _run_stack(rank, world, dev, True, False, reference_order=True, passes=3, quant=None)
        assert torch.isfinite(ref16[0]).all() and torch.equal(ref16[0], ref16[1]) and torch.equal(ref16[0], ref16[2]), \
            "the lazy all-reduce changed the 16-bit model's reference-order result under TP"
        assert torch.equal(outs["sync"], outs["async"]), "side-stream all-reduce changed the result"
        assert torch.equal(outs["sync"], outs["fused"]), "fused all-reduce + norm kernel changed the result"
        gathered = [None] * world
        dist.all_gather_object(gathered, outs["fused"])
        assert all(torch.equal(gathered[0], t) for t in gathered), "ranks disagree"
        # vocab-parallel LM head: every rank computes its slice of the logits, the all-gather puts them together
        dist.barrier()
        logits = _run_stack(rank, world, dev, True, False, with_lm_head=True)
        assert logits.shape == (64, 128256), logits.shape
        gathered = [None] * world
        dist.all_gather_object(gathered, logits)
        assert all(torch.equal(gathered[0], t) for t in gathered), "ranks disagree on the logits"
        assert not tp.ca_comm.timed_out()
        dist.barrier()
        tp.ca_comm.close()
        # numpy: plain pickling, no shared-memory fds
        q.put((rank, "ok", (outs["fused"].numpy(), logits.numpy()) if rank == 0 else None))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.timeout(600)
def test_tp2_stack_three_collective_forms_agree_and_match_tp1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(30)
    tp2 = None
    for rank, msg, out in res:
        assert msg == "ok", f"rank {rank}: {msg}"
        if out is not None:
            tp2, tp2_logits = torch.from_numpy(out[0]), torch.from_numpy(out[1])
    from sglang_npu_amd import distributed as D
    D.set_tp_group(D.GroupCoordinator(None, 0, 1, torch.device("cuda", 0)))
    tp1 = _run_stack(0, 1, torch.device("cuda", 0), False, False)
    err, scale = float((tp1 - tp2).abs().max()), float(tp1.abs().max())
    assert err <= 2.0 ** -5 * scale, f"TP=2 vs TP=1 hidden states: {err:.3e} (max {scale:.3e})"
    tp1_logits = _run_stack(0, 1, torch.device("cuda", 0), False, False, with_lm_head=True)
    err, scale = float((tp1_logits - tp2_logits).abs().max()), float(tp1_logits.abs().max())
    assert err <= 2.0 ** -4 * scale, f"TP=2 vs TP=1 logits: {err:.3e} (max {scale:.3e})"
    # greedy tokens agree wherever the TP=1 top-2 margin exceeds that noise
    top2 = tp1_logits.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * err
    assert torch.equal(tp1_logits.argmax(-1)[clear], tp2_logits.argmax(-1)[clear])
