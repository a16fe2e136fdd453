#!/usr/bin/env python3
"""Headline benchmark: output tokens/s of a Llama-3-8B-shaped FP8 (w8a8) model decoding bs=64 requests
with RadixAttention-style token-level paged KV (BASELINE.json configs[1]/[2]) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    N > 1, either way one rank per GPU, TP = N over RCCL/xGMI:
      * the plain command above: this process becomes the LAUNCHER (launch_ranks below) -- it starts the N ranks as child
        processes before anything touches the GPU, relays rank 0's JSON line and fails if a rank fails
        (the reference's harness spawns its TP ranks itself too: bench_one_batch.py:509-545);
      * python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: the ranks are given (RANK / WORLD_SIZE set).

One "step" = one decode step of the whole batch: for each of the 32 layers
  fused-add RMSNorm -> per-token FP8 quant -> qkv GEMM (fp8_scaled_mm) -> RoPE -> KV-pool write ->
  paged decode attention -> o_proj GEMM (+ all-reduce) -> RMSNorm -> quant -> gate_up GEMM -> SiLU*mul ->
  quant -> down GEMM (+ all-reduce), then final norm, LM head, greedy argmax.
Every op of the hot path is this repo's HIP library (sglang_npu_amd/lib/libsgl_mi355.so); the step is
captured once into a HIP graph and replayed, like the reference's decode path
(model_runner.py:1663-1669).  Inputs (weights, KV pool, page table) are resident in HBM before the
timed region.  Data is synthetic: random-init weights with the reference's dummy-loader recipe, a
random-permutation page table (worst case for HBM), context length CTX per request.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (paged decode attention):
achieved = algorithmic bytes per launch / its average launch duration, measured with HIP events around every launch of it
in an instrumented (eager) pass over the same K steps (`roofline.method` says which figure is primary and gives the
in-step difference-of-graphs figure beside it).
`roofline_gemm` = the four decode GEMMs at M = bs (HBM fraction) and at M = 4096 (MFMA fraction), event-timed
in this run over the model's own per-layer weights.
`value` is the fused call order (`--call-order fused`: norm+quant / RoPE+KV-write / SiLU+quant producers, GEMM
epilogues inside their consumers); `dropin_ms_per_step` is the SAME model driven in the reference's call order
(RMSNorm -> LinearMethodBase.apply [quant + GEMM] -> RoPE -> AttentionBackend.forward(save_kv_cache=True) -> ...,
models/llama.py:94-98,186-191,245-268), i.e. what an untouched SGLang model file gets from the drop-in classes.
`cpu_baseline` = the CPU oracle (a port of the reference's CPU algorithm, oracle/) timed on the
host cores on a bounded sample (1 of the 32 layers' hot path, 3 warm-ups + median of 10), scaled to the full step; its GEMM leg is
the faster of the C port and the reference's own fallback formula (fp8_utils.py:479-507) in plain torch on the same cores;
`reference_cpu_container` quotes the compiled reference's own timing in the build container
(tools/time_reference_cpu.py -> profiles/r02_reference_cpu_container.json).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory: the HIP runtime then skips a host-visible kernarg fetch per launch.  Nothing changes for the
# graph-replayed decode step; the eagerly launched prefill (TTFT) pass is ~4 % shorter (9.9-10.0 -> 9.5 ms, same box).  Must be in
# the environment before the runtime initialises; an explicit setting of the caller wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILED_STATS_GLOB = "r[0-9][0-9]_bench_tp1_kernel_stats.csv"  # rocprofv3 --kernel-trace --stats of `bench.py`, one per round
PMC_SUMMARY = "r05_decode_pmc_instep.json"  # in-step counters of the profiled bench command, refreshed per round
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def profiled_attn_us():
    """Average busy time of the decode attention kernel in the newest committed rocprofv3 summary of this command
    (profiles/rNN_bench_tp1_kernel_stats.csv), or None -- read from the file, not a literal."""
    import csv
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", PROFILED_STATS_GLOB)))[-1]
        for row in csv.DictReader(open(path)):
            if row.get("Name", "").find("decode_mfma_pair_kernel") >= 0:
                return {"us": round(float(row["AverageNs"]) / 1e3, 2), "launches": int(row["Calls"]), "source": os.path.basename(path)}
    except (OSError, IndexError, KeyError, ValueError):
        pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=2048)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b", "llama2-7b", "qwen2-0.5b"])
    ap.add_argument("--quant", default="w8a8_fp8", choices=["w8a8_fp8", "awq", "none"])
    ap.add_argument("--layers", type=int, default=None, help="override the layer count (debug only; invalidates the number)")
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8_e4m3", "fp8_e5m2"],
                    help="KV pool dtype (server_args.py --kv-cache-dtype); the headline number is 'auto' = the model dtype")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="debug only: run ONE rank's share of a TP=N step on this GPU: plain collectives are identities, the "
                         "fused all-reduce + add + RMSNorm kernels run on a one-rank communicator (the launches of a real "
                         "rank, minus the peers' bytes; per-rank kernel rehearsal on a 1-GPU box; invalidates the number)")
    ap.add_argument("--call-order", default="reference", choices=["fused", "reference"],
                    help="which call order `value` reports: the reference's operator order (default: what an untouched SGLang "
                         "model file gets from the drop-in classes) or this repo's fused producers; the other one is reported "
                         "beside it (value_fused / value_dropin)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE configs 2, 4 and 5 (one rank emulated) that the default N=1 "
                         "invocation reports beside the headline")
    return ap.parse_args()


def build(args, device, tp):
    from sglang_npu_amd import model as M
    from sglang_npu_amd.harness import (MHATokenToKVPool, ModelRunnerLike, ReqToTokenPool, ServerArgs,
                                        install_attention_backend)
    cfg = {"llama3-8b": M.LLAMA3_8B, "llama3-70b": M.LLAMA3_70B, "llama2-7b": M.LLAMA2_7B,
           "qwen2-0.5b": M.QWEN2_05B}[args.model]
    dtype = torch.float16 if args.quant == "awq" else torch.bfloat16
    quant = None if args.quant == "none" else args.quant
    if args.ctx + args.steps + args.warmup + 8 > cfg.context_len:
        # positions index the rotary table (context_len rows): a run past the model's context would read beyond it -- the
        # rotary ops, like upstream's, do not check positions on the device.  Off-default sweeps (--ctx 8192) get a longer table.
        import dataclasses
        cfg = dataclasses.replace(cfg, context_len=args.ctx + args.steps + args.warmup + 8)
        print(f"[bench] --ctx {args.ctx} + steps exceeds the model's context length: rotary table extended to "
              f"{cfg.context_len} positions", file=sys.stderr)
    net = M.LlamaForCausalLM(cfg, quant, dtype, str(device), num_layers=args.layers).load_dummy_weights()
    n_layers = len(net.layers)
    B, ctx = args.batch, args.ctx
    max_len = ctx + args.steps + args.warmup + 8
    n_tok = B * max_len + 1
    hkv = cfg.get_num_kv_heads(tp)
    r2t_pool = ReqToTokenPool(B, max_len, str(device))
    kv_dtype = {"fp8_e4m3": torch.float8_e4m3fn, "fp8_e5m2": torch.float8_e5m2}.get(args.kv_dtype, dtype)
    kv_pool = MHATokenToKVPool(n_tok, 1, kv_dtype, hkv, cfg.head_dim, n_layers, str(device))
    g = torch.Generator(device=device).manual_seed(1234)
    for l in range(n_layers):  # KV of the already-decoded context: N(0,1) like the reference's kernel tests
        if kv_dtype == dtype:
            kv_pool.k_buffer[l].normal_(generator=g)
            kv_pool.v_buffer[l].normal_(generator=g)
        else:  # uint8 storage of e4m3 / e5m2 values
            for buf in (kv_pool.k_buffer[l], kv_pool.v_buffer[l]):
                buf.copy_(torch.randn(buf.shape, device=device, generator=g, dtype=torch.bfloat16)
                          .to(kv_dtype).view(torch.uint8))
    # token-level page table = one random permutation of the pool (slot 0 stays the padding slot)
    perm = (torch.randperm(n_tok - 1, device=device, generator=g) + 1).to(torch.int32)
    r2t_pool.req_to_token.copy_(perm[: B * max_len].view(B, max_len))
    runner = ModelRunnerLike(cfg, r2t_pool, kv_pool, str(device), device.index or 0, tp, ServerArgs())
    backend = install_attention_backend(runner)
    return net, cfg, runner, backend, max_len


class DecodeLoop:
    """Static-buffer decode loop (the shape of cuda_graph_runner.py: static inputs, metadata refreshed
    before every replay)."""

    def __init__(self, net, runner, backend, B, ctx, device, use_graph=True):
        from sglang_npu_amd.harness import ForwardBatch, ForwardMode
        self.net, self.runner, self.backend, self.B, self.device = net, runner, backend, B, device
        self.input_ids = torch.randint(0, 10000, (B,), device=device)
        self.seq_lens = torch.full((B,), ctx, dtype=torch.int64, device=device)  # includes the token being decoded
        self.positions = self.seq_lens - 1
        self.req_pool_indices = torch.arange(B, dtype=torch.int64, device=device)
        self.out_cache_loc = torch.zeros(B, dtype=torch.int64, device=device)
        self.r2t = runner.req_to_token_pool.req_to_token
        self.rows = torch.arange(B, device=device)
        self.fb = ForwardBatch(ForwardMode.DECODE, B, self.input_ids, self.req_pool_indices, self.seq_lens,
                               self.out_cache_loc, B * ctx, None, self.positions,
                               req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                               attn_backend=backend)
        self.next_ids = torch.zeros(B, dtype=torch.int64, device=device)
        self.cur_len, self.max_len = ctx, self.r2t.size(1)  # host copy of the (uniform) sequence length: bounds check
        self.graph = None
        self.use_graph = use_graph
        self._refresh()

    def _refresh(self):
        # what the scheduler does between steps: the new token's slot is the page-table entry at seq_len-1
        self.positions.copy_(self.seq_lens - 1)
        self.out_cache_loc.copy_(self.r2t[self.rows, self.positions].to(torch.int64))

    def _forward(self):
        logits = self.net(self.input_ids, self.positions, self.fb)
        from sglang_npu_amd.layers import greedy_sample
        self.next_ids.copy_(greedy_sample(logits))

    def capture(self):
        self.backend.init_cuda_graph_state(self.B, self.B)
        self.backend.init_forward_metadata_capture_cuda_graph(self.B, self.B, self.req_pool_indices, self.seq_lens,
                                                              None, self.fb.forward_mode, None)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._forward()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # capture on the stream the warm-up passes ran on (cuda_graph_runner.py:526-534 does the same): per-stream state
        # such as the split-K scratch then already has its final size
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            self._forward()

    def rewind(self, ctx):
        self.seq_lens.fill_(ctx)
        self.cur_len = ctx
        self._refresh()

    def step(self):
        if self.cur_len + 1 > self.max_len:  # the gather in _refresh would read past the request's page-table row
            raise RuntimeError(f"decode loop ran past the page table ({self.cur_len + 1} > {self.max_len}): rewind() first")
        if self.graph is not None:
            self.backend.init_forward_metadata_replay_cuda_graph(self.B, self.req_pool_indices, self.seq_lens,
                                                                 0, None, self.fb.forward_mode, None, None)
            self.graph.replay()
        else:
            self.backend.init_forward_metadata(self.fb)
            self._forward()
        # feed the sampled token back and advance every request by one position
        self.input_ids.copy_(self.next_ids % 10000)
        self.seq_lens += 1
        self.cur_len += 1
        self._refresh()


def time_attention_kernel(loop, steps):
    """Average duration of the dominant kernel (paged decode attention), HIP events around each launch
    of it on the launching stream, over `steps` eager steps of the same loop.  Every `ops.decode_attention*` entry is
    wrapped: the plain launch, and the forms with extra work INSIDE the same launch (qkv epilogue + RoPE + KV write in the
    prologue; row-absmax epilogue; kv-split merge + FP8 quant) -- that work is then part of the timed launch while the
    algorithmic bytes stay the KV bytes alone (the fraction is, if anything, understated).  A form that declines a shape
    (returns False / None without launching) is not counted."""
    from sglang_npu_amd import ops
    time_attention_kernel.merged_launches = False
    names = [n for n in dir(ops) if n.startswith("decode_attention") and callable(getattr(ops, n))]
    declines = {"decode_attention_qkv_partials": (False, None), "decode_attention_paged_absmax": (False, None),
                "decode_attention_paged_merged": (False,), "decode_attention_paged_newkv": (False,)}
    real = {n: getattr(ops, n) for n in names}
    pairs = []

    def wrap(name, fn):
        def timed(*a, **kw):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            done = fn(*a, **kw)
            en.record()
            if not any(done is d for d in declines.get(name, ())):
                pairs.append((st, en))
                if name == "decode_attention_paged_merged":
                    time_attention_kernel.merged_launches = True
            return done
        return timed

    for n in names:
        setattr(ops, n, wrap(n, real[n]))
    g = loop.graph
    loop.graph = None
    try:
        for _ in range(steps):
            # keep the GPU BEHIND the host for the whole eager step: the events bracket a launch on the device timeline, and where
            # the device waits for the host (small kernels, Python between two launches) the start event would be stamped before
            # the kernel has even been enqueued -- host time in a kernel's duration.  A spin of ~3 ms in front of the step lets
            # the host queue the step ahead (a step is ~5 ms of host time, ~6 ms of device time).
            if hasattr(torch.cuda, "_sleep"):
                torch.cuda._sleep(6_000_000)
            loop.step()
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(ops, n, real[n])
        loop.graph = g
    if not pairs:
        raise RuntimeError("time_attention_kernel: no decode attention launch went through sglang_npu_amd.ops.decode_attention*")
    durations = sorted(s.elapsed_time(e) for s, e in pairs)
    # drop the slowest 5 % (first-touch / clock ramp) but keep the mean honest otherwise
    keep = durations[: max(1, int(len(durations) * 0.95))]
    return sum(keep) / len(keep), len(durations)


def time_ttft(net, runner, backend, device, input_len=1024, reps=7, prefix_len=0):
    """p50 time-to-first-token at bs=1: one EXTEND (prefill) pass of `input_len` new tokens with an empty
    prefix through the whole model + greedy sample, device-synchronised (bench_one_batch.py:380-405).
    prefix_len > 0: the new tokens follow that many cached ones (a radix-cache hit / a later chunk of a chunked prefill); the pool
    rows of the prefix hold what the bench put there (N(0,1))."""
    from sglang_npu_amd.harness import ForwardBatch, ForwardMode
    from sglang_npu_amd.layers import greedy_sample
    r2t = runner.req_to_token_pool.req_to_token
    input_len = min(input_len, r2t.size(1) - 1)
    prefix_len = max(0, min(prefix_len, r2t.size(1) - 1 - input_len))
    ids = torch.randint(0, 10000, (input_len,), device=device)
    pos = torch.arange(prefix_len, prefix_len + input_len, device=device)
    rpi = torch.zeros(1, dtype=torch.int64, device=device)
    seq = torch.full((1,), prefix_len + input_len, dtype=torch.int64, device=device)
    loc = r2t[0, prefix_len:prefix_len + input_len].to(torch.int64)
    zero = torch.zeros(1, dtype=torch.int64, device=device)
    ext = torch.full((1,), input_len, dtype=torch.int64, device=device)
    fb = ForwardBatch(ForwardMode.EXTEND, 1, ids, rpi, seq, loc, input_len, seq.cpu(), pos,
                      extend_num_tokens=input_len, extend_seq_lens=ext, extend_prefix_lens=zero + prefix_len,
                      extend_start_loc=zero.clone(), extend_prefix_lens_cpu=[prefix_len], extend_seq_lens_cpu=[input_len],
                      req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                      attn_backend=backend)
    times = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        backend.init_forward_metadata(fb)
        logits = net(ids, pos, fb)
        tok = greedy_sample(logits[-1:])
        tok.item()
        if i >= 2:
            times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    return times[len(times) // 2], input_len


def time_ttft_graph(net, runner, backend, device, input_len=128, reps=9, prefix_len=0):
    """p50 time-to-first-token of a SHORT prompt (bs=1, `input_len` tokens, empty prefix) replayed from a HIP graph captured
    at that token count (harness.PrefillGraphRunner): the eager pass is host-bound at this size (VERDICT r3 ask 7).
    prefix_len > 0: the same behind that many cached tokens (graph captured for that prefix bound)."""
    from sglang_npu_amd.harness import PrefillGraphRunner
    r2t = runner.req_to_token_pool.req_to_token
    ids = torch.randint(0, 10000, (input_len,), device=device)
    slots = r2t[0, prefix_len:prefix_len + input_len].to(torch.int64)
    pre = r2t[0, :prefix_len].to(torch.int64) if prefix_len else None
    pg = PrefillGraphRunner(net, runner, backend, device, buckets=(input_len,), prefix_buckets=(prefix_len,))
    times = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, tok = pg.run(ids, slots, pre)
        tok.item()
        if i >= 2:
            times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    return times[len(times) // 2]


def _effective_cpus() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a one-GPU box shares a
    128-core host: 128 OpenMP threads on a 16-core share would only measure oversubscription)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(cfg, B, ctx, n_layers_full):
    """Time the CPU oracle on ONE layer's hot path (attention over ctx tokens + 4 quant-GEMMs) on the host cores this
    process may use, and scale to the full step.  Decode attention runs the oracle's BLOCKED form (the structure of
    decode.cpp:942-985, within 2x of the compiled reference per core; the token-at-a-time form is the checker).
    3 warm-ups + median of 10 per op, bounded to about a minute."""
    import oracle
    lib_path = oracle.build(native=True, out=os.path.join("/tmp", f"libsgl_oracle_native_{os.getpid()}.so"), force=True)
    lib = oracle.load(lib_path)
    cores = min(int(lib.orc_num_threads()), _effective_cpus())
    lib.orc_set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    Hq, Hkv, D, H, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size
    n_tok = B * ctx + 1
    q = torch.randn(B, Hq, D, generator=g).bfloat16()
    kb = torch.randn(n_tok, Hkv, D, generator=g).bfloat16()
    vb = torch.randn(n_tok, Hkv, D, generator=g).bfloat16()
    r2t = (torch.randperm(n_tok - 1, generator=g) + 1).to(torch.int32).view(B, ctx)
    o = torch.zeros(B, Hq, D, dtype=torch.bfloat16)
    logits = torch.zeros(B, Hq, 8, D + 1)
    rpi, seq = torch.arange(B), torch.full((B,), ctx)
    budget_end = time.perf_counter() + 60.0

    def med(fn, warm=3, reps=10):
        """3 warm-ups, then the median of 10 runs (fewer, but never under 3, if the budget of the whole baseline runs out)."""
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() > budget_end and len(ts) >= 3:
                break
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    t_attn, n_attn = med(lambda: oracle.decode_attention(q, kb, vb, o, None, None, None, logits, r2t, rpi, seq,
                                                         D ** -0.5, 0.0, lib=lib, blocked=True))
    t_lin, n_lin = 0.0, 10
    for (K, N) in [(H, (Hq + 2 * Hkv) * D), (Hq * D, H), (H, 2 * I), (I, H)]:
        x = torch.randn(B, K, generator=g).bfloat16()
        w = ((torch.rand(N, K, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
        sb = torch.rand(N, generator=g) * 1e-2
        xq = torch.empty(B, K, dtype=torch.uint8)
        xs = torch.empty(B)

        def lin():
            oracle.per_token_quant_fp8(x, xq, xs, lib=lib)
            oracle.fp8_scaled_mm(xq.view(torch.float8_e4m3fn), w.t(), xs, sb, torch.bfloat16, lib=lib)

        t, n = med(lin)
        t_lin += t
        n_lin = min(n_lin, n)
    # The reference's own CPU formula for the quant-GEMM (fp8_utils.py:479-507 `_apply_fallback_scaled_mm`: the unscaled
    # product in fp32, then `* x_scale * weight_scale.t()` (+ bias) -> dtype) with plain torch on the same host cores.
    # `torch._scaled_mm` itself is a scalar loop on CPU in this torch (60 s for one gate_up call), so the product is
    # `torch.matmul` on operands dequantised ONCE outside the timed region (weights are static): in fp32 (the formula's
    # accumulator type, bit-comparable) and in bf16 (AMX where the host has it; e4m3 values are exact in bf16, the product is
    # rounded to bf16 before the scales -- the fastest thing plain torch can do, not bit-comparable).  VERDICT r3 #6.
    gemm_torch = {}
    try:
        old_threads = torch.get_num_threads()
        torch.set_num_threads(cores)
        t32 = t16 = 0.0
        n32 = 10
        for (K, N) in [(H, (Hq + 2 * Hkv) * D), (Hq * D, H), (H, 2 * I), (I, H)]:
            x = torch.randn(B, K, generator=g).bfloat16()
            w = ((torch.rand(N, K, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
            sb = (torch.rand(N, 1, generator=g) * 1e-2)
            wf, wb = w.float(), w.bfloat16()

            def quant():  # torch_per_token_quant_fp8 of sgl-kernel/tests/test_per_token_quant_fp8.py:14-22
                xf = x.float()
                s_ = xf.abs().amax(dim=1, keepdim=True).clamp(min=1e-10) / 448.0
                return (xf / s_).clamp(-448.0, 448.0).to(torch.float8_e4m3fn), s_

            def lin32():
                q_, s_ = quant()
                return ((q_.float() @ wf.t()) * s_ * sb.t()).bfloat16()

            def lin16():
                q_, s_ = quant()
                return ((q_.bfloat16() @ wb.t()).float() * s_ * sb.t()).bfloat16()

            a_, n_ = med(lin32)
            t32 += a_
            n32 = min(n32, n_)
            a_, n_ = med(lin16)
            t16 += a_
            n32 = min(n32, n_)
            del wf, wb
        torch.set_num_threads(old_threads)
        gemm_torch = {"gemm_torch_ms": round(t32 * 1e3, 2), "gemm_torch_bf16_ms": round(t16 * 1e3, 2),
                      "gemm_port_ms": round(t_lin * 1e3, 2), "gemm_torch_reps": n32}
    except Exception as e:  # noqa: BLE001  (a torch build without CPU fp8 casts: the port's number stands)
        gemm_torch = {"gemm_torch_error": repr(e)[:200]}
    t_lin_best = min([t_lin] + [gemm_torch[k] / 1e3 for k in ("gemm_torch_ms", "gemm_torch_bf16_ms") if k in gemm_torch])
    t_layer = t_attn + t_lin_best
    which = ("the C port" if t_lin_best == t_lin else
             "torch.matmul in bf16 on dequantised operands" if gemm_torch.get("gemm_torch_bf16_ms", 1e30) / 1e3 == t_lin_best
             else "torch.matmul in fp32 on dequantised operands")
    out = {"value": round(B / (t_layer * n_layers_full), 3), "unit": "tokens/s", "cores": cores, "kind": "port",
           "sample": f"1 of {n_layers_full} layers (decode attention {t_attn * 1e3:.1f} ms, blocked form of decode.cpp:942-985, "
                     f"+ 4 quant-GEMMs {t_lin_best * 1e3:.1f} ms [{which}; the reference's fallback formula "
                     f"fp8_utils.py:479-507; C port {t_lin * 1e3:.1f} ms] at bs={B}, ctx={ctx}; 3 warm-ups, median of "
                     f"{min(n_attn, n_lin)}), scaled x{n_layers_full}; norms/LM head not counted; host threads = "
                     f"min(OpenMP threads, affinity, cgroup quota)"}
    out.update(gemm_torch)
    return out


def reference_cpu_container():
    """The compiled reference's own CPU timing, taken in the build container (the reference cannot travel)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_cpu_container.json")))
        p = [x for x in d["points"] if x["op"] == "decode_attention_cpu" and x["S"] == 2048][0]
        return {"op": "decode_attention_cpu (sgl-kernel/csrc/cpu/decode.cpp, compiled from /root/reference)",
                "shape": f"bs={p['B']} ctx={p['S']} {p['Hq']}/{p['Hkv']}/{p['D']} bf16", "ms": p["ms"],
                "kv_GBps": p["kv_GBps"], "cores": d["cores"], "cpu": d["cpu"], "method": d["method"],
                "source": "tools/time_reference_cpu.py -> profiles/r02_reference_cpu_container.json (build container)"}
    except (OSError, KeyError, IndexError, ValueError):
        return None


def time_extend_kernel(cfg, device, tp):
    """`roofline_extend`: the ragged prefix + extend ("prefill") attention kernel of north_star, event-timed in this run like
    `roofline_gemm` (one HIP graph of 8 launches per case, median of 5 replays / 8).  Cases: the TTFT shape
    (one request, 1024 new tokens, no prefix), 4096 new tokens, and a radix-cache hit (4 requests x 512 new tokens behind
    2048 cached tokens each, gathered through a random-permutation page table) -- geometry of this model per rank, bf16,
    through the Triton-form entry point (extend_attention.py:306-438).  flops = 4 B Hq D (L P + L (L + 1) / 2) against the
    2.5 PFLOP/s dense bf16 MFMA peak; the prefix stage's gathered K/V bytes (B P Hkv 2 D 2) over the whole launch are a lower
    bound of its gather rate.  Round 4 adds a short suffix behind a long cached prefix (one request, 128 new tokens after 4096:
    a later chunk of a chunked prefill, a multi-turn radix hit) -- the launch that has too few (query block, head group) items
    to fill the chip; the attention backend passes the host's prefix bound + scratch and the kernel cuts every item's keys into
    ranges over several workgroups (`us` = that call, `us_unsplit` = the same call without them)."""
    from sglang_npu_amd import ops
    Hq, Hkv, D = cfg.num_attention_heads // tp, cfg.get_num_kv_heads(tp), cfg.head_dim
    out = {"unit": "TFLOP/s", "peak": 2500.0, "bound": "mfma", "kernel": "extend_mfma_kernel (csrc/attention_extend.hip)",
           "geometry": f"Hq={Hq} Hkv={Hkv} D={D} bf16, causal", "cases": []}
    scratch = ops.ExtendPartsScratch(device)
    for (B, L, P, name) in [(1, 1024, 0, "ttft_1024"), (1, 4096, 0, "prefill_4096"), (4, 512, 2048, "radix_hit_4x512_after_2048"),
                            (1, 128, 4096, "suffix_128_after_4096")]:
        g = torch.Generator(device=device).manual_seed(B * 1000 + L + P)
        n_tok = B * (L + P) + 1
        kb = torch.randn(n_tok, Hkv, D, device=device, generator=g).bfloat16()
        vb = torch.randn(n_tok, Hkv, D, device=device, generator=g).bfloat16()
        perm = (torch.randperm(n_tok - 1, device=device, generator=g) + 1).to(torch.int32)
        q = torch.randn(B * L, Hq, D, device=device, generator=g).bfloat16()
        ke = torch.randn(B * L, Hkv, D, device=device, generator=g).bfloat16()
        ve = torch.randn(B * L, Hkv, D, device=device, generator=g).bfloat16()
        o = torch.zeros(B * L, Hq, D, dtype=torch.bfloat16, device=device)
        qo_indptr = (torch.arange(B + 1, device=device) * L).to(torch.int32)
        kv_indptr = (torch.arange(B + 1, device=device) * P).to(torch.int32)
        kv_indices = perm[: B * P].contiguous() if P else torch.zeros(1, dtype=torch.int32, device=device)

        def timed(kw):
            def run(n=8):
                for _ in range(n):
                    ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True, None, L,
                                             D ** -0.5, 0.0, **kw)

            s = torch.cuda.Stream(device=device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run(2)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                run()
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                graph.replay()
                en.record()
                torch.cuda.synchronize()
                ts.append(st.elapsed_time(en) * 1e3 / 8)
            ts.sort()
            del graph
            return ts[len(ts) // 2]

        # as the attention backend calls it: the host's bound of the prefix lengths + its scratch
        us = timed(dict(max_prefix_len=P, parts_scratch=scratch))
        flops = 4.0 * B * Hq * D * (L * P + L * (L + 1) / 2)
        case = {"name": name, "B": B, "extend_len": L, "prefix_len": P, "us": round(us, 2),
                "TFLOPs": round(flops / us / 1e6, 1), "frac_mfma": round(flops / us / 1e6 / 2500.0, 4)}
        if P:
            case["prefix_kv_gathered_GBps_lower_bound"] = round(B * P * Hkv * 2 * D * 2 / us / 1e3, 1)
        if name.startswith("suffix"):
            case["us_unsplit"] = round(timed({}), 2)
        out["cases"].append(case)
    out["note"] = ("event-timed in this run: one HIP graph of 8 launches per case, median of 5 replays / 8; counters of the same "
                   "kernel (SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_INST_ANY): profiles/r05_extend_pmc.txt; per-tile instruction budget from the ISA: profiles/r05_extend_valu_budget.txt")
    return out


def time_awq_decode_gemms(layers, B, device):
    """`roofline_gemm` of the AWQ config (SURVEY 8d config 4, M in {1, 64, 512}): the four INT4 g128 dequant-GEMMs of a layer, each
    as one captured graph that runs it once per layer on that layer's own packed weights, through the kernel AWQLinearMethod.apply
    picks for the row count (M <= 64: the k-packed weight streamer; M = 512: 128 x 128 tiles on the fp16 MFMA with the INT4
    unpacked in registers); bytes = K*N/2 (nibbles) + K/G*N/2 (zeros) + K/G*N*2 (scales) + 2*M*K + 2*M*N, flops = 2*M*N*K
    against the 2.5 PFLOP/s dense fp16 MFMA peak."""
    from sglang_npu_amd import ops
    names = [("qkv", lambda l: l.self_attn.qkv_proj), ("o", lambda l: l.self_attn.o_proj),
             ("gate_up", lambda l: l.mlp.gate_up_proj), ("down", lambda l: l.mlp.down_proj)]
    out = {"unit_hbm": "GB/s", "peak_hbm": HBM_PEAK_GBPS, "unit_mfma": "TFLOP/s", "peak_mfma": 2500.0,
           "quant": "awq int4 g128 (k-packed copy, fp16 activations)", "shapes": []}
    g = torch.Generator(device=device).manual_seed(7)
    tot_us, tot_bytes = 0.0, 0
    for name, pick in names:
        lins = [pick(l) for l in layers]
        wp, sz, G = lins[0].awq_packed
        N, K = int(lins[0].awq_out_features), int(getattr(lins[0], "input_size_per_partition", lins[0].input_size))
        row = {"name": name, "K": K, "N": N}
        for M, key in ((1, "m1"), (B, "decode"), (512, "m512")):
            x = torch.randn(M, K, device=device, generator=g).half()
            fn = ops.awq_gemm_packed if M <= 64 else ops.awq_gemm_packed_tiled

            def run():
                for lin in lins:
                    fn(x, lin.awq_packed[0], lin.awq_packed[1], lin.awq_packed[2])

            s = torch.cuda.Stream(device=device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                run()
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                graph.replay()
                en.record()
                torch.cuda.synchronize()
                ts.append(st.elapsed_time(en) * 1e3 / len(lins))
            ts.sort()
            us = ts[len(ts) // 2]
            nbytes = K * N // 2 + (K // G) * N // 2 + (K // G) * N * 2 + 2 * M * K + 2 * M * N
            flops = 2.0 * M * N * K
            row[key] = {"M": M, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1),
                        "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4),
                        "TFLOPs": round(flops / us / 1e6, 1), "frac_mfma": round(flops / us / 1e6 / 2500.0, 4)}
            if key == "decode":
                tot_us += us
                tot_bytes += nbytes
            del graph
        out["shapes"].append(row)
    out["decode_all_four"] = {"us": round(tot_us, 2), "frac_hbm": round(tot_bytes / tot_us / 1e3 / HBM_PEAK_GBPS, 4)}
    out["note"] = ("event-timed in this run: one HIP graph per (shape, M) with one call per layer on that layer's packed weights, "
                   "median of 5 replays / layers; includes the split-K finalize launch where the kernel uses it")
    return out


def time_decode_gemms(net, cfg, B, device, tp):
    """`roofline_gemm`: the four FP8 decode GEMMs of a layer, each as ONE captured HIP graph that runs it once per
    layer on that layer's own weights (32 different weight matrices: nothing is served from L2 / Infinity Cache the
    way a loop over one matrix would be), timed with HIP events over 5 replays; M = B rows for the HBM fraction
    (bytes = M*K + K*N + 2*M*N + 4*(M+N)) and M = 1024 (the TTFT pass) and 4096 for the MFMA fraction
    (2*M*N*K flops against the 5 PFLOP/s dense FP8 peak); M = 1 and M = 512 complete SURVEY 8d's list {1, 64, 512, 4096} (every
    row carries both fractions: which one binds follows from M)."""
    from sglang_npu_amd import ops
    layers = list(net.layers)
    if getattr(layers[0].mlp.gate_up_proj, "awq_packed", None) is not None:
        return time_awq_decode_gemms(layers, B, device)
    w0 = getattr(layers[0].mlp.gate_up_proj, "weight", None)
    if w0 is None or not (w0.dtype == torch.float8_e4m3fn or ops.is_wshuffled(w0)):  # row-major [K, N] view or fragment-major uint8
        return None
    names = [("qkv", lambda l: l.self_attn.qkv_proj), ("o", lambda l: l.self_attn.o_proj),
             ("gate_up", lambda l: l.mlp.gate_up_proj), ("down", lambda l: l.mlp.down_proj)]
    out = {"unit_hbm": "GB/s", "peak_hbm": HBM_PEAK_GBPS, "unit_mfma": "TFLOP/s", "peak_mfma": 5000.0, "shapes": []}
    g = torch.Generator(device=device).manual_seed(7)
    for name, pick in names:
        lins = [pick(l) for l in layers]
        K, N = ops.fp8_weight_kn(lins[0].weight)
        row = {"name": name, "K": int(K), "N": int(N)}
        # M in {1, 64, 512, 4096} (sgl-kernel/benchmark/bench_fp8_gemm.py:22-34), + 1024 (the TTFT pass)
        for M, key in ((1, "m1"), (B, "decode"), (512, "m512"), (1024, "prefill_1024"), (4096, "prefill")):
            a = ((torch.rand(M, K, device=device, generator=g) - 0.5) * 8).to(torch.float8_e4m3fn)
            sa = torch.rand(M, 1, device=device, generator=g) * 1e-2 + 1e-3

            def run():
                for lin in lins:
                    ops.fp8_scaled_mm(a, lin.weight, sa, lin.weight_scale, torch.bfloat16)

            def run_silu():  # what the prefill pass launches for gate_up: SiLU * mul in the GEMM's epilogue ([M, N/2] written)
                for lin in lins:
                    ops.fp8_scaled_mm_silu_mul(a, lin.weight, sa, lin.weight_scale, torch.bfloat16)

            fused = (name == "gate_up" and M > B
                     and ops.fp8_scaled_mm_silu_mul(a, lins[0].weight, sa, lins[0].weight_scale, torch.bfloat16) is not None)
            s = torch.cuda.Stream(device=device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                run()
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                graph.replay()
                en.record()
                torch.cuda.synchronize()
                ts.append(st.elapsed_time(en) * 1e3 / len(lins))
            ts.sort()
            us = ts[len(ts) // 2]
            nbytes = M * K + K * N + 2 * M * N + 4 * (M + N)
            flops = 2.0 * M * N * K
            row[key] = {"M": M, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1),
                        "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4),
                        "TFLOPs": round(flops / us / 1e6, 1), "frac_mfma": round(flops / us / 1e6 / 5000.0, 4)}
            del graph
            if fused:  # the same flops, timed the same way, in the form the model runs at this size
                with torch.cuda.stream(s):
                    run_silu()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):
                    run_silu()
                graph.replay()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    st.record()
                    graph.replay()
                    en.record()
                    torch.cuda.synchronize()
                    ts.append(st.elapsed_time(en) * 1e3 / len(lins))
                ts.sort()
                us = ts[len(ts) // 2]
                row[key]["with_silu_mul_epilogue"] = {"us": round(us, 2), "TFLOPs": round(flops / us / 1e6, 1),
                                                      "frac_mfma": round(flops / us / 1e6 / 5000.0, 4)}
                del graph
        out["shapes"].append(row)
    dec_bytes = sum(r["decode"]["M"] * r["K"] + r["K"] * r["N"] + 2 * r["decode"]["M"] * r["N"] for r in out["shapes"])
    dec_us = sum(r["decode"]["us"] for r in out["shapes"])
    out["decode_all_four"] = {"us": round(dec_us, 2), "frac_hbm": round(dec_bytes / dec_us / 1e3 / HBM_PEAK_GBPS, 4)}
    out["note"] = ("event-timed in this run: one HIP graph per shape with one call per layer on that layer's weights, "
                   "median of 5 replays / layers; includes split-K finalize launches where the kernel uses them")
    return out


def allreduce_latency_vs_size(tp_group, device, world):
    """All-reduce latency against message size, per data plane (the sizes of
    /root/reference/test/srt/test_custom_allreduce.py:59-68 extended to 128 MiB): RCCL through torch.distributed, the P2P
    kernel over IPC buffers while the message fits its staging area, QuickReduce when ROCM_QUICK_REDUCE_QUANTIZATION enables
    it.  bf16, integer-valued payload (every backend must return the exact sum), 3 warm-ups + median of 20 event-timed
    calls, max over ranks.  `bound_us` = 2 (n-1)/n x bytes over the 7 xGMI links x 153 GB/s a GPU has (each rank moves
    that much in a reduce-scatter + all-gather).  Outside the timed region of the headline."""
    import torch.distributed as dist
    sizes = [512 << (2 * i) for i in range(10)]  # 512 B, 2 KiB, ... 128 MiB
    planes = [("rccl", lambda x: (dist.all_reduce(x, group=tp_group.device_group), x)[1])]
    ca, qr = tp_group.ca_comm, tp_group.qr_comm
    if ca is not None and not ca.disabled:
        planes.append(("p2p", lambda x: ca.custom_all_reduce(x)))
    if qr is not None and not qr.disabled:
        planes.append(("quickreduce", lambda x: qr.quick_all_reduce(x) if qr.should_quick_allreduce(x) else None))
    link_bw = 7 * 153e9
    rows = []
    for nbytes in sizes:
        n = nbytes // 2
        g = torch.Generator(device=device).manual_seed(1 + tp_group.rank_in_group)
        src = torch.randint(-3, 4, (n,), device=device, generator=g).to(torch.bfloat16)
        ref = src.clone()
        dist.all_reduce(ref, group=tp_group.device_group)
        row = {"bytes": nbytes, "bound_us": round(2 * (world - 1) / world * nbytes / link_bw * 1e6, 3)}
        for name, fn in planes:
            # every rank issues the same group collectives whatever fails locally: a local exception in the
            # timing loop becomes a status, the ranks agree on it (MIN) BEFORE the MAX all-reduce of the medians
            err, med, exact, takes = None, 0.0, None, True
            try:
                x = src.clone()
                out = fn(x)
                if out is None:  # this plane does not take the size (staging area / thresholds): a shape-only decision,
                    takes = False  # identical on all ranks
                else:
                    torch.cuda.synchronize(device)
                    exact = bool(torch.equal(out, ref)) if name != "quickreduce" else None
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {str(e)[:120]}"
            if not takes:
                continue
            ok = torch.tensor([0 if err else 1], device=device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=tp_group.device_group)
            if int(ok.item()) == 1:
                try:
                    ts = []
                    for it in range(23):
                        x.copy_(src)
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        fn(x)
                        b.record()
                        b.synchronize()
                        if it >= 3:
                            ts.append(a.elapsed_time(b) * 1e3)
                    ts.sort()
                    med = ts[len(ts) // 2]
                except Exception as e:  # noqa: BLE001  (RCCL raises on every rank or none; the P2P kernels fail closed)
                    err = f"{type(e).__name__}: {str(e)[:120]}"
                t = torch.tensor([med if err is None else -1.0], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=tp_group.device_group)
                bad = torch.tensor([1 if err else 0], device=device, dtype=torch.int32)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=tp_group.device_group)
                if int(bad.item()) == 0:
                    row[name + "_us"] = round(float(t.item()), 2)
                    if exact is not None:
                        row[name + "_exact"] = exact
                    continue
            row[name + "_error"] = err or "failed on another rank"
        rows.append(row)
    return {"dtype": "bf16", "world": world, "link_model": "7 xGMI links x 153 GB/s per GPU", "method": "3 warm-ups + median of 20 "
            "event-timed calls per (size, data plane), max over ranks", "planes": [p[0] for p in planes], "points": rows,
            "quickreduce_parity": "parity-unpinned: the QuickReduce codec is checked bit for bit against oracle/quick_reduce.py, a "
                                  "restatement of quick_all_reduce.cuh:71-440 that no reference fixture pins, and against the "
                                  "reference test's bound (test_quick_allreduce.py:131-165); rccl / p2p are exact sums"}


def other_configs():
    """BASELINE configs 2 (bf16), 4 (Llama-2-7B AWQ) and 5 (Llama-3-70B FP8, ONE rank of TP = 8 with the collectives
    stubbed) as short child runs of this script, started BEFORE this process touches the GPU; each contributes its
    whole-step time, TTFT and attention roofline fraction to the headline's JSON line (`other_configs`)."""
    import subprocess
    runs = [("config 2: llama3-8b bf16 TP=1", ["--quant", "none"]),
            ("config 4: llama2-7b AWQ INT4 g128 TP=1", ["--model", "llama2-7b", "--quant", "awq"]),
            ("config 5 (ONE rank of TP=8 on one GPU: the real rank's kernel sequence incl. the fused all-reduce + norm on a one-rank "
             "communicator, no peers' bytes -- not a job number): llama3-70b w8a8_fp8",
             ["--model", "llama3-70b", "--emulate-tp", "8"])]
    out = []
    for name, extra in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
               "--no-other-configs"] + extra
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            out.append({"config": name, "ms_per_step": d["ms_per_step"], "tokens_per_s": d["value"], "call_order": d.get("call_order"),
                        "dropin_ms_per_step": d.get("dropin_ms_per_step"), "fused_ms_per_step": d.get("fused_ms_per_step"),
                        "ttft_ms_p50": d.get("ttft_ms_p50"), "decode_attention_frac_hbm": d["roofline"]["frac"],
                        "decode_attention_us": d["roofline"].get("avg_launch_us"),
                        "workload": d["config"]["workload"]})
            if "launch_includes" in d["roofline"]:
                out[-1]["decode_attention_launch_includes"] = d["roofline"]["launch_includes"]
            rg = d.get("roofline_gemm")
            if isinstance(rg, dict) and "decode_all_four" in rg:  # the config's own GEMM roofline (AWQ: config 4)
                out[-1]["roofline_gemm_decode"] = {"all_four": rg["decode_all_four"],
                                                   "shapes": {r["name"]: {k: v for k, v in r.items() if isinstance(v, dict)}
                                                              for r in rg.get("shapes", []) if "decode" in r}}
        except Exception as e:  # a side measurement: never take the headline down with it
            out.append({"config": name, "error": f"{type(e).__name__}: {str(e)[:200]}"})
    return out


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tail(path, n=25):
    try:
        return "".join(open(path, errors="replace").readlines()[-n:])
    except OSError:
        return ""


def _run_rank_children(cmd, n, env, tag, timeout_s, log_dir):
    """Start `cmd` once per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, each child the leader of its
    own process group), wait for all of them.  The first non-zero exit or the deadline ends the others: SIGTERM to exactly
    the process groups started here, SIGKILL ten seconds later.  Returns (ok, reason, [(stdout path, stderr path)])."""
    import signal
    import subprocess
    port = _free_port()
    procs, files = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_PORT=str(port))
        out_p, err_p = os.path.join(log_dir, f"{tag}.rank{r}.out"), os.path.join(log_dir, f"{tag}.rank{r}.err")
        files.append((out_p, err_p))
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=e, stdout=open(out_p, "w"), stderr=open(err_p, "w"),
                                      stdin=subprocess.DEVNULL, start_new_session=True))
    deadline = time.time() + timeout_s
    reason, beat = None, time.time() + 30
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            reason = f"{tag}: rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            reason = f"{tag}: not finished after {timeout_s:.0f} s"
            break
        if time.time() > beat:  # (a silent command is taken for a hung one)
            print(f"[bench launcher] {tag}: {sum(c is None for c in codes)} of {n} ranks running", file=sys.stderr, flush=True)
            beat = time.time() + 30
        time.sleep(0.2)
    if reason is not None:
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            live = [p for p in procs if p.poll() is None]
            for p in live:
                try:
                    os.killpg(p.pid, sig)  # the group this launcher created for that rank, nothing else
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.time() + grace
            while time.time() < t_end and any(p.poll() is None for p in live):
                time.sleep(0.1)
    return reason is None, reason, files


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment: be the launcher.

    The reference's harness spawns one process per TP rank itself (bench_one_batch.py:509-545); so does this.  The parent
    never initialises the GPU (no HIP call, no torch.cuda call): it only starts children --
      phase 0  N short-lived probes (sglang_npu_amd._probe_graph_ar): can this job's process-group all-reduce be captured
               into a HIP graph?  (asked in throw-away processes: a failed capture leaves a sticky HIP error);
      phase 1  the N ranks: this script again with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set and the
               probe's answer in SGL_MI355_BENCH_AR_GRAPH.
    Rank 0's JSON line is relayed on stdout with a `launcher` object added.  Exit code 0 only if every rank exits 0 -- except
    that a failure AFTER rank 0 has checkpointed the finished headline measurement (SGL_MI355_BENCH_PARTIAL) still prints
    that line, marked `incomplete`: a side measurement must not take the timed K steps down with it."""
    import shutil
    import tempfile
    n, t0 = args.gpus, time.time()
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what hipIpcGetMemHandle / RCCL need on this host driver
    if env.get("SGL_MI355_SHARE_GPU"):  # rehearsal on a 1-GPU box: RCCL refuses two ranks on one device, gloo carries the group
        env.setdefault("SGL_MI355_DIST_BACKEND", "gloo")
    log_dir = tempfile.mkdtemp(prefix="sgl_mi355_bench_")
    budget = float(env.get("SGL_MI355_BENCH_LAUNCH_TIMEOUT", "1100"))
    info = {"mode": "self-launched: the parent started one child process per rank and never touched the GPU "
                    "(bench_one_batch.py:509-545 does the same)", "ranks": n}
    try:
        graph_ok = False
        if not args.no_graph:
            ok, why, files = _run_rank_children([sys.executable, "-m", "sglang_npu_amd._probe_graph_ar"], n, env, "probe",
                                                min(240.0, budget / 3), log_dir)
            graph_ok = ok
            info["probe_graph_capturable_allreduce"] = ok
            if not ok:
                info["probe_note"] = why
                print(f"[bench launcher] {why}: the step runs eagerly\n{_tail(files[0][1], 8)}", file=sys.stderr, flush=True)
        partial = os.path.join(log_dir, "rank0_partial.json")
        env_r = dict(env, SGL_MI355_BENCH_AR_GRAPH="1" if graph_ok else "0", SGL_MI355_BENCH_PARTIAL=partial,
                     SGL_MI355_BENCH_T0=str(t0), SGL_MI355_BENCH_BUDGET_S=str(budget))
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
        ok, why, files = _run_rank_children(cmd, n, env_r, "ranks", max(60.0, budget - (time.time() - t0)), log_dir)
        for r, (_, err_p) in enumerate(files):  # the ranks' diagnostics go where ours go
            txt = _tail(err_p, 60 if not ok else 12)
            if txt.strip():
                print(f"----- rank {r} stderr (tail) -----\n{txt}", file=sys.stderr, flush=True)
        line = None
        try:
            line = [l for l in open(files[0][0]).read().splitlines() if l.startswith("{")][-1]
        except (OSError, IndexError):
            pass
        if line is None and os.path.exists(partial):
            line = open(partial).read().strip() or None
            info["incomplete"] = f"{why or 'rank 0 printed no line'}; this is rank 0's checkpoint after the timed region"
        if line is None:
            print(f"[bench launcher] no result: {why or 'rank 0 printed no JSON line'}", file=sys.stderr, flush=True)
            return 1
        d = json.loads(line)
        if not ok and "incomplete" not in info:
            info["incomplete"] = why
        info["wall_s"] = round(time.time() - t0, 1)
        d["launcher"] = info
        print(json.dumps(d), flush=True)
        return 0
    finally:
        shutil.rmtree(log_dir, ignore_errors=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    others = None
    if (world == 1 and args.gpus == 1 and not args.no_other_configs and args.model == "llama3-8b" and
            args.quant == "w8a8_fp8" and args.emulate_tp == 0 and args.layers is None and args.kv_dtype == "auto"):
        others = other_configs()
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))  # becomes the parent of N rank processes; never touches the GPU itself
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE={world}: they must agree")
    use_graph = not args.no_graph
    if use_graph and world > 1 and os.environ.get("SGL_MI355_BENCH_AR_GRAPH") in ("0", "1"):
        use_graph = os.environ["SGL_MI355_BENCH_AR_GRAPH"] == "1"  # the launcher's phase 0 already asked
        if not use_graph and rank == 0:
            print("[bench] all-reduce is not graph-capturable with this backend; running the step eagerly", file=sys.stderr)
    elif use_graph and world > 1:
        # Ask a throw-away child per rank whether the communicator's all-reduce can be stream-captured
        # (RCCL can; a failed capture would leave a sticky HIP error in THIS process, so we never try here).
        import subprocess
        env = dict(os.environ)
        env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
        try:
            rc = subprocess.run([sys.executable, "-m", "sglang_npu_amd._probe_graph_ar"], env=env, cwd=ROOT,
                                timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        except Exception:
            rc = 1
        use_graph = rc == 0
        if not use_graph and rank == 0:
            print("[bench] all-reduce is not graph-capturable with this backend; running the step eagerly", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if os.environ.get("SGL_MI355_SHARE_GPU"):  # rehearsal on a 1-GPU box: every rank uses device 0
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
                         f"(a rehearsal with all ranks on ONE GPU: SGL_MI355_SHARE_GPU=1)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    from sglang_npu_amd import _lib
    from sglang_npu_amd.distributed import init_distributed_environment
    _lib.lib()  # fail loudly if the HIP library is missing
    tp_group = init_distributed_environment(device=device)
    if args.emulate_tp > 1 and world == 1:
        from sglang_npu_amd.distributed import GroupCoordinator, set_tp_group
        tp_group = GroupCoordinator(None, 0, args.emulate_tp, device)
        tp_group.stub_all_reduce = True  # the plain all-reduce and all-gather become identities ...
        # ... but the row-parallel layers still hand their output / split-K partials to the fused all-reduce + add + RMSNorm
        # (+ FP8 quant) kernel, on a communicator whose only rank is this one: the launches of a real rank (no standalone
        # finalize kernel), minus the peers' bytes
        from sglang_npu_amd.distributed import CustomAllreduce
        if os.environ.get("SGL_MI355_EMULATE_PLAIN_STUB", "0") in ("", "0"):
            tp_group.ca_comm = CustomAllreduce.single_rank(device)
            tp_group.fuse_under_stub = True
        set_tp_group(tp_group)
    tp = tp_group.world_size
    dist_on = world > 1  # a real multi-process job (false under --emulate-tp)

    net, cfg, runner, backend, max_len = build(args, device, tp)
    loop = DecodeLoop(net, runner, backend, args.batch, args.ctx, device)
    can_fuse = net.fuse_quant          # only the FP8 config has fused producers
    want_fused = can_fuse and args.call_order == "fused"

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def set_call_order(fused):
        """Switch the model between the fused producers and the reference's operator order, rewind the requests to
        ctx (the page table has room for ctx + steps + warmup + 8 positions only) and re-capture the step."""
        net.fuse_quant = bool(fused and can_fuse)
        loop.rewind(args.ctx)
        if use_graph:
            loop.capture()

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            loop.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loop.step()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    kv_esz = 1 if args.kv_dtype in ("fp8_e4m3", "fp8_e5m2") else 2
    partial_path = os.environ.get("SGL_MI355_BENCH_PARTIAL") if rank == 0 else None

    def checkpoint(d):
        """Rank 0 leaves the line as it stands for the launcher (launch_ranks): once the timed region is done, a failure in a
        later side measurement on any rank must not lose it."""
        if partial_path:
            tmp = partial_path + ".tmp"
            with open(tmp, "w") as f:
                f.write(json.dumps(d))
            os.replace(tmp, partial_path)

    # ---- the contract's timed region: W warm-up steps, then exactly K steps between barriers
    set_call_order(want_fused)
    elapsed = timed(args.warmup, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * args.steps / elapsed
    order_name = ("this repo's fused entry points (norm+quant / RoPE+KV-write / SiLU+quant producers, GEMM epilogues inside their "
                  "consumers)" if want_fused else
                  "the reference's operator order, i.e. what an untouched SGLang models/llama.py gets from the drop-in classes")
    out = {
        "metric": "output tokens/s (decode, whole model step) + p50 TTFT", "value": round(value, 1), "unit": "tokens/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("fp8_e4m3 (GEMM) / bf16 (attention" + (", KV)" if kv_esz == 2 else "), " + args.kv_dtype + " KV"))
        if args.quant == "w8a8_fp8" else args.quant,
        "data": "synthetic (dummy-loader random weights, N(0,1) KV, random-permutation page table)",
        "config": {"workload": f"{args.model} {args.quant} decode bs={args.batch} ctx={args.ctx} TP={tp} "
                               f"(token-level paged KV, {'HIP graph replay' if loop.graph is not None else 'eager launches'}); "
                               f"`value` = {order_name}",
                   "global_batch": args.batch, "seq_len": args.ctx, "layers": len(net.layers),
                   "parallelism": f"tp{tp}" + (" (ONE rank emulated on one GPU, fused all-reduce + norm on a one-rank communicator: not a job number)"
                                               if args.emulate_tp > 1 and not dist_on else "")},
        "call_order": "fused" if want_fused else "reference",
        "world_size_seen": torch.distributed.get_world_size() if dist_on else 1,
        "dist_backend": torch.distributed.get_backend() if dist_on else None,
    }
    checkpoint(out)

    # ---- the other call order, and a device-synchronised per-step median (bench_one_batch.py:380-424 shape)
    other = None
    if can_fuse:
        n_other = min(args.steps, 10)
        set_call_order(not want_fused)
        other = timed(2, n_other) / n_other * 1e3
        set_call_order(want_fused)
    fused_ms, dropin_ms = (ms_per_step, other) if want_fused else (other, ms_per_step)
    out.update({
        "value_dropin": round(args.batch / dropin_ms * 1e3, 1) if dropin_ms else None,
        "value_fused": round(args.batch / fused_ms * 1e3, 1) if fused_ms else None,
        "fused_ms_per_step": round(fused_ms, 4) if fused_ms is not None else None,
        "dropin_ms_per_step": round(dropin_ms, 4) if dropin_ms is not None else None,
        "dropin_tokens_per_s": round(args.batch / dropin_ms * 1e3, 1) if dropin_ms else None,
        "dropin_note": "value_dropin / dropin_ms_per_step: the model in the reference's operator order (RMSNorm -> apply() "
                       "[per-token quant + fp8_scaled_mm] -> RoPE -> attn_backend.forward(save_kv_cache=True) -> ... ), what "
                       "SGLang's untouched models/llama.py gets from the drop-in classes (which hand FP8 companions and still-unfinished "
                       "GEMM epilogues / all-reduces to each other as tensors, sglang_npu_amd/deferred.py); value_fused / fused_ms_per_step: the "
                       "same model through this repo's fused entry points (needs its own model file); whichever is not `value` "
                       "ran min(steps,10) steps after 2 warm-ups"})
    checkpoint(out)
    sync_ts = []
    loop.rewind(args.ctx)  # the page table has room for ctx + steps + warmup + 8 positions only
    for _ in range(min(args.steps, 10) + 2):
        barrier()
        t0 = time.perf_counter()
        loop.step()
        torch.cuda.synchronize()
        sync_ts.append((time.perf_counter() - t0) * 1e3)
    sync_ts = sorted(sync_ts[2:])
    median_synced_ms = max_over_ranks(sync_ts[len(sync_ts) // 2])
    loop.rewind(args.ctx + args.warmup + args.steps)   # where the instrumented passes below expect to start

    # ---- all-reduce overhead (N > 1): the same step with the TP all-reduce stubbed to identity, outside the timed
    # region above ((t_with - t_without) / t_with)
    ar_info = None
    if dist_on:
        try:
            tp_group.stub_all_reduce = True
            len_after = loop.cur_len
            loop.rewind(args.ctx)  # the page table has room for ctx + steps + warmup + 8 positions only
            if loop.graph is not None:
                loop.capture()
            n_stub = min(args.steps, 6)
            for _ in range(2):
                loop.step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n_stub):
                loop.step()
            barrier()
            stub_s = time.perf_counter() - t1
            t = torch.tensor([stub_s], device=device, dtype=torch.float64)
            tp_group.stub_all_reduce = False
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            stub_ms = float(t.item()) / n_stub * 1e3
            ar_info = {"ms_per_step_allreduce_stubbed": round(stub_ms, 4),
                       "overhead_frac": round(max(0.0, (ms_per_step - stub_ms) / ms_per_step), 4),
                       "backend": ("p2p kernel over IPC buffers (verified against the process group at start-up), fused "
                                   "with the next add+RMSNorm" if tp_group.ca_comm is not None and not tp_group.ca_comm.disabled
                                   else "rccl (torch.distributed)"),
                       "collectives_per_step": 2 * len(net.layers) + 1,
                       "message_bytes": args.batch * cfg.hidden_size * 2,
                       "step_under_graph_replay": loop.graph is not None}
        finally:
            tp_group.stub_all_reduce = False
            loop.rewind(len_after)
        # which data plane each collective of ONE step takes: counted where Python dispatches (one eager pass)
        from sglang_npu_amd.distributed import DISPATCH_COUNTS
        DISPATCH_COUNTS.clear()
        if loop.graph is not None:
            loop.capture()  # (the instrumented passes below run the real step again) 2 warm-up passes + the captured one
            passes = 3
        else:
            loop.backend.init_forward_metadata(loop.fb)
            loop._forward()
            passes = 1
        torch.cuda.synchronize()
        ar_info["dispatch_per_step"] = {k: v // passes for k, v in DISPATCH_COUNTS.items()}
        try:
            ar_info["latency_vs_size"] = allreduce_latency_vs_size(tp_group, device, world)
        except Exception as e:
            ar_info["latency_vs_size"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        out["allreduce"] = ar_info
        checkpoint(out)

    # ---- roofline of the dominant kernel: paged decode attention (HBM-bound)
    ctx_mid = args.ctx + args.warmup + args.steps + args.steps // 2  # mean sequence length of the instrumented pass
    hq, hkv, d = cfg.num_attention_heads // tp, cfg.get_num_kv_heads(tp), cfg.head_dim
    attn_ms, n_launch = time_attention_kernel(loop, min(args.steps, 8))
    ctx_mid = args.ctx + args.warmup + args.steps + min(args.steps, 8) / 2
    # ... and its cost INSIDE the graph-replayed step, as the difference of two event-timed runs of the same loop over the
    # same sequence lengths: the step as it is, and the step with the decode attention launch left out
    # (MI355AttnBackend.measure_skip_decode_kernel).  The eager launches timed above sit between idle gaps (the host
    # cannot issue ten kernels per layer as fast as the GPU runs them) and read 4-12 % long, box by box; the profiler's
    # average busy time of the kernel in the replayed step is what this difference tracks (it also carries the one kernel
    # boundary that disappears with the launch, ~1.6 us).
    attn_instep_ms = None
    if loop.graph is not None and not dist_on and hasattr(backend, "measure_skip_decode_kernel"):
        try:
            n_d = min(args.steps, 16)
            loop.rewind(args.ctx)
            loop.capture()
            t_full = timed(2, n_d) / n_d * 1e3
            backend.measure_skip_decode_kernel = True
            loop.rewind(args.ctx)
            loop.capture()
            t_skip = timed(2, n_d) / n_d * 1e3
            attn_instep_ms = (t_full - t_skip) / len(net.layers)
        finally:
            backend.measure_skip_decode_kernel = False
            loop.rewind(args.ctx)
            loop.capture()
    alg_bytes = args.batch * ctx_mid * hkv * 2 * d * kv_esz + 4 * args.batch * ctx_mid + 2 * args.batch * hq * 2 * d
    eager_event_ms = attn_ms
    frac_eager = alg_bytes / (eager_event_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    frac_instep, instep_used = None, False
    if attn_instep_ms is not None and attn_instep_ms > 0:
        # the lengths of the difference runs: ctx + 2 warm-ups + n_d steps, mean ctx + 2 + n_d / 2
        ctx_d = args.ctx + 2 + min(args.steps, 16) / 2
        alg_d = args.batch * ctx_d * hkv * 2 * d * kv_esz + 4 * args.batch * ctx_d + 2 * args.batch * hq * 2 * d
        frac_instep = alg_d / (attn_instep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        # the difference of two short timed runs carries their noise and second-order effects (the skipped pass
        # feeds uninitialised rows downstream, one kernel boundary disappears).  It is the primary figure only while it
        # agrees with the conservative one -- events around each eager launch, which read long by the host's launch gaps,
        # never short: the in-step time must lie within [0.85, 1.02] x the eager time.  Otherwise the eager figure stands.
        if 0.85 * eager_event_ms <= attn_instep_ms <= 1.02 * eager_event_ms:
            alg_bytes, attn_ms, instep_used = alg_d, attn_instep_ms, True
    achieved = alg_bytes / (attn_ms * 1e-3) / 1e9
    # HBM traffic of that kernel from the committed rocprofv3 --pmc passes (bench.py cannot run the profiler on
    # itself): measured bytes / algorithmic bytes of the same kernel at the same geometry, applied to this launch.
    traffic, traffic_src = None, None
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                          PMC_SUMMARY)))
        if tp == 1 and args.batch == 64 and (hq, hkv, d) == (32, 8, 128) and kv_esz == 2:
            traffic = int(alg_bytes * pmc["traffic_over_algorithmic"])
            traffic_src = (f"profiles/{PMC_SUMMARY}: FETCH_SIZE x2 (gfx950) + WRITE_SIZE of this kernel's launches INSIDE "
                           f"the profiled `bench.py --steps 8 --warmup 2` step (tools/exp/prof_bench_r05.sh, separate --pmc "
                           f"passes), ratio {pmc['traffic_over_algorithmic']} to the algorithmic bytes of the lengths the run visits")
    except (OSError, KeyError, ValueError):
        pass
    out.update({
        "median_step_ms_synced": round(median_synced_ms, 4),
        "roofline": {"bound": "hbm", "kernel": "decode_mfma_pair_kernel / decode_mfma_kernel (paged decode attention)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "avg_launch_us": round(attn_ms * 1e3, 2), "launches_timed": n_launch,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "method": ("in-step: (graph-replayed step - the same step without the decode attention launch) / layers, "
                                "HIP events on the launch stream around min(steps,16) replays each; accepted because it lies "
                                "within [0.85, 1.02] x the eager per-launch event time"
                                if instep_used else
                                "HIP events around each eager launch of the kernel on the launch stream"
                                + ("" if frac_instep is None else " (the in-step difference disagreed with it by more than the "
                                                                   "stated tolerance and is only reported)")),
                     "eager_event_launch_us": round(eager_event_ms * 1e3, 2),
                     "frac_eager_events": round(frac_eager, 4),
                     "instep_difference_us": round(attn_instep_ms * 1e3, 2) if attn_instep_ms else None,
                     "frac_instep_difference": round(frac_instep, 4) if frac_instep is not None else None,
                     "rocprofv3_avg_busy_us_same_command": profiled_attn_us()},
    })
    checkpoint(out)
    if time_attention_kernel.merged_launches:
        out["roofline"]["launch_includes"] = ("kv-split merge + per-token FP8 quant of the output (one launch: "
                                              "sgl_mi355_decode_attention_merged); the bytes are the KV bytes alone")
    try:
        out["roofline_gemm"] = time_decode_gemms(net, cfg, args.batch, device, tp)
    except Exception as e:  # a diagnostic: never take the headline down with it
        out["roofline_gemm"] = {"error": f"{type(e).__name__}: {e}"}
    if args.quant != "awq" and not dist_on:
        try:
            out["roofline_extend"] = time_extend_kernel(cfg, device, tp)
        except Exception as e:  # a diagnostic: never take the headline down with it
            out["roofline_extend"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        ttft_ms, ttft_len = time_ttft(net, runner, backend, device)
        if dist_on:
            t = torch.tensor([ttft_ms], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ttft_ms = float(t.item())
        out["ttft_ms_p50"] = round(ttft_ms, 3)
        out["config"]["ttft"] = f"bs=1, input_len={ttft_len}, empty prefix, eager"
        if not dist_on:  # short prompts: eager (host-bound) and replayed from a captured graph of that token count
            t128, _ = time_ttft(net, runner, backend, device, input_len=128)
            out["ttft_ms_p50_128_eager"] = round(t128, 3)
            out["ttft_ms_p50_128"] = round(time_ttft_graph(net, runner, backend, device, 128), 3)
            out["config"]["ttft_128"] = ("bs=1, input_len=128, empty prefix; ttft_ms_p50_128 = replay of a HIP graph captured at 128 "
                                         "tokens (harness.PrefillGraphRunner), _eager = op-by-op launches")
            # the headline prompt length replayed the same way (informational: ttft_ms_p50 stays the eager pass, which is what a
            # drop-in backend gets from SGLang -- cuda_graph_runner.py captures decode only)
            out["ttft_ms_p50_graph"] = round(time_ttft_graph(net, runner, backend, device, ttft_len, reps=5), 3)
            # a short message behind a cached conversation (radix hit): 128 new tokens after as long a prefix as the bench's
            # context holds; the extend kernel's KV-range parts against the same pass without them
            pfx = max(0, min(args.ctx - 128, 4096))
            if pfx >= 1024:
                t_hit, _ = time_ttft(net, runner, backend, device, input_len=128, prefix_len=pfx)
                g_hit = time_ttft_graph(net, runner, backend, device, 128, reps=7, prefix_len=pfx)
                parts, backend._extend_parts = backend._extend_parts, None
                try:
                    t_hit_plain, _ = time_ttft(net, runner, backend, device, input_len=128, prefix_len=pfx)
                    g_hit_plain = time_ttft_graph(net, runner, backend, device, 128, reps=7, prefix_len=pfx)
                finally:
                    backend._extend_parts = parts
                out["ttft_ms_p50_128_after_prefix"] = {
                    "prefix_len": pfx, "eager_ms": round(t_hit, 3), "eager_ms_without_kv_range_parts": round(t_hit_plain, 3),
                    "graph_ms": round(g_hit, 3), "graph_ms_without_kv_range_parts": round(g_hit_plain, 3),
                    "note": "the eager pass is host-bound (~4.6 ms of launches); graph_ms = the same pass replayed from a graph "
                            "captured for this prefix bound (harness.PrefillGraphRunner)"}
    except Exception as e:
        out["ttft_ms_p50"] = None
        out["config"]["ttft"] = f"failed: {type(e).__name__}: {e}"
    checkpoint(out)
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, args.batch, args.ctx, cfg.num_hidden_layers)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        out["reference_cpu_container"] = reference_cpu_container()
    if others is not None:
        out["other_configs"] = others
    # ---- BASELINE config 5 on the real thing: when this IS an 8-GPU job (or a rehearsal asks for it), the same loop on the
    # Llama-3-70B shape with TP = world, after everything above (the headline numbers are final by now).  Every rank
    # takes the same decisions (environment + collectives only), so a failure anywhere skips it everywhere.
    want_70b = (dist_on and args.model == "llama3-8b" and args.quant == "w8a8_fp8" and args.layers is None
                and os.environ.get("SGL_MI355_BENCH_NO_70B", "0") in ("", "0")
                and (world == 8 or os.environ.get("SGL_MI355_BENCH_EXTRA_70B", "0") not in ("", "0")))
    if want_70b:
        # the launcher's wall-clock budget (launch_ranks): rank 0's clock decides for everybody whether there is time left
        t_b0, t_bud = float(os.environ.get("SGL_MI355_BENCH_T0", "0") or 0), float(os.environ.get("SGL_MI355_BENCH_BUDGET_S", "0") or 0)
        go = torch.tensor([0.0 if (t_b0 and t_bud and time.time() - t_b0 > 0.55 * t_bud) else 1.0], device=device, dtype=torch.float64)
        torch.distributed.broadcast(go, src=0)
        if float(go.item()) != 1.0:
            want_70b = False
            out["config5_llama3_70b"] = {"skipped": "more than 55 % of the launcher's time budget was spent before this stage"}
    if want_70b:
        import copy
        res = {"workload": f"llama3-70b w8a8_fp8 decode bs={args.batch} ctx={args.ctx} TP={tp}, random weights, the same loop"}
        ok = 1.0
        try:
            del loop
            net = runner = backend = None
            torch.cuda.empty_cache()
            a70 = copy.copy(args)
            a70.model = "llama3-70b"
            # the fused all-reduce + norm staging area is bound to ONE row length per communicator (4096 from the 8B model):
            # drop the binding while nothing is in flight on any rank, or every 70B collective would take the unfused path
            torch.cuda.synchronize()
            torch.distributed.barrier()
            if tp_group.ca_comm is not None and not tp_group.ca_comm.disabled:
                tp_group.ca_comm.rebind_fused_norm()
            torch.distributed.barrier()
            net, cfg70, runner, backend, _ = build(a70, device, tp)
            loop = DecodeLoop(net, runner, backend, args.batch, args.ctx, device)
            can_fuse = net.fuse_quant
            from sglang_npu_amd.distributed import DISPATCH_COUNTS as _DC0
            _DC0.clear()
        except Exception as e:  # e.g. out of memory on one rank
            ok = 0.0
            res["error"] = f"{type(e).__name__}: {str(e)[:200]}"
        t = torch.tensor([ok], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        if float(t.item()) == 1.0:
            try:
                n70 = min(args.steps, 16)
                first = want_fused and can_fuse   # the same call order as the headline `value`, then the other one
                set_call_order(first)
                ms70 = timed(3, n70) / n70 * 1e3
                tp_group.stub_all_reduce = True
                set_call_order(first)
                stub70 = timed(2, min(n70, 6)) / min(n70, 6) * 1e3
                tp_group.stub_all_reduce = False
                from sglang_npu_amd.distributed import DISPATCH_COUNTS as _DC
                res["dispatch_since_capture"] = dict(_DC)  # (counted where Python dispatches: warm-ups + captures of this stage)
                res.update({"call_order": "fused" if first else "reference",
                            "ms_per_step": round(ms70, 4), "tokens_per_s": round(args.batch / ms70 * 1e3, 1),
                            "ms_per_step_allreduce_stubbed": round(stub70, 4),
                            "allreduce_overhead_frac": round(max(0.0, (ms70 - stub70) / ms70), 4),
                            "layers": len(net.layers), "collectives_per_step": 2 * len(net.layers) + 1,
                            "message_bytes": args.batch * cfg70.hidden_size * 2,
                            "graph": loop.graph is not None})
                out["config5_llama3_70b"] = res
                checkpoint(out)
                if can_fuse:
                    set_call_order(not first)
                    ms_o = timed(3, min(n70, 8)) / min(n70, 8) * 1e3
                    res["fused_ms_per_step" if not first else "dropin_ms_per_step"] = round(ms_o, 4)
                    res["fused_tokens_per_s" if not first else "dropin_tokens_per_s"] = round(args.batch / ms_o * 1e3, 1)
            except Exception as e:
                res["error"] = f"{type(e).__name__}: {str(e)[:200]}"
            finally:
                tp_group.stub_all_reduce = False
        elif "error" not in res:
            res["error"] = "another rank could not build the model"
        out["config5_llama3_70b"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
