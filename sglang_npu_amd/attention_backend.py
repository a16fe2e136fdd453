"""MI355X attention backend -- the drop-in behind SGLang's ``AttentionBackend`` interface.

Interface mirrored: python/sglang/srt/layers/attention/base_attn_backend.py:14-117 (same method
names, arguments and return conventions).  Behaviour mirrored: TritonAttnBackend,
python/sglang/srt/layers/attention/triton_backend.py:40-732 (metadata, KV write before the
kernel, output allocation), with the HIP kernels of this package underneath:

  forward_decode  -> ops.set_kv_buffer + ops.decode_attention / decode_attention_fwd
  forward_extend  -> ops.set_kv_buffer + ops.extend_attention_fwd

Differences that are deliberate (MI355X-first):
  * decode reads ``req_to_token`` directly (the kernel gathers page-table rows itself), so the
    per-step ``kv_indices`` flatten pass of triton_backend.py:172-188 is skipped unless
    ``flat_kv_indices=True`` (kept for speculative-decoding style callers that hand in indices);
  * the split count is chosen on the host from the batch geometry only (no device round trip,
    graph-replay safe): one split as soon as every CU has a workgroup, otherwise enough splits
    to fill the 256 CUs, never more than ``--triton-attention-num-kv-splits``.

Registration: SGLang has no backend registry (server_args.py:1264-1280 closed ``choices``,
model_runner.py:1384-1471 if/elif); see ``harness.install_attention_backend`` and INTEGRATION.md.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import torch

from . import deferred, ops
from .harness import AttentionType, ForwardBatch, ForwardMode

NUM_CUS = 256


# kv-split decode: merge (and per-token FP8 quant) inside the attention launch (sgl_mi355_decode_attention_merged) instead
# of a stage-2 launch, while a request has at most FUSE_SPLIT_MERGE_MAX_WGS workgroups (kv heads x head blocks x splits).
# Measured as HIP graphs at bs=64, ctx=2048 (tools/bench_decode_split_merge.py, us, two launches -> one): one rank of
# Llama-3-70B TP=8 (8/1 heads, 4 splits) 23.7 -> 22.3, of Llama-3-8B TP=8 (4/1) 21.8 -> 20.5, TP=4 (8/2, 2 splits) 33.1 ->
# 32.0; Llama-3-8B at bs=16 (32/8 heads, 2 splits = 16 workgroups and a 4096-wide row per request) 34.6 -> 37.1: the last
# workgroup's serial merge outweighs the launch there.  SGL_MI355_NO_SPLIT_MERGE_FUSION=1 keeps the separate launches.
FUSE_SPLIT_MERGE = os.environ.get("SGL_MI355_NO_SPLIT_MERGE_FUSION", "0") in ("", "0")
FUSE_SPLIT_MERGE_MAX_WGS = 8
# SGL_MI355_DECODE_QUANT_FUSION=1: one-split decode (more items than CUs) in which the workgroup that finishes a request's
# last head block also does the per-token FP8 quant of its row (sgl_mi355_decode_attention_quant) instead of a quant
# launch.  OFF by default: with one workgroup per CU every request completes in the launch's last microseconds, so the
# hand-over (write-through stores acknowledged -> counter -> coherent read-back, ~4 us: tools/exp/grid_sync_probe.hip)
# lands on the critical path where the 4.8 us quant launch was -- headline step 6.146 ms with the separate launch, 6.164
# with this (attention in-step 95.5 -> 97.7 us, same box, profiles/r03_decode_quant_in_launch.txt).
FUSE_DECODE_QUANT = bool(os.environ.get("SGL_MI355_DECODE_QUANT_FUSION"))
# round 5, OPT-IN (SGL_MI355_DECODE_KV_WRITE_FUSION=1): forward_decode(save_kv_cache=True) writes the step's K / V rows from
# INSIDE the attention launch when the batch runs the pairs-of-items kernel (ops.decode_attention_paged_newkv).  Parity-green
# (tests/test_decode_newkv_gpu.py) but no faster: one launch fewer per layer, 6.357 / 6.373 ms against 6.377 ms per step on the
# same box (profiles/r05_decode_kv_write_fusion.txt) -- the write and the new token's q.k sit at the head of the kernel's
# dependency chain and cost what the 2.5 us set_kv launch did.  Default: set_kv_buffer + attention, the reference's two calls.
FUSE_DECODE_KV_WRITE = bool(os.environ.get("SGL_MI355_DECODE_KV_WRITE_FUSION"))
# extend launches with few, long items (one short request behind a long cached prefix; the heaviest query blocks of a single
# 1024-token prefill) cut every item's keys into up to four ranges over as many workgroups (csrc/attention_extend.hip PARTS).
# SGL_MI355_EXTEND_PARTS=0: never.
EXTEND_PARTS = os.environ.get("SGL_MI355_EXTEND_PARTS", "1") not in ("", "0")


@dataclass
class ForwardMetadata:
    attn_logits: Optional[torch.Tensor]
    attn_lse: Optional[torch.Tensor]
    max_extend_len: Optional[int]
    num_kv_splits: int
    kv_indptr: Optional[torch.Tensor]
    kv_indices: Optional[torch.Tensor]
    qo_indptr: Optional[torch.Tensor]
    custom_mask: Optional[torch.Tensor] = None
    mask_indptr: Optional[torch.Tensor] = None
    # sliding-window layers (triton_backend.py:45-52): the last W+1 tokens of every request, flattened
    window_kv_indptr: Optional[torch.Tensor] = None
    window_kv_indices: Optional[torch.Tensor] = None
    window_num_kv_splits: int = 1
    window_attn_logits: Optional[torch.Tensor] = None
    window_attn_lse: Optional[torch.Tensor] = None
    # extend: the longest cached prefix of the batch when the host knows it (extend_prefix_lens_cpu); lets the extend kernel cut
    # few, long items into KV-range parts (ops.extend_attention_fwd, round 4)
    max_prefix_len: Optional[int] = None


class AttentionBackend:
    """Same surface as base_attn_backend.py:14-117."""

    def init_forward_metadata(self, forward_batch: ForwardBatch):
        raise NotImplementedError()

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens, encoder_lens,
                                                 forward_mode, spec_info):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum, encoder_lens,
                                                forward_mode, spec_info, seq_lens_cpu):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache: bool = True, **kwargs):
        """Dispatch exactly as base_attn_backend.py:57-89."""
        if forward_batch.forward_mode.is_idle():
            return q.new_empty(q.shape[0], layer.tp_q_head_num * layer.v_head_dim)
        elif forward_batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
        else:
            return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()

    def support_triton(self):
        return True


class MI355AttnBackend(AttentionBackend):
    def __init__(self, model_runner, skip_prefill: bool = False, flat_kv_indices: bool = False):
        super().__init__()
        self.measure_skip_decode_kernel = False  # see forward_decode
        self.skip_prefill = skip_prefill
        self.flat_kv_indices = flat_kv_indices
        max_bs = model_runner.req_to_token_pool.size
        self.device = model_runner.device
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        tp = getattr(model_runner, "tp_size", 1)
        self.num_head = model_runner.model_config.num_attention_heads // tp
        self.num_kv_head = model_runner.model_config.get_num_kv_heads(tp)
        self.max_kv_splits = model_runner.server_args.triton_attention_num_kv_splits
        self.v_head_dim = model_runner.token_to_kv_pool.get_value_buffer(0).shape[-1]
        self.max_context_len = model_runner.model_config.context_len
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = sw if (sw is not None and sw > 0) else None  # triton_backend.py:64-68
        self.kv_indptr = torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device)
        if self.sliding_window_size is not None:  # triton_backend.py:81-82
            self.window_kv_indptr = torch.zeros_like(self.kv_indptr)
        if not skip_prefill:
            self.qo_indptr = torch.zeros((max_bs + 1,), dtype=torch.int32, device=self.device)
        # arrival counters of the in-launch split merge (ops.decode_attention_paged_merged): zero here, left zero by every call
        self._merge_counters = torch.zeros((max_bs,), dtype=torch.int32, device=self.device)
        # partials + arrival counters of the extend kernel's KV-range parts (this backend's launches are on one stream)
        self._extend_parts = (ops.ExtendPartsScratch(self.device) if (not skip_prefill and EXTEND_PARTS
                                                                     and torch.device(self.device).type == "cuda") else None)
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._graph = None  # static buffers once init_cuda_graph_state has run

    def _fuse_split_merge(self, splits: int, bs: int) -> bool:
        group = max(1, self.num_head // self.num_kv_head)
        return (FUSE_SPLIT_MERGE and bs <= self._merge_counters.numel()
                and self.num_kv_head * ((group + 15) // 16) * splits <= FUSE_SPLIT_MERGE_MAX_WGS)

    # ---------------------------------------------------------------- split policy (host only)
    def choose_num_kv_splits(self, bs: int, max_seq_len: Optional[int] = None) -> int:
        group = max(1, self.num_head // self.num_kv_head)
        wgs = bs * self.num_kv_head * ((group + 15) // 16)
        if wgs >= NUM_CUS:
            # One split, pairs of items per workgroup: a round is 2 x NUM_CUS items.  A batch just past one round (65..96
            # requests x 8 kv heads) would pay two full rounds for it; two or four kv-splits cut the items finer, and the
            # half-empty last round costs a half / a quarter as much (+ the merge).  Measured, 32/8/128, us, 1 / 2 / 4 splits
            # (tools/exp/decode_splits_probe.py): bs 65 at ctx 512 / 1024 / 2048 / 8192: 43.6 / 40.1 / 45.1, 75.1 / 65.9 / 65.6,
            # 135.9 / 116.3 / 110.9, 515 / 434 / 394; bs 80: 44.3 / 44.5 / 52.0, 76.4 / 71.4 / 76.6, 140.1 / 126.2 / 129.8,
            # 524 / 460 / 459; bs 96 at 2048: 152.5 / 148.8 / 154.0; at ctx 256 splitting never pays (27.6 / 27.9 / 35.6).
            # Just past TWO rounds the same holds on a smaller scale (bs 129 / 136 / 144 at ctx 2048: 227.8 / 225.5 / 225.7 ->
            # 202.2 / 208.6 / 218.4 with two splits; from bs 160 one split is ahead again).
            full = int(wgs // (2 * NUM_CUS))                     # whole rounds of pairs
            excess = wgs / (2.0 * NUM_CUS) - full                # fraction of one more round
            if excess > 0.0 and self.max_kv_splits >= 2 and (max_seq_len is None or max_seq_len >= 512):
                if full == 1 and excess <= 0.5:
                    if excess <= 0.15 and self.max_kv_splits >= 4 and max_seq_len is not None and max_seq_len >= 2048:
                        return 4
                    if excess <= 0.25 or (max_seq_len is not None and max_seq_len >= 2048):
                        return 2
                elif full == 2 and excess <= 0.125 and (max_seq_len is None or max_seq_len >= 1024):
                    return 2
            return 1
        # one workgroup per CU: measured at bs=64, ctx=2048 (tools/sweep_decode_small.py) 64 (request, kv-head) pairs
        # run 23.5 us with 4 splits vs 32 with 8, 128 pairs 33.8 us with 2 splits vs 40 with 4 -- every extra split adds
        # a prologue and merge work, and a workgroup already keeps 4 independent wave pipelines in flight
        splits = min(self.max_kv_splits, -(-NUM_CUS // wgs))
        if max_seq_len is not None:
            splits = min(splits, max(1, max_seq_len // 256))  # keep >= 256 tokens per split
        return max(1, splits)

    def _scratch(self, bs: int, splits: int):
        """fp32 partials of the kv-splits.  Direct form: one tensor [bs, H, splits, Dv+1] with the LSE in
        the last column (the layout of decode_attention_cpu); flat form: Triton's logits + lse pair."""
        if splits == 1:
            return None, None
        if not self.flat_kv_indices:
            return torch.empty((bs, self.num_head, splits, self.v_head_dim + 1), dtype=torch.float32,
                               device=self.device), None
        logits = torch.empty((bs, self.num_head, splits, self.v_head_dim), dtype=torch.float32, device=self.device)
        lse = torch.empty((bs, self.num_head, splits), dtype=torch.float32, device=self.device)
        return logits, lse

    def _window_buffer(self, lens: torch.Tensor, req_pool_indices: torch.Tensor, bs: int,
                       out: Optional[torch.Tensor] = None):
        """update_sliding_window_buffer[_cuda_graph] (triton_backend.py:927-983): per request the last
        min(len, W + 1) page-table entries, flattened; `out` = the static graph buffer when capturing/replaying."""
        window_lens = torch.clamp(lens, max=self.sliding_window_size + 1)
        indptr = self.window_kv_indptr
        indptr[1:bs + 1] = torch.cumsum(window_lens, dim=0)
        indptr = indptr[:bs + 1]
        if out is None:
            out = torch.empty(max(int(indptr[-1].item()), 1), dtype=torch.int32, device=self.device)
        start = (lens - window_lens).to(torch.int32)
        ops.create_kv_indices(self.req_to_token, req_pool_indices, window_lens, indptr, start, out)
        return indptr, out, window_lens

    def _flat_scratch(self, bs: int, splits: int):
        if splits == 1:
            return None, None
        return (torch.empty((bs, self.num_head, splits, self.v_head_dim), dtype=torch.float32, device=self.device),
                torch.empty((bs, self.num_head, splits), dtype=torch.float32, device=self.device))

    # ---------------------------------------------------------------- metadata
    def init_forward_metadata(self, forward_batch: ForwardBatch):
        """Once per step (model_runner.py:1553-1554 / :1572-1573), as triton_backend.py:160-336."""
        bs = forward_batch.batch_size
        if forward_batch.spec_info is not None:
            raise NotImplementedError("MI355AttnBackend: speculative decoding metadata is not implemented")
        deferred.hint_decode = forward_batch.forward_mode.is_decode()  # (what the pass that follows is: see deferred.hint_decode)
        if forward_batch.forward_mode.is_decode_or_idle():
            max_len = int(forward_batch.seq_lens_cpu.max()) if forward_batch.seq_lens_cpu is not None else None
            splits = self.choose_num_kv_splits(bs, max_len)
            kv_indptr = kv_indices = None
            if self.flat_kv_indices:
                kv_indptr = self.kv_indptr
                kv_indptr[1:bs + 1] = torch.cumsum(forward_batch.seq_lens, dim=0)
                kv_indptr = kv_indptr[:bs + 1]
                kv_indices = torch.empty(forward_batch.seq_lens_sum, dtype=torch.int32, device=self.device)
                ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens,
                                      kv_indptr, None, kv_indices)
            logits, lse = self._scratch(bs, splits)
            md = ForwardMetadata(logits, lse, None, splits, kv_indptr, kv_indices, None)
            if self.sliding_window_size is not None:  # triton_backend.py:187-203
                md.window_kv_indptr, md.window_kv_indices, _ = self._window_buffer(
                    forward_batch.seq_lens, forward_batch.req_pool_indices, bs)
                md.window_num_kv_splits = self.choose_num_kv_splits(bs, min(max_len or 1 << 30,
                                                                            self.sliding_window_size + 1))
                md.window_attn_logits, md.window_attn_lse = self._flat_scratch(bs, md.window_num_kv_splits)
            self.forward_metadata = md
        else:
            kv_indptr = self.kv_indptr
            kv_indptr[1:bs + 1] = torch.cumsum(forward_batch.extend_prefix_lens, dim=0)
            kv_indptr = kv_indptr[:bs + 1]
            n_prefix = (sum(forward_batch.extend_prefix_lens_cpu) if forward_batch.extend_prefix_lens_cpu is not None
                        else int(forward_batch.extend_prefix_lens.sum().item()))
            kv_indices = torch.empty(max(n_prefix, 1), dtype=torch.int32, device=self.device)
            ops.create_kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.extend_prefix_lens,
                                  kv_indptr, None, kv_indices)
            qo_indptr = self.qo_indptr
            qo_indptr[1:bs + 1] = torch.cumsum(forward_batch.extend_seq_lens, dim=0)
            qo_indptr = qo_indptr[:bs + 1]
            max_extend_len = (max(forward_batch.extend_seq_lens_cpu) if forward_batch.extend_seq_lens_cpu is not None
                              else int(torch.max(forward_batch.extend_seq_lens).item()))
            md = ForwardMetadata(None, None, max_extend_len, 1, kv_indptr, kv_indices, qo_indptr)
            if forward_batch.extend_prefix_lens_cpu is not None:
                md.max_prefix_len = max(forward_batch.extend_prefix_lens_cpu) if bs else 0
            if self.sliding_window_size is not None:  # triton_backend.py:301-311
                md.window_kv_indptr, md.window_kv_indices, _ = self._window_buffer(
                    forward_batch.extend_prefix_lens, forward_batch.req_pool_indices, bs)
            self.forward_metadata = md

    # ---------------------------------------------------------------- graph capture / replay
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int, kv_indices_buf: Optional[torch.Tensor] = None):
        """triton_backend.py:338-388: allocate every buffer a captured decode step touches."""
        splits = self.max_kv_splits
        g = {
            "attn_logits": torch.zeros((max_num_tokens, self.num_head, splits, self.v_head_dim + 1),
                                       dtype=torch.float32, device=self.device),
            "attn_lse": torch.zeros((max_num_tokens, self.num_head, splits), dtype=torch.float32, device=self.device),
        }
        if self.flat_kv_indices:
            g["kv_indices"] = kv_indices_buf if kv_indices_buf is not None else torch.zeros(
                (max_num_tokens * self.max_context_len,), dtype=torch.int32, device=self.device)
        if self.sliding_window_size is not None:  # triton_backend.py:373-381
            g["window_kv_indices"] = torch.zeros((max_num_tokens * (self.sliding_window_size + 1),), dtype=torch.int32,
                                                 device=self.device)
            g["window_attn_logits"] = torch.zeros((max_num_tokens, self.num_head, splits, self.v_head_dim),
                                                  dtype=torch.float32, device=self.device)
        self._graph = g

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens, encoder_lens,
                                                 forward_mode, spec_info):
        assert encoder_lens is None, "Not supported"
        if not forward_mode.is_decode_or_idle() or spec_info is not None:
            raise ValueError(f"Invalid forward mode: {forward_mode=} for CUDA Graph capture.")
        deferred.hint_decode = forward_mode.is_decode()
        splits = self.choose_num_kv_splits(bs)  # depends on bs only: identical at capture and replay
        g = self._graph
        logits = lse = None
        if splits > 1:
            dv = self.v_head_dim if self.flat_kv_indices else self.v_head_dim + 1
            logits = g["attn_logits"].view(-1)[: bs * self.num_head * splits * dv].view(bs, self.num_head, splits, dv)
            if self.flat_kv_indices:
                lse = g["attn_lse"].view(-1)[: bs * self.num_head * splits].view(bs, self.num_head, splits)
        kv_indptr = kv_indices = None
        if self.flat_kv_indices:
            kv_indptr = self.kv_indptr
            kv_indptr[1:bs + 1] = torch.cumsum(seq_lens, dim=0)
            kv_indptr = kv_indptr[:bs + 1]
            kv_indices = g["kv_indices"]
            ops.create_kv_indices(self.req_to_token, req_pool_indices, seq_lens, kv_indptr, None, kv_indices)
        md = ForwardMetadata(logits, lse, None, splits, kv_indptr, kv_indices, None)
        if self.sliding_window_size is not None:  # triton_backend.py:420-434
            md.window_kv_indptr, md.window_kv_indices, _ = self._window_buffer(
                seq_lens, req_pool_indices, bs, out=g["window_kv_indices"])
            md.window_num_kv_splits = splits
            if splits > 1:
                md.window_attn_logits = g["window_attn_logits"].view(-1)[: bs * self.num_head * splits * self.v_head_dim] \
                    .view(bs, self.num_head, splits, self.v_head_dim)
                md.window_attn_lse = g["attn_lse"].view(-1)[: bs * self.num_head * splits].view(bs, self.num_head, splits)
        self.forward_metadata = md

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum, encoder_lens,
                                                forward_mode, spec_info, seq_lens_cpu):
        if not forward_mode.is_decode_or_idle() or spec_info is not None:
            raise ValueError(f"Invalid forward mode: {forward_mode=} for CUDA Graph replay.")
        if self.flat_kv_indices:  # refresh the flattened table in place (triton_backend.py:539-552)
            kv_indptr = self.kv_indptr
            kv_indptr[1:bs + 1] = torch.cumsum(seq_lens[:bs], dim=0)
            ops.create_kv_indices(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], kv_indptr[:bs + 1], None,
                                  self._graph["kv_indices"])
        # the direct form reads req_to_token / req_pool_indices / seq_lens, all graph inputs updated by the
        # graph runner itself: nothing to do here.
        if self.sliding_window_size is not None:  # triton_backend.py:554-568
            self._window_buffer(seq_lens[:bs], req_pool_indices[:bs], bs, out=self._graph["window_kv_indices"])

    def get_cuda_graph_seq_len_fill_value(self):
        return 1

    # ---------------------------------------------------------------- forward
    @staticmethod
    def _plain_kv_write(layer, forward_batch, q) -> bool:
        """set_kv_buffer of this step is a plain copy into a 16-bit pool (no scales, no cast) and the heads are uniform: the
        forms the fused RoPE + KV-write from GEMM partials covers."""
        pool = forward_batch.token_to_kv_pool
        return (layer.qk_head_dim == layer.v_head_dim and getattr(layer, "k_scale", None) is None
                and getattr(layer, "v_scale", None) is None
                and getattr(pool, "store_dtype", None) == getattr(pool, "dtype", 0) == q.dtype
                and forward_batch.out_cache_loc is not None)

    @staticmethod
    def _rope_and_write(root, layer, forward_batch, q_size: int, kv_size: int):
        """q / k / v are the column ranges of a FINISHED qkv projection behind a lazy handle with RoPE recorded (deferred.py): rotate
        q / k in place and write the KV rows in one launch; the handle then holds what the reference's in-place rotary_emb leaves."""
        qkv = root.pending_local()
        if qkv is None:  # still split-K partials (a prefill-sized raw split-K form): finish the GEMM, then rotate + write as below
            qkv = root.pending_partials().finalize()
        positions, rot = root._rope[0], root._rope[1]
        pool = forward_batch.token_to_kv_pool
        q, k, v = qkv[:, :q_size], qkv[:, q_size:q_size + kv_size], qkv[:, q_size + kv_size:]
        ops.apply_rope_and_set_kv_buffer(positions, q, k, v, layer.qk_head_dim, rot.cos_sin_cache, pool.get_key_buffer(layer.layer_id),
                                         pool.get_value_buffer(layer.layer_id), forward_batch.out_cache_loc, rot.is_neox_style)
        root._rope = None
        root.resolve(qkv)
        return q, k.view(-1, layer.tp_k_head_num, layer.qk_head_dim), v.view(-1, layer.tp_v_head_num, layer.v_head_dim)

    def forward_decode(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache=True, fp8_out: bool = False):
        """fp8_out (MI355X extension, passed through RadixAttention's **kwargs): the caller wants the per-token FP8
        quantisation of the output (the w8a8 o_proj input).  When the kv-splits are merged anyway, the merge kernel
        quantises in the same pass and (q_fp8, scale) is returned instead of the 16-bit tensor; otherwise the flag is
        ignored and the caller quantises."""
        if save_kv_cache and k is not None and self._plain_kv_write(layer, forward_batch, q):
            q_size, kv_size = layer.tp_q_head_num * layer.qk_head_dim, layer.tp_k_head_num * layer.qk_head_dim
            if q.__class__ is not torch.Tensor:
                # q / k / v are the column ranges of a qkv projection still in split-K partials, RoPE recorded (deferred.py:
                # untouched model code did qkv.split, rotary_emb, RadixAttention's views): finish the GEMM, rotate and write
                # the KV rows in ONE launch (what this repo's fused call order does), then attend on the real q
                root = deferred.qkv_root(q, k, v, q_size, kv_size)
                if root is not None and root.pending_partials() is not None:
                    positions, rot = root._rope[0], root._rope[1]
                    pool = forward_batch.token_to_kv_pool
                    q = ops.rope_set_kv_from_partials(root.pending_partials(), positions, layer.tp_q_head_num, layer.tp_k_head_num,
                                                      layer.qk_head_dim, rot.cos_sin_cache, pool.get_key_buffer(layer.layer_id),
                                                      pool.get_value_buffer(layer.layer_id), forward_batch.out_cache_loc,
                                                      rot.is_neox_style)
                    root.consume()
                    k = v = None
                    save_kv_cache = False
                elif root is not None:
                    q, k, v = self._rope_and_write(root, layer, forward_batch, q_size, kv_size)
                    save_kv_cache = False
            else:
                # a plain q: the view of a finished qkv tensor.  Tell the projection that produced it that this backend could
                # have taken the epilogue (its next output then comes as partials, quantization.W8A8Fp8LinearMethod.apply)
                prod = getattr(getattr(q, "_base", None), "_sgl_mi355_epilogue_producer", None)
                if prod is not None and getattr(prod, "head_size", None) == layer.qk_head_dim and deferred.DEFERRED_EPILOGUES:
                    prod._sgl_mi355_defer_epilogue = True
        q, k, v = deferred.materialize(q), deferred.materialize(k), deferred.materialize(v)
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        md = self.forward_metadata
        sliding = (getattr(layer, "sliding_window_size", None) is not None and layer.sliding_window_size > -1
                   and md.window_kv_indices is not None)
        # one split, more (request, kv head) items than CUs, 16-bit pool: the launch quantises the rows as well
        quant_in_launch = (fp8_out and FUSE_DECODE_QUANT and not self.flat_kv_indices and not sliding
                           and isinstance(md.num_kv_splits, int) and md.num_kv_splits == 1
                           and layer.qk_head_dim == layer.v_head_dim and q.shape[0] <= self._merge_counters.numel()
                           and q.shape[0] * layer.tp_k_head_num > NUM_CUS
                           and forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id).dtype == q.dtype)
        if self.measure_skip_decode_kernel:
            # measurement aid (bench.py): the step WITHOUT the decode attention launch (and without set_kv_buffer where that is
            # still a launch of its own at this point; with the qkv hand-over above the KV write has already happened), so that the
            # kernel's in-step cost can be taken as the difference of two graph-replayed steps; the output is uninitialised
            # (in the form the skipped launch would have returned: the quantised pair when it quantises as well)
            if quant_in_launch:
                return (torch.empty((q.shape[0], q.shape[1]), dtype=torch.float8_e4m3fn, device=q.device),
                        torch.ones((q.shape[0], 1), dtype=torch.float32, device=q.device))
            return q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        if quant_in_launch:
            kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
            vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
            if save_kv_cache:
                forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
                save_kv_cache = False
            q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
            done = ops.decode_attention_paged_quant(q3, kb, vb, torch.empty_like(q3), self.req_to_token,
                                                    forward_batch.req_pool_indices, forward_batch.seq_lens,
                                                    self._merge_counters, layer.scaling, layer.logit_cap)
            if done is not False:
                return done
        if (fp8_out and not self.flat_kv_indices and isinstance(md.num_kv_splits, int) and md.num_kv_splits > 1
                and md.attn_logits is not None and layer.qk_head_dim == layer.v_head_dim
                and (layer.tp_q_head_num * layer.v_head_dim) % 8 == 0
                and not (getattr(layer, "sliding_window_size", None) is not None and layer.sliding_window_size > -1
                         and md.window_kv_indices is not None)):
            if save_kv_cache:
                forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
            kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
            vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
            q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
            if self._fuse_split_merge(md.num_kv_splits, q3.shape[0]):
                # one launch: the workgroup that publishes a request's last partial merges and quantises its row
                done = ops.decode_attention_paged_merged(q3, kb, vb, None, self.req_to_token, forward_batch.req_pool_indices,
                                                         forward_batch.seq_lens, md.attn_logits[:q3.shape[0]],
                                                         md.num_kv_splits, self._merge_counters, layer.scaling,
                                                         layer.logit_cap, fp8_out=True)
                if done is not False:
                    return done
            ops.decode_attention_paged(q3, kb, vb, None, self.req_to_token, forward_batch.req_pool_indices,
                                       forward_batch.seq_lens, md.attn_logits[:q3.shape[0]], md.num_kv_splits,
                                       layer.scaling, layer.logit_cap)
            return ops.decode_merge_quant_fp8(md.attn_logits[:q3.shape[0]], md.num_kv_splits, q.dtype)
        if layer.qk_head_dim != layer.v_head_dim:
            o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        else:
            o = torch.empty_like(q)
        if (save_kv_cache and FUSE_DECODE_KV_WRITE and not sliding and md.kv_indices is None and k is not None
                and isinstance(md.num_kv_splits, int) and md.num_kv_splits == 1 and layer.qk_head_dim == layer.v_head_dim
                and getattr(layer, "k_scale", None) is None and getattr(layer, "v_scale", None) is None
                and getattr(forward_batch.token_to_kv_pool, "store_dtype", None) == getattr(forward_batch.token_to_kv_pool, "dtype", 0)):
            # opt-in: set_kv_buffer + decode attention as ONE launch (base_attn_backend.py:57-89 hands both to this call):
            # the step's K / V rows go to the pool from inside the attention kernel, which takes the new token into the softmax
            # straight from the tensors.  out_cache_loc of a decode batch IS the page-table entry of position seq_len - 1.
            # Declines (False, nothing written) outside the pairs-of-items kernel; pools with their own write (scales, ...) keep it.
            kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
            vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
            q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
            k3 = k.view(-1, layer.tp_k_head_num, layer.qk_head_dim)
            v3 = v.view(-1, layer.tp_k_head_num, layer.v_head_dim)
            if ops.decode_attention_paged_newkv(q3, kb, vb, o.view(-1, layer.tp_q_head_num, layer.v_head_dim), k3, v3,
                                                forward_batch.out_cache_loc, self.req_to_token, forward_batch.req_pool_indices,
                                                forward_batch.seq_lens, layer.scaling, layer.logit_cap):
                return o
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        md = self.forward_metadata
        kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
        vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        o3 = o.view(-1, layer.tp_q_head_num, layer.v_head_dim)
        sw = getattr(layer, "sliding_window_size", None)
        if sw is not None and sw > -1 and md.window_kv_indices is not None:  # triton_backend.py:711-713
            ops.decode_attention_fwd(q3, kb, vb, o3, md.window_kv_indptr, md.window_kv_indices, md.window_attn_logits,
                                     md.window_attn_lse, None, md.window_num_kv_splits, layer.scaling, layer.logit_cap)
        elif md.kv_indices is not None:
            ops.decode_attention_fwd(q3, kb, vb, o3, md.kv_indptr, md.kv_indices, md.attn_logits, md.attn_lse, None,
                                     md.num_kv_splits, layer.scaling, layer.logit_cap)
        else:
            if (isinstance(md.num_kv_splits, int) and md.num_kv_splits > 1 and md.attn_logits is not None
                    and layer.qk_head_dim == layer.v_head_dim and self._fuse_split_merge(md.num_kv_splits, q3.shape[0])):
                # the in-launch merge of the kv-splits.  FP8 companion (ops.attach_fp8_companion, the reference call order): once the
                # w8a8 o_proj that receives this tensor has asked (layer.emit_fp8_companion, set through the producer tag below),
                # the merging workgroup also quantises the row per token -- `o` AND (q, scale) from the one launch, bit-identical
                # to sgl_per_token_quant_fp8 on `o`
                want_comp = (ops.FP8_COMPANIONS and getattr(layer, "emit_fp8_companion", False)
                             and (layer.tp_q_head_num * layer.v_head_dim) % 8 == 0)
                done = ops.decode_attention_paged_merged(q3, kb, vb, o3, self.req_to_token, forward_batch.req_pool_indices,
                                                         forward_batch.seq_lens, md.attn_logits[:q3.shape[0]],
                                                         md.num_kv_splits, self._merge_counters, layer.scaling,
                                                         layer.logit_cap, fp8_out=want_comp)
                if done is not False:
                    if want_comp:
                        ops.attach_fp8_companion(o, done[0], done[1])
                    elif ops.FP8_COMPANIONS:
                        o._sgl_mi355_producer = layer
                    return o
            ops.decode_attention_paged(q3, kb, vb, o3, self.req_to_token, forward_batch.req_pool_indices,
                                       forward_batch.seq_lens, md.attn_logits, md.num_kv_splits, layer.scaling,
                                       layer.logit_cap)
        return o

    def forward_decode_absmax(self, q, layer, forward_batch: ForwardBatch, row_absmax: torch.Tensor):
        """MI355X extension: forward_decode (KV already in the pool) that also leaves row_absmax[t] = max |output[t]| over
        all heads (float32 [T], zeroed by the caller) -- the absmax pass of the w8a8 o_proj's per-token input quant, taken
        in the attention kernel's epilogue (ops.decode_attention_paged_absmax).  Returns the [T, Hq * D] output, or None
        (nothing launched) when the batch is outside that kernel's form."""
        md = self.forward_metadata
        sw = getattr(layer, "sliding_window_size", None)
        if (self.flat_kv_indices or md.kv_indices is not None or md.num_kv_splits != 1
                or not isinstance(md.num_kv_splits, int) or layer.qk_head_dim != layer.v_head_dim
                or (sw is not None and sw > -1)):
            return None
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
        vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
        o = torch.empty_like(q)
        done = ops.decode_attention_paged_absmax(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), kb, vb, o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            row_absmax, self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, layer.scaling,
            layer.logit_cap)
        return o if done else None

    def forward_decode_qkv_partials(self, part, positions, cos_sin_cache, is_neox, layer, forward_batch: ForwardBatch):
        """MI355X extension: decode attention taking the layer's qkv projection while it is still split-K partial sums
        (ops.GemmPartials) -- the kernel's prologue applies the GEMM epilogue, RoPE and the KV-pool write
        (RotaryEmbedding.forward_cuda + MHATokenToKVPool.set_kv_buffer + forward_decode in one launch).  Returns the
        [T, Hq * D] output, or None when this batch is outside the fused kernel's form (the caller then runs the
        separate ops); nothing has been written in that case."""
        md = self.forward_metadata
        sw = getattr(layer, "sliding_window_size", None)
        if (self.flat_kv_indices or md.kv_indices is not None or md.num_kv_splits != 1
                or not isinstance(md.num_kv_splits, int) or layer.qk_head_dim != layer.v_head_dim
                or (sw is not None and sw > -1) or positions.dtype != torch.int64):
            return None
        kb = forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id)
        vb = forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id)
        o = torch.empty((part.M, layer.tp_q_head_num * layer.v_head_dim), dtype=part.out_dtype, device=kb.device)
        done = ops.decode_attention_qkv_partials(
            part, positions, cos_sin_cache, is_neox, forward_batch.out_cache_loc, kb, vb,
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim), self.req_to_token, forward_batch.req_pool_indices,
            forward_batch.seq_lens, layer.tp_q_head_num, layer.scaling, layer.logit_cap)
        return o if done else None

    def forward_extend(self, q, k, v, layer, forward_batch: ForwardBatch, save_kv_cache=True, **kwargs):
        if save_kv_cache and k is not None and self._plain_kv_write(layer, forward_batch, q):
            q_size, kv_size = layer.tp_q_head_num * layer.qk_head_dim, layer.tp_k_head_num * layer.qk_head_dim
            if q.__class__ is not torch.Tensor:
                # the column ranges of a finished qkv projection behind a lazy handle, RoPE recorded (deferred.py): RoPE + KV-pool
                # write as one launch; the extend kernel then reads the rotated K / V as tensors
                root = deferred.qkv_root(q, k, v, q_size, kv_size)
                if root is not None and not root.needs_allreduce:
                    q, k, v = self._rope_and_write(root, layer, forward_batch, q_size, kv_size)
                    save_kv_cache = False
            else:
                prod = getattr(getattr(q, "_base", None), "_sgl_mi355_epilogue_producer", None)
                if prod is not None and getattr(prod, "head_size", None) == layer.qk_head_dim and deferred.DEFERRED_EPILOGUES:
                    prod._sgl_mi355_defer_epilogue = True
        # (anything else lazy, e.g. a qkv projection still in split-K partials: the extend kernel reads K / V as tensors -- finish it)
        q, k, v = deferred.materialize(q), deferred.materialize(k), deferred.materialize(v)
        if layer.qk_head_dim != layer.v_head_dim:
            o = q.new_empty((q.shape[0], layer.tp_q_head_num * layer.v_head_dim))
        else:
            o = torch.empty_like(q)
        if save_kv_cache:
            forward_batch.token_to_kv_pool.set_kv_buffer(layer, forward_batch.out_cache_loc, k, v)
        causal = layer.attn_type != AttentionType.ENCODER_ONLY
        md = self.forward_metadata
        sw = getattr(layer, "sliding_window_size", None)
        if sw is not None and sw > -1 and md.window_kv_indices is not None:  # triton_backend.py:656-665
            sliding_window_size, kv_indptr, kv_indices = sw, md.window_kv_indptr, md.window_kv_indices
        else:
            sliding_window_size, kv_indptr, kv_indices = -1, md.kv_indptr, md.kv_indices
        # Positional call exactly as triton_backend.py:666-684: the 16th positional argument of
        # extend_attention_fwd is `skip_prefix_custom_mask` (extend_attention.py:306-324), so in this reference
        # snapshot the window size never reaches the kernel's SLIDING_WINDOW_SIZE -- the window acts through
        # window_kv_indices only.  Kept identical on purpose (same inputs -> same results).
        ops.extend_attention_fwd(
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
            # (the reference makes k and v contiguous here, triton_backend.py:667-668; the HIP kernel takes the token /
            #  head strides of the qkv split as they are -- two copies less per layer)
            k if k.stride(-1) == 1 else k.contiguous(), v if v.stride(-1) == 1 else v.contiguous(),
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            forward_batch.token_to_kv_pool.get_key_buffer(layer.layer_id),
            forward_batch.token_to_kv_pool.get_value_buffer(layer.layer_id),
            md.qo_indptr, kv_indptr, kv_indices, md.custom_mask, causal, md.mask_indptr, md.max_extend_len,
            layer.scaling, layer.logit_cap, sliding_window_size,
            max_prefix_len=md.max_prefix_len if kv_indices is md.kv_indices else None, parts_scratch=self._extend_parts)
        return o
