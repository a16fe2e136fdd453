"""Synthetic-weight Llama/Qwen2-shaped decoder stack wired to the MI355X backend.

Only used by bench.py / tests / smoke: SGLang's own model files stay untouched.  The
module tree and the per-layer call order follow the reference so the hot path is exercised exactly
where it sits there:
  LlamaMLP.forward             python/sglang/srt/models/llama.py:94-98
  LlamaAttention.forward       python/sglang/srt/models/llama.py:186-191
  LlamaDecoderLayer.forward    python/sglang/srt/models/llama.py:245-268
  LlamaModel / ForCausalLM     python/sglang/srt/models/llama.py:330ff
Weights: the dummy loader's recipe (model_loader/weight_utils.py:752-781): every float parameter
~ U(-1e-3, 1e-3) from a generator seeded 1234 per parameter (drawn at FULL size, then TP-sliced, so
shards are consistent across ranks); FP8 weights are then per-channel quantised like
w8a8_fp8.py:119-125; AWQ tensors follow sgl-kernel/tests/test_awq_dequant.py:80-102.

With ``fuse_quant=True`` (FP8 only) the layer uses the backend's fused norm+quant / silu+quant
kernels so each GEMM input is produced directly in FP8 -- same arithmetic as the unfused sequence.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import ops
from .distributed import (AllReduceHandle, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                          get_tp_group)
from .harness import ForwardBatch, ModelConfig, RadixAttention
from .layers import RMSNorm, RotaryEmbedding, SiluAndMul, VocabParallelEmbedding
from .linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from .parameter import tracked
from .quantization import AWQConfig, W8A8Fp8Config


def _dummy(shape, dtype, device, low=-1e-3, high=1e-3, seed=1234):
    """initialize_dummy_weights: values depend only on numel + dtype (weight_utils.py:752-781)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.empty(shape, dtype=torch.float16 if dtype in (torch.bfloat16, torch.float16) else torch.float32,
                    device=device)
    t.uniform_(low, high, generator=g)
    return t.to(dtype)


# Row-parallel layers hand their all-reduce to the TP group's side stream and return an AllReduceHandle; the next
# norm waits on it (north_star: "RCCL all-reduce overlapped on a side HIP stream").  SGL_MI355_SYNC_AR=1: in-stream.
ASYNC_AR = not os.environ.get("SGL_MI355_SYNC_AR")


def _arrived(h):
    """The tensor behind a pending all-reduce (fences the current stream on it), or h itself."""
    return h.wait() if isinstance(h, AllReduceHandle) else h


# With the P2P communicator on (SGL_MI355_CUSTOM_AR=1) row-parallel layers skip their collective and the next norm runs
# all-reduce + residual add + RMSNorm (+ FP8 quant) as ONE kernel (upstream seam: can_fuse_mlp_allreduce /
# RMSNorm.forward_with_allreduce_fusion).  SGL_MI355_NO_AR_FUSION=1 keeps them apart.
FUSE_AR_NORM = not os.environ.get("SGL_MI355_NO_AR_FUSION")


# SGL_MI355_QKV_ATTN_FUSION=1: the qkv GEMM's epilogue + RoPE + KV write run in the decode attention kernel's prologue when
# the batch allows (MI355AttnBackend.forward_decode_qkv_partials).  OFF by default: measured slower than the separate
# RoPE/KV-write launch (96.0 vs 91.3 us for the pair at bs=64 ctx=2048, profiles/README.md round 2) -- the prologue's
# dependent round trips sit in front of every workgroup's K/V stream, the separate kernel runs wide and costs < 1 us.
FUSE_QKV_ATTN = bool(os.environ.get("SGL_MI355_QKV_ATTN_FUSION"))


# LM head (logits_processor.py:430-505): rows up to which the 16-bit weight streamer runs instead of the library GEMM.
# Round 3: on the fragment-major copy of an UNTIED head (ops.linear16_shuffle_weight, built once after loading; the
# row-major tensor stays for batches above 128 rows) the streamer is ahead of the library at every M <= 128
# (profiles/r03_lm_head_points.jsonl; M = 128 on its 128-row form: 223 vs 234 us); on a row-major weight it loses above 32
# rows (213 vs 201 us at M = 64).
LM_HEAD_STREAMER_MAX_ROWS = 32
LM_HEAD_SHUFFLED_MAX_ROWS = int(os.environ.get("SGL_MI355_LM_HEAD_SHUFFLED_MAX_ROWS", "128"))  # (M = 128: 223 us vs 234 library)
SHUFFLE_LM_HEAD = not os.environ.get("SGL_MI355_NO_LM_HEAD_SHUFFLE")

# SGL_MI355_ATTN_QUANT_FUSION=1: the per-token FP8 quant between the decode attention and o_proj folded into the two
# (attention epilogue: row absmax by atomic max; o_proj GEMM: quantise while staging; bit-identical results).  OFF by
# default: it removes a launch but every one of the 256 GEMM workgroups then converts its whole 64 x 1024 activation slice
# (64-fold redundant VALU work on 5 waves per CU): o_proj + neighbours 17.5 -> 20.3 us per layer, the step 6.09 -> 6.22 ms
# (tools/bench_attn_quant_fusion.py).
FUSE_ATTN_QUANT = bool(os.environ.get("SGL_MI355_ATTN_QUANT_FUSION"))


def _unreduced(h) -> bool:
    return isinstance(h, torch.Tensor) and getattr(h, "_sglang_needs_allreduce_fusion", False)


# fewest decode rows for which GEMM epilogues are deferred into the consumer kernels (tuning: SGL_MI355_DEFER_MIN_ROWS)
DEFER_MIN_ROWS = int(os.environ.get("SGL_MI355_DEFER_MIN_ROWS", "32"))
# short prefills (extend passes of at most this many new tokens) defer the o_proj / down_proj epilogues too (0: never)
DEFER_EXTEND_MAX_ROWS = int(os.environ.get("SGL_MI355_DEFER_EXTEND_MAX_ROWS", "128"))
# widest per-rank gate_up (2 * intermediate / tp) that goes through split-K partials + fused SiLU (tuning: SGL_MI355_GATE_UP_PARTIALS_MAX_N)
GATE_UP_PARTIALS_MAX_N = int(os.environ.get("SGL_MI355_GATE_UP_PARTIALS_MAX_N", "4096"))


# prefill: the gate_up GEMM applies SiLU * mul in its epilogue (ops.fp8_scaled_mm_silu_mul) instead of writing [T, 2 I] for a
# separate activation kernel.  SGL_MI355_NO_SILU_GEMM_FUSION=1 keeps the two launches.
FUSE_SILU_GEMM = os.environ.get("SGL_MI355_NO_SILU_GEMM_FUSION", "0") in ("", "0")


class LlamaMLP(torch.nn.Module):
    def __init__(self, hidden_size, intermediate_size, quant_config, dtype):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, params_dtype=dtype,
                                                       quant_config=quant_config)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, params_dtype=dtype, quant_config=quant_config)
        self.act_fn = SiluAndMul()

    def forward(self, x):
        # models/llama.py:94-98, verbatim: no flags -- what the drop-in classes make of it is their business (deferred.py)
        gate_up, _ = self.gate_up_proj(x)
        x = self.act_fn(gate_up)
        x, _ = self.down_proj(x)
        return x

    def forward_fp8(self, xq, xs, out_dtype, defer: bool = False):
        """Same computation with the activations kept in FP8 between the kernels (fused producers)."""
        part = None
        if defer and xq.shape[0] <= 128 and self.gate_up_proj.output_size_per_partition <= GATE_UP_PARTIALS_MAX_N:
            # narrow per-rank gate_up (Llama-3-8B at TP = 8: 4096 -> 3584): the split-K kernel + the epilogue inside
            # silu.mul beats the latency-bound single-pass GEMM (one rank's step 2.61 -> 2.57 ms); at 7168 columns it loses
            part = self.gate_up_proj.forward_prequantized_partials(xq, xs, out_dtype)
        act = None
        if part is None and FUSE_SILU_GEMM and xq.shape[0] > 128:
            # prefill: SiLU * mul in the gate_up GEMM's epilogue (the [T, 2 I] product never goes to HBM), then the row quant
            act = self.gate_up_proj.forward_prequantized_silu_mul(xq, xs, out_dtype)
        if part is not None:
            aq, a_s = ops.silu_and_mul_quant_fp8_from_partials(part)
        elif act is not None:
            a2 = act.view(-1, act.shape[-1])
            aq = torch.empty_like(a2, dtype=torch.float8_e4m3fn)
            a_s = torch.empty((a2.shape[0], 1), dtype=torch.float32, device=a2.device)
            ops.sgl_per_token_quant_fp8(a2, aq, a_s)
        else:
            gate_up, _ = self.gate_up_proj.forward_prequantized(xq, xs, out_dtype)
            aq, a_s = ops.silu_and_mul_quant_fp8(gate_up)
        if defer:  # leave the down_proj epilogue to the next norm (ops.GemmPartials)
            part = self.down_proj.forward_prequantized_partials(aq, a_s, out_dtype, can_fuse_mlp_allreduce=FUSE_AR_NORM)
            if part is not None:
                return part
        x, _ = self.down_proj.forward_prequantized(aq, a_s, out_dtype, async_reduce=ASYNC_AR,
                                                  can_fuse_mlp_allreduce=FUSE_AR_NORM)
        return x


class LlamaAttention(torch.nn.Module):
    def __init__(self, cfg: ModelConfig, layer_id: int, quant_config, dtype, device):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.total_num_heads, self.total_num_kv_heads = cfg.num_attention_heads, cfg.num_key_value_heads
        self.num_heads = self.total_num_heads // tp
        self.num_kv_heads = max(1, self.total_num_kv_heads // tp)
        self.head_dim = cfg.head_dim
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_proj = QKVParallelLinear(cfg.hidden_size, self.head_dim, self.total_num_heads, self.total_num_kv_heads,
                                          params_dtype=dtype, quant_config=quant_config)
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, cfg.hidden_size, params_dtype=dtype,
                                        quant_config=quant_config)
        self.rotary_emb = RotaryEmbedding(self.head_dim, self.head_dim, cfg.context_len, cfg.rope_theta, True, dtype,
                                          device)
        self.attn = RadixAttention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads, layer_id)

    def forward(self, positions, hidden_states, forward_batch: ForwardBatch):
        qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q, k = self.rotary_emb(positions, q, k)
        attn_output = self.attn(q, k, v, forward_batch)
        output, _ = self.o_proj(attn_output)  # (models/llama.py:186-191, verbatim)
        return output

    def forward_fp8(self, positions, xq, xs, forward_batch: ForwardBatch, out_dtype, defer: bool = False):
        """FP8-input variant: qkv GEMM on the pre-quantised activation, RoPE fused with the KV-pool write
        (the backend is then called with save_kv_cache=False).  defer=True additionally leaves GEMM epilogues to
        the consumer kernel where a split-K form exists (ops.GemmPartials): qkv -> RoPE/KV write, o_proj -> the next
        norm; the returned hidden state may then be a GemmPartials."""
        pool = forward_batch.token_to_kv_pool
        kb, vb = pool.get_key_buffer(self.attn.layer_id), pool.get_value_buffer(self.attn.layer_id)
        part = (self.qkv_proj.forward_prequantized_partials(xq, xs, out_dtype)
                if defer and forward_batch.forward_mode.is_decode() else None)
        attn_output = None
        if part is not None and FUSE_QKV_ATTN and forward_batch.forward_mode.is_decode():
            # ... and all of that inside the decode attention kernel's prologue, when the batch has the shape for it
            fused = getattr(forward_batch.attn_backend, "forward_decode_qkv_partials", None)
            if fused is not None:
                attn_output = fused(part, positions, self.rotary_emb.cos_sin_cache, self.rotary_emb.is_neox_style,
                                    self.attn, forward_batch)
        if attn_output is not None:
            pass
        elif part is not None:  # qkv epilogue + RoPE + KV write in one kernel
            q = ops.rope_set_kv_from_partials(part, positions, self.num_heads, self.num_kv_heads, self.head_dim,
                                              self.rotary_emb.cos_sin_cache, kb, vb, forward_batch.out_cache_loc,
                                              self.rotary_emb.is_neox_style)
            k = v = None
        else:
            qkv, _ = self.qkv_proj.forward_prequantized(xq, xs, out_dtype)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            ops.apply_rope_and_set_kv_buffer(positions, q, k, v, self.head_dim, self.rotary_emb.cos_sin_cache, kb, vb,
                                             forward_batch.out_cache_loc, self.rotary_emb.is_neox_style)
        # o_proj's per-token input quant without a launch of its own: the attention epilogue takes the row absmax, the
        # o_proj GEMM quantises while it stages its activations (bit-identical to the quant kernel in between)
        absmax = getattr(forward_batch, "attn_row_absmax", None)
        if (attn_output is None and defer and FUSE_ATTN_QUANT and absmax is not None and k is None
                and forward_batch.forward_mode.is_decode() and get_tensor_model_parallel_world_size() == 1):
            fn = getattr(forward_batch.attn_backend, "forward_decode_absmax", None)
            row = absmax[self.attn.layer_id, :q.shape[0]]
            o16 = fn(q, self.attn, forward_batch, row) if fn is not None else None
            if o16 is not None:
                part = self.o_proj.forward_a16_partials(o16, row, out_dtype)
                if part is not None:
                    return part
                attn_output = o16  # the attention ran; quantise its output the ordinary way
        # fp8_out: when the backend merges kv-splits anyway, the merge kernel also does o_proj's input quant
        if attn_output is None:
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False,
                                    fp8_out=forward_batch.forward_mode.is_decode())
        if isinstance(attn_output, tuple):
            aq, a_s = attn_output
        elif defer:
            a2 = attn_output.view(-1, attn_output.shape[-1]).contiguous()  # apply_fp8_linear's own quant step
            aq = torch.empty_like(a2, dtype=torch.float8_e4m3fn)
            a_s = torch.empty((a2.shape[0], 1), dtype=torch.float32, device=a2.device)
            ops.sgl_per_token_quant_fp8(a2, aq, a_s)
        else:
            output, _ = self.o_proj(attn_output, async_reduce=ASYNC_AR, can_fuse_mlp_allreduce=FUSE_AR_NORM)
            return output
        if defer:  # leave the o_proj epilogue to post_attention_layernorm
            part = self.o_proj.forward_prequantized_partials(aq, a_s, out_dtype, can_fuse_mlp_allreduce=FUSE_AR_NORM)
            if part is not None:
                return part
        output, _ = self.o_proj.forward_prequantized(aq, a_s, out_dtype, async_reduce=ASYNC_AR,
                                                      can_fuse_mlp_allreduce=FUSE_AR_NORM)
        return output


class LlamaDecoderLayer(torch.nn.Module):
    def __init__(self, cfg: ModelConfig, layer_id: int, quant_config, dtype, device):
        super().__init__()
        self.self_attn = LlamaAttention(cfg, layer_id, quant_config, dtype, device)
        self.mlp = LlamaMLP(cfg.hidden_size, cfg.intermediate_size, quant_config, dtype)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)

    def forward(self, positions, hidden_states, forward_batch, residual):
        # models/llama.py:245-268, verbatim (the reference call order): plain calls, no tags looked at, no handles waited for
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual

    def forward_fp8(self, positions, hidden_states, forward_batch, residual, defer: bool = False):
        """Fused-producer variant for the FP8 config: (add +) RMSNorm emits the per-token FP8 activation
        directly, so no standalone quant kernel runs before qkv / gate_up / down.  `hidden_states` may be an
        ops.GemmPartials left by the previous layer's down_proj (defer=True)."""
        def norm_quant(norm, h, res):
            h = _arrived(h)  # a row-parallel GEMM's all-reduce still running on the side stream
            if _unreduced(h) and res is not None:  # all-reduce + add + norm + quant in one kernel
                return norm.forward_with_allreduce_fusion(h, res, quant_fp8=True)[0]
            if isinstance(h, ops.GemmPartials) and h.needs_allreduce:  # ... and the collective: one kernel does it all
                return norm.forward_with_allreduce_fusion(h, res, quant_fp8=True)[0]
            if isinstance(h, ops.GemmPartials):  # the producer GEMM left its epilogue to this kernel
                return ops.rmsnorm_quant_fp8_from_partials(h, res, norm.weight.data, norm.variance_epsilon)
            q, s_, _ = ops.rmsnorm_quant_fp8(h, norm.weight.data, norm.variance_epsilon, residual=res)
            return q, s_

        hidden_states = _arrived(hidden_states)
        dt = residual.dtype if residual is not None else hidden_states.dtype
        if residual is None:
            residual = hidden_states.clone()
            xq, xs = norm_quant(self.input_layernorm, hidden_states, None)
        else:
            xq, xs = norm_quant(self.input_layernorm, hidden_states, residual)
        hidden_states = self.self_attn.forward_fp8(positions, xq, xs, forward_batch, dt, defer)
        xq, xs = norm_quant(self.post_attention_layernorm, hidden_states, residual)
        hidden_states = self.mlp.forward_fp8(xq, xs, dt, defer)
        return hidden_states, residual


class LlamaForCausalLM(torch.nn.Module):
    def __init__(self, cfg: ModelConfig, quantization: Optional[str] = None, dtype=torch.bfloat16, device="cuda:0",
                 num_layers: Optional[int] = None, with_lm_head: bool = True, fuse_quant: bool = True):
        super().__init__()
        self.cfg, self.dtype, self.device_str = cfg, dtype, device
        self.fuse_quant = fuse_quant and quantization == "w8a8_fp8"
        self.defer_epilogues = not os.environ.get("SGL_MI355_NO_DEFER")  # GEMM epilogues inside the consumer kernels
        self._attn_absmax = None  # [layers, 64] float32, see FUSE_ATTN_QUANT
        self._attn_absmax_retired = []  # earlier, smaller buffers that captured graphs may still address
        self.quant_config = None
        if quantization == "w8a8_fp8":
            self.quant_config = W8A8Fp8Config(is_checkpoint_fp8_serialized=False)
        elif quantization == "awq":
            self.quant_config = AWQConfig(4, 128, True)
            # synthetic weights are never reloaded: keep only the k-packed INT4 copy (quantization.py)
            self.quant_config.release_checkpoint_layout = not os.environ.get("SGL_MI355_AWQ_KEEP_CHECKPOINT_LAYOUT")
        elif quantization is not None:
            raise ValueError(f"unknown quantization {quantization}")
        tp = get_tensor_model_parallel_world_size()
        self.vocab_per_rank = cfg.vocab_size // tp
        n_layers = num_layers if num_layers is not None else cfg.num_hidden_layers
        with torch.device(device):
            self.layers = torch.nn.ModuleList(
                [LlamaDecoderLayer(cfg, i, self.quant_config, dtype, device) for i in range(n_layers)])
            self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.embed = None
        self.embed_tokens = None  # (TP = 1 only) the table as a plain tensor, for tests that read it
        self.lm_head = None
        self._lm_head_fm = None  # ops.TrackedCopy16 of an untied head (its fragment-major copy for the 16-bit streamer)
        self.with_lm_head = with_lm_head

    # ------------------------------------------------------------------ synthetic weights
    @torch.no_grad()
    def load_dummy_weights(self):
        dev, dt, cfg = self.device_str, self.dtype, self.cfg
        rank = get_tensor_model_parallel_rank()
        for layer in self.layers:
            at, mlp = layer.self_attn, layer.mlp
            qkv_n = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
            specs = [
                (at.qkv_proj, qkv_n, cfg.hidden_size),
                (at.o_proj, cfg.hidden_size, cfg.num_attention_heads * cfg.head_dim),
                (mlp.gate_up_proj, 2 * cfg.intermediate_size, cfg.hidden_size),
                (mlp.down_proj, cfg.hidden_size, cfg.intermediate_size),
            ]
            for lin, n_full, k_full in specs:
                if isinstance(self.quant_config, AWQConfig):
                    self._fill_awq(lin)
                    continue
                # the FULL (unsharded, fused-on-disk) matrix goes through the layer's checkpoint loader, which takes
                # this rank's rows / columns (and replicates KV heads when tp > num_kv_heads) -- linear.py weight_loader
                lin.weight.weight_loader(lin.weight, _dummy((n_full, k_full), dt, dev))
                lin.quant_method.process_weights_after_loading(lin)
            layer.input_layernorm.weight.data = _dummy((cfg.hidden_size,), dt, dev, 0.5, 1.5)
            layer.post_attention_layernorm.weight.data = _dummy((cfg.hidden_size,), dt, dev, 0.5, 1.5, seed=4321)
        self.norm.weight.data = _dummy((cfg.hidden_size,), dt, dev, 0.5, 1.5)
        # vocab-parallel table (vocab_parallel_embedding.py): the FULL dummy table goes through the layer's loader, which
        # keeps this rank's rows; forward = masked gather + all-reduce
        self.embed = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size, params_dtype=dt).to(dev)
        self.embed.weight.weight_loader(self.embed.weight, _dummy((cfg.vocab_size, cfg.hidden_size), dt, dev, -1.0, 1.0, seed=99))
        self.embed_tokens = self.embed.weight.data if get_tensor_model_parallel_world_size() == 1 else None
        if self.embed_tokens is not None:
            self.embed_tokens = self.embed_tokens[:cfg.vocab_size]
        if self.with_lm_head:
            full = _dummy((cfg.vocab_size, cfg.hidden_size), dt, dev, -2e-2, 2e-2, seed=77)
            # (a tracked parameter: the fragment-major copy next to it follows in-place updates, parameter.py / ops.TrackedCopy16)
            self.lm_head = tracked(torch.nn.Parameter)(
                full[rank * self.vocab_per_rank:(rank + 1) * self.vocab_per_rank].contiguous(), requires_grad=False)
            self._lm_head_fm = None
            if SHUFFLE_LM_HEAD and ops.linear16_shuffle_supported(self.lm_head.shape[0], self.lm_head.shape[1]):
                self._lm_head_fm = ops.TrackedCopy16(self.lm_head)
        return self

    def _fill_awq(self, lin):
        g = torch.Generator(device=self.device_str)
        g.manual_seed(1234)
        imax = torch.iinfo(torch.int32).max
        lin.qweight.data = torch.randint(0, imax, lin.qweight.shape, dtype=torch.int32, device=self.device_str, generator=g)
        lin.qzeros.data = torch.randint(0, imax, lin.qzeros.shape, dtype=torch.int32, device=self.device_str, generator=g)
        lin.scales.data = (torch.rand(lin.scales.shape, device=self.device_str, generator=g) * 2e-3).to(lin.scales.dtype)
        lin.quant_method.process_weights_after_loading(lin)

    @property
    def lm_head_shuffled(self):
        """The valid fragment-major copy of the LM head (ops.ShuffledWeight16; re-shuffled in place when the head was
        written since the copy was made) or None."""
        t = self._lm_head_fm
        if t is None:
            return None
        if t.src is not self.lm_head:
            self._lm_head_fm = None
            return None
        fm = t.get()
        if fm is None:
            self._lm_head_fm = None
        return fm

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, positions, forward_batch: ForwardBatch):
        # positions index the rotary table on the device, unchecked (as in rotary_embedding.py:79-260): refuse on the host what
        # would read past it whenever the batch carries host-side lengths
        lens_cpu = getattr(forward_batch, "seq_lens_cpu", None)
        if lens_cpu is not None and lens_cpu.numel() and int(lens_cpu.max()) > self.cfg.context_len:
            raise RuntimeError(f"sequence of {int(lens_cpu.max())} tokens exceeds the model's context length "
                               f"({self.cfg.context_len}): positions would index past the rotary table")
        hidden_states = self.embed(input_ids)
        residual = None
        fused = self.fuse_quant and (forward_batch.forward_mode.is_decode() or forward_batch.forward_mode.is_extend())
        # (under tensor parallelism the row-parallel GEMMs decline -- the all-reduce needs their finished output --
        #  while the column-parallel qkv / gate_up still hand their partials to RoPE / SiLU)
        # ... and only for more than 32 rows: the split-K kernels the partials come from are the best GEMM there, while
        # smaller batches have faster single-pass kernels (whole step, ms: bs=1 3.59 vs 4.88 deferred, bs=16 4.24 vs 4.62,
        # bs=32 5.24 vs 5.23, bs=48 6.58 vs 6.44, bs=64 7.17 vs 7.06)
        # ... and (round 4) for SHORT prefills too: up to 128 new tokens the GEMMs are the same split-K weight streamers,
        # and o_proj / down_proj leave their epilogues to the next norm (two finalize launches of ~5 us less per layer; the
        # qkv GEMM keeps its own epilogue there: the extend kernel reads the new tokens' K / V as tensors, not from the pool)
        rows_in = input_ids.shape[0]
        defer = (fused and self.defer_epilogues and rows_in > DEFER_MIN_ROWS
                 and (forward_batch.forward_mode.is_decode()
                      or (forward_batch.forward_mode.is_extend() and rows_in <= DEFER_EXTEND_MAX_ROWS)))
        if defer and FUSE_ATTN_QUANT:  # one zeroed row-absmax vector per layer (the attention kernels max into it)
            rows = input_ids.shape[0]
            if self._attn_absmax is None or self._attn_absmax.shape[1] < rows:
                # sized by the batch.  A graph captured for a smaller batch holds the OLD buffer's address (its zero_() and
                # the attention kernels' atomic max): replaced buffers are kept alive, never freed, and the
                # buffer never grows under capture (the first, eager, pass of a batch size allocates it)
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("the attention row-absmax buffer cannot grow while a graph is being captured: run "
                                       "one eager step at this batch size first")
                if self._attn_absmax is not None:
                    self._attn_absmax_retired.append(self._attn_absmax)
                self._attn_absmax = torch.zeros((len(self.layers), max(64, rows)), dtype=torch.float32,
                                                device=hidden_states.device)
            self._attn_absmax.zero_()
            forward_batch.attn_row_absmax = self._attn_absmax
        else:
            forward_batch.attn_row_absmax = None
        for layer in self.layers:
            if fused:
                hidden_states, residual = layer.forward_fp8(positions, hidden_states, forward_batch, residual, defer)
            else:
                hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states = _arrived(hidden_states)
        normed = False
        if isinstance(hidden_states, ops.GemmPartials):
            if hidden_states.needs_allreduce and residual is not None:
                # the last down_proj left its epilogue AND its all-reduce to the final norm (decode only: no row selection)
                hidden_states, _ = self.norm.forward_with_allreduce_fusion(hidden_states, residual)
                normed = True
            else:
                tagged = hidden_states.needs_allreduce
                hidden_states = hidden_states.finalize()
                if tagged:
                    hidden_states._sglang_needs_allreduce_fusion = True
        pending_ar = _unreduced(hidden_states)  # the last layer's down_proj left its all-reduce to the final norm
        if forward_batch.forward_mode.is_extend() and forward_batch.extend_seq_lens is not None:
            # LogitsProcessor (logits_processor.py:430-470): prefill only needs the last position of every request
            last = torch.cumsum(forward_batch.extend_seq_lens, dim=0) - 1
            hidden_states = hidden_states[last]  # (a new tensor: the tag does not travel with it)
            residual = residual[last] if residual is not None else None
        if normed:
            pass
        elif pending_ar:
            if residual is not None and residual.is_contiguous() and hidden_states.is_contiguous():
                hidden_states, _ = self.norm.forward_with_allreduce_fusion(hidden_states, residual)
            else:
                hidden_states, _ = self.norm(get_tp_group().all_reduce(hidden_states.contiguous()), residual)
        else:
            hidden_states, _ = self.norm(hidden_states, residual)
        if not self.with_lm_head:
            return hidden_states
        # LM head (logits_processor.py:430-505): the 16-bit weight streamer on the fragment-major copy up to 128 rows (round 3:
        # M = 1 / 16 / 64 155 / 159 / 184 us on the 128256 x 4096 head against 183-187 / 187-188 / 210 for the library GEMM),
        # the row-major streamer when there is no copy, the library GEMM above that
        rows = hidden_states.shape[0]
        head_fm = self.lm_head_shuffled if rows <= min(128, LM_HEAD_SHUFFLED_MAX_ROWS) else None
        if head_fm is not None:
            logits = ops.linear16(hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous(), head_fm)
        elif rows <= LM_HEAD_STREAMER_MAX_ROWS and \
                ops.linear16_supported(rows, self.lm_head.shape[0], self.lm_head.shape[1]):
            logits = ops.linear16(hidden_states, self.lm_head)
        else:
            logits = torch.matmul(hidden_states, self.lm_head.t())
        if get_tensor_model_parallel_world_size() > 1:
            logits = get_tp_group().all_gather(logits, dim=-1)
        return logits


LLAMA3_8B = ModelConfig(num_attention_heads=32, num_key_value_heads=8, head_dim=128, hidden_size=4096,
                        intermediate_size=14336, num_hidden_layers=32, vocab_size=128256, context_len=8192,
                        rms_norm_eps=1e-5, rope_theta=500000.0)
LLAMA3_70B = ModelConfig(num_attention_heads=64, num_key_value_heads=8, head_dim=128, hidden_size=8192,
                         intermediate_size=28672, num_hidden_layers=80, vocab_size=128256, context_len=8192,
                         rms_norm_eps=1e-5, rope_theta=500000.0)
LLAMA2_7B = ModelConfig(num_attention_heads=32, num_key_value_heads=32, head_dim=128, hidden_size=4096,
                        intermediate_size=11008, num_hidden_layers=32, vocab_size=32000, context_len=4096,
                        rms_norm_eps=1e-5, rope_theta=10000.0)
QWEN2_05B = ModelConfig(num_attention_heads=14, num_key_value_heads=2, head_dim=64, hidden_size=896,
                        intermediate_size=4864, num_hidden_layers=24, vocab_size=151936, context_len=32768,
                        rms_norm_eps=1e-6, rope_theta=1000000.0)
