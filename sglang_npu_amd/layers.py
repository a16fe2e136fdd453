"""Elementwise layers around the hot path, same class names / forward signatures as the reference.

  RMSNorm          python/sglang/srt/layers/layernorm.py:59-172   (forward(x, residual=None))
  SiluAndMul       python/sglang/srt/layers/activation.py:59-83
  RotaryEmbedding  python/sglang/srt/layers/rotary_embedding.py:79-260 (fp32 cos/sin cache, neox)
  Sampler (greedy) python/sglang/srt/layers/sampler.py:72-75       (ops.argmax: first maximal index, like torch.argmax)
  VocabParallelEmbedding  python/sglang/srt/layers/vocab_parallel_embedding.py:153-486 (original vocabulary only)
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import ops
from .deferred import DeferredEpilogue, rope_target


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype=torch.bfloat16):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size
        # set by the first W8A8Fp8LinearMethod.apply that receives this layer's output (ops.take_fp8_companion): from then on
        # the norm kernel also emits the per-token FP8 quantisation of its output, in the same pass
        self.emit_fp8_companion = False

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if x.__class__ is not torch.Tensor and isinstance(x, DeferredEpilogue):
            # a row-parallel FP8 GEMM of this backend still in split-K partials (deferred.py): its epilogue runs inside the norm
            # kernel -- finalize + fused_add_rmsnorm (+ the per-token FP8 quant the next FP8 linear asked for), bit-identical
            if x.needs_allreduce and x.is_pending():
                # a row-parallel layer's unreduced output under tensor parallelism: the collective, the residual add and the norm
                # as one kernel where the P2P communicator takes the shape (forward_with_allreduce_fusion falls back to
                # all-reduce + norm otherwise) -- what upstream reaches through can_fuse_mlp_allreduce (layernorm.py:191-216)
                src = x.pending_partials()
                if src is None:
                    src = x.pending_local()
                    src._sglang_needs_allreduce_fusion = True
                else:
                    src.needs_allreduce = True
                res = self.forward_with_allreduce_fusion(src, residual)
                x.resolve(res[0] if isinstance(res, tuple) else res)
                return res
            part = x.pending_partials()
            if (part is not None and residual is not None and residual.is_cuda and residual.is_contiguous()
                    and tuple(residual.shape) == (part.M, part.N) and residual.dtype == part.out_dtype
                    and self.weight.dtype == part.out_dtype):
                if ops.FP8_COMPANIONS and self.emit_fp8_companion:
                    out, q, s = ops.fused_add_rmsnorm_from_partials(part, residual, self.weight.data, self.variance_epsilon, True)
                    ops.attach_fp8_companion(out, q, s)
                else:
                    out = ops.fused_add_rmsnorm_from_partials(part, residual, self.weight.data, self.variance_epsilon)
                    out._sgl_mi355_producer = self
                x.resolve(out)  # (the reference's fused_add_rmsnorm is in place: x now IS the normed row)
                return out, residual
            x = x.materialize()
        elif residual is not None:
            prod = getattr(x, "_sgl_mi355_epilogue_producer", None)
            if prod is not None:
                prod._sgl_mi355_defer_epilogue = True  # from the next pass on this GEMM leaves its epilogue to this norm
        if ops.FP8_COMPANIONS and x.is_cuda and x.dim() == 2 and x.is_contiguous():
            if self.emit_fp8_companion:
                # one kernel: (add +) norm -> 16-bit `out` + (q, scale) = sgl_per_token_quant_fp8(out), bit for bit
                q, s, out = ops.rmsnorm_quant_fp8(x, self.weight.data, self.variance_epsilon, residual=residual, want_out=True)
                ops.attach_fp8_companion(out, q, s)
                return out if residual is None else (out, residual)
            if residual is not None:
                ops.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
                x._sgl_mi355_producer = self
                return x, residual
            out = ops.rmsnorm(x, self.weight.data, self.variance_epsilon)
            out._sgl_mi355_producer = self
            return out
        if residual is not None:
            ops.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return ops.rmsnorm(x, self.weight.data, self.variance_epsilon)

    def forward_with_allreduce_fusion(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, quant_fp8: bool = False):
        """layers/layernorm.py:191-216: `x` is a row-parallel GEMM's output whose all-reduce was skipped
        (RowParallelLinear.forward(..., can_fuse_mlp_allreduce=True), linear.py:1285-1303); do the collective, the
        residual add and the norm here -- in one kernel when the P2P communicator takes the shape, else the plain
        sequence.  quant_fp8: return ((q, scale), residual) for the FP8 linear that follows."""
        from .distributed import get_tp_group, tensor_model_parallel_all_reduce
        tp = get_tp_group()
        ca = tp.ca_comm
        if isinstance(x, ops.GemmPartials):  # the row-parallel GEMM left its epilogue here as well (needs_allreduce)
            if (residual is not None and tp.world_size > 1 and tp.fused_collectives_on and ca is not None
                    and ca.should_fuse_norm_shape(x.M, x.N, x.out_dtype)):
                if not quant_fp8 and ops.FP8_COMPANIONS:  # the reference call order (see the tensor form below)
                    if self.emit_fp8_companion:
                        out, q, s = ca.fused_add_rmsnorm_partials(x, residual, self.weight.data, self.variance_epsilon,
                                                                  with_fp8_companion=True)
                        return ops.attach_fp8_companion(out, q, s), residual
                    out = ca.fused_add_rmsnorm_partials(x, residual, self.weight.data, self.variance_epsilon)
                    out._sgl_mi355_producer = self
                    return out, residual
                r = ca.fused_add_rmsnorm_partials(x, residual, self.weight.data, self.variance_epsilon, quant_fp8)
                return r, residual
            x = x.finalize()
        if residual is not None and tp.world_size > 1 and tp.fused_collectives_on and ca is not None and ca.should_fuse_norm(x):
            if not quant_fp8 and ops.FP8_COMPANIONS:
                # the reference call order under TP: the 16-bit result goes to an FP8 linear next -- once that linear has asked
                # (emit_fp8_companion, see forward), the same launch also quantises it (out AND (q, scale))
                if self.emit_fp8_companion:
                    out, q, s = ca.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon, with_fp8_companion=True)
                    return ops.attach_fp8_companion(out, q, s), residual
                out = ca.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
                out._sgl_mi355_producer = self
                return out, residual
            r = ca.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon, quant_fp8)
            return r, residual
        x = tensor_model_parallel_all_reduce(x)
        if quant_fp8:
            return self.forward_quant_fp8(x, residual)
        return self.forward(x, residual)

    def forward_quant_fp8(self, x, residual=None):
        """Fused (add +) norm + per-token FP8 quant; returns ((q, scale), residual)."""
        q, s, _ = ops.rmsnorm_quant_fp8(x, self.weight.data, self.variance_epsilon, residual=residual)
        return (q, s), (residual if residual is not None else None)


class SiluAndMul(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.emit_fp8_companion = False  # see RMSNorm

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.__class__ is not torch.Tensor and isinstance(x, DeferredEpilogue):
            act = x.silu_act if x.is_pending() else None
            if act is not None:
                return act  # the gate_up GEMM ran with this activation in its epilogue (deferred.py); x stays a valid handle
            x = x.materialize()
        else:
            prod = getattr(x, "_sgl_mi355_epilogue_producer", None)
            if prod is not None and getattr(prod, "_sgl_mi355_may_fuse_silu", False):
                prod._sgl_mi355_fuse_silu = True  # from the next pass on that GEMM computes SiLU * up itself
        if ops.FP8_COMPANIONS and x.is_cuda and x.is_contiguous():
            if self.emit_fp8_companion:
                out, q, s = ops.silu_and_mul_with_quant_fp8(x)
                return ops.attach_fp8_companion(out, q, s)
            out = ops.silu_and_mul(x)
            out._sgl_mi355_producer = self
            return out
        return ops.silu_and_mul(x)


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool = True, dtype=torch.bfloat16, device=None):
        super().__init__()
        self.head_size, self.rotary_dim, self.is_neox_style = head_size, rotary_dim, is_neox_style
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1)  # kept in fp32 (rotary_embedding.py:98-100)
        self.register_buffer("cos_sin_cache", cache.to(device) if device is not None else cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        if query.__class__ is not torch.Tensor:
            root = rope_target(query, key)
            if (root is not None and self.rotary_dim == self.head_size and query.shape[1] % self.head_size == 0
                    and key.shape[1] % self.head_size == 0 and positions.is_cuda):
                # q / k are column ranges of a qkv projection still in split-K partials (deferred.py): the rotation is recorded
                # and happens where the partials are finished -- inside the attention backend's RoPE + KV-write launch, or with
                # this very method on the finished tensor if anybody else reads q / k first.  In place, so q / k are returned.
                root._rope = (positions, self, (query._c0, query._c1), (key._c0, key._c1))
                return query, key
        ops.apply_rope_with_cos_sin_cache_inplace(positions, query, key, self.head_size, self.cos_sin_cache,
                                                  self.is_neox_style)
        return query, key


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    """Sampler.forward with temperature 0 (sampler.py:72-75): argmax over the vocabulary."""
    return ops.argmax(logits.view(-1, logits.shape[-1]))


DEFAULT_VOCAB_PADDING_SIZE = 64  # vocab_parallel_embedding.py:36


def pad_vocab_size(vocab_size: int, pad_to: int = DEFAULT_VOCAB_PADDING_SIZE) -> int:
    """vocab_parallel_embedding.py:44-46."""
    return ((vocab_size + pad_to - 1) // pad_to) * pad_to


def vocab_shard_range(org_vocab_size: int, padded_vocab_size: int, rank: int, world: int):
    """The original-vocabulary part of VocabParallelEmbedding._get_indices (vocab_parallel_embedding.py:49-62, 273-322):
    the padded vocabulary is cut into `world` equal parts; a rank holds the real entries of its part.
    Returns (org_vocab_start, org_vocab_end, rows_per_partition)."""
    if padded_vocab_size % world != 0:
        raise ValueError(f"padded vocabulary {padded_vocab_size} is not divisible by the TP size {world}")
    per = padded_vocab_size // world
    start = min(rank * per, org_vocab_size)
    end = min((rank + 1) * per, org_vocab_size)
    return start, end, per


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding cut along the vocabulary over the TP ranks (vocab_parallel_embedding.py:153-486, original vocabulary
    only -- no added / LoRA entries): `weight` [padded_vocab / tp, H] holds rows [org_vocab_start, org_vocab_end) of the
    table (+ zero padding); forward = masked gather + masked fill (one kernel) + all-reduce over the TP group (:462-486)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype=torch.bfloat16,
                 padding_size: int = DEFAULT_VOCAB_PADDING_SIZE):
        super().__init__()
        from .distributed import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
        self.tp_size, self.tp_rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.org_vocab_size, self.embedding_dim = num_embeddings, embedding_dim
        # padded so that every rank gets whole blocks of `padding_size` rows (:203-216 pads to padding_size and asserts
        # the divisibility; padding_size * tp makes it hold for any vocabulary)
        self.num_embeddings_padded = pad_vocab_size(num_embeddings, padding_size * self.tp_size)
        self.org_vocab_start_index, self.org_vocab_end_index, self.num_embeddings_per_partition = vocab_shard_range(
            num_embeddings, self.num_embeddings_padded, self.tp_rank, self.tp_size)
        self.weight = torch.nn.Parameter(torch.zeros(self.num_embeddings_per_partition, embedding_dim, dtype=params_dtype),
                                         requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param, loaded_weight: torch.Tensor):
        """The FULL [org_vocab, H] table -> this rank's rows, zero padding behind them (:405-460)."""
        assert loaded_weight.shape[0] == self.org_vocab_size, (loaded_weight.shape, self.org_vocab_size)
        n = self.org_vocab_end_index - self.org_vocab_start_index
        param.data[:n].copy_(loaded_weight[self.org_vocab_start_index:self.org_vocab_end_index])
        param.data[n:].fill_(0)

    def forward(self, input_: torch.Tensor) -> torch.Tensor:
        from .distributed import tensor_model_parallel_all_reduce
        out = ops.vocab_parallel_embedding(input_, self.weight.data, self.org_vocab_start_index, self.org_vocab_end_index)
        if self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
