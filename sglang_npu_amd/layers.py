"""Elementwise layers around the hot path, same class names / forward signatures as the reference.

  RMSNorm          python/sglang/srt/layers/layernorm.py:59-172   (forward(x, residual=None))
  SiluAndMul       python/sglang/srt/layers/activation.py:59-83
  RotaryEmbedding  python/sglang/srt/layers/rotary_embedding.py:79-260 (fp32 cos/sin cache, neox)
  Sampler (greedy) python/sglang/srt/layers/sampler.py:72-75       (ops.argmax: first maximal index, like torch.argmax)
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from . import ops


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype=torch.bfloat16):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
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
            if (residual is not None and tp.world_size > 1 and not tp.stub_all_reduce and ca is not None
                    and ca.should_fuse_norm_shape(x.M, x.N, x.out_dtype)):
                r = ca.fused_add_rmsnorm_partials(x, residual, self.weight.data, self.variance_epsilon, quant_fp8)
                return r, residual
            x = x.finalize()
        if residual is not None and tp.world_size > 1 and not tp.stub_all_reduce and ca is not None and ca.should_fuse_norm(x):
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
    def forward(self, x: torch.Tensor) -> torch.Tensor:
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
        ops.apply_rope_with_cos_sin_cache_inplace(positions, query, key, self.head_size, self.cos_sin_cache,
                                                  self.is_neox_style)
        return query, key


def greedy_sample(logits: torch.Tensor) -> torch.Tensor:
    """Sampler.forward with temperature 0 (sampler.py:72-75): argmax over the vocabulary."""
    return ops.argmax(logits.view(-1, logits.shape[-1]))
