"""Tensor-parallel linear layers that dispatch to a quant method.

Interface mirrored: python/sglang/srt/layers/linear.py
  LinearBase.__init__ (:125-146), ColumnParallelLinear.forward (:401-413),
  MergedColumnParallelLinear (:416-), QKVParallelLinear (:728-), RowParallelLinear.forward (:1285-1309)
``quant_method = quant_config.get_quant_method(layer, prefix)``; ``out = quant_method.apply(layer, x, bias)``;
row-parallel: bias only on rank 0 (:1298), then all-reduce (:1302-1303).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                          tensor_model_parallel_all_reduce)
from .quantization import QuantizationConfig, UnquantizedLinearMethod


class LinearBase(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, params_dtype: torch.dtype = torch.bfloat16,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = ""):
        super().__init__()
        self.input_size, self.output_size, self.params_dtype = input_size, output_size, params_dtype
        self.quant_method = UnquantizedLinearMethod() if quant_config is None else \
            quant_config.get_quant_method(self, prefix=prefix)


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size, output_sizes: List[int], bias: bool = False, params_dtype=torch.bfloat16,
                 quant_config=None, prefix: str = ""):
        super().__init__(input_size, sum(output_sizes), params_dtype, quant_config, prefix)
        tp = get_tensor_model_parallel_world_size()
        for s in output_sizes:
            assert s % tp == 0, f"output size {s} is not divisible by tp={tp}"
        self.output_partition_sizes = [s // tp for s in output_sizes]
        self.output_size_per_partition = sum(self.output_partition_sizes)
        self.quant_method.create_weights(self, input_size, self.output_partition_sizes, input_size, self.output_size,
                                         params_dtype)
        self.bias = torch.nn.Parameter(torch.zeros(self.output_size_per_partition, dtype=params_dtype),
                                       requires_grad=False) if bias else None

    def forward(self, x):
        return self.quant_method.apply(self, x, self.bias), None

    def forward_prequantized(self, qinput, x_scale, out_dtype):
        return self.quant_method.apply_prequantized(self, qinput, x_scale, out_dtype, self.bias), None

    def forward_prequantized_partials(self, qinput, x_scale, out_dtype):
        """Split-K partials for a fused consumer (ops.GemmPartials) or None."""
        fn = getattr(self.quant_method, "apply_prequantized_partials", None)
        return fn(self, qinput, x_scale, out_dtype, self.bias) if fn is not None else None


class MergedColumnParallelLinear(ColumnParallelLinear):
    """gate_up_proj: two column-parallel matrices stored as one."""


class QKVParallelLinear(ColumnParallelLinear):
    """Q/K/V projections fused; KV heads are replicated when there are fewer of them than ranks."""

    def __init__(self, hidden_size, head_size, total_num_heads, total_num_kv_heads, bias=False,
                 params_dtype=torch.bfloat16, quant_config=None, prefix: str = ""):
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = total_num_heads // tp
        if tp >= total_num_kv_heads:
            self.num_kv_heads, rep = 1, tp // total_num_kv_heads
        else:
            self.num_kv_heads, rep = total_num_kv_heads // tp, 1
        sizes = [total_num_heads * head_size, total_num_kv_heads * rep * head_size, total_num_kv_heads * rep * head_size]
        super().__init__(hidden_size, sizes, bias, params_dtype, quant_config, prefix)


class RowParallelLinear(LinearBase):
    def __init__(self, input_size, output_size, bias: bool = False, reduce_results: bool = True,
                 params_dtype=torch.bfloat16, quant_config=None, prefix: str = ""):
        super().__init__(input_size, output_size, params_dtype, quant_config, prefix)
        tp = get_tensor_model_parallel_world_size()
        assert input_size % tp == 0
        self.input_size_per_partition = input_size // tp
        self.reduce_results = reduce_results
        self.quant_method.create_weights(self, self.input_size_per_partition, [output_size], input_size, output_size,
                                         params_dtype)
        self.bias = torch.nn.Parameter(torch.zeros(output_size, dtype=params_dtype), requires_grad=False) if bias else None

    def forward(self, x):
        bias_ = None if (get_tensor_model_parallel_rank() > 0) else self.bias
        out = self.quant_method.apply(self, x, bias_)
        if self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out, None

    def forward_prequantized_partials(self, qinput, x_scale, out_dtype):
        """Split-K partials for a fused consumer; only without tensor parallelism (the all-reduce needs the
        completed output)."""
        fn = getattr(self.quant_method, "apply_prequantized_partials", None)
        if fn is None or get_tensor_model_parallel_world_size() > 1:
            return None
        return fn(self, qinput, x_scale, out_dtype, self.bias)

    def forward_prequantized(self, qinput, x_scale, out_dtype):
        bias_ = None if (get_tensor_model_parallel_rank() > 0) else self.bias
        out = self.quant_method.apply_prequantized(self, qinput, x_scale, out_dtype, bias_)
        if self.reduce_results and get_tensor_model_parallel_world_size() > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out, None
