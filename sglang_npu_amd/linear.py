"""Tensor-parallel linear layers that dispatch to a quant method.

Interface mirrored: python/sglang/srt/layers/linear.py
  LinearBase.__init__ (:125-146), ColumnParallelLinear.forward (:401-413),
  MergedColumnParallelLinear (:416-), QKVParallelLinear (:728-), RowParallelLinear.forward (:1285-1309)
``quant_method = quant_config.get_quant_method(layer, prefix)``; ``out = quant_method.apply(layer, x, bias)``;
row-parallel: bias only on rank 0 (:1298), then all-reduce (:1302-1303).

Checkpoint loading follows the reference too: each layer hands ``weight_loader=self.weight_loader`` to
``create_weights`` (:306-319), the parameters keep it, and a model loader calls
``param.weight_loader(param, full_tensor[, shard_id])`` (models/llama.py:600-625); the layer then works out this
rank's slice (``weight_loader_v2`` of :383-400, 674-726, 877-913, 1264-1283) and lets the parameter copy it.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .distributed import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size, get_tp_group,
                          tensor_model_parallel_all_reduce)
from . import deferred
from .deferred import DeferredEpilogue
from .quantization import QuantizationConfig, UnquantizedLinearMethod


class LinearBase(torch.nn.Module):
    # hand-over switches of deferred.py, read on every apply(): class-level defaults so that the lookups never fall through to
    # nn.Module.__getattr__ (a raised AttributeError per miss, ~1 us each, seven per decoder layer in an eager pass)
    _sgl_mi355_may_defer = False       # may leave its decode-time epilogue to the consumer (row-parallel, qkv)
    _sgl_mi355_defer_epilogue = False  # ... and the consumer has asked
    _sgl_mi355_may_fuse_silu = False   # gate_up: SiLU * up may run in the prefill GEMM's epilogue
    _sgl_mi355_fuse_silu = False       # ... and SiluAndMul has asked
    _sgl_mi355_is_qkv = False
    _sgl_mi355_partials_ok = None      # optional predicate (rows, dtype) -> bool

    def __init__(self, input_size: int, output_size: int, params_dtype: torch.dtype = torch.bfloat16,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = ""):
        super().__init__()
        self.input_size, self.output_size, self.params_dtype = input_size, output_size, params_dtype
        self.tp_rank, self.tp_size = get_tensor_model_parallel_rank(), get_tensor_model_parallel_world_size()
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
        self.output_sizes = list(output_sizes)
        self.quant_method.create_weights(self, input_size, self.output_partition_sizes, input_size, self.output_size,
                                         params_dtype, weight_loader=self.weight_loader)
        self.bias = torch.nn.Parameter(torch.zeros(self.output_size_per_partition, dtype=params_dtype),
                                       requires_grad=False) if bias else None

    def weight_loader(self, param, loaded_weight: torch.Tensor, loaded_shard_id=None):
        """linear.py:383-400 / 674-726: one full (unsharded) checkpoint tensor -> this rank's rows.  With a shard id
        (an index into ``output_sizes``) the tensor is that sub-matrix alone; without one it is the whole fused
        matrix and is split here."""
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        if len(self.output_sizes) == 1:
            param.load_column_parallel_weight(loaded_weight, tp_rank=self.tp_rank)
            return
        if loaded_shard_id is None:  # fused on disk: walk the logical sub-matrices (:638-672)
            off = 0
            for i, size in enumerate(self.output_sizes):
                lo, n = off, size
                if getattr(param, "packed_dim", None) == param.output_dim:
                    n, lo = param.adjust_shard_indexes_for_packing(shard_size=n, shard_offset=lo)
                self.weight_loader(param, loaded_weight.narrow(param.output_dim, lo, n), i)
                off += size
            return
        param.load_merged_column_weight(loaded_weight, shard_id=loaded_shard_id,
                                        shard_offset=sum(self.output_sizes[:loaded_shard_id]) // self.tp_size,
                                        shard_size=self.output_sizes[loaded_shard_id] // self.tp_size,
                                        tp_rank=self.tp_rank, tp_size=self.tp_size)

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

    def __init__(self, input_size, output_sizes: List[int], bias: bool = False, params_dtype=torch.bfloat16,
                 quant_config=None, prefix: str = ""):
        super().__init__(input_size, output_sizes, bias, params_dtype, quant_config, prefix)
        # [gate | up] of equal width: SiLU(gate) * up may run in the prefill GEMM's epilogue once the SiluAndMul that consumes
        # the output has asked (quantization.W8A8Fp8LinearMethod.apply, deferred.py)
        self._sgl_mi355_may_fuse_silu = len(output_sizes) == 2 and output_sizes[0] == output_sizes[1]
        self._sgl_mi355_fuse_silu = False

    def forward_prequantized_silu_mul(self, qinput, x_scale, out_dtype):
        """SiluAndMul(forward_prequantized(...)[0]) in the GEMM's own launch where the quant method has that form
        (w8a8 FP8 at prefill sizes), else None."""
        fn = getattr(self.quant_method, "apply_prequantized_silu_mul", None)
        return fn(self, qinput, x_scale, out_dtype, self.bias) if fn is not None else None


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
        self.head_size, self.total_num_heads, self.total_num_kv_heads = head_size, total_num_heads, total_num_kv_heads
        self.num_kv_head_replicas = rep
        sizes = [total_num_heads * head_size, total_num_kv_heads * rep * head_size, total_num_kv_heads * rep * head_size]
        super().__init__(hidden_size, sizes, bias, params_dtype, quant_config, prefix)
        # column-parallel, no collective: at decode sizes the epilogue may be left to the attention backend's fused RoPE +
        # KV-write (deferred.py: the output travels as a lazy tensor through split / rotary_emb / RadixAttention); any TP size
        self._sgl_mi355_may_defer = True
        self._sgl_mi355_defer_epilogue = False
        self._sgl_mi355_is_qkv = True

    def forward(self, x):
        out = self.quant_method.apply(self, x, self.bias)
        if deferred.DEFERRED_EPILOGUES and type(out) is torch.Tensor and out.dim() == 2 and out.is_cuda:
            # outside the FP8 split-K window (prefill, small decode batches, 16-bit and AWQ weights) the finished tensor goes out
            # behind a lazy handle all the same once the attention backend has asked: rotary_emb then RECORDS the rotation and the
            # backend runs RoPE + KV-pool write as one launch (ops.apply_rope_and_set_kv_buffer) instead of two -- and nobody
            # can see the unrotated q / k meanwhile, every access goes through the handle (deferred.py)
            # (only where it cannot cost more host time than it saves on the GPU: under graph capture, or in a pass long enough
            #  to be GPU-bound -- the lazy split / views are ~15 us of Python per layer against one ~5 us launch)
            if self._sgl_mi355_defer_epilogue and (out.shape[0] >= 256 or torch.cuda.is_current_stream_capturing()):
                return DeferredEpilogue(local=out), None
            out._sgl_mi355_epilogue_producer = self  # (the backend finds it through q._base on the first plain pass)
            if out._base is not None:
                out._base._sgl_mi355_epilogue_producer = self
        return out, None

    def weight_loader(self, param, loaded_weight: torch.Tensor, loaded_shard_id=None):
        """linear.py:877-913: shard ids "q" / "k" / "v"; the checkpoint's k / v hold ``total_num_kv_heads`` heads and
        rank r takes head ``r // num_kv_head_replicas``."""
        if loaded_shard_id is None:  # fused qkv on disk (:832-875)
            hs, off = self.head_size, 0
            for sid, n in (("q", self.total_num_heads * hs), ("k", self.total_num_kv_heads * hs),
                           ("v", self.total_num_kv_heads * hs)):
                lo, m = off, n
                if getattr(param, "packed_dim", None) == param.output_dim:
                    m, lo = param.adjust_shard_indexes_for_packing(shard_size=m, shard_offset=lo)
                self.weight_loader(param, loaded_weight.narrow(param.output_dim, lo, m), sid)
                off += n
            return
        assert loaded_shard_id in ("q", "k", "v")
        q_size, kv_size = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        offset = {"q": 0, "k": q_size, "v": q_size + kv_size}[loaded_shard_id]
        param.load_qkv_weight(loaded_weight, num_heads=self.num_kv_head_replicas, shard_id=loaded_shard_id,
                              shard_offset=offset, shard_size=q_size if loaded_shard_id == "q" else kv_size,
                              tp_rank=self.tp_rank)


class RowParallelLinear(LinearBase):
    def __init__(self, input_size, output_size, bias: bool = False, reduce_results: bool = True,
                 params_dtype=torch.bfloat16, quant_config=None, prefix: str = ""):
        super().__init__(input_size, output_size, params_dtype, quant_config, prefix)
        tp = get_tensor_model_parallel_world_size()
        assert input_size % tp == 0
        self.input_size_per_partition = input_size // tp
        self.reduce_results = reduce_results
        self.quant_method.create_weights(self, self.input_size_per_partition, [output_size], input_size, output_size,
                                         params_dtype, weight_loader=self.weight_loader)
        self.bias = torch.nn.Parameter(torch.zeros(output_size, dtype=params_dtype), requires_grad=False) if bias else None
        # no collective behind this GEMM: its decode-time epilogue may be left to the RMSNorm that consumes the output
        # (quantization.W8A8Fp8LinearMethod.apply, deferred.py); under TP the all-reduce needs the finished output
        self._sgl_mi355_may_defer = True
        self._sgl_mi355_defer_epilogue = False

    def _sgl_mi355_partials_ok(self, rows: int, dtype) -> bool:
        """May W8A8Fp8LinearMethod.apply leave this GEMM as split-K partials (deferred.py)?  Without a collective: always.  Under
        TP only where the next norm can take them into the fused all-reduce + add + RMSNorm kernel, and only where the GEMM
        would run split-K + finalize anyway (the rule of forward_prequantized_partials below)."""
        if get_tensor_model_parallel_world_size() == 1:
            return True
        tp = get_tp_group()
        ca = tp.ca_comm
        return (self.reduce_results and tp.fused_collectives_on and ca is not None
                and ca.should_fuse_norm_shape(rows, self.output_size, dtype)
                and self.output_size * self.input_size_per_partition >= ((20 if rows > 32 else 40) << 20))

    def weight_loader(self, param, loaded_weight: torch.Tensor):
        """linear.py:1264-1283: this rank's slice along the parameter's input dimension; per-output-channel scales
        have none and are copied whole."""
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        if hasattr(param, "input_dim"):
            param.load_row_parallel_weight(loaded_weight, tp_rank=self.tp_rank)
        else:
            param.load_row_parallel_weight(loaded_weight)

    def _can_fuse(self, out) -> bool:
        ca = get_tp_group().ca_comm
        return (self.reduce_results and get_tensor_model_parallel_world_size() > 1 and get_tp_group().fused_collectives_on
                and ca is not None and out.dim() == 2 and ca.should_fuse_norm(out))

    def _reduce(self, out, async_reduce: bool, can_fuse_mlp_allreduce: bool = False):
        if can_fuse_mlp_allreduce and self._can_fuse(out):
            # linear.py:1302: the collective is left to the next norm (RMSNorm.forward_with_allreduce_fusion); the
            # partial sum travels tagged the way upstream tags it (communicator.py:190-199, deepseek_v2.py:1924)
            out._sglang_needs_allreduce_fusion = True
            return out
        return self._reduce_now(out, async_reduce)

    def _reduce_now(self, out, async_reduce: bool):
        """linear.py:1302-1303.  async_reduce: the collective goes to the TP group's side stream and an
        AllReduceHandle comes back instead of the tensor; its consumer (the next norm) calls wait(), which fences the
        main stream on the collective's event -- whatever the main stream launches in between overlaps it."""
        if not (self.reduce_results and get_tensor_model_parallel_world_size() > 1):
            return out
        if async_reduce and out.is_cuda:
            return get_tp_group().all_reduce_async(out)
        return tensor_model_parallel_all_reduce(out)

    def forward(self, x, async_reduce: bool = False, can_fuse_mlp_allreduce: bool = False):
        bias_ = None if (get_tensor_model_parallel_rank() > 0) else self.bias
        out = self.quant_method.apply(self, x, bias_)
        if type(out) is not torch.Tensor and isinstance(out, DeferredEpilogue) and get_tensor_model_parallel_world_size() > 1:
            if self.reduce_results and not async_reduce and not can_fuse_mlp_allreduce:
                out.needs_allreduce = True  # split-K partials of this rank's addend: whoever finishes them owes the collective
                return out, None
            out = out.materialize()  # (a caller with its own plan for the collective gets the finished local sum)
        if (self.reduce_results and not async_reduce and not can_fuse_mlp_allreduce and deferred.DEFERRED_EPILOGUES
                and get_tensor_model_parallel_world_size() > 1 and type(out) is torch.Tensor):
            # the call untouched model code makes (models/llama.py:97,190): the all-reduce is this layer's.  When the RMSNorm that
            # consumed the previous output has asked and the P2P communicator takes the shape, hand it the unreduced sum as a
            # lazy tensor (deferred.py): it runs all-reduce + add + norm as one kernel; anybody else gets the plain collective
            if self._sgl_mi355_defer_epilogue and self._can_fuse(out):
                return DeferredEpilogue(local=out, needs_allreduce=True), None
            res = self._reduce_now(out, False)
            res._sgl_mi355_epilogue_producer = self
            return res, None
        return self._reduce(out, async_reduce, can_fuse_mlp_allreduce), None

    def forward_prequantized_partials(self, qinput, x_scale, out_dtype, can_fuse_mlp_allreduce: bool = False):
        """Split-K partials for a fused consumer.  Under tensor parallelism the all-reduce needs the completed output --
        unless the next norm runs it as all-reduce + add + RMSNorm in one kernel (can_fuse_mlp_allreduce and the P2P
        communicator takes the shape): that kernel runs the GEMM epilogue itself while it stages this rank's rows
        (CustomAllreduce.fused_add_rmsnorm_partials), and the partials travel tagged `needs_allreduce`."""
        fn = getattr(self.quant_method, "apply_prequantized_partials", None)
        if fn is None:
            return None
        if get_tensor_model_parallel_world_size() > 1:
            tp = get_tp_group()
            ca = tp.ca_comm
            if not (can_fuse_mlp_allreduce and self.reduce_results and tp.fused_collectives_on and ca is not None
                    and qinput.dim() == 2
                    and ca.should_fuse_norm_shape(qinput.shape[0], self.output_size, out_dtype)):
                return None
            # only where the GEMM would run split-K + finalize anyway (gemm_fp8.hip run_gemm: from 20 Mi weights at more
            # than 32 rows, 40 Mi below): elsewhere the single-pass kernels are faster than forcing the split-K form
            # (Llama-3-8B TP = 2 o_proj, 2048 x 4096: 7.6 us single pass, 11.1 us split-K + finalize)
            rows = qinput.shape[0]
            if self.output_size * self.input_size_per_partition < ((20 if rows > 32 else 40) << 20):
                return None
            part = fn(self, qinput, x_scale, out_dtype, None if get_tensor_model_parallel_rank() > 0 else self.bias)
            if part is not None:
                part.needs_allreduce = True
            return part
        return fn(self, qinput, x_scale, out_dtype, self.bias)

    def forward_a16_partials(self, x16, row_absmax, out_dtype):
        """forward_prequantized_partials on 16-bit activations whose per-token absmax is known: the GEMM quantises while
        staging (ops.fp8_scaled_mm_partials_a16).  None when the method or the shape has no such form, or under TP."""
        fn = getattr(self.quant_method, "apply_a16_partials", None)
        if fn is None or get_tensor_model_parallel_world_size() > 1:
            return None
        return fn(self, x16, row_absmax, out_dtype, self.bias)

    def forward_prequantized(self, qinput, x_scale, out_dtype, async_reduce: bool = False,
                             can_fuse_mlp_allreduce: bool = False):
        bias_ = None if (get_tensor_model_parallel_rank() > 0) else self.bias
        out = self.quant_method.apply_prequantized(self, qinput, x_scale, out_dtype, bias_)
        return self._reduce(out, async_reduce, can_fuse_mlp_allreduce), None
