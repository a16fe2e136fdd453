"""Quant-linear methods of the MI355X backend -- the drop-in behind SGLang's quant-linear operator API.

Interface mirrored (same method names / argument meaning):
  QuantizationConfig / LinearMethodBase   python/sglang/srt/layers/quantization/base_config.py:15-80,113-203
  W8A8Fp8Config / W8A8Fp8LinearMethod      python/sglang/srt/layers/quantization/w8a8_fp8.py:34-189
  apply_fp8_linear                          python/sglang/srt/layers/quantization/fp8_utils.py:510-533,653-704
  AWQConfig / AWQLinearMethod               python/sglang/srt/layers/quantization/awq.py:75-150,319-418
  QUANTIZATION_METHODS registry             python/sglang/srt/layers/quantization/__init__.py:74-121

``create_weights`` registers the same attributes on the layer (``weight``, ``weight_scale`` /
``qweight``, ``qzeros``, ``scales``) with the same shapes, dtypes AND loader metadata: they are
``ModelWeightParameter`` / ``ChannelQuantScaleParameter`` / ``PackedvLLMParameter`` /
``GroupQuantScaleParameter`` instances (SGLang's own classes when SGLang is importable, same-named local
ones otherwise -- ``parameter.classes()``) carrying the ``weight_loader`` the layer passes in
``extra_weight_attrs`` (w8a8_fp8.py:153-173, awq.py:354-394), so SGLang's checkpoint loaders shard and fill
them unchanged.  The method classes keep the reference's names because ``layers/linear.py:44-61,315``
picks ``weight_loader_v2`` by ``quant_method.__class__.__name__``.
``process_weights_after_loading`` stores ``weight.t()``
(K-major [K,N] view) exactly like w8a8_fp8.py:115,132; ``apply`` runs the HIP kernels.  gfx950 uses
OCP e4m3fn (is_fp8_fnuz() is False there), so no fnuz normalisation step exists here.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import os

import torch

from . import deferred, ops
from .parameter import classes as _param_classes, is_linear_layer, raw_data, tracked

# FP8 weights are re-laid fragment-major after loading when the shape allows (N % 16 == 0, K % 512 == 0): decode GEMMs
# 31.5 -> 25.9 us (gate_up, M = 64), the headline step 6.34 -> 6.16 ms.  SGL_MI355_NO_WSHUFFLE=1 keeps row-major weights.
# batches of more than this many rows leave row-parallel epilogues to the consumer (model.DEFER_MIN_ROWS, same switch: below it
# the single-pass GEMM kernels beat split-K + consumer epilogue)
DEFER_MIN_ROWS = int(os.environ.get("SGL_MI355_DEFER_MIN_ROWS", "32"))
PRESHUFFLE_FP8_WEIGHTS = not os.environ.get("SGL_MI355_NO_WSHUFFLE")


class QuantizeMethodBase:
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError()

    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError()

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return


class LinearMethodBase(QuantizeMethodBase):
    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError()

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError()


class QuantizationConfig:
    def get_name(self) -> str:
        raise NotImplementedError()

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError()

    @classmethod
    def get_min_capability(cls) -> int:
        raise NotImplementedError()

    @staticmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError()

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError()

    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError()

    def get_scaled_act_names(self) -> List[str]:
        return []


# ----------------------------------------------------------------------------- unquantised
# The unquantised decode linears of a bf16 model run the 16-bit weight streamer on a fragment-major copy (round 3): unsplit
# for wide N, split-K slabs + finalize for narrow N.  Llama-3-8B, bf16, M = 1 / 16 / 64, us, streamer vs hipBLASLt
# (profiles/r03_linear16_shapes_splitk.txt): qkv 12.6 / 13.3 / 16.6-16.9 vs 14.3 / 14.4 / 17.9; o 10.6 / 11.6 / 14.8 vs 15.5 /
# 15.5 / 14.9-15.0; gate_up 39.0 / 40.0 / 41.9 vs 60.2 / 62.4 / 53.2; down 24.9 / 25.6 / 32.1 vs 26.5 / 27.4 / 41.4.
# SGL_MI355_LINEAR16_MIN_N raises the narrowest layer that gets the copy (A/B aid).
LINEAR16_MIN_N = int(os.environ.get("SGL_MI355_LINEAR16_MIN_N", "16"))
# rows up to which an AWQ layer runs its 64-row decode streamer in two passes instead of the tiled prefill kernel (0: never)
AWQ_TWO_PASS_MAX_ROWS = int(os.environ.get("SGL_MI355_AWQ_TWO_PASS_MAX_ROWS", "128"))
# rows up to which an unquantised linear runs the 16-bit streamer (65..128: its 128-row form)
LINEAR16_MAX_ROWS = min(128, int(os.environ.get("SGL_MI355_LINEAR16_MAX_ROWS", "128")))
# above that: the tiled 16-bit kernel on the same fragment-major copy (round 4, csrc/gemm_bf16.hip gemm16_tiled_kernel), or the
# library GEMM on the row-major weight (F.linear -> hipBLASLt, what the reference runs: unquant.py:111-123).
LINEAR16_TILED = os.environ.get("SGL_MI355_LINEAR16_TILED", "0") not in ("", "0")


class UnquantizedLinearMethod(LinearMethodBase):
    """layers/quantization/unquant.py: F.linear.  Batches of up to 128 rows run ops.linear16 on a fragment-major copy of the
    weight built once after loading (round 3).  Larger batches (prefill): the library GEMM (hipBLASLt through torch) by
    default, or with SGL_MI355_LINEAR16_TILED=1 the tiled 16-bit kernel on the same copy (round 4) -- within 0.86-1.30x of
    the library on the Llama shapes at 256..8192 rows (profiles/r04_linear16_tiled_vs_library.txt: ahead on one shape class,
    5-14 % behind on most), so the library GEMM stays the default."""

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        # (tracked: uses of `.data` are counted, which is how an in-place update of this weight is noticed -- parameter.py)
        layer.register_parameter("weight", tracked(_param_classes().ModelWeightParameter)(
            data=torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype), input_dim=1,
            output_dim=0, weight_loader=extra_weight_attrs.get("weight_loader")))

    def process_weights_after_loading(self, layer) -> None:
        old = getattr(layer, "_weight_fm", None)
        layer._weight_fm = None
        w = raw_data(layer.weight)
        if (w.is_cuda and w.dim() == 2 and w.dtype in (torch.bfloat16, torch.float16) and w.shape[0] >= LINEAR16_MIN_N
                and ops.linear16_shuffle_supported(w.shape[0], w.shape[1]) and not os.environ.get("SGL_MI355_NO_LINEAR16_SHUFFLE")):
            # a second copy (the row-major weight stays for batches above 128 rows and for reloads): +2 N K bytes per layer.
            # ops.TrackedCopy16 keeps it valid against in-place updates of the parameter: SGLang's
            # update_weights_from_tensor / _from_distributed / _model_load_weights_direct write `param.data.copy_` and do NOT
            # come back through this hook.  A re-run of this hook re-shuffles into the existing storage (graphs hold it).
            if old is not None and old.src is layer.weight:
                old.epoch = None  # force a re-shuffle into the existing storage
                if old.get() is not None:
                    layer._weight_fm = old
            if layer._weight_fm is None:
                layer._weight_fm = ops.TrackedCopy16(layer.weight)

    @staticmethod
    def weight_fm(layer):
        """The layer's valid fragment-major copy (ops.ShuffledWeight16) or None."""
        t = getattr(layer, "_weight_fm", None)
        if t is None:
            return None
        if t.src is not layer.weight:  # the parameter OBJECT was replaced: this copy belongs to nothing
            layer._weight_fm = None
            return None
        fm = t.get()
        if fm is None:
            layer._weight_fm = None
        return fm

    def apply(self, layer, x, bias=None):
        fm = self.weight_fm(layer)
        if fm is not None and x.is_cuda and x.dtype == fm.dtype:
            x2 = x.reshape(-1, x.shape[-1])
            rows = x2.shape[0]
            if (0 < rows <= LINEAR16_MAX_ROWS or (rows > 128 and LINEAR16_TILED)) and x2.stride(-1) == 1:
                # the hand-overs of deferred.py for an unquantised layer (same protocol as W8A8Fp8LinearMethod.apply): a
                # row-parallel or qkv layer whose consumer has asked leaves its split-K partial sums unfinished at decode sizes
                # (from one row on: narrow 16-bit layers run split-K + finalize at every decode size)
                may_defer = (deferred.DEFERRED_EPILOGUES and getattr(layer, "_sgl_mi355_may_defer", False) and x.dim() == 2
                             and rows <= 128)
                if may_defer and layer._sgl_mi355_partials_ok is not None:
                    may_defer = layer._sgl_mi355_partials_ok(rows, x.dtype)
                if may_defer and not deferred.hint_decode and (layer._sgl_mi355_is_qkv or not torch.cuda.is_current_stream_capturing()):
                    may_defer = False
                if may_defer and getattr(layer, "_sgl_mi355_defer_epilogue", False):
                    part = ops.linear16_partials(x2, fm, bias)
                    if part is not None:
                        return ops.defer_epilogue(part)
                out = ops.linear16(x2, fm, bias).reshape(x.shape[:-1] + (fm.N,))
                if may_defer:
                    out._sgl_mi355_epilogue_producer = layer
                    if out._base is not None:
                        out._base._sgl_mi355_epilogue_producer = layer
                return out
        return torch.nn.functional.linear(x, layer.weight, bias)


# ----------------------------------------------------------------------------- FP8 w8a8
def apply_fp8_linear(input: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp8_utils.py:653-704 (the cutlass branch): dynamic per-token activation quant, then
    fp8_scaled_mm(qinput, weight, x_scale, weight_scale, out_dtype=input.dtype, bias)."""
    if input_scale is not None:
        raise NotImplementedError("static activation scales (per-tensor) are not part of this path")
    input_2d = input.view(-1, input.shape[-1])
    output_shape = [*input.shape[:-1], ops.fp8_weight_kn(weight)[1]]
    qinput = torch.empty_like(input_2d, dtype=torch.float8_e4m3fn)
    x_scale = torch.empty((input_2d.shape[0], 1), dtype=torch.float32, device=input.device)
    ops.sgl_per_token_quant_fp8(input_2d.contiguous(), qinput, x_scale)
    output = ops.fp8_scaled_mm(qinput, weight, x_scale, weight_scale, out_dtype=input.dtype, bias=bias)
    return output.view(*output_shape)


def per_channel_quant_fp8_weight(weight: torch.Tensor):
    """Weight-side quantisation of an unquantised checkpoint, w8a8_fp8.py:119-125:
    per output channel, scale = rowmax/448, q = cast(w / scale).  Load-time, off the hot path."""
    w = weight.float()
    amax = w.abs().amax(dim=1, keepdim=True).clamp(min=1e-12)
    scale = amax / 448.0
    q = (w / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale


class W8A8Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False):
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized

    def get_name(self) -> str:
        return "w8a8_fp8"

    def get_supported_act_dtypes(self):
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 89

    @staticmethod
    def get_config_filenames():
        return []

    @classmethod
    def from_config(cls, config):
        quant_method = config.get("quant_method", "") if isinstance(config, dict) else ""
        return cls(is_checkpoint_fp8_serialized="compressed-tensors" in quant_method or "w8a8_fp8" in quant_method)

    def get_quant_method(self, layer, prefix: str):
        # w8a8_fp8.py:80-92: `isinstance(layer, LinearBase)` -- SGLang's LinearBase or the harness one
        if is_linear_layer(layer):
            return W8A8Fp8LinearMethod(self)
        return None


class W8A8Fp8LinearMethod(LinearMethodBase):
    def __init__(self, quantization_config: W8A8Fp8Config):
        self.quantization_config = quantization_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        # w8a8_fp8.py:136-173: weight [N_part, K] (fp8 if the checkpoint is serialised in fp8, else
        # params_dtype and quantised after loading), weight_scale [N_part, 1] fp32
        P = _param_classes()
        serialized = self.quantization_config.is_checkpoint_fp8_serialized
        weight_dtype = torch.float8_e4m3fn if serialized else params_dtype
        weight_loader = extra_weight_attrs.get("weight_loader")
        n = sum(output_partition_sizes)
        self.logical_widths = output_partition_sizes
        layer.register_parameter("weight", P.ModelWeightParameter(
            data=torch.empty(n, input_size_per_partition, dtype=weight_dtype), input_dim=1, output_dim=0,
            weight_loader=weight_loader))
        if serialized:
            layer.register_parameter("weight_scale", P.ChannelQuantScaleParameter(
                data=torch.empty((n, 1), dtype=torch.float32), output_dim=0, weight_loader=weight_loader))
        else:
            layer.weight_scale = None  # made by process_weights_after_loading from the 16-bit weight (:119-133)
        layer.input_scale = None

    def process_weights_after_loading(self, layer) -> None:
        # w8a8_fp8.py:113-133
        weight = layer.weight
        if ops.is_wshuffled(weight) or (weight.dtype == torch.float8_e4m3fn and weight.dim() == 2 and weight.stride(0) == 1
                                        and weight.shape[0] > 1 and getattr(layer, "_sgl_mi355_weight_is_kn", False)):
            # Re-entered on a layer this hook has already processed (the reference re-runs it after a reload,
            # model_loader/loader.py:456, model_runner.py:731).  The parameter is the fragment-major uint8 [N / 16, 16 K] tensor
            # (or the K-major [K, N] view) this hook stored: a real reload through the weight loaders fails on that shape before
            # it gets here, so the bytes are the ones already laid out -- nothing to do, and the STORAGE must not change
            # (captured HIP graphs hold its raw pointer).  In-place FP8 weight updates are unsupported.
            return
        if self.quantization_config.is_checkpoint_fp8_serialized:
            weight_scale = layer.weight_scale.detach()
        else:
            weight, weight_scale = per_channel_quant_fp8_weight(layer.weight)
        n, k = weight.shape
        if PRESHUFFLE_FP8_WEIGHTS and weight.is_cuda and weight.dtype == torch.float8_e4m3fn and ops.fp8_shuffle_supported(n, k):
            # MI355X repack (this hook is where the reference repacks too, e.g. aiter's shuffle_weight for its ROCm MoE
            # weights, fp8.py:780-783): fragment-major bytes, so that a decode wave's weight loads are 1 KiB contiguous
            # instead of 16 rows x 64 B.  The parameter becomes a uint8 tensor [N / 16, 16 K] (round 4): the layout is
            # carried by dtype and shape, survives .data / detach / deepcopy / state_dict, and anything that expects the
            # [N, K] or [K, N] matrix (a loader, another kernel) fails on the shape instead of reading shuffled bytes.
            layer.weight = torch.nn.Parameter(ops.fp8_shuffle_weight(weight.contiguous()), requires_grad=False)
            layer._sgl_mi355_weight_is_kn = False
        else:
            layer.weight = torch.nn.Parameter(weight.t(), requires_grad=False)  # K-major [K, N] view
            layer._sgl_mi355_weight_is_kn = True
        layer.weight_scale = torch.nn.Parameter(weight_scale.contiguous(), requires_grad=False)
        layer.input_scale = None

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None):
        # the producer of `x` (this backend's RMSNorm / SiluAndMul) may have quantised it already, in its own pass: the first
        # half of apply_fp8_linear is then done (ops.take_fp8_companion: only while x is untouched since; bit-identical)
        comp = ops.take_fp8_companion(x) if layer.input_scale is None else None
        # A row-parallel layer with no collective behind it (linear.RowParallelLinear marks itself) whose output goes straight
        # into this backend's RMSNorm: at decode sizes the GEMM is a split-K weight streamer, and once that norm has asked
        # (layers.RMSNorm.forward -> _sgl_mi355_defer_epilogue) the epilogue is left to it -- the output travels through the
        # model code as a DeferredEpilogue tensor (deferred.py; bit-identical, one launch less per GEMM).
        rows = x.shape[0] if x.dim() == 2 else 0
        # (33..128 rows: the decode streamers' split-K form; from 256 rows: the tiled kernel's raw split-K form for a narrow
        #  output with a long K -- down_proj, a TP-sharded qkv behind a hidden size of 8192 -- where ops.fp8_scaled_mm_partials has it)
        may_defer = (deferred.DEFERRED_EPILOGUES and getattr(layer, "_sgl_mi355_may_defer", False) and layer.input_scale is None
                     and x.is_cuda and (DEFER_MIN_ROWS < rows <= 128 or rows >= 256))
        if may_defer and layer._sgl_mi355_partials_ok is not None:
            may_defer = layer._sgl_mi355_partials_ok(x.shape[0], x.dtype)
        if may_defer and rows <= 128 and not deferred.hint_decode and (layer._sgl_mi355_is_qkv
                                                                       or not torch.cuda.is_current_stream_capturing()):
            # an extend pass: nobody there takes a qkv projection's partials, and an eager prefill of <= 128 rows is host-bound
            # -- the lazy objects would cost more Python time than the finalize launch they save (deferred.hint_decode)
            may_defer = False
        if may_defer and getattr(layer, "_sgl_mi355_defer_epilogue", False):
            if comp is None:
                x2 = x if x.is_contiguous() else x.contiguous()
                q = torch.empty_like(x2, dtype=torch.float8_e4m3fn)
                sc = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device)
                ops.sgl_per_token_quant_fp8(x2, q, sc)
                comp = (q, sc)
            part = ops.fp8_scaled_mm_partials(comp[0], layer.weight, comp[1], layer.weight_scale, x.dtype, bias)
            if part is not None:
                return ops.defer_epilogue(part)
        # A merged gate_up projection at prefill sizes whose output goes straight into this backend's SiluAndMul: once that has
        # asked (_sgl_mi355_fuse_silu), SiLU(gate) * up runs in the GEMM's epilogue and the [T, 2I] matrix is never written
        # (deferred.py: the lazy output carries the activation; the matrix is computed only if somebody else reads it)
        may_silu = (deferred.DEFERRED_EPILOGUES and getattr(layer, "_sgl_mi355_may_fuse_silu", False) and layer.input_scale is None
                    and x.dim() == 2 and x.is_cuda and x.shape[0] > 128 and ops.is_wshuffled(layer.weight))
        if may_silu and getattr(layer, "_sgl_mi355_fuse_silu", False):
            if comp is None:
                x2 = x if x.is_contiguous() else x.contiguous()
                q = torch.empty_like(x2, dtype=torch.float8_e4m3fn)
                sc = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device)
                ops.sgl_per_token_quant_fp8(x2, q, sc)
                comp = (q, sc)
            act = ops.fp8_scaled_mm_silu_mul(comp[0], layer.weight, comp[1], layer.weight_scale, x.dtype, bias)
            if act is not None:
                q_, s_, w_, ws_, dt_ = comp[0], comp[1], layer.weight, layer.weight_scale, x.dtype
                lazy = ops.DeferredEpilogue(compute=lambda: ops.fp8_scaled_mm(q_, w_, s_, ws_, out_dtype=dt_, bias=bias),
                                            like=((x.shape[0], 2 * act.shape[-1]), x.dtype, x.device))
                lazy.silu_act = act
                return lazy
        if comp is not None:
            out = ops.fp8_scaled_mm(comp[0], layer.weight, comp[1], layer.weight_scale, out_dtype=x.dtype, bias=bias)
            out = out.view(*x.shape[:-1], out.shape[-1])
        else:
            out = apply_fp8_linear(x, layer.weight, layer.weight_scale, input_scale=layer.input_scale, bias=bias)
        if may_silu:
            out._sgl_mi355_epilogue_producer = layer  # SiluAndMul.forward tells the layer (see above)
        if may_defer:
            # the consumer that receives this tensor (RMSNorm), or a view of it whose ._base it is (the attention backend behind
            # qkv.split), tells the layer that it could have taken the epilogue (see above)
            out._sgl_mi355_epilogue_producer = layer
            if out._base is not None:
                out._base._sgl_mi355_epilogue_producer = layer
        return out

    def apply_prequantized(self, layer, qinput: torch.Tensor, x_scale: torch.Tensor, out_dtype: torch.dtype,
                           bias: Optional[torch.Tensor] = None):
        """The second half of apply_fp8_linear (fp8_utils.py:696-704) for callers whose producer already emitted
        the per-token FP8 activation (fused norm+quant / silu+quant kernels of this backend)."""
        q2 = qinput.view(-1, qinput.shape[-1])
        out = ops.fp8_scaled_mm(q2, layer.weight, x_scale, layer.weight_scale, out_dtype=out_dtype, bias=bias)
        return out.view(*qinput.shape[:-1], out.shape[-1])


    def apply_prequantized_silu_mul(self, layer, qinput: torch.Tensor, x_scale: torch.Tensor, out_dtype: torch.dtype,
                                    bias: Optional[torch.Tensor] = None):
        """SiluAndMul(apply_prequantized(...)) in one launch for a merged gate_up layer at prefill sizes
        (ops.fp8_scaled_mm_silu_mul), or None when the shape has no such form (the caller then makes the two calls)."""
        q2 = qinput.view(-1, qinput.shape[-1])
        out = ops.fp8_scaled_mm_silu_mul(q2, layer.weight, x_scale, layer.weight_scale, out_dtype, bias)
        return None if out is None else out.view(*qinput.shape[:-1], out.shape[-1])

    def apply_prequantized_partials(self, layer, qinput: torch.Tensor, x_scale: torch.Tensor, out_dtype: torch.dtype,
                                    bias: Optional[torch.Tensor] = None):
        """apply_prequantized without the epilogue: ops.GemmPartials for a fused consumer, or None when the shape has
        no split-K decode path (the caller then uses apply_prequantized)."""
        q2 = qinput.view(-1, qinput.shape[-1])
        return ops.fp8_scaled_mm_partials(q2, layer.weight, x_scale, layer.weight_scale, out_dtype, bias)


    def apply_a16_partials(self, layer, x16: torch.Tensor, row_absmax: torch.Tensor, out_dtype: torch.dtype,
                           bias: Optional[torch.Tensor] = None):
        """apply_prequantized_partials with the per-token quant done inside the GEMM (the absmax comes from the producer
        of x16, e.g. the decode attention's epilogue); ops.GemmPartials or None."""
        return ops.fp8_scaled_mm_partials_a16(x16.view(-1, x16.shape[-1]), row_absmax, layer.weight, layer.weight_scale,
                                              out_dtype, bias)


# ----------------------------------------------------------------------------- AWQ INT4
class AWQConfig(QuantizationConfig):
    """awq.py:75-150."""

    def __init__(self, weight_bits: int = 4, group_size: int = 128, zero_point: bool = True):
        if weight_bits != 4:
            raise ValueError(f"Currently, only 4-bit weight quantization is supported for AWQ, but got {weight_bits} bits.")
        self.weight_bits, self.group_size, self.zero_point = weight_bits, group_size, zero_point
        self.pack_factor = 32 // weight_bits

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self):
        # the reference is fp16-only (awq.py:111-112); the kernels here also take bf16
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @staticmethod
    def get_config_filenames():
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config):
        def get(keys, default=None):
            for k in keys:
                if k in config:
                    return config[k]
            if default is None:
                raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")
            return default
        return cls(get(["w_bit", "bits"]), get(["q_group_size", "group_size"]), get(["zero_point"], True))

    def get_quant_method(self, layer, prefix: str):
        if is_linear_layer(layer):  # awq.py:127-136
            return AWQLinearMethod(self)
        return None


class AWQLinearMethod(LinearMethodBase):
    """awq.py:319-418."""

    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        g = self.quant_config.group_size
        if g == -1:
            g = input_size_per_partition
        if input_size_per_partition % g != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        n = sum(output_partition_sizes)
        if n % self.quant_config.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        pf = self.quant_config.pack_factor
        P = _param_classes()
        weight_loader = extra_weight_attrs.get("weight_loader")
        # awq.py:354-394: K is the slow dimension (input_dim 0), eight output columns per int32 (packed_dim 1)
        layer.register_parameter("qweight", P.PackedvLLMParameter(
            data=torch.empty(input_size_per_partition, n // pf, dtype=torch.int32), input_dim=0, output_dim=1,
            packed_dim=1, packed_factor=pf, weight_loader=weight_loader))
        layer.register_parameter("qzeros", P.PackedvLLMParameter(
            data=torch.empty(input_size_per_partition // g, n // pf, dtype=torch.int32), input_dim=0, output_dim=1,
            packed_dim=1, packed_factor=pf, weight_loader=weight_loader))
        layer.register_parameter("scales", P.GroupQuantScaleParameter(
            data=torch.empty(input_size_per_partition // g, n, dtype=params_dtype), input_dim=0, output_dim=1,
            weight_loader=weight_loader))

    def process_weights_after_loading(self, layer) -> None:
        layer.qweight = torch.nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = torch.nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = torch.nn.Parameter(layer.scales.data, requires_grad=False)
        # Decode-time copy in the k-packed layout (csrc/awq_packed.hip): the reference's own hook for "transpose /
        # repack freely" (base_config.py process_weights_after_loading).  Costs K*N/2 + K*N/32 extra bytes per
        # layer; SGL_MI355_AWQ_NO_REPACK=1 keeps only the checkpoint layout.
        layer.awq_packed = None
        K, N = layer.qweight.shape[0], layer.qweight.shape[1] * self.quant_config.pack_factor
        G = K // layer.scales.shape[0]
        if layer.qweight.is_cuda and not os.environ.get("SGL_MI355_AWQ_NO_REPACK") and \
                ops.awq_packable(K, N, G, layer.scales.dtype):
            layer.awq_packed = ops.awq_repack(layer.qweight.data, layer.scales.data, layer.qzeros.data) + (G,)
            layer.awq_out_features = N
            # The k-packed copy serves every fp16 call (decode streamer and prefill tiles), so the checkpoint-layout tensors
            # are dead weight in HBM afterwards (K*N/2 + K*N/32 bytes per layer, 3.6 GB on Llama-2-7B).  They are released
            # only on request -- SGL_MI355_AWQ_RELEASE_CHECKPOINT_LAYOUT=1 or AWQConfig.release_checkpoint_layout -- because a
            # live weight reload (update_weights_from_disk: the loaders copy into these parameters, then this hook runs
            # again) needs them.
            if (os.environ.get("SGL_MI355_AWQ_RELEASE_CHECKPOINT_LAYOUT") or
                    getattr(self.quant_config, "release_checkpoint_layout", False)) and layer.scales.dtype == torch.float16:
                for name in ("qweight", "qzeros", "scales"):
                    getattr(layer, name).data = torch.empty(0, dtype=getattr(layer, name).dtype, device=layer.qweight.device)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None):
        # awq.py:401-418 computes awq_dequantize(...) then x @ W; here the dequant is fused into the GEMM
        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        pack_factor = self.quant_config.pack_factor
        packed = getattr(layer, "awq_packed", None)
        n_out = layer.awq_out_features if packed is not None else qweight.shape[-1] * pack_factor
        out_shape = x.shape[:-1] + (n_out,)
        reshaped_x = x.reshape(-1, x.shape[-1])
        if packed is not None and qweight.numel() == 0 and reshaped_x.dtype != torch.float16:
            raise RuntimeError("AWQLinearMethod.apply: the checkpoint-layout tensors were released "
                               "(SGL_MI355_AWQ_RELEASE_CHECKPOINT_LAYOUT); only fp16 activations are served")
        if packed is not None and reshaped_x.dtype == torch.float16:
            rows = reshaped_x.shape[0]
            if rows <= 64:  # decode: weight-streaming kernel on the k-packed copy
                # the hand-overs of deferred.py (same protocol as W8A8Fp8LinearMethod.apply): a row-parallel or qkv layer whose
                # consumer has asked leaves its split-K partial sums unfinished
                may_defer = (deferred.DEFERRED_EPILOGUES and getattr(layer, "_sgl_mi355_may_defer", False) and x.dim() == 2
                             and reshaped_x.stride(-1) == 1)
                if may_defer and layer._sgl_mi355_partials_ok is not None:
                    may_defer = layer._sgl_mi355_partials_ok(rows, x.dtype)
                if may_defer and not deferred.hint_decode and (layer._sgl_mi355_is_qkv or not torch.cuda.is_current_stream_capturing()):
                    may_defer = False
                if may_defer and getattr(layer, "_sgl_mi355_defer_epilogue", False):
                    part = ops.awq_gemm_packed_partials(reshaped_x, packed[0], packed[1], packed[2], bias)
                    if part is not None:
                        return ops.defer_epilogue(part, pool=ops._awq_workspace)
                out = ops.awq_gemm_packed(reshaped_x, packed[0], packed[1], packed[2], bias)
                if may_defer:  # (tagged for the consumer that will ask: on the tensor AND on what its views call their base)
                    res = out.reshape(out_shape)
                    res._sgl_mi355_epilogue_producer = out._sgl_mi355_epilogue_producer = layer
                    return res
            elif rows <= AWQ_TWO_PASS_MAX_ROWS:
                # 65..128 rows: two passes of the 64-row streamer (rows are independent: the same bits as one call would give)
                # -- the tiled kernel below puts a narrow layer on N / 128 CUs at this size (Llama-2-7B, bs = 128: 29.6 -> see
                # profiles/r03_sweep.txt)
                out = torch.cat([ops.awq_gemm_packed(reshaped_x[:64], packed[0], packed[1], packed[2], bias),
                                 ops.awq_gemm_packed(reshaped_x[64:], packed[0], packed[1], packed[2], bias)], dim=0)
            else:  # prefill: 128 x 128 tiles on the fp16 MFMA, INT4 unpacked in registers -- no fp16 weight copy anywhere
                out = ops.awq_gemm_packed_tiled(reshaped_x, packed[0], packed[1], packed[2], bias)
        else:
            out = ops.awq_gemm(reshaped_x, qweight, scales, qzeros, bias)
        return out.reshape(out_shape)


QUANTIZATION_METHODS = {
    "w8a8_fp8": W8A8Fp8Config,
    "awq": AWQConfig,
}


def get_quantization_config(quantization: str):
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}. Available methods: {list(QUANTIZATION_METHODS)}")
    return QUANTIZATION_METHODS[quantization]
