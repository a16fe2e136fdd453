"""Loader-aware parameters for the quant-linear methods.

SGLang's checkpoint loaders never copy into a layer's tensors themselves: they call
``param.weight_loader(param, loaded_weight[, shard_id])`` (models/llama.py:622), and the layer's
``weight_loader`` / ``weight_loader_v2`` (layers/linear.py:333-413, 484-726, 877-1120, 1212-1283) slices
the checkpoint tensor by what the parameter says about itself: ``output_dim`` / ``input_dim``, for
packed INT4 ``packed_dim`` / ``packed_factor``, and -- in the v2 loaders -- the parameter's own
``load_{column_parallel,row_parallel,merged_column,qkv}_weight`` methods, picked with ``isinstance``
against the classes of ``python/sglang/srt/layers/parameter.py:29-458``.

So inside SGLang the quant methods of this backend must create *SGLang's* parameter classes
(``classes()`` returns them whenever ``sglang.srt.layers.parameter`` is importable).  Where SGLang is
absent (the GPU box, this repo's own harness layers) the same-named local classes below carry the same
metadata and the same four ``load_*`` methods, so ``sglang_npu_amd.linear`` drives them identically.
"""
from __future__ import annotations

import importlib
from types import SimpleNamespace
from typing import Callable, Optional

import torch
from torch.nn import Parameter


class BasevLLMParameter(Parameter):
    """A parameter that remembers the layer's weight loader (parameter.py:29-70)."""

    def __new__(cls, data: torch.Tensor, **kwargs):
        return super().__new__(cls, data=data, requires_grad=False)

    def __init__(self, data: torch.Tensor, weight_loader: Optional[Callable] = None):
        self._weight_loader = weight_loader

    @property
    def weight_loader(self):
        return self._weight_loader

    def _copy_exact(self, dst: torch.Tensor, src: torch.Tensor):
        if dst.shape != src.shape:
            raise AssertionError(f"weight loading: parameter slice {tuple(dst.shape)} vs checkpoint {tuple(src.shape)}")
        dst.copy_(src)

    # unsharded defaults: the whole tensor is this rank's
    def load_column_parallel_weight(self, loaded_weight, **kw):
        self._copy_exact(self.data, loaded_weight)

    def load_row_parallel_weight(self, loaded_weight, **kw):
        self._copy_exact(self.data, loaded_weight)

    def load_merged_column_weight(self, loaded_weight, **kw):
        self._copy_exact(self.data, loaded_weight)

    def load_qkv_weight(self, loaded_weight, **kw):
        self._copy_exact(self.data, loaded_weight)


class _ColumnvLLMParameter(BasevLLMParameter):
    """Sharded along ``output_dim`` (parameter.py:73-214)."""

    def __init__(self, output_dim: int, **kwargs):
        self._output_dim = output_dim
        super().__init__(**kwargs)

    @property
    def output_dim(self):
        return self._output_dim

    def _packed_on_output(self):
        return getattr(self, "packed_dim", None) == self.output_dim

    def load_column_parallel_weight(self, loaded_weight, tp_rank: int = 0, use_presharded_weights: bool = False):
        if not use_presharded_weights:
            n = self.data.shape[self.output_dim]
            loaded_weight = loaded_weight.narrow(self.output_dim, tp_rank * n, n)
        self._copy_exact(self.data, loaded_weight)

    def load_merged_column_weight(self, loaded_weight, shard_offset=None, shard_size=None, tp_rank: int = 0,
                                  use_presharded_weights: bool = False, **kw):
        if self._packed_on_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size=shard_size,
                                                                             shard_offset=shard_offset)
        dst = self.data.narrow(self.output_dim, shard_offset, shard_size)
        if not use_presharded_weights:
            loaded_weight = loaded_weight.narrow(self.output_dim, tp_rank * shard_size, shard_size)
        self._copy_exact(dst, loaded_weight)

    def load_qkv_weight(self, loaded_weight, tp_rank: int = 0, use_presharded_weights: bool = False, shard_offset=None,
                        shard_size=None, shard_id=None, num_heads: int = 1, **kw):
        if self._packed_on_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size=shard_size,
                                                                             shard_offset=shard_offset)
        # q is split over all ranks; a k/v head is shared by `num_heads` (= kv replicas) consecutive ranks
        src_block = tp_rank if shard_id == "q" else tp_rank // num_heads
        dst = self.data.narrow(self.output_dim, shard_offset, shard_size)
        if not use_presharded_weights:
            loaded_weight = loaded_weight.narrow(self.output_dim, src_block * shard_size, shard_size)
        self._copy_exact(dst, loaded_weight)


class RowvLLMParameter(BasevLLMParameter):
    """Sharded along ``input_dim`` (parameter.py:217-269)."""

    def __init__(self, input_dim: int, **kwargs):
        self._input_dim = input_dim
        super().__init__(**kwargs)

    @property
    def input_dim(self):
        return self._input_dim

    def load_row_parallel_weight(self, loaded_weight, tp_rank: int = 0, use_presharded_weights: bool = False):
        if not use_presharded_weights:
            n = self.data.shape[self.input_dim]
            loaded_weight = loaded_weight.narrow(self.input_dim, tp_rank * n, n)
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        self._copy_exact(self.data, loaded_weight)


class ModelWeightParameter(_ColumnvLLMParameter, RowvLLMParameter):
    """A linear weight: column- and row-shardable (parameter.py:272-278)."""


class GroupQuantScaleParameter(_ColumnvLLMParameter, RowvLLMParameter):
    """Scales of group-quantised weights (parameter.py:281-287)."""


class ChannelQuantScaleParameter(_ColumnvLLMParameter):
    """Per-output-channel scales (parameter.py:290-296)."""


class PackedvLLMParameter(ModelWeightParameter):
    """Weights packed several to a word along ``packed_dim`` (parameter.py:416-458): shard offsets and
    sizes given in logical columns are divided by ``packed_factor`` before slicing."""

    def __init__(self, packed_factor, packed_dim: int, marlin_tile_size: Optional[int] = None, **kwargs):
        self._packed_factor, self._packed_dim, self._marlin_tile_size = packed_factor, packed_dim, marlin_tile_size
        super().__init__(**kwargs)

    @property
    def packed_dim(self):
        return self._packed_dim

    @property
    def packed_factor(self):
        return self._packed_factor

    @property
    def pack_factor(self):  # the name the legacy loaders read (linear.py:545-546)
        return self._packed_factor

    @property
    def marlin_tile_size(self):
        return self._marlin_tile_size

    def adjust_shard_indexes_for_packing(self, shard_size, shard_offset):
        size, off = shard_size // self._packed_factor, shard_offset // self._packed_factor
        if self._marlin_tile_size is not None:
            size, off = size * self._marlin_tile_size, off * self._marlin_tile_size
        return size, off


# ---------------------------------------------------------------------------- write tracking
# Layers that keep a re-laid COPY of a 16-bit weight next to the row-major parameter (UnquantizedLinearMethod.weight_fm, the
# LM head) must notice every way the parameter's bytes can change after the copy was made.  SGLang updates weights in place
# without re-running process_weights_after_loading (model_runner.py:831-900 update_weights_from_distributed / _from_tensor,
# :1777 _model_load_weights_direct -> default_weight_loader -> `param.data.copy_`), so neither the storage pointer nor the
# loader hook is enough.  What every such path has in common is that it goes through `param.data` (an alias whose writes the
# parameter's own version counter does not see) or writes the parameter itself (`param.copy_`, which bumps `_version`).
# tracked(cls) makes a subclass of a parameter class whose `data` attribute counts every read and assignment of the alias:
# write_epoch() = (those, _version, storage pointer) changes whenever the bytes MAY have changed -- conservative (a pure read
# through `.data` also counts; the copy is then rebuilt once, which is only a weight-sized pass).
_TENSOR_DATA = torch.Tensor.data
_EPOCH = "_sgl_mi355_data_uses"
_tracked_cache = {}


def raw_data(p: torch.Tensor) -> torch.Tensor:
    """`p.data` without counting as a possible write (this backend's own reads)."""
    return _TENSOR_DATA.__get__(p)


def tracked(cls):
    """A subclass of the parameter class ``cls`` (same name, isinstance-compatible) whose ``.data`` uses are counted."""
    sub = _tracked_cache.get(cls)
    if sub is None:
        def _get(self):
            self.__dict__[_EPOCH] = self.__dict__.get(_EPOCH, 0) + 1
            return _TENSOR_DATA.__get__(self)

        def _set(self, value):
            self.__dict__[_EPOCH] = self.__dict__.get(_EPOCH, 0) + 1
            _TENSOR_DATA.__set__(self, value)

        sub = type(cls.__name__, (cls,), {"data": property(_get, _set), "__module__": cls.__module__,
                                          "_sgl_mi355_tracked": True})
        _tracked_cache[cls] = sub
    return sub


def write_epoch(p: torch.Tensor):
    """Changes whenever the bytes of ``p`` may have changed (see above).  For a parameter that is not tracked() only
    in-place writes through the tensor itself and storage replacement are seen."""
    return (p.__dict__.get(_EPOCH, 0) if hasattr(p, "__dict__") else 0, p._version, _TENSOR_DATA.__get__(p).data_ptr())


_LOCAL = SimpleNamespace(
    BasevLLMParameter=BasevLLMParameter, _ColumnvLLMParameter=_ColumnvLLMParameter, RowvLLMParameter=RowvLLMParameter,
    ModelWeightParameter=ModelWeightParameter, GroupQuantScaleParameter=GroupQuantScaleParameter,
    ChannelQuantScaleParameter=ChannelQuantScaleParameter, PackedvLLMParameter=PackedvLLMParameter, source="local")
_cached = None


def classes(refresh: bool = False):
    """The parameter classes ``create_weights`` must instantiate: SGLang's own when SGLang is importable
    (its loaders dispatch with ``isinstance``, linear.py:390-400, 681-695), else the local ones."""
    global _cached
    if _cached is not None and not refresh:
        return _cached
    try:
        m = importlib.import_module("sglang.srt.layers.parameter")
        _cached = SimpleNamespace(
            BasevLLMParameter=m.BasevLLMParameter, _ColumnvLLMParameter=m._ColumnvLLMParameter,
            RowvLLMParameter=m.RowvLLMParameter, ModelWeightParameter=m.ModelWeightParameter,
            GroupQuantScaleParameter=m.GroupQuantScaleParameter, ChannelQuantScaleParameter=m.ChannelQuantScaleParameter,
            PackedvLLMParameter=m.PackedvLLMParameter, source="sglang")
    except Exception:  # SGLang absent or not importable in this process: same-named local classes
        _cached = _LOCAL
    return _cached


def is_linear_layer(layer) -> bool:
    """``isinstance(layer, LinearBase)`` of the reference registry gate (w8a8_fp8.py:80-92, awq.py:127-136), for
    SGLang's LinearBase and this repo's harness one alike -- by class name along the MRO, so that neither package
    has to import the other."""
    return any(c.__name__ == "LinearBase" for c in type(layer).__mro__)
