"""A row-parallel FP8 GEMM whose epilogue has not run yet, handed through UNTOUCHED model code as a tensor.

In the reference's call order (models/llama.py:  `hidden_states = self.mlp(hidden_states)` ... next layer:
`hidden_states, residual = self.input_layernorm(hidden_states, residual)`, and the same between `o_proj` and
`post_attention_layernorm`) the output of `o_proj` / `down_proj` goes straight into this backend's RMSNorm, which
overwrites it (layernorm.py:82-85, fused_add_rmsnorm is in place).  At decode sizes the GEMM is a split-K weight streamer
whose partial sums are finished by a separate `finalize` launch (~5 us) that the norm kernel can do on its way in
(sgl_mi355_fused_add_rmsnorm_from_partials, bit-identical) -- this repo's own fused entry points have done so since round 2.

`DeferredEpilogue` lets the drop-in classes do the same without touching the model: `W8A8Fp8LinearMethod.apply` returns one
(once the RMSNorm that consumed this layer's previous output has asked for it) and `RMSNorm.forward` consumes it.  It is a
wrapper tensor subclass: shape / dtype / device are real, there is no storage, and ANY torch operation on it from anybody else
first runs the plain finalize launch and proceeds on the real tensor (`__torch_dispatch__`) -- so it is never observable as
anything but the GEMM's output; `data_ptr()` raises instead of returning garbage.  The partial sums live in the stream's
split-K workspace: the workspace pool finishes a still-pending tensor BEFORE it hands the buffer to the next GEMM
(ops._ScratchPool.get), under graph capture too (the finalize launch is captured where it happens).
SGL_MI355_NO_DEFERRED_EPILOGUE=1 switches the mechanism off (the linear finishes its own output as before)."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.utils._pytree import tree_map

DEFERRED_EPILOGUES = not os.environ.get("SGL_MI355_NO_DEFERRED_EPILOGUE")


class DeferredEpilogue(torch.Tensor):
    __torch_function__ = torch._C._disabled_torch_function_impl  # only __torch_dispatch__ below sees operations

    @staticmethod
    def __new__(cls, part, on_resolve=None):
        return torch.Tensor._make_wrapper_subclass(cls, (part.M, part.N), dtype=part.out_dtype, device=part.ws.device,
                                                   requires_grad=False)

    def __init__(self, part, on_resolve=None):
        self._part = part        # ops.GemmPartials (anything with M, N, out_dtype, ws and finalize())
        self._value: Optional[torch.Tensor] = None
        self._on_resolve = on_resolve

    # ---- the consumer's side (RMSNorm.forward)
    def pending_partials(self):
        """The partial sums if nobody has finished them yet, else None."""
        return self._part if self._value is None else None

    def resolve(self, value: torch.Tensor) -> None:
        """The consumer finished the GEMM inside its own kernel; `value` is what this tensor now holds under the reference's
        in-place semantics (fused_add_rmsnorm overwrites its input with the normed row)."""
        self._value, self._part = value, None
        if self._on_resolve is not None:
            self._on_resolve(self)
            self._on_resolve = None

    # ---- everybody else
    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self.resolve(self._part.finalize())
        return self._value

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(x):
            return x.materialize() if isinstance(x, cls) else x
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))

    def __repr__(self):  # (the default repr would dispatch and materialise)
        state = "pending" if self._value is None else "resolved"
        return f"DeferredEpilogue({tuple(self.shape)}, {self.dtype}, {state})"


def materialize(x):
    """x itself, or the real tensor behind a DeferredEpilogue (for this package's own ops, which take raw pointers)."""
    return x.materialize() if isinstance(x, DeferredEpilogue) else x
