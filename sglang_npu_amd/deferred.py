"""Lazy tensors: work that one drop-in class leaves for the next, handed through UNTOUCHED model code as a tensor.

SGLang's model files call the drop-in classes one after another (models/llama.py:94-98, 186-191, 245-268):
`qkv_proj -> split -> rotary_emb -> attn -> o_proj -> post_attention_layernorm -> gate_up_proj -> act_fn -> down_proj -> next
input_layernorm`.  This repo's own fused call order saves launches by letting a consumer finish its producer's work (the norm
kernel runs the GEMM epilogue of `o_proj` / `down_proj` on its way in, the attention backend's RoPE + KV-write launch that of
`qkv_proj`, ...).  The classes here let the REFERENCE call order do the same without touching the model: the producer returns a
wrapper tensor subclass -- real shape / dtype / device, no storage -- and the consumer, one of this backend's own classes,
recognises it and takes the unfinished work.  ANY other torch operation on such a tensor first runs the plain sequence (finalize,
all-reduce, RoPE ...) and proceeds on the real tensor (`__torch_dispatch__`); this package's raw-pointer ops do the same when they
take its pointer (ops._ptr), and so does anybody's `data_ptr()` / `tolist()` / `numpy()` (answered by the finished tensor).  So a lazy tensor is never observable
as anything but the value the reference would hold at that point, bit for bit.  A producer only goes lazy after its consumer has
ASKED (the consumer tags the producer on the first plain pass), i.e. from the second eager pass and in every graph captured after
the usual warm-up.

* `DeferredEpilogue(part)`: a row-parallel FP8 GEMM (`o_proj`, `down_proj`; 33..128 rows) still in split-K partial sums;
  consumer `RMSNorm.forward(x, residual)` -> sgl_mi355_fused_add_rmsnorm_from_partials (finalize + fused_add_rmsnorm [+ the FP8
  companion], one launch).  The partial sums live in the stream's split-K workspace: the workspace pool finishes a pending tensor
  BEFORE it hands the buffer to the next GEMM (ops._ScratchPool.get), under graph capture too.
* the same with `needs_allreduce` under tensor parallelism: `RowParallelLinear.forward(x)` called without flags, as llama.py does,
  owns the all-reduce (linear.py:1302-1303); the lazy tensor carries this rank's addend (finished sum or partials) and the norm
  runs all-reduce + add + norm as ONE kernel on the P2P communicator (RMSNorm.forward_with_allreduce_fusion -- the role upstream
  reserves for `can_fuse_mlp_allreduce`, which llama.py does not pass); anybody else gets the plain all-reduce first.
* the qkv projection: `qkv.split(...)` on the lazy tensor gives lazy column ranges (`DeferredCols`), `RotaryEmbedding.forward`
  RECORDS the rotation, RadixAttention's `.view(-1, heads, head_dim)` stays lazy, and `MI355AttnBackend.forward_decode` finishes
  GEMM + RoPE + KV-pool write in one launch (rope_set_kv_from_partials).  Outside the split-K window the projection is finished
  by its own GEMM but still travels behind a handle (`local=`), so that RoPE + KV write are one launch
  (apply_rope_and_set_kv_buffer).  Anybody else who reads q / k / v first gets finalize + the recorded RoPE, the reference's own
  sequence; reading the qkv handle AFTER the backend consumed the partials raises (there is no tensor any more).
* the prefill gate_up projection (`compute=`): once SiluAndMul has asked, the GEMM runs with SiLU(gate) * up in its epilogue, the
  lazy tensor carries that activation (`silu_act`), and anybody else who reads the [T, 2I] matrix gets it computed then (a
  closure over the same FP8 operands launches the plain GEMM: rare, correct, one GEMM slower).
SGL_MI355_NO_DEFERRED_EPILOGUE=1 switches all of it off (every class finishes its own output as before)."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.utils._pytree import tree_map

DEFERRED_EPILOGUES = not os.environ.get("SGL_MI355_NO_DEFERRED_EPILOGUE")
# What the forward pass that is about to run is, as far as this backend knows: MI355AttnBackend.init_forward_metadata sets it
# once per batch (model_runner.py:1553-1573 calls that before the model).  A HINT for speed only -- every lazy tensor finishes
# correctly whoever consumes it -- used where a hand-over can only pay in one mode: the qkv projection left as split-K partials
# is finished by the attention backend's RoPE + KV-write launch in DECODE; an extend pass would finish it the plain way and
# only pay the host time of the lazy objects (eager 128-token prefill: 5.7 -> 7.0 ms before this hint).
hint_decode = True


_aten = torch.ops.aten


def _dispatch(func, args, kwargs):
    """Every operation on a lazy tensor of this module lands here.  Two are answered lazily, because untouched model code does
    them between the qkv projection and the attention backend (models/llama.py:186-189, radix_attention.py:94-101):
    `qkv.split([q, kv, kv], dim=-1)` on a pending root and `.view(-1, heads, head_dim)` on one of its column ranges.  Everything
    else runs on the finished tensors."""
    kwargs = kwargs or {}
    if func is _aten.split_with_sizes.default and args and isinstance(args[0], DeferredEpilogue) and args[0].is_pending():
        root, sizes = args[0], [int(x) for x in args[1]]
        dim = args[2] if len(args) > 2 else kwargs.get("dim", 0)
        if root.dim() == 2 and dim in (-1, 1) and sum(sizes) == root.shape[1] and all(x > 0 for x in sizes):
            out, off = [], 0
            for n in sizes:
                out.append(DeferredCols(root, off, off + n, (root.shape[0], n)))
                off += n
            return out
    if func in (_aten.view.default, _aten._unsafe_view.default, _aten.reshape.default) and args \
            and isinstance(args[0], DeferredCols) and args[0]._root.is_pending():
        t, shape = args[0], [int(x) for x in args[1]]
        m, n = t._root.shape[0], t._c1 - t._c0
        if shape.count(-1) == 1:
            known = 1
            for x in shape:
                known *= x if x != -1 else 1
            if known > 0 and (m * n) % known == 0:
                shape[shape.index(-1)] = m * n // known
        if tuple(shape) == (m, n) or (len(shape) == 3 and shape[0] == m and shape[1] * shape[2] == n and min(shape) > 0):
            return DeferredCols(t._root, t._c0, t._c1, tuple(shape))

    def unwrap(x):
        return x.materialize() if isinstance(x, (DeferredEpilogue, DeferredCols)) else x
    return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs))


class DeferredEpilogue(torch.Tensor):
    __torch_function__ = torch._C._disabled_torch_function_impl  # only __torch_dispatch__ below sees operations

    @staticmethod
    def __new__(cls, part=None, on_resolve=None, local=None, needs_allreduce=False, compute=None, like=None):
        if part is not None:
            shape, dtype, device = (part.M, part.N), part.out_dtype, part.ws.device
        elif local is not None:
            shape, dtype, device = tuple(local.shape), local.dtype, local.device
        else:
            shape, dtype, device = like
        return torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, device=device, requires_grad=False)

    def __init__(self, part=None, on_resolve=None, local=None, needs_allreduce=False, compute=None, like=None):
        self._part = part        # ops.GemmPartials (anything with M, N, out_dtype, ws and finalize()) ...
        self._local = local      # ... or this rank's finished GEMM output whose tensor-parallel all-reduce has not run
        self._compute = compute  # ... or a closure that launches the GEMM (prefill gate_up whose SiLU * up went into the GEMM's
        self.silu_act = None     #     epilogue instead: `silu_act` holds that result for SiluAndMul, the matrix itself was never written)
        self.needs_allreduce = needs_allreduce  # RowParallelLinear under TP: the collective belongs to whoever finishes this
        self._value: Optional[torch.Tensor] = None
        self._on_resolve = on_resolve
        self._rope = None        # (positions, RotaryEmbedding, q columns, k columns): RoPE recorded, not applied yet (qkv form)
        self._consumed = False   # the attention backend took the partials (RoPE + KV write); there is no tensor any more

    # ---- the consumer's side (RMSNorm.forward, MI355AttnBackend.forward_decode)
    def is_pending(self) -> bool:
        return self._value is None and not self._consumed

    def pending_partials(self):
        """The partial sums if nobody has finished them yet, else None."""
        return self._part if self.is_pending() else None

    def pending_local(self):
        """This rank's unreduced GEMM output if nobody has all-reduced it yet, else None."""
        return self._local if self.is_pending() else None

    def consume(self) -> None:
        """The consumer used the partial sums for something that leaves no tensor behind (the qkv projection whose q went
        into the attention launch rotated, k / v into the KV pool).  Reading the handle afterwards is an error, not garbage."""
        self._consumed, self._part = True, None
        if self._on_resolve is not None:
            self._on_resolve(self)
            self._on_resolve = None

    def resolve(self, value: torch.Tensor) -> None:
        """The consumer finished the GEMM inside its own kernel; `value` is what this tensor now holds under the reference's
        in-place semantics (fused_add_rmsnorm overwrites its input with the normed row)."""
        self._value, self._part, self._local, self._compute = value, None, None, None
        if self._on_resolve is not None:
            self._on_resolve(self)
            self._on_resolve = None

    # ---- everybody else
    def materialize(self) -> torch.Tensor:
        if self._consumed:
            raise RuntimeError("this qkv projection's output was consumed by the fused RoPE + KV-write of the attention backend "
                               "(deferred.py) and is read again afterwards: set SGL_MI355_NO_DEFERRED_EPILOGUE=1 for this model")
        if self._value is None:
            value = (self._part.finalize() if self._part is not None else self._local if self._local is not None
                     else self._compute())
            rope = self._rope
            if self.needs_allreduce:  # linear.py:1302-1303, the collective untouched model code expects from the linear itself
                from .distributed import tensor_model_parallel_all_reduce
                value = tensor_model_parallel_all_reduce(value)
            self.resolve(value)
            if rope is not None:  # rotary_emb(positions, q, k) was called on the column ranges: in place, as the reference does
                positions, rot, (q0, q1), (k0, k1) = rope
                rot.forward(positions, value[:, q0:q1], value[:, k0:k1])
        return self._value

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        return _dispatch(func, args, kwargs)

    def split(self, split_size=None, dim=0, split_size_or_sections=None):
        """`qkv.split([q, kv, kv], dim=-1)` without the trip through the dispatcher (host time: 20 -> 8 us; same answer as
        _dispatch gives for aten.split_with_sizes)."""
        if split_size is None:
            split_size = split_size_or_sections
        if (isinstance(split_size, (list, tuple)) and dim in (-1, 1) and self._value is None and not self._consumed
                and self.dim() == 2 and sum(split_size) == self.shape[1] and min(split_size) > 0):
            m, out, off = self.shape[0], [], 0
            for n in split_size:
                out.append(DeferredCols(self, off, off + n, (m, n)))
                off += n
            return tuple(out)
        return torch.Tensor.split(self.materialize(), split_size, dim)

    def __repr__(self):  # (the default repr would dispatch and materialise)
        state = "consumed" if self._consumed else "pending" if self._value is None else "resolved"
        return f"DeferredEpilogue({tuple(self.shape)}, {self.dtype}, {state})"


class DeferredCols(torch.Tensor):
    """Columns [c0, c1) of a pending DeferredEpilogue -- what `qkv.split(...)` hands the model -- optionally viewed as
    [tokens, heads, head_dim] (RadixAttention.forward).  Strides are those of the real view (row stride = the GEMM's N)."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, root, c0, c1, shape):
        n_total = root.shape[1]
        strides = (n_total, 1) if len(shape) == 2 else (n_total, shape[2], 1)
        return torch.Tensor._make_wrapper_subclass(cls, shape, strides=strides, dtype=root.dtype, device=root.device,
                                                   requires_grad=False)

    def __init__(self, root, c0, c1, shape):
        self._root, self._c0, self._c1 = root, c0, c1

    def materialize(self) -> torch.Tensor:
        v = self._root.materialize()[:, self._c0:self._c1]
        return v if tuple(v.shape) == tuple(self.shape) else v.view(self.shape)

    def view(self, *shape):
        """`k.view(-1, heads, head_dim)` (RadixAttention.forward) without the trip through the dispatcher."""
        if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)):
            shape = tuple(shape[0])
        root = self._root
        if len(shape) == 3 and root._value is None and not root._consumed:
            m, n = root.shape[0], self._c1 - self._c0
            a, b, c = shape
            if a == -1 and b > 0 and c > 0:
                a = m * n // (b * c)
            if a == m and b * c == n and b > 0 and c > 0:
                return DeferredCols(root, self._c0, self._c1, (a, b, c))
        return self.materialize().view(*shape)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        return _dispatch(func, args, kwargs)

    def __repr__(self):
        return f"DeferredCols([{self._c0}:{self._c1}] of {self._root!r}, shape {tuple(self.shape)})"


def _passthrough(name):
    def method(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*args, **kwargs)
    method.__name__ = name
    method.__doc__ = f"Tensor.{name} of the finished tensor (these do not go through the dispatcher: answered here, never garbage)."
    return method


for _name in ("data_ptr", "tolist", "numpy", "untyped_storage", "storage_offset", "__array__", "__dlpack__"):
    setattr(DeferredEpilogue, _name, _passthrough(_name))
    setattr(DeferredCols, _name, _passthrough(_name))


def rope_target(query, key):
    """RotaryEmbedding.forward(positions, query, key) on two column ranges of ONE pending root without a RoPE recorded yet:
    the root, else None."""
    if not (isinstance(query, DeferredCols) and isinstance(key, DeferredCols)):
        return None
    root = query._root
    if key._root is not root or not root.is_pending() or root._rope is not None or query.dim() != 2 or key.dim() != 2:
        return None
    return root


def qkv_root(q, k, v, q_size: int, kv_size: int):
    """attn_backend.forward(q, k, v, ...) on the three column ranges [q | k | v] of ONE pending root whose RoPE is recorded
    on exactly the q and k ranges: the root, else None."""
    if not (isinstance(q, DeferredCols) and isinstance(k, DeferredCols) and isinstance(v, DeferredCols)):
        return None
    root = q._root
    if k._root is not root or v._root is not root or not root.is_pending() or root._rope is None:
        return None
    n = root.shape[1]
    if (q._c0, q._c1, k._c0, k._c1, v._c0, v._c1) != (0, q_size, q_size, q_size + kv_size, q_size + kv_size, n) \
            or n != q_size + 2 * kv_size:
        return None
    if root._rope[2] != (0, q_size) or root._rope[3] != (q_size, q_size + kv_size):
        return None
    return root


def materialize(x):
    """x itself, or the real tensor behind a lazy one (for this package's own ops, which take raw pointers)."""
    return x.materialize() if isinstance(x, (DeferredEpilogue, DeferredCols)) else x
