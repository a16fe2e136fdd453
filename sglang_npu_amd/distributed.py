"""Tensor-parallel group + all-reduce for the MI355X backend.

Interface mirrored:
  GroupCoordinator.all_reduce            python/sglang/srt/distributed/parallel_state.py:463-568
  tensor_model_parallel_all_reduce       python/sglang/srt/distributed/communication_op.py:11-13
  get_tensor_model_parallel_{rank,world_size}

One process per GPU; the data plane is RCCL over xGMI through ``torch.distributed`` (backend
"nccl" IS RCCL on ROCm); on CPU tensors (tests) the same code runs over ``gloo``.  ws == 1 returns
the input untouched exactly like parallel_state.py:478-479.

MI355X-first addition (north_star): the all-reduce may be issued on a side HIP stream so that it
overlaps the work the main stream runs next; ``all_reduce_async`` returns a handle whose ``wait()``
fences the main stream on the collective's completion event.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import logging

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)

# How many collectives each data plane took since the last clear() -- bench.py reports it per step so that the first
# multi-GPU run shows which backend every one of the 2 x layers + 1 all-reduces went through (keys: 'quickreduce',
# 'p2p', 'p2p+norm', 'p2p+norm(partials)', 'rccl', 'stub'; 'all_gather(rccl)' for the LM head's gather).  Counted where Python dispatches, i.e. once per capture of a
# graph, not per replay.
import collections  # noqa: E402
DISPATCH_COUNTS: "collections.Counter[str]" = collections.Counter()


class AllReduceHandle:
    def __init__(self, tensor, event=None, stream=None):
        self.tensor, self.event, self.stream = tensor, event, stream

    def wait(self) -> torch.Tensor:
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return self.tensor


def _exchange_ipc_handles(group, world_size: int, create, open_peers) -> Optional[str]:
    """The collective part of bringing up a P2P communicator, written so that EVERY rank issues the same two object
    all-gathers whatever fails locally: `create()` -> this rank's IPC handle (bytes), `open_peers(blob)` maps the
    others' buffers.  A local exception becomes a status the peers see; the second gather is both the barrier (nobody
    launches a P2P kernel before every rank has mapped every peer) and the agreement.  Returns None when every rank
    succeeded, else a message naming the first failing rank (identical on all ranks)."""
    handle, err = b"", None
    try:
        handle = create()
    except Exception as e:  # hipIpcGetMemHandle refused, out of memory, missing entry point, ...
        err = f"create: {type(e).__name__}: {e}"
    gathered = [None] * world_size
    dist.all_gather_object(gathered, (err, handle), group=group)
    failed = [(r, g[0]) for r, g in enumerate(gathered) if g[0] is not None]
    if not failed:
        try:
            open_peers(b"".join(g[1] for g in gathered))
        except Exception as e:  # a peer's buffer cannot be mapped from here (one-directional IPC / xGMI trouble)
            err = f"open_peers: {type(e).__name__}: {e}"
    second = [None] * world_size
    dist.all_gather_object(second, err, group=group)
    failed = failed or [(r, e) for r, e in enumerate(second) if e is not None]
    return None if not failed else f"rank {failed[0][0]}: {failed[0][1]}"


def _agree(tp: "GroupCoordinator", ok: bool) -> bool:
    """True only if `ok` on every rank of the group (one MIN all-reduce of the same shape and dtype on all ranks)."""
    dev = tp.device if tp.device is not None else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=tp.device_group)
    return int(flag.item()) == 1


def _run_lockstep(tp: "GroupCoordinator", steps) -> bool:
    """Run verification `steps` ((name, fn) pairs; fn() -> bool issues its group collectives FIRST, then the P2P kernel it
    checks, then compares) so that no rank ever launches a P2P kernel its peers have given up on: the ranks agree
    (all-reduce MIN) before every step and leave together at the first step after a failure anywhere.  A local exception
    counts as a failed step.  The P2P kernels themselves are bounded (fail-closed timeout), so a step always returns."""
    ok = True
    for _name, fn in steps:
        if not _agree(tp, ok):
            return False
        try:
            ok = bool(fn())
        except Exception:  # noqa: BLE001 -- any local error is a "no" the peers must hear about
            ok = False
    return _agree(tp, ok)


class CustomAllreduce:
    """P2P all-reduce over IPC buffers -- same role and method names as
    python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:35-421
    (``disabled``, ``should_custom_ar(inp)``, ``custom_all_reduce(inp) -> Optional[Tensor]``, ``close()``).
    Out of place, graph-capturable, bit-identical on every rank.  Opt-in (``SGL_MI355_CUSTOM_AR=1``): RCCL
    stays the default data plane."""

    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group, device: torch.device, max_size: int = 16 * 1024 * 1024, rank: Optional[int] = None,
                 world_size: Optional[int] = None, exchange: bool = True):
        import ctypes
        from . import _lib
        self.disabled = True
        self._comm = None
        self.init_error: Optional[str] = None
        self._norm_h: Optional[int] = None  # row length the fused all-reduce + norm staging area is bound to
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self.max_size = max_size
        self.device = device
        self._lib, self._ct = _lib.lib(), ctypes
        if self.world_size > 8 or (self.world_size == 1 and exchange):
            return
        if not exchange:  # the caller wires the peers itself (connect_local)
            self._create()
            return
        # Every rank runs the SAME sequence of group collectives whatever happens locally (a one-directional IPC / xGMI
        # problem must end in "all ranks fall back to RCCL", never in one rank waiting in a collective its peers skipped).
        self.init_error = _exchange_ipc_handles(
            group, self.world_size,
            create=lambda: (self._create(), self._ipc_handle())[1],
            open_peers=lambda blob: _lib.check(self._lib.sgl_mi355_ar_open_peers(self._comm, ctypes.c_char_p(blob))))
        if self.init_error is not None:
            self.close()
            return
        self.disabled = False

    def _create(self):
        from . import _lib
        comm = self._ct.c_void_p()
        _lib.check(self._lib.sgl_mi355_ar_create(self._ct.c_int(self.rank), self._ct.c_int(self.world_size),
                                                 self._ct.c_int64(self.max_size), self._ct.byref(comm)))
        self._comm = comm

    def _ipc_handle(self) -> bytes:
        from . import _lib
        handle = self._ct.create_string_buffer(64)
        _lib.check(self._lib.sgl_mi355_ar_get_ipc_handle(self._comm, handle))
        return bytes(handle.raw)

    @classmethod
    def connect_local(cls, world_size: int, device: torch.device, max_size: int = 16 * 1024 * 1024):
        """All `world_size` ranks inside ONE process (they share a device or reach each other by peer access): the
        peers are wired by pointer (sgl_mi355_ar_set_peers_local), no IPC handles and no process group.  Used to run
        the 6- and 8-rank protocol on a one-GPU box, where one process per rank is not possible."""
        import ctypes
        from . import _lib
        comms = [cls(None, device, max_size, rank=r, world_size=world_size, exchange=False) for r in range(world_size)]
        arr = (ctypes.c_void_p * world_size)(*[c._comm for c in comms])
        for c in comms:
            _lib.check(c._lib.sgl_mi355_ar_set_peers_local(c._comm, arr))
            c.disabled = False
        return comms

    @classmethod
    def single_rank(cls, device: torch.device, max_size: int = 16 * 1024 * 1024):
        """A communicator of ONE rank (its only peer is itself): the all-reduce kernels run their whole protocol -- staging
        stores, flag barriers, the reduction, the fused add + RMSNorm (+ FP8 quant) -- on the one GPU, with nothing to fetch
        from peers.  For rehearsing ONE rank of a TP = N job on a 1-GPU box with the kernel SEQUENCE of the real job (bench.py
        --emulate-tp N): the row-parallel layers hand their split-K partials to the fused all-reduce + norm kernel exactly as a
        real rank does, instead of a finalize launch in front of an identity."""
        c = cls(None, device, max_size, rank=0, world_size=1, exchange=False)
        c.disabled = False
        return c

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        if self.disabled or not inp.is_cuda or not inp.is_contiguous():
            return False
        nbytes = inp.numel() * inp.element_size()
        return nbytes % 16 == 0 and 0 < nbytes <= self.max_size and \
            inp.dtype in (torch.bfloat16, torch.float16, torch.float32)

    def custom_all_reduce(self, inp: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.should_custom_ar(inp):
            return None
        from . import _lib
        if self.timed_out():
            # fail closed: a peer never arrived at a barrier (its output of that call is NaN-filled); the communicator
            # stays unusable -- the role of the std::runtime_error of custom_all_reduce_hip.cuh:512-519
            self.disabled = True
            raise RuntimeError("custom all-reduce: a peer did not reach the barrier in time; the affected outputs were "
                               "filled with NaN and this communicator is disabled")
        DISPATCH_COUNTS["p2p"] += 1
        out = torch.empty_like(inp)
        code = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[inp.dtype]
        ct = self._ct
        _lib.check(self._lib.sgl_mi355_ar_all_reduce(
            self._comm, ct.c_void_p(inp.data_ptr()), ct.c_void_p(out.data_ptr()),
            ct.c_int64(inp.numel() * inp.element_size()), ct.c_int(code),
            ct.c_void_p(torch.cuda.current_stream(inp.device).cuda_stream)))
        return out

    def should_fuse_norm(self, inp: torch.Tensor) -> bool:
        """Shapes the fused all-reduce + add + RMSNorm kernel takes (sgl_mi355_ar_fused_add_rmsnorm)."""
        if self.disabled or not inp.is_cuda or inp.dim() != 2 or not inp.is_contiguous():
            return False
        H = inp.size(1)
        return inp.dtype in (torch.bfloat16, torch.float16) and H % (8 * self.world_size) == 0 and H <= 16384 and \
            0 < inp.numel() * 2 <= self.max_size and self._norm_h in (None, H)

    def should_fuse_norm_shape(self, T: int, H: int, dtype: torch.dtype) -> bool:
        """should_fuse_norm for a [T, H] addend that does not exist as a tensor yet (split-K partials)."""
        return (not self.disabled) and dtype in (torch.bfloat16, torch.float16) and H % (8 * self.world_size) == 0 and \
            H <= 16384 and 0 < T * H * 2 <= self.max_size and self._norm_h in (None, H)

    def rebind_fused_norm(self):
        """Forget the row length the fused-norm staging area is bound to (sgl_mi355_ar_rebind_norm).  Collective in
        spirit: only while no call of this communicator is in flight on any rank (after a group barrier)."""
        from . import _lib
        if self._comm is not None:
            _lib.check(self._lib.sgl_mi355_ar_rebind_norm(self._comm))
        self._norm_h = None

    def fused_add_rmsnorm_partials(self, part, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                   quant_fp8: bool = False, with_fp8_companion: bool = False):
        """fused_add_rmsnorm whose addend is still an ops.GemmPartials (the row-parallel GEMM's split-K sums): its epilogue
        runs while the row is staged (sgl_mi355_ar_fused_add_rmsnorm_partials).  Bit-identical to part.finalize() followed
        by fused_add_rmsnorm; the finalize launch is gone."""
        from . import _lib
        if self.timed_out():
            self.disabled = True
            raise RuntimeError("custom all-reduce: a peer did not reach the barrier in time; the affected outputs were "
                               "filled with NaN and this communicator is disabled")
        DISPATCH_COUNTS["p2p+norm(partials)"] += 1
        T, H = part.M, part.N
        self._norm_h = H
        ct = self._ct
        dev = residual.device
        out = q = s = None
        if quant_fp8 or with_fp8_companion:
            q = torch.empty((T, H), dtype=torch.float8_e4m3fn, device=dev)
            s = torch.empty((T, 1), dtype=torch.float32, device=dev)
        if not quant_fp8 or with_fp8_companion:  # (with_fp8_companion: BOTH from the one launch, as in fused_add_rmsnorm below)
            out = torch.empty((T, H), dtype=part.out_dtype, device=dev)
        ptr = lambda t: ct.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check(self._lib.sgl_mi355_ar_fused_add_rmsnorm_partials(
            self._comm, ptr(part.ws), ct.c_int64(part.num_slices), ptr(part.x_scale), ptr(part.w_scale), ptr(part.bias),
            ptr(residual), ptr(weight), ptr(out), ptr(q), ptr(s), ct.c_int64(T), ct.c_int64(H), ct.c_float(eps),
            ct.c_int(0 if part.out_dtype == torch.bfloat16 else 1),
            ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if with_fp8_companion:
            return out, q, s
        return (q, s) if quant_fp8 else out

    def fused_add_rmsnorm(self, inp: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                          quant_fp8: bool = False, with_fp8_companion: bool = False):
        """all_reduce(inp) + residual -> residual (in place); RMSNorm of the sum -> a new tensor, or with quant_fp8 its
        per-token e4m3 quantisation (q, scale).  One kernel; bit-identical to custom_all_reduce + fused_add_rmsnorm
        (+ sgl_per_token_quant_fp8).  The role of flashinfer_allreduce_residual_rmsnorm behind
        RMSNorm.forward_with_allreduce_fusion (layers/layernorm.py:191-216).
        with_fp8_companion (round 5): BOTH results from the one launch -- returns (out, q, scale); the drop-in RMSNorm attaches
        (q, scale) to `out` for the FP8 linear that follows (ops.attach_fp8_companion)."""
        from . import _lib
        if self.timed_out():
            self.disabled = True
            raise RuntimeError("custom all-reduce: a peer did not reach the barrier in time; the affected outputs were "
                               "filled with NaN and this communicator is disabled")
        DISPATCH_COUNTS["p2p+norm"] += 1
        T, H = inp.shape
        self._norm_h = H
        ct = self._ct
        out = q = s = None
        if quant_fp8 or with_fp8_companion:
            q = torch.empty((T, H), dtype=torch.float8_e4m3fn, device=inp.device)
            s = torch.empty((T, 1), dtype=torch.float32, device=inp.device)
        if not quant_fp8 or with_fp8_companion:
            out = torch.empty_like(inp)
        _lib.check(self._lib.sgl_mi355_ar_fused_add_rmsnorm(
            self._comm, ct.c_void_p(inp.data_ptr()), ct.c_void_p(residual.data_ptr()), ct.c_void_p(weight.data_ptr()),
            ct.c_void_p(out.data_ptr()) if out is not None else None, ct.c_void_p(q.data_ptr()) if q is not None else None,
            ct.c_void_p(s.data_ptr()) if s is not None else None, ct.c_int64(T), ct.c_int64(H), ct.c_float(eps),
            ct.c_int(0 if inp.dtype == torch.bfloat16 else 1),
            ct.c_void_p(torch.cuda.current_stream(inp.device).cuda_stream)))
        if with_fp8_companion:
            return out, q, s
        return (q, s) if quant_fp8 else out

    @contextlib.contextmanager
    def capture(self):
        """Reference surface (custom_all_reduce.py:248-262 registers graph buffers when the capture ends).  Inputs are
        staged into the IPC buffer by the kernel itself here, so there is nothing to register: a no-op context."""
        yield

    def timed_out(self) -> bool:
        flag = self._ct.c_int(0)
        self._lib.sgl_mi355_ar_timed_out(self._comm, self._ct.byref(flag))
        return bool(flag.value)

    def close(self):
        if self._comm is not None:
            self._lib.sgl_mi355_ar_destroy(self._comm)
            self._comm = None
            self.disabled = True


class QuickReduceRegime:
    """device_communicators/quick_all_reduce.py:47-52."""
    FP, INT8, INT6, INT4, NONE = 0, 1, 2, 3, 4
    _NAMES = {"FP": 0, "INT8": 1, "INT6": 2, "INT4": 3, "NONE": 4}


class QuickAllReduce:
    """QuickReduce for prefill-size messages -- same role, switches and method names as
    python/sglang/srt/distributed/device_communicators/quick_all_reduce.py:56-273 (``disabled``,
    ``should_quick_allreduce(inp)``, ``quick_all_reduce(inp, out=None)``, ``close()``):
    * opt-in through ``ROCM_QUICK_REDUCE_QUANTIZATION`` = FP | INT8 | INT6 | INT4 (default NONE = disabled, :183-199);
    * ``ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB`` caps the message size (default 2 GiB as ``ops.qr_max_size()``, :205-215);
    * the per-(dtype, world size, regime) minimum sizes of ``_QR_MIN_SIZE`` (:62-71): below them the custom all-reduce
      or RCCL is faster and ``should_quick_allreduce`` says no;
    * ``ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16`` (:177-181) is read for the threshold table only: the kernels here do the
      codec arithmetic in fp32 whatever the input dtype, so there is nothing to cast.
    It runs on the P2P communicator's IPC staging area (``CustomAllreduce``; the reference allocates a second one), in
    chunks, by ``sgl_mi355_ar_quick_all_reduce``.  World sizes 2 / 4 / 8 as upstream (6 works too)."""

    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]
    _SUPPORTED_DTYPES = [torch.float16, torch.bfloat16]
    _MB = 1024 * 1024
    # quick_all_reduce.py:62-71, [FP, INT8, INT6, INT4]; world size 6 takes the row of 8
    _QR_MIN_SIZE = {
        (torch.float16, 2): [1, 2, 2, 1], (torch.float16, 4): [1, 16, 4, 2], (torch.float16, 8): [16, 4, 4, 2],
        (torch.bfloat16, 2): [2, 8, 8, 8], (torch.bfloat16, 4): [8, 64, 64, 16], (torch.bfloat16, 8): [16, 2048, 2048, 2048],
    }

    def __init__(self, ca_comm: Optional[CustomAllreduce], regime: Optional[str] = None, max_size_mb: Optional[int] = None):
        self.disabled = True
        self.ca_comm = ca_comm
        if ca_comm is None or ca_comm._comm is None or ca_comm.world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        regime = regime if regime is not None else os.environ.get("ROCM_QUICK_REDUCE_QUANTIZATION", "NONE")
        if regime not in QuickReduceRegime._NAMES:
            logger.warning("quick all-reduce: invalid quantization level %r (supported: %s)", regime,
                           list(QuickReduceRegime._NAMES))
            return
        if regime == "NONE":
            return
        self.qr_quant_level = QuickReduceRegime._NAMES[regime]
        self.use_fp16_kernels = int(os.environ.get("ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16", 1))
        mb = max_size_mb if max_size_mb is not None else int(os.environ.get("ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB", 0))
        self.qr_max_size = mb * self._MB if mb > 0 else 2048 * self._MB
        self.world_size, self.rank = ca_comm.world_size, ca_comm.rank
        self.disabled = False

    def should_quick_allreduce(self, inp: torch.Tensor) -> bool:
        if self.disabled or self.ca_comm.disabled or inp.dtype not in self._SUPPORTED_DTYPES or not inp.is_cuda:
            return False
        nbytes = inp.numel() * inp.element_size()
        if nbytes % 64 != 0 or not inp.is_contiguous():  # whole 32-value blocks (upstream: multiples of 16 bytes)
            return False
        dtype = torch.float16 if self.use_fp16_kernels else inp.dtype
        ws = 8 if self.world_size == 6 else self.world_size
        return self._QR_MIN_SIZE[(dtype, ws)][self.qr_quant_level] * self._MB <= nbytes <= self.qr_max_size

    def quick_all_reduce(self, inp: torch.Tensor, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Out of place; graph-capturable (static IPC staging area, as upstream :240-243)."""
        from . import _lib
        ca = self.ca_comm
        if ca.timed_out():
            ca.disabled = True
            raise RuntimeError("quick all-reduce: a peer did not reach the barrier in time; the affected outputs were "
                               "filled with NaN and this communicator is disabled")
        DISPATCH_COUNTS["quickreduce"] += 1
        if out is None:
            out = torch.empty_like(inp)
        ct = ca._ct
        _lib.check(ca._lib.sgl_mi355_ar_quick_all_reduce(
            ca._comm, ct.c_void_p(inp.data_ptr()), ct.c_void_p(out.data_ptr()),
            ct.c_int64(inp.numel() * inp.element_size()), ct.c_int(0 if inp.dtype == torch.bfloat16 else 1),
            ct.c_int(self.qr_quant_level), ct.c_void_p(torch.cuda.current_stream(inp.device).cuda_stream)))
        return out

    def close(self):
        self.disabled = True


class GroupCoordinator:
    def __init__(self, group: Optional[dist.ProcessGroup], rank: int, world_size: int, device: Optional[torch.device]):
        self.device_group = group
        self.rank_in_group = rank
        self.world_size = world_size
        self.device = device
        self._side_stream = None
        self.ca_comm: Optional[CustomAllreduce] = None
        self.qr_comm: Optional["QuickAllReduce"] = None
        # measurement aid (overhead = (step with AR - step with AR stubbed to identity) / step)
        self.stub_all_reduce = False
        # one-rank rehearsal of a TP = N job (bench.py --emulate-tp): the plain collectives are identities (stub_all_reduce) but
        # the FUSED all-reduce + add + RMSNorm still runs, on a CustomAllreduce.single_rank communicator -- the launches of a
        # real rank, minus the peers' bytes
        self.fuse_under_stub = False

    @property
    def fused_collectives_on(self) -> bool:
        """May a row-parallel layer leave its collective to the next norm's fused kernel?"""
        return (not self.stub_all_reduce) or self.fuse_under_stub

    @property
    def side_stream(self):
        if self._side_stream is None and self.device is not None and self.device.type == "cuda":
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """SUM over the TP ranks (in place, like the pynccl path parallel_state.py:563-568)."""
        if self.world_size == 1 or self.stub_all_reduce:
            if self.world_size > 1:
                DISPATCH_COUNTS["stub"] += 1
            return input_
        # dispatch order of parallel_state.py:519-542: QuickReduce, then the custom P2P all-reduce, if they accept the
        # tensor, else RCCL
        qr = getattr(self, "qr_comm", None)
        if qr is not None and not qr.disabled and qr.should_quick_allreduce(input_):
            return qr.quick_all_reduce(input_)
        if self.ca_comm is not None and not self.ca_comm.disabled:
            out = self.ca_comm.custom_all_reduce(input_)
            if out is not None:
                return out
        DISPATCH_COUNTS["rccl"] += 1
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def all_reduce_async(self, input_: torch.Tensor) -> AllReduceHandle:
        """Issue the all-reduce on the side stream; the caller's stream is only fenced at wait().  Same dispatch as
        all_reduce (stub / custom P2P / RCCL); the handle's tensor is the result (the P2P kernel is out of place)."""
        if self.world_size == 1 or self.stub_all_reduce:
            return AllReduceHandle(input_)
        if not input_.is_cuda:
            return AllReduceHandle(self.all_reduce(input_))
        if torch.cuda.is_current_stream_capturing() and not (
                self.ca_comm is not None and not self.ca_comm.disabled and self.ca_comm.should_custom_ar(input_)):
            # Under stream capture the process group's collective stays IN-STREAM: a fork to the side stream and back buys
            # nothing inside a graph (the consumer is the next node anyway) and a library collective captured across two
            # streams is the one pattern bench.py's capturability probe does not cover (round 5; first 8-GPU contact).
            return AllReduceHandle(self.all_reduce(input_))
        side = self.side_stream
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        side.wait_event(ready)
        with torch.cuda.stream(side):
            out = self.all_reduce(input_)
            done = torch.cuda.Event()
            done.record(side)
        input_.record_stream(side)
        out.record_stream(torch.cuda.current_stream())
        return AllReduceHandle(out, done, side)

    def all_gather(self, input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """parallel_state.py all_gather: concatenate the ranks' tensors along `dim`."""
        if self.world_size == 1 or self.stub_all_reduce:
            return input_
        if dim < 0:
            dim += input_.dim()
        inp = input_.contiguous()
        if inp.dim() == 0:
            inp = inp.view(1)
        DISPATCH_COUNTS["all_gather(rccl)"] += 1
        flat = torch.empty((self.world_size * inp.shape[0],) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        dist.all_gather_into_tensor(flat, inp, group=self.device_group)
        out = flat.view((self.world_size,) + tuple(input_.shape)).movedim(0, dim)
        shape = list(input_.shape)
        shape[dim] *= self.world_size
        return out.reshape(shape)


_TP: Optional[GroupCoordinator] = None


def init_distributed_environment(backend: Optional[str] = None, device: Optional[torch.device] = None) -> GroupCoordinator:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun sets them) and
    make the whole world one TP group (single node, TP only -- the BASELINE configs)."""
    global _TP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        _TP = GroupCoordinator(None, 0, 1, device)
        return _TP
    if backend is None:
        # SGL_MI355_DIST_BACKEND is a rehearsal aid: "gloo" lets several ranks share ONE GPU (RCCL refuses
        # duplicate devices), e.g. to exercise the N>1 code path on a 1-GPU box.
        backend = os.environ.get("SGL_MI355_DIST_BACKEND") or (
            "nccl" if (device is not None and device.type == "cuda") else "gloo")
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    _TP = GroupCoordinator(dist.group.WORLD, rank, world, device)
    # P2P all-reduce: SGL_MI355_CUSTOM_AR=1 on, =0 off, unset = "verify": build it, check it against the process
    # group's own all-reduce on this very node (integer payloads: exact), and keep it only if EVERY rank agrees --
    # otherwise RCCL stays the data plane.  (The kernel has been exercised with ranks sharing one GPU and with all ranks
    # in one process; this start-up check is what covers the first run across xGMI.)
    mode = os.environ.get("SGL_MI355_CUSTOM_AR", "verify")
    if mode != "0" and device is not None and device.type == "cuda" and world in CustomAllreduce._SUPPORTED_WORLD_SIZES:
        _bring_up_custom_ar(_TP, lambda: CustomAllreduce(dist.group.WORLD, device), verify=(mode != "1"))
    return _TP


def _preflight_library() -> None:
    """Everything CustomAllreduce's constructor does BEFORE its first group collective and that can fail on one rank only."""
    from . import _lib
    lib = _lib.lib()
    for sym in ("sgl_mi355_ar_create", "sgl_mi355_ar_get_ipc_handle", "sgl_mi355_ar_open_peers", "sgl_mi355_ar_all_reduce"):
        getattr(lib, sym)


def _bring_up_custom_ar(tp: "GroupCoordinator", make_comm, verify: bool = True, steps_of=None, preflight=None) -> bool:
    """Build the P2P communicator (`make_comm()`), verify it against the process group and install it on `tp` only if
    EVERY rank agrees; otherwise close it everywhere and leave RCCL as the data plane.  All ranks issue the same group
    collectives in the same order whatever fails locally (a rank that bailed out early used to pair its
    `flag` all-reduce with a peer's bf16 all-reduce and hang the job at init)."""
    ca, err = None, None
    # Pre-flight: what can fail on ONE rank before the constructor's first collective -- the kernel library not
    # loading, an ABI mismatch -- is found out and agreed on FIRST.  A rank that raised there used to go straight to the
    # agreement below while its peers sat in the constructor's object all-gather: a hang at init.
    pre_err = None
    try:
        (preflight or _preflight_library)()
    except Exception as e:  # noqa: BLE001
        pre_err = f"{type(e).__name__}: {e}"
    if not _agree(tp, pre_err is None):
        if tp.rank_in_group == 0:
            print(f"[sglang_npu_amd] P2P all-reduce not used ({pre_err or 'the kernel library did not load on some rank'}); "
                  f"using RCCL", flush=True)
        return False
    try:
        ca = make_comm()  # lockstep inside (_exchange_ipc_handles)
        err = ca.init_error
    except Exception as e:  # (after the pre-flight only a failure BEHIND the constructor's collectives can land here)
        err = f"{type(e).__name__}: {e}"
    ok = _agree(tp, ca is not None and not ca.disabled)
    if ok and verify:
        qr = QuickAllReduce(ca)
        steps = (steps_of or _verification_steps)(ca, qr, tp)
        ok = _run_lockstep(tp, steps)  # (ends with a group agreement behind device synchronisations: nothing in flight)
        if ok and hasattr(ca, "rebind_fused_norm"):
            ca.rebind_fused_norm()     # the self-check used a row length of its own; the model binds its own H
    if ok:
        tp.ca_comm = ca
        tp.qr_comm = QuickAllReduce(ca)  # disabled unless ROCM_QUICK_REDUCE_QUANTIZATION asks for a regime
        return True
    if ca is not None:
        ca.close()  # not leaked: the IPC staging area goes back
    if tp.rank_in_group == 0:
        print(f"[sglang_npu_amd] P2P all-reduce not used ({err or 'start-up verification failed on some rank'}); "
              f"using RCCL", flush=True)
    return False


def _verification_steps(ca: "CustomAllreduce", qr: "QuickAllReduce", tp: "GroupCoordinator"):
    """What the model is going to call, each against the process group's own all-reduce + the plain ops on integer-valued
    payloads (sums exact in bf16): one-shot and two-shot all-reduce (both halves of the double buffer), the fused
    all-reduce + add + RMSNorm on a finished addend AND on split-K partial sums (+ FP8 quant, the form the decode layers
    use), and -- when ROCM_QUICK_REDUCE_QUANTIZATION enables it -- QuickReduce within the reference test's bound
    (test_quick_allreduce.py:131-165) and identical on every rank."""
    from . import ops
    dev, rank, world = tp.device, tp.rank_in_group, tp.world_size
    grp = tp.device_group
    steps = []

    # Every step issues its group collectives UNCONDITIONALLY and with fixed shapes -- operands that could not be built
    # locally are replaced by zeros of the declared shape, the P2P call sits in a try of its own -- so that a local
    # exception can never pair this rank's next agreement with a peer's reference all-reduce.
    def reference_sum(shape, dtype, make):
        x, failed = None, False
        try:
            x = make()
        except Exception:  # noqa: BLE001
            failed = True
        ref = x.clone() if x is not None else torch.zeros(shape, device=dev, dtype=dtype)
        dist.all_reduce(ref, group=grp)
        return x, ref, failed

    def plain(n):
        def fn():
            def make():
                g = torch.Generator(device=dev).manual_seed(100 + rank)
                return torch.randint(-3, 4, (n,), device=dev, generator=g).to(torch.bfloat16)
            x, ref, failed = reference_sum((n,), torch.bfloat16, make)
            if failed:
                return False
            good = True
            for _ in range(2):  # both halves of the double buffer
                out = ca.custom_all_reduce(x)
                torch.cuda.synchronize(dev)
                good = good and out is not None and not ca.timed_out() and torch.equal(out, ref)
            return good
        return fn
    steps.append(("one-shot 8 KiB", plain(4096)))
    steps.append(("two-shot 1 MiB", plain(1 << 19)))

    T, H = 64, 1024 * world

    def operands():
        built = {}

        def make():
            g = torch.Generator(device=dev).manual_seed(7)
            built["part"] = torch.randint(-2, 3, (T, H), device=dev, generator=g).to(torch.bfloat16) * (rank + 1)
            built["res"] = torch.randint(-2, 3, (T, H), device=dev, generator=g).to(torch.bfloat16)
            built["w"] = torch.ones(H, device=dev, dtype=torch.bfloat16)
            return built["part"]
        part, red, failed = reference_sum((T, H), torch.bfloat16, make)  # (the collective is issued either way)
        if failed:
            raise RuntimeError("verification operands could not be built on this rank")
        return part, built["res"], built["w"], red

    def fused_norm():
        part, res, w, red = operands()
        res_ref = res.clone()
        ops.fused_add_rmsnorm(red, res_ref, w, 1e-5)  # in place: red <- norm, res_ref <- sum
        res2 = res.clone()
        got = ca.fused_add_rmsnorm(part, res2, w, 1e-5)
        torch.cuda.synchronize(dev)
        return (not ca.timed_out()) and torch.equal(got, red) and torch.equal(res2, res_ref)
    steps.append(("fused all-reduce + add + RMSNorm", fused_norm))

    def fused_norm_partials():
        part, res, w, red = operands()
        res_ref = res.clone()
        q_ref, s_ref, _ = ops.rmsnorm_quant_fp8(red, w, 1e-5, residual=res_ref)
        # the same addend as three split-K slices of integer-valued fp32 partial sums with unit scales
        slabs = torch.stack([part.float() - 1.0, torch.ones_like(part, dtype=torch.float32),
                             torch.zeros_like(part, dtype=torch.float32)]).contiguous()
        gp = ops.GemmPartials(slabs, 3, torch.ones(T, device=dev), torch.ones(H, device=dev), None, T, H, torch.bfloat16)
        res2 = res.clone()
        q, s_ = ca.fused_add_rmsnorm_partials(gp, res2, w, 1e-5, quant_fp8=True)
        torch.cuda.synchronize(dev)
        return (not ca.timed_out()) and torch.equal(q.view(torch.uint8), q_ref.view(torch.uint8)) and \
            torch.equal(s_.view(-1), s_ref.view(-1)) and torch.equal(res2, res_ref)
    steps.append(("fused all-reduce + add + RMSNorm + FP8 quant on split-K partials", fused_norm_partials))

    if not qr.disabled:
        def quick():
            n = 32 * 4096

            def make():
                g = torch.Generator(device=dev).manual_seed(1000 + rank)
                return torch.randint(1, 24, (n,), device=dev, generator=g).to(torch.bfloat16).float()
            xf, exact, failed = reference_sum((n,), torch.float32, make)
            out = None
            if not failed:
                try:
                    out = qr.quick_all_reduce(xf.to(torch.bfloat16))
                    torch.cuda.synchronize(dev)
                except Exception:  # noqa: BLE001 -- the two collectives below are still issued
                    out = None
            local_ok = out is not None
            if out is None:
                out = torch.zeros(n, device=dev, dtype=torch.bfloat16)
            lo, hi = out.float(), out.float()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=grp)  # (issued by every rank whatever `out` holds)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=grp)
            if not local_ok:
                return False
            same = torch.equal(lo, hi)  # every rank decoded the same bytes
            close = bool(((out.float() - exact).abs() <= 1.25 * world + 0.5 * world * exact.abs()).all())
            exact_fp = qr.qr_quant_level != QuickReduceRegime.FP or torch.equal(out.float(), exact)
            return (not ca.timed_out()) and same and close and exact_fp
        steps.append(("QuickReduce", quick))
    return steps


def get_tp_group() -> GroupCoordinator:
    global _TP
    if _TP is None:
        _TP = GroupCoordinator(None, 0, 1, None)
    return _TP


def set_tp_group(g: GroupCoordinator) -> None:
    global _TP
    _TP = g


def get_tensor_model_parallel_world_size() -> int:
    return get_tp_group().world_size


def get_tensor_model_parallel_rank() -> int:
    return get_tp_group().rank_in_group


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    return get_tp_group().all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    return get_tp_group().all_gather(input_, dim)
