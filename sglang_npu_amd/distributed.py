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

import os
from typing import Optional

import torch
import torch.distributed as dist


class AllReduceHandle:
    def __init__(self, tensor, event=None, stream=None):
        self.tensor, self.event, self.stream = tensor, event, stream

    def wait(self) -> torch.Tensor:
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return self.tensor


class GroupCoordinator:
    def __init__(self, group: Optional[dist.ProcessGroup], rank: int, world_size: int, device: Optional[torch.device]):
        self.device_group = group
        self.rank_in_group = rank
        self.world_size = world_size
        self.device = device
        self._side_stream = None

    @property
    def side_stream(self):
        if self._side_stream is None and self.device is not None and self.device.type == "cuda":
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """SUM over the TP ranks (in place, like the pynccl path parallel_state.py:563-568)."""
        if self.world_size == 1:
            return input_
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def all_reduce_async(self, input_: torch.Tensor) -> AllReduceHandle:
        """Issue the all-reduce on the side stream; the caller's stream is only fenced at wait()."""
        if self.world_size == 1:
            return AllReduceHandle(input_)
        if not input_.is_cuda:
            dist.all_reduce(input_, group=self.device_group)
            return AllReduceHandle(input_)
        side = self.side_stream
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        side.wait_event(ready)
        with torch.cuda.stream(side):
            dist.all_reduce(input_, group=self.device_group)
            done = torch.cuda.Event()
            done.record(side)
        input_.record_stream(side)
        return AllReduceHandle(input_, done, side)

    def all_gather(self, input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """parallel_state.py all_gather: concatenate the ranks' tensors along `dim`."""
        if self.world_size == 1:
            return input_
        if dim < 0:
            dim += input_.dim()
        inp = input_.contiguous()
        if inp.dim() == 0:
            inp = inp.view(1)
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
    return _TP


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
