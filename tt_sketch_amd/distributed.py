"""Multi-GPU sharding of the sketch path: one process per GPU, ONE collective.

The sketch is linear in the input for fixed DRMs (reference sketch_dispatch.py:85-139,
``SketchContainer.__add__`` sketch_container.py:61-69, ``SketchedTensorTrain.__add__``
sketch.py:292-301), so independent additive pieces of the input -- summands of a ``TensorSum``,
nnz shards of a ``SparseTensor`` (``SparseTensor.split``), items of a stream -- are dealt to
the ranks, every rank sketches its share with the SAME DRMs into the packed buffer
``[Psi_0 .. Psi_{d-1}, Omega_0 .. Omega_{d-2}]`` and a single sum over ranks finishes the job:
RCCL ``ncclAllReduce`` (fp64, over xGMI) through the C ABI on GPUs, or any ``torch.distributed``
process group on host buffers (used by the CPU tests with ``gloo``).
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from . import _native as nat
from .device import DevArray
from .sketch_container import SketchContainer
from .tensor import SparseTensor, Tensor, TensorSum


def shard_bounds(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) share of ``n_units`` for ``rank`` (first ranks get the extra)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_tensor(tensor: Tensor, rank: int, world: int) -> TensorSum:
    """This rank's additive share of ``tensor``: summands of a TensorSum, or nnz ranges of a
    SparseTensor (reference tensor.py:215-234).  May be an empty TensorSum."""
    if isinstance(tensor, SparseTensor):
        lo, hi = shard_bounds(tensor.nnz, rank, world)
        part = SparseTensor(tensor.shape, np.asarray(tensor.indices)[:, lo:hi], np.asarray(tensor.entries)[lo:hi])
        return TensorSum([part], shape=tensor.shape)
    if isinstance(tensor, TensorSum):
        lo, hi = shard_bounds(tensor.num_summands, rank, world)
        return TensorSum(list(tensor.tensors[lo:hi]), shape=tensor.shape)
    raise ValueError(f"{type(tensor).__name__} has no additive decomposition; shard a TensorSum or SparseTensor")


def allreduce_container(local: SketchContainer, group=None) -> SketchContainer:
    """Sum host-resident partial sketches over a ``torch.distributed`` group (one all_reduce of
    the packed buffer)."""
    import torch
    import torch.distributed as dist
    buf = torch.from_numpy(local.pack())
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return local.unpack(buf.numpy())


class RcclComm:
    """Thin handle on the library's RCCL communicator (one per process)."""

    def __init__(self, rank: int, world: int, broadcast_bytes):
        """``broadcast_bytes(payload_or_None) -> bytes`` carries rank 0's 128-byte id to all ranks
        over any host channel (e.g. a gloo broadcast)."""
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            nat.call("ttsk_comm_unique_id", uid)
        raw = broadcast_bytes(bytes(uid.raw) if rank == 0 else None)
        uid = (ctypes.c_char * 128).from_buffer_copy(raw)
        nat.call("ttsk_comm_init", uid, rank, world)
        self.rank, self.world = rank, world

    def allreduce_sum(self, buf: DevArray, stream: int = 0) -> None:
        if not buf.is_contiguous():
            raise ValueError("allreduce needs the packed (contiguous) sketch buffer")
        nat.call("ttsk_comm_allreduce_sum", ctypes.c_void_p(buf.ptr), ctypes.c_size_t(buf.size), stream)

    def reduce_sum(self, buf: DevArray, root: int = 0, stream: int = 0) -> None:
        nat.call("ttsk_comm_reduce_sum", ctypes.c_void_p(buf.ptr), ctypes.c_size_t(buf.size), root, stream)

    def close(self) -> None:
        nat.call("ttsk_comm_destroy")
