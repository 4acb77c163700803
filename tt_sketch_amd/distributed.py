"""Multi-GPU sharding of the sketch path: one process per GPU, ONE collective.

The sketch is linear in the input for fixed DRMs (reference sketch_dispatch.py:85-139,
``SketchContainer.__add__`` sketch_container.py:61-69, ``SketchedTensorTrain.__add__``
sketch.py:292-301), so independent additive pieces of the input -- summands of a ``TensorSum``,
nnz shards of a ``SparseTensor`` (``SparseTensor.split``), items of a stream -- are dealt to
the ranks, every rank sketches its share with the SAME DRMs into the packed buffer
``[Psi_0 .. Psi_{d-1}, Omega_0 .. Omega_{d-2}]`` and a single sum over ranks finishes the job
(``stream_sketch_sharded``): RCCL ``ncclAllReduce`` (fp64, over xGMI) through the C ABI.

A single tensor that has no additive pieces is sharded over the sketch RANK instead
(``blocked_stream_sketch_sharded``): the DRMs are cut into rank slices, block (i, j) of the sketch
needs only left slice i and right slice j (reference sketch.py:364-397), the blocks are dealt to the
ranks and assembled by placement (:446-473) after ONE all-gather -- no sum at all.

The collectives are reached through a small communicator protocol (``rank``, ``world``,
``allreduce_sum``, ``allgather``): ``RcclComm`` on the GPUs, ``HostComm`` around any pair of
host callables (the CPU tests wrap a ``gloo`` group with it; nothing in this package imports torch).
"""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from .device import DevArray, copy_into
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


# --------------------------------------------------------------------------- communicators
class HostComm:
    """Communicator over host buffers: ``allreduce(buf) -> summed buf`` and
    ``allgather(buf) -> [buf of rank 0, ...]`` are supplied by the caller (a ``gloo`` group in the
    CPU tests, MPI, ...)."""

    on_device = False

    def __init__(self, rank: int, world: int, allreduce: Callable[[np.ndarray], np.ndarray],
                 allgather: Optional[Callable[[np.ndarray], List[np.ndarray]]] = None,
                 reduce: Optional[Callable[[np.ndarray, int], np.ndarray]] = None):
        self.rank, self.world = int(rank), int(world)
        self._allreduce, self._allgather, self._reduce = allreduce, allgather, reduce

    @classmethod
    def over_files(cls, rank: int, world: int, directory: Optional[str] = None, timeout: float = 300.0) -> "HostComm":
        """Collectives of one node through ``FileRendezvous`` (host buffers written to a shared directory): for
        launches without RCCL -- several ranks on ONE GPU in the tests, or a host-only rehearsal.  Not a fast path:
        every rank reads every other rank's buffer."""
        from .rendezvous import FileRendezvous
        rdv = FileRendezvous(rank, world, directory, timeout)

        def gather(buf):
            return [np.frombuffer(b, dtype=np.float64).copy() for b in rdv.allgather(np.ascontiguousarray(buf).tobytes())]

        def reduce(buf):
            parts = gather(buf)
            out = parts[0]
            for p in parts[1:]:
                out = out + p                      # rank order: every rank forms the same sum
            return out.reshape(np.shape(buf))

        comm = cls(rank, world, reduce, gather)
        comm.close = rdv.close
        return comm

    def allreduce_sum(self, buf: np.ndarray) -> np.ndarray:
        return np.asarray(self._allreduce(np.ascontiguousarray(buf, dtype=np.float64)))

    def reduce_sum(self, buf: np.ndarray, root: int = 0) -> np.ndarray:
        """The sum on ``root``; the other ranks get their own buffer back (as an RCCL reduce leaves it).  Carried by the
        all-reduce the caller supplied, or by ``reduce`` when one was given."""
        fn = getattr(self, "_reduce", None)
        if fn is not None:
            return np.asarray(fn(np.ascontiguousarray(buf, dtype=np.float64), root))
        out = self.allreduce_sum(buf)
        return out if self.rank == root else np.ascontiguousarray(buf, dtype=np.float64)

    def allgather(self, buf: np.ndarray) -> List[np.ndarray]:
        if self._allgather is None:
            raise ValueError("this HostComm was built without an allgather")
        return [np.asarray(b) for b in self._allgather(np.ascontiguousarray(buf, dtype=np.float64))]


class RcclComm:
    """Handle on the library's RCCL communicator (one per process, one process per GPU)."""

    on_device = True

    def __init__(self, rank: int, world: int, broadcast_bytes):
        """``broadcast_bytes(payload_or_None) -> bytes`` carries rank 0's 128-byte id to all ranks
        over any host channel (``FileRendezvous.broadcast``)."""
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            nat.call("ttsk_comm_unique_id", uid)
        raw = broadcast_bytes(bytes(uid.raw) if rank == 0 else None)
        uid = (ctypes.c_char * 128).from_buffer_copy(raw)
        self.rank, self.world = rank, world
        self._scalar = None
        self._rdv = None
        # RCCL writes a version banner to STDOUT when a communicator comes up; a caller's stdout (bench.py: one JSON
        # line) is not the library's to write to -- it goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            nat.call("ttsk_comm_init", uid, rank, world)
            self.max_over_ranks(0.0)          # first collective: whatever RCCL sets up (and prints) lazily
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    @classmethod
    def from_env(cls, device: Optional[int] = None) -> "RcclComm":
        """One rank of a ``torch.distributed.run`` / ``torchrun`` style launch (RANK, LOCAL_RANK,
        WORLD_SIZE in the environment): selects GPU LOCAL_RANK and exchanges the id through files."""
        from .rendezvous import FileRendezvous
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC is the only mode the host driver supports
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        nat.call("ttsk_init", int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device))
        rdv = FileRendezvous(rank, world)
        try:
            comm = cls(rank, world, rdv.broadcast)
        except Exception:
            rdv.close()
            raise
        comm._rdv = rdv
        return comm

    def allreduce_sum(self, buf: DevArray, stream: int = 0) -> DevArray:
        if not buf.is_contiguous():
            raise ValueError("allreduce needs the packed (contiguous) sketch buffer")
        nat.call("ttsk_comm_allreduce_sum", ctypes.c_void_p(buf.ptr), ctypes.c_size_t(buf.size), stream)
        return buf

    def reduce_sum(self, buf: DevArray, root: int = 0, stream: int = 0) -> DevArray:
        nat.call("ttsk_comm_reduce_sum", ctypes.c_void_p(buf.ptr), ctypes.c_size_t(buf.size), root, stream)
        return buf

    def allgather(self, send: DevArray, stream: int = 0) -> DevArray:
        """(world, send.size) array holding every rank's ``send``."""
        if not send.is_contiguous():
            raise ValueError("allgather needs a contiguous buffer")
        recv = DevArray.empty((self.world, send.size), stream=stream)
        nat.call("ttsk_comm_allgather", ctypes.c_void_p(send.ptr), ctypes.c_void_p(recv.ptr),
                 ctypes.c_size_t(send.size), stream)
        return recv

    def max_over_ranks(self, value: float, stream: int = 0) -> float:
        """max of a host scalar over the ranks (blocking): the bench's max-over-ranks clock."""
        if self._scalar is None:
            self._scalar = DevArray.empty((2,), stream=stream)
        host = np.array([float(value), 0.0])
        nat.call("ttsk_h2d", ctypes.c_void_p(self._scalar.ptr), ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(16), stream)
        nat.call("ttsk_comm_allreduce_max", ctypes.c_void_p(self._scalar.ptr), ctypes.c_size_t(2), stream)
        nat.call("ttsk_d2h", ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(self._scalar.ptr), ctypes.c_size_t(16), stream)
        return float(host[0])

    def barrier(self, stream: int = 0) -> None:
        """Every rank has drained ALL its library streams and reached this point."""
        nat.call("ttsk_sync", -1)
        self.max_over_ranks(0.0, stream)

    def close(self) -> None:
        try:
            nat.call("ttsk_comm_destroy")
        finally:
            if self._rdv is not None:
                self._rdv.close()
                self._rdv = None


# --------------------------------------------------------------------------- packed sketches
def _layout(shape, left_rank, right_rank):
    lr, rr = (1,) + tuple(left_rank), tuple(right_rank) + (1,)
    shapes = [(lr[mu], shape[mu], rr[mu]) for mu in range(len(shape))]
    shapes += [(left_rank[mu], right_rank[mu]) for mu in range(len(shape) - 1)]
    return shapes, int(sum(int(np.prod(s)) for s in shapes))


def pack_device(Psi: Sequence[DevArray], Omega: Sequence[DevArray], stream: int = 0) -> DevArray:
    """The packed buffer ``[Psi_0 .. Psi_{d-1}, Omega_0 .. Omega_{d-2}]``.  The one-call TT path already
    produces its sketch in this form (``tt_fused.TTSketchPlan.views``): then no byte is copied."""
    arrs = list(Psi) + list(Omega)
    off, same = arrs[0].offset if arrs else 0, bool(arrs)
    for a in arrs:
        if a.buf is not arrs[0].buf or a.offset != off or not a.is_contiguous():
            same = False
            break
        off += a.size
    total = sum(a.size for a in arrs)
    if same:
        return DevArray(arrs[0].buf, arrs[0].offset, (total,), (1,))
    buf = DevArray.empty((total,), stream=stream)
    off = 0
    for a in arrs:
        if a.size:
            copy_into(buf[off:off + a.size].reshape(a.shape), a, stream)
        off += a.size
    return buf


def unpack_device(buf: DevArray, shape, left_rank, right_rank) -> SketchContainer:
    """A container whose arrays are views into ``buf`` (they stay in HBM until read on the host)."""
    shapes, total = _layout(shape, left_rank, right_rank)
    if buf.size != total:
        raise ValueError(f"packed sketch has {buf.size} entries, the layout needs {total}")
    out, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(buf[off:off + n].reshape(s))
        off += n
    d = len(shape)
    return SketchContainer(out[:d], out[d:], tuple(shape), tuple(left_rank), tuple(right_rank))


def allreduce_container(local: SketchContainer, comm, root: Optional[int] = None) -> SketchContainer:
    """Sum the ranks' partial sketches with ONE collective of the packed buffer
    (``SketchContainer.__add__`` across ranks, reference sketch_container.py:61-69).  ``root``: a single REDUCE to that
    rank (half the link traffic of the all-reduce; what a job needs whose assembly -- ``to_tt`` -- runs on one rank):
    the container is then the whole sum on ``root`` and the rank's own partial sketch elsewhere."""
    if getattr(comm, "on_device", False):
        Psi, Om = local.device_arrays()
        buf = pack_device(Psi, Om)
        buf = comm.allreduce_sum(buf) if root is None else comm.reduce_sum(buf, root)
        return unpack_device(buf, local.shape, local.left_rank, local.right_rank)
    return local.unpack(comm.allreduce_sum(local.pack()) if root is None else comm.reduce_sum(local.pack(), root))


def _device_sketch(tensor, left_drm, right_drm) -> SketchContainer:
    from .sketch_dispatch import SketchMethod, general_sketch
    return general_sketch(tensor, left_drm, right_drm, SketchMethod.streaming)


def stream_sketch_sharded(tensor: Tensor, left_rank, right_rank, comm, seed: Optional[int] = None,
                          left_drm_type=None, right_drm_type=None, left_drm=None, right_drm=None,
                          sketch_fn: Optional[Callable] = None, root: Optional[int] = None):
    """``stream_sketch`` of a TensorSum / SparseTensor whose additive pieces are dealt over the ranks of
    ``comm``: every rank sketches its share with the same DRMs, one all-reduce sums the packed partial
    sketches, every rank returns the ``SketchedTensorTrain`` of the WHOLE tensor
    (reference sketch.py:154-229 for the argument policy, sketch_dispatch.py:85-147 for the sum).
    ``root``: one REDUCE instead -- only that rank holds the sketch of the whole tensor (the others return their partial
    sketch), at half the link traffic.

    The DRMs must be identical on all ranks: pass ``seed`` (the device samplers are pure functions of
    it) or the DRM objects.  ``sketch_fn(shard, left_drm, right_drm) -> SketchContainer`` replaces the
    device sketch (the CPU tests run the oracle through it)."""
    from .sketch import (SketchedTensorTrain, _pick_types, _right_seed, process_tt_rank)
    d = len(tensor.shape)
    lr, rr = np.array(left_rank), np.array(right_rank)
    left_bigger, right_bigger = bool(np.all(lr > rr)), bool(np.all(lr < rr))
    if not (left_bigger or right_bigger):
        raise ValueError("Left ranks or right ranks must be conistently larger or smaller than the "
                         f"other. Left rank: {left_rank}, right rank: {right_rank}")
    if seed is None and (left_drm is None or right_drm is None):
        raise ValueError("stream_sketch_sharded needs `seed` (or both DRMs): every rank must build the same DRMs")
    ltype, rtype = _pick_types(left_drm_type, right_drm_type)
    if left_drm is None:
        left_rank = process_tt_rank(left_rank, tensor.shape, trim=right_bigger)
        left_drm = ltype(left_rank, transpose=False, shape=tensor.shape, seed=seed)
    elif left_drm.rank != left_rank:
        raise ValueError(f"Left rank {left_rank} does not match the rank of the DRM {left_drm.rank}.")
    if right_drm is None:
        right_rank = process_tt_rank(right_rank, tensor.shape, trim=left_bigger)
        right_drm = rtype(right_rank, transpose=True, shape=tensor.shape, seed=_right_seed(seed, d))
    elif tuple(right_drm.rank[::-1]) != right_rank:
        raise ValueError(f"Right rank {right_rank} does not match the rank of the DRM {right_drm.rank}.")
    mine = shard_tensor(tensor, comm.rank, comm.world)
    fn = sketch_fn or _device_sketch
    if mine.num_summands:
        local = fn(mine, left_drm, right_drm)
    else:
        local = SketchContainer.zero(tensor.shape, tuple(left_drm.rank), tuple(right_drm.rank[::-1]))
    total = allreduce_container(local, comm, root)
    return SketchedTensorTrain(total, left_drm, right_drm)


# --------------------------------------------------------------------------- rank-sharded (blocked) sketch
def block_owner(i: int, j: int, n_right: int, world: int) -> int:
    """Blocks are dealt round robin in row-major order."""
    return (i * n_right + j) % world


def blocked_stream_sketch_sharded(tensor: Tensor, left_drm, right_drm, left_rank_slices, right_rank_slices,
                                  comm, sketch_fn: Optional[Callable] = None) -> SketchContainer:
    """``blocked_stream_sketch`` (reference sketch.py:493-525) with the blocks dealt over the ranks:
    block (i, j) = streaming sketch with left DRM slice i and right DRM slice j (:364-397), computed by
    rank ``block_owner(i, j)``; ONE all-gather of the ranks' packed blocks; placement at the rank
    offsets (:446-473) on every rank.  The blocks are disjoint pieces of Psi / Omega -- nothing is
    summed, so the result is bit-identical to the single-process blocked sketch."""
    from .drm_base import CanSlice
    from .sketch import _assemble_blocked_stream_sketches
    for drm in (left_drm, right_drm):
        if not isinstance(drm, CanSlice):
            raise ValueError(f"Blocked sketch not supported for DRM {type(drm).__name__}")
    fn = sketch_fn or _device_sketch
    nl, nr = len(left_rank_slices) - 1, len(right_rank_slices) - 1
    lefts = [left_drm.slice(a, b) for a, b in zip(left_rank_slices[:-1], left_rank_slices[1:])]
    rights = [right_drm.slice(a, b) for a, b in zip(right_rank_slices[:-1], right_rank_slices[1:])]
    layouts: Dict[Tuple[int, int], Tuple[list, int]] = {}
    per_rank = [0] * comm.world
    for i in range(nl):
        for j in range(nr):
            lrk = tuple(b - a for a, b in zip(left_rank_slices[i], left_rank_slices[i + 1]))
            rrk = tuple(b - a for a, b in zip(right_rank_slices[j], right_rank_slices[j + 1]))
            layouts[(i, j)] = (lrk, rrk, _layout(tensor.shape, lrk, rrk)[1])
            per_rank[block_owner(i, j, nr, comm.world)] += layouts[(i, j)][2]
    width = max(per_rank) if per_rank else 0
    mine = [(i, j) for i in range(nl) for j in range(nr) if block_owner(i, j, nr, comm.world) == comm.rank]
    on_dev = getattr(comm, "on_device", False)
    if on_dev:
        send = DevArray.zeros((max(width, 1),))
        off = 0
        for key in mine:
            Psi, Om = fn(tensor, lefts[key[0]], rights[key[1]]).device_arrays()
            n = layouts[key][2]
            blk = pack_device(Psi, Om)
            copy_into(send[off:off + n], blk)
            off += n
        gathered = comm.allgather(send)                    # (world, width) on the device
        rows = [gathered[r] for r in range(comm.world)]
    else:
        send = np.zeros(max(width, 1))
        off = 0
        for key in mine:
            n = layouts[key][2]
            send[off:off + n] = fn(tensor, lefts[key[0]], rights[key[1]]).pack()
            off += n
        rows = comm.allgather(send)
    blocks, cursor = {}, [0] * comm.world
    for i in range(nl):
        for j in range(nr):
            lrk, rrk, n = layouts[(i, j)]
            r = block_owner(i, j, nr, comm.world)
            piece = rows[r][cursor[r]:cursor[r] + n]
            cursor[r] += n
            if on_dev:
                blocks[(i, j)] = unpack_device(piece, tensor.shape, lrk, rrk)
            else:
                blocks[(i, j)] = SketchContainer.zero(tensor.shape, lrk, rrk).unpack(np.asarray(piece))
    return _assemble_blocked_stream_sketches(left_rank_slices, right_rank_slices, tensor.shape, blocks)
