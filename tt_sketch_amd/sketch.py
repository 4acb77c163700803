"""Public sketching API: ``stream_sketch``, ``orthogonal_sketch``, ``hmt_sketch``,
``SketchedTensorTrain``, blocked sketches.

Same call signatures, defaults and error behaviour as the reference's ``tt_sketch/sketch.py``
(:44-229 entry points, :232-361 ``SketchedTensorTrain``, :364-525 blocked sketch and
assembly); the work happens in ``sketch_dispatch.general_sketch`` on the GPU and the
Omega-pseudoinverse assembly (:400-443) runs as device Jacobi-SVD + MFMA GEMM.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence, Tuple, Type

import numpy as np
import numpy.typing as npt

from .device import DevArray, as_dev, contract, sync, to_host
from .drm import ALL_DRM, DenseGaussianDRM, SparseGaussianDRM, TensorTrainDRM
from .drm_base import DRM, CanIncreaseRank, CanSlice
from .sketch_container import SketchContainer
from .sketch_dispatch import SketchMethod, general_sketch
from .sketching_methods.abstract_methods import (CansketchCP, CansketchDense, CansketchSparse,
                                                 CansketchTT)
from .tensor import Tensor, TensorTrain
from .utils import ArrayList, TTRank, pinv_dev_many, process_tt_rank, refine_left, refine_right

DEFAULT_DRM = {
    CansketchDense: DenseGaussianDRM,
    CansketchSparse: SparseGaussianDRM,
    CansketchTT: TensorTrainDRM,
    CansketchCP: TensorTrainDRM,
}

BlockedSketch = Dict[Tuple[int, int], SketchContainer]


def _fresh_seed() -> int:
    return int(np.random.SeedSequence().generate_state(1)[0])


def _right_seed(seed: int, d: int) -> int:
    # The reference derives the right seed from hash(str(d)), which changes from process to
    # process (sketch.py:132,210).  A fixed odd multiplier keeps the two DRMs
    # independent *and* reproducible.
    return (int(seed) + 0x9E3779B1 * (d + 1)) % 2**32


def _pick_types(left_drm_type, right_drm_type):
    """A missing DRM type defaults to the other side's, then to TensorTrainDRM."""
    given = [t for t in (left_drm_type, right_drm_type) if t is not None]
    fallback = given[0] if given else TensorTrainDRM
    return (left_drm_type or fallback), (right_drm_type or (left_drm_type or fallback))


def hmt_sketch(tensor: Tensor, rank: TTRank, seed: Optional[int] = None,
               drm_type: Optional[Type[DRM]] = None, drm: Optional[DRM] = None,
               return_drm: bool = False):
    """One-sided (Halko-Martinsson-Tropp style) sketch; returns a left-orthogonal TensorTrain
    (reference sketch.py:44-78)."""
    if seed is None:
        seed = _fresh_seed()
    if drm is None:
        drm_type = TensorTrainDRM if drm_type is None else drm_type
        rank = process_tt_rank(rank, tensor.shape, trim=True)
        drm = drm_type(rank, transpose=True, shape=tensor.shape, seed=seed)
    elif tuple(drm.rank[::-1]) != rank:
        raise ValueError(f"Right rank {rank} does not match the rank of the DRM {drm.rank}.")
    sketch = general_sketch(tensor, None, drm, method=SketchMethod.hmt)
    sketched = TensorTrain(sketch.device_arrays()[0])     # cores stay in HBM (DevArray; np.asarray(core) copies out)
    return (sketched, drm) if return_drm else sketched


def orthogonal_sketch(tensor: Tensor, left_rank: TTRank, right_rank: TTRank,
                      seed: Optional[int] = None, left_drm_type: Optional[Type[DRM]] = None,
                      right_drm_type: Optional[Type[DRM]] = None, left_drm: Optional[DRM] = None,
                      right_drm: Optional[DRM] = None, return_drm: bool = False):
    """Two-sided sketch with an orthogonalisation after every core; returns a TensorTrain
    (reference sketch.py:81-151).  Requires right_rank > left_rank elementwise."""
    left_drm, right_drm = _orth_drms(tensor, left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm)
    sketch = general_sketch(tensor, left_drm, right_drm, method=SketchMethod.orthogonal)
    sketched = TensorTrain(sketch.device_arrays()[0])     # cores stay in HBM (DevArray; np.asarray(core) copies out)
    return (sketched, left_drm, right_drm) if return_drm else sketched


def _stream_drms(tensor: Tensor, left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm):
    """argument policy of ``stream_sketch`` (reference sketch.py:170-219): rank direction, trimming of the smaller side
    only, default DRM types and seeds, rank checks of explicit DRMs"""
    d = len(tensor.shape)
    lr, rr = np.array(left_rank), np.array(right_rank)
    left_bigger, right_bigger = bool(np.all(lr > rr)), bool(np.all(lr < rr))
    if not (left_bigger or right_bigger):
        raise ValueError("Left ranks or right ranks must be conistently larger or smaller than the "
                         f"other. Left rank: {left_rank}, right rank: {right_rank}")
    if seed is None:
        seed = _fresh_seed()
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
        raise ValueError(
            f"Right rank {right_rank} does not match the rank of the DRM {right_drm.rank}.")
    return left_drm, right_drm


def stream_sketch(tensor: Tensor, left_rank: TTRank, right_rank: TTRank, seed: Optional[int] = None,
                  left_drm_type: Optional[Type[DRM]] = None,
                  right_drm_type: Optional[Type[DRM]] = None, left_drm: Optional[DRM] = None,
                  right_drm: Optional[DRM] = None, return_drm: bool = False):
    """Streaming two-sided sketch; returns a ``SketchedTensorTrain`` (reference sketch.py:154-229).
    One side's ranks must dominate the other's elementwise; only the smaller side is trimmed."""
    left_drm, right_drm = _stream_drms(tensor, left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm)
    sketch = general_sketch(tensor, left_drm, right_drm, method=SketchMethod.streaming)
    sketched = SketchedTensorTrain(sketch, left_drm, right_drm)
    return (sketched, left_drm, right_drm) if return_drm else sketched


def stream_sketch_batch(tensors: Sequence[Tensor], left_rank: TTRank, right_rank: TTRank, seed: Optional[int] = None,
                        left_drm_type: Optional[Type[DRM]] = None, right_drm_type: Optional[Type[DRM]] = None,
                        left_drm: Optional[DRM] = None, right_drm: Optional[DRM] = None, return_drm: bool = False):
    """``[stream_sketch(t, ...) for t in tensors]`` with ONE pair of DRMs -- the streaming setting of the reference
    (sketch.py:292-301: further tensors are sketched with the DRMs of the first) -- as one batched pass on the device
    where the inputs allow it: TensorTrains of one signature (mode sizes, TT ranks) with TensorTrainDRMs go through
    ``ttsk_tt_sketch_batch`` in slices of 32 tensors (every chain product is one launch over the whole slice; the DRM
    cores are read once per launch), anything else falls back to one ``stream_sketch`` per tensor.  Returns the list of
    ``SketchedTensorTrain`` (device-resident; the sketches of a batched pass share one packed buffer)."""
    tensors = list(tensors)
    if not tensors:
        return ([], left_drm, right_drm) if return_drm else []
    left_drm, right_drm = _stream_drms(tensors[0], left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm)
    from . import tt_fused
    batched = (type(left_drm) is TensorTrainDRM and type(right_drm) is TensorTrainDRM and not left_drm.transpose
               and right_drm.transpose and all(type(t) is TensorTrain for t in tensors)
               and all(tuple(t.shape) == tuple(tensors[0].shape) and tuple(t.rank) == tuple(tensors[0].rank) for t in tensors))
    out = []
    if batched:
        import ctypes
        if tuple(left_drm.shape) != tuple(tensors[0].shape) or tuple(right_drm.shape) != tuple(tensors[0].shape):
            raise ValueError(f"Shape {left_drm.shape} of DRM doesn't match tensor's shape {tensors[0].shape}")
        plan = tt_fused.TTSketchPlan(tensors[0].shape, tensors[0].rank, left_drm, right_drm)
        stride = plan.size + (plan.size & 1)
        buf = DevArray.empty((len(tensors) * stride,))
        keep, flat = [], []
        for t in tensors:
            ptrs, k = plan.core_pointers(t)
            keep.append(k)
            flat += [ptrs[i] for i in range(plan.d)]
        plan.run_batch((ctypes.c_void_p * len(flat))(*flat), len(tensors), buf, stride)
        for b in range(len(tensors)):
            Psi, Om = plan.views(buf[b * stride:b * stride + plan.size])
            out.append(SketchedTensorTrain(SketchContainer(Psi, Om), left_drm, right_drm))
    else:
        lrank, rrank = left_drm.rank, tuple(right_drm.rank[::-1])
        for t in tensors:
            out.append(stream_sketch(t, lrank, rrank, left_drm=left_drm, right_drm=right_drm))
    return (out, left_drm, right_drm) if return_drm else out


ORTH_BATCH_SLICE = 16        # tensors per ttsk_tt_orth_sketch_batch call (four chains run at a time; outputs of a call share one buffer)


def _orth_batch(tensors, left_drm, right_drm, method, one):
    """the tensors through ``ttsk_tt_orth_sketch_batch`` in slices where that applies; ``one(t)``: the per-tensor call, for
    everything else and for tensors whose fast factorisations were rejected (rank-deficient Omega, ill-conditioned
    unfolding: the verdict comes back per tensor)"""
    from . import tt_fused
    from .sketch_dispatch import robust_reruns
    out = []
    for b0 in range(0, len(tensors), ORTH_BATCH_SLICE):
        part = tensors[b0:b0 + ORTH_BATCH_SLICE]
        res = tt_fused.try_orth_sketch_batch(part, left_drm, right_drm, method) if len(part) > 1 else None
        if res is None:
            out += [one(t) for t in part]
            continue
        outs, status = res
        bad = status.get().view(np.int32)[:len(part)]
        for t, (cores, _), rejected in zip(part, outs, bad):
            if rejected:
                robust_reruns[method.value] = robust_reruns.get(method.value, 0) + 1
                out.append(one(t))
            else:
                out.append(TensorTrain(cores))
    return out


def orthogonal_sketch_batch(tensors: Sequence[Tensor], left_rank: TTRank, right_rank: TTRank, seed: Optional[int] = None,
                            left_drm_type: Optional[Type[DRM]] = None, right_drm_type: Optional[Type[DRM]] = None,
                            left_drm: Optional[DRM] = None, right_drm: Optional[DRM] = None, return_drm: bool = False):
    """``[orthogonal_sketch(t, ...) for t in tensors]`` with ONE pair of DRMs (the setting of reference sketch.py:292-301;
    the caller with many same-shaped inputs is tt_gmres.py:293-299).  TensorTrains of one signature with TensorTrainDRMs
    run as concurrent chains on the device (``ttsk_tt_orth_sketch_batch``: a single sketch is ~90 dependent launches on
    small operands, four of them side by side fill the gaps); anything else is one ``orthogonal_sketch`` per tensor."""
    tensors = list(tensors)
    if not tensors:
        return ([], left_drm, right_drm) if return_drm else []
    left_drm, right_drm = _orth_drms(tensors[0], left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm)
    lrank, rrank = left_drm.rank, tuple(right_drm.rank[::-1])
    one = lambda t: orthogonal_sketch(t, lrank, rrank, left_drm=left_drm, right_drm=right_drm)
    out = _orth_batch(tensors, left_drm, right_drm, SketchMethod.orthogonal, one)
    return (out, left_drm, right_drm) if return_drm else out


def _orth_drms(tensor, left_rank, right_rank, seed, left_drm_type, right_drm_type, left_drm, right_drm):
    """argument policy of ``orthogonal_sketch`` (reference sketch.py:99-140) without the sketch"""
    d = len(tensor.shape)
    if not bool(np.all(np.array(left_rank) < np.array(right_rank))):
        raise ValueError("The right rank needs to be larger than the left rank. "
                         f"Left rank: {left_rank}, right rank: {right_rank}")
    if seed is None:
        seed = _fresh_seed()
    ltype, rtype = _pick_types(left_drm_type, right_drm_type)
    if left_drm is None:
        left_rank = process_tt_rank(left_rank, tensor.shape, trim=True)
        left_drm = ltype(left_rank, transpose=False, shape=tensor.shape, seed=seed)
    elif left_drm.rank != left_rank:
        raise ValueError(f"Left rank {left_rank} does not match the rank of the DRM {left_drm.rank}.")
    if right_drm is None:
        right_rank = process_tt_rank(right_rank, tensor.shape, trim=False)
        right_drm = rtype(right_rank, transpose=True, shape=tensor.shape, seed=_right_seed(seed, d))
    elif tuple(right_drm.rank[::-1]) != right_rank:
        raise ValueError(
            f"Right rank {right_rank} does not match the rank of the DRM {right_drm.rank}.")
    return left_drm, right_drm


def hmt_sketch_batch(tensors: Sequence[Tensor], rank: TTRank, seed: Optional[int] = None,
                     drm_type: Optional[Type[DRM]] = None, drm: Optional[DRM] = None, return_drm: bool = False):
    """``[hmt_sketch(t, ...) for t in tensors]`` with ONE DRM; batched as ``orthogonal_sketch_batch``."""
    tensors = list(tensors)
    if not tensors:
        return ([], drm) if return_drm else []
    if seed is None:
        seed = _fresh_seed()
    if drm is None:
        drm_type = TensorTrainDRM if drm_type is None else drm_type
        rank = process_tt_rank(rank, tensors[0].shape, trim=True)
        drm = drm_type(rank, transpose=True, shape=tensors[0].shape, seed=seed)
    elif tuple(drm.rank[::-1]) != rank:
        raise ValueError(f"Right rank {rank} does not match the rank of the DRM {drm.rank}.")
    rrank = tuple(drm.rank[::-1])
    one = lambda t: hmt_sketch(t, rrank, drm=drm)
    out = _orth_batch(tensors, None, drm, SketchMethod.hmt, one)
    return (out, drm) if return_drm else out


class SketchedTensorTrain(Tensor):
    """Result of ``stream_sketch``: the sketch plus the DRMs that produced it.  Cheap to turn
    into a TT (``to_tt``) and cheap to update with further summands (``+``)
    (reference sketch.py:232-361)."""

    def __init__(self, sketch_: SketchContainer, left_drm: DRM, right_drm: DRM) -> None:
        self.sketch_ = sketch_
        self.left_drm = left_drm
        self.right_drm = right_drm
        self.shape = sketch_.shape

    left_rank = property(lambda self: self.left_drm.rank)
    right_rank = property(lambda self: self.right_drm.rank[::-1])
    Psi_cores = property(lambda self: self.sketch_.Psi_cores)
    Omega_mats = property(lambda self: self.sketch_.Omega_mats)

    @property
    def size(self) -> int:
        return int(sum(a.size for a in self.Psi_cores) + sum(a.size for a in self.Omega_mats))

    def C_cores(self, direction="auto") -> ArrayList:
        return assemble_sketched_tt(self.sketch_, direction=direction)

    @property
    def T(self) -> "SketchedTensorTrain":
        return SketchedTensorTrain(self.sketch_.T, self.right_drm.T, self.left_drm.T)

    def to_tt(self) -> TensorTrain:
        # device-resident cores: sketching or contracting the result again needs no PCIe round trip;
        # every host-side TensorTrain method copies what it needs (tensor._host)
        return TensorTrain(assemble_sketched_tt(self.sketch_, device=True))

    def to_numpy(self) -> npt.NDArray[np.float64]:
        return self.to_tt().to_numpy()

    def __repr__(self) -> str:
        return (f"<Sketched tensor train of shape {self.shape} with left-rank {self.left_rank} and "
                f"right-rank {self.right_rank} at {hex(id(self))}>")

    def __add__(self, other: Tensor) -> "SketchedTensorTrain":
        """Streaming update: sketch ``other`` with the same DRMs and add (reference :292-301)."""
        extra = stream_sketch(other, self.left_rank, self.right_rank, left_drm=self.left_drm,
                              right_drm=self.right_drm)
        return SketchedTensorTrain(self.sketch_ + extra.sketch_, self.left_drm, self.right_drm)

    def increase_rank(self, tensor: Tensor, new_left_rank: TTRank,
                      new_right_rank: TTRank) -> "SketchedTensorTrain":
        """Grow the sketch ranks by sketching only the new blocks (reference :303-353)."""
        new_left_rank = process_tt_rank(new_left_rank, tensor.shape, trim=False)
        new_right_rank = process_tt_rank(new_right_rank, tensor.shape, trim=False)
        for drm in (self.left_drm, self.right_drm):
            if not isinstance(drm, CanSlice):
                raise ValueError(f"Increasing rank is not supported for DRM {type(drm).__name__}")
        zeros = (0,) * (len(tensor.shape) - 1)
        left_slices = [zeros, self.left_drm.rank, new_left_rank]
        right_slices = [zeros, self.right_drm.rank[::-1], new_right_rank]
        left_drm = self.left_drm.increase_rank(new_left_rank)      # type: ignore[attr-defined]
        right_drm = self.right_drm.increase_rank(new_right_rank)   # type: ignore[attr-defined]
        blocks = _blocked_stream_sketch_components(tensor, left_drm, right_drm, left_slices,
                                                   right_slices, excluded_entries=[(0, 0)])
        blocks[(0, 0)] = self.sketch_
        sketch = _assemble_blocked_stream_sketches(left_slices, right_slices, tensor.shape, blocks)
        return SketchedTensorTrain(sketch, left_drm, right_drm)

    def __mul__(self, other: float) -> "SketchedTensorTrain":
        return SketchedTensorTrain(self.sketch_ * other, self.left_drm, self.right_drm)

    def dot(self, other: Tensor, reverse=False) -> float:
        return self.to_tt().dot(other, reverse)


def _blocked_stream_sketch_components(tensor, left_drm, right_drm, left_rank_slices,
                                      right_rank_slices, excluded_entries=None) -> BlockedSketch:
    """One streaming sketch per (left block, right block) of DRM rank slices (reference :364-397)."""
    skip = set(excluded_entries or [])
    lefts = [left_drm.slice(a, b) for a, b in zip(left_rank_slices[:-1], left_rank_slices[1:])]
    rights = [right_drm.slice(a, b) for a, b in zip(right_rank_slices[:-1], right_rank_slices[1:])]
    out: BlockedSketch = {}
    for i, ldrm in enumerate(lefts):
        for j, rdrm in enumerate(rights):
            if (i, j) not in skip:
                out[(i, j)] = general_sketch(tensor, ldrm, rdrm, method=SketchMethod.streaming)
    return out


def nat_streams() -> int:
    from . import _native
    return _native.NUM_STREAMS


def _assemble_one_call(Psi, Om, direction):
    """The d - 1 (pseudo-inverse, product) pairs through ``ttsk_tt_assemble``: one library call that deals them over the
    library's streams and joins them, instead of 3 (d - 1) calls from here.  None if an operand is not a plain array."""
    import ctypes
    from . import _native as nat
    d = len(Psi)
    if d < 2 or len(Om) != d - 1 or not all(isinstance(a, DevArray) for a in list(Psi) + list(Om)):
        return None
    if os.environ.get("TTSK_ASSEMBLE_ONE_CALL", "1") == "0":
        return None
    if any(s != 0 and s in nat._dirty and nat._joined_into.get(s) != 0 for s in range(nat.NUM_STREAMS)):
        sync()                                    # operands may still be in flight on another stream
    Psi = [p.contiguous() for p in Psi]
    Om = [o.contiguous() for o in Om]
    lr, rr = [int(o.shape[0]) for o in Om], [int(o.shape[1]) for o in Om]
    n = [int(p.shape[1]) for p in Psi]
    for mu, p in enumerate(Psi):
        want = (1 if mu == 0 else lr[mu - 1], n[mu], 1 if mu == d - 1 else rr[mu])
        if tuple(p.shape) != want:
            return None
    # every output and the pseudo-inverses in ONE allocation (a dozen pool round trips cost the host more than the device idles
    # for); pieces of equal shape are equally spaced
    from .tt_fused import _carve
    if direction == "right":
        shapes = [(1 if mu == 0 else lr[mu - 1], n[mu], lr[mu]) for mu in range(d - 1)]
    else:
        shapes = [(rr[mu - 1], n[mu], 1 if mu == d - 1 else rr[mu]) for mu in range(1, d)]
    arrs = _carve(shapes + [(rr[k], lr[k]) for k in range(d - 1)])
    cores = arrs[:d - 1] + [Psi[-1]] if direction == "right" else [Psi[0]] + arrs[:d - 1]
    work = arrs[d - 1:]
    I64, P = ctypes.c_int64, ctypes.c_void_p
    nat.call("ttsk_tt_assemble", d, (I64 * d)(*n), (I64 * (d - 1))(*lr), (I64 * (d - 1))(*rr), (P * d)(*[p.ptr for p in Psi]),
             (P * (d - 1))(*[o.ptr for o in Om]), (P * d)(*[c.ptr for c in cores]), (P * (d - 1))(*[w.ptr for w in work]),
             0 if direction == "right" else 1, 0)
    nat.call("ttsk_sync", 0)                      # operands and the pseudo-inverses are released after this
    return cores


def assemble_sketched_tt(sketch: SketchContainer, direction="auto", device: bool = False) -> ArrayList:
    """TT cores from a streaming sketch: C_mu = Psi_mu pinv(Omega_mu) ("right") or
    pinv(Omega_{mu-1}) Psi_mu ("left") (reference sketch.py:400-443)."""
    if direction == "auto":
        bigger = np.all(np.array(sketch.left_rank) > np.array(sketch.right_rank))
        direction = "left" if bigger else "right"
    # device-resident: the d-1 (pinv, product) pairs are independent -> one library stream each, so
    # that the one-workgroup Jacobi kernels (~2 ms at rank 50 x 100) run side by side
    Psi, Om = sketch.device_arrays()
    if direction not in ("right", "left"):
        raise ValueError(f"Unknown direction {direction}")
    one = _assemble_one_call(Psi, Om, direction)
    if one is not None:
        return one if device else [np.asarray(to_host(C)) for C in one]
    nstreams = max(1, min(len(Om), nat_streams()))
    sync()
    pending, keep = [], []          # `keep`: operands stay allocated until the streams have drained
    pinvs = pinv_dev_many(Om, streams=range(nstreams))      # pinv k on stream k % nstreams, verdicts read afterwards
    if direction == "right":
        for k, (P, O) in enumerate(zip(Psi[:-1], Om)):
            r1, n, r2 = P.shape
            st = k % nstreams
            Pc, Oi = P.contiguous(st), pinvs[k]
            keep += [Pc, Oi]
            M = Pc.reshape(r1 * n, r2)
            Oc = as_dev(O, st).contiguous(st)
            keep.append(Oc)
            pending.append(refine_right(contract("ij,jk->ik", M, Oi, stream=st), M, Oc, Oi, stream=st).reshape(r1, n, O.shape[0]))
        pending.append(Psi[-1])
    elif direction == "left":
        pending.append(Psi[0])
        for k, (P, O) in enumerate(zip(Psi[1:], Om)):
            r1, n, r2 = P.shape
            st = k % nstreams
            Pc, Oi = P.contiguous(st), pinvs[k]
            keep += [Pc, Oi]
            M = Pc.reshape(r1, n * r2)
            Oc = as_dev(O, st).contiguous(st)
            keep.append(Oc)
            pending.append(refine_left(contract("ij,jk->ik", Oi, M, stream=st), Oc, M, Oi, stream=st).reshape(O.shape[1], n, r2))
    else:
        raise ValueError(f"Unknown direction {direction}")
    sync()
    if device:          # cores stay on the device (TensorTrain copies them to the host on demand)
        return [C if isinstance(C, DevArray) else as_dev(C) for C in pending]
    return [np.asarray(to_host(C)) for C in pending]


def _assemble_blocked_stream_sketches(left_rank_slices, right_rank_slices, shape,
                                      sketch_dict: BlockedSketch) -> SketchContainer:
    """Place every block at its rank offsets (reference :446-473)."""
    out = SketchContainer.zero(shape, tuple(left_rank_slices[-1]), tuple(right_rank_slices[-1]))
    for (i, j), blk in sketch_dict.items():
        l0, l1 = (0,) + tuple(left_rank_slices[i]), (1,) + tuple(left_rank_slices[i + 1])
        r0, r1 = tuple(right_rank_slices[j]) + (0,), tuple(right_rank_slices[j + 1]) + (1,)
        for mu, P in enumerate(blk.Psi_cores):
            out.Psi_cores[mu][l0[mu]:l1[mu], :, r0[mu]:r1[mu]] = P
        for mu, O in enumerate(blk.Omega_mats):
            out.Omega_mats[mu][l0[mu + 1]:l1[mu + 1], r0[mu]:r1[mu]] = O
    return out


def get_drm_capabilities():
    """Which capability mixins each shipped DRM has (reference :476-490)."""
    caps = (CanSlice, CanIncreaseRank, CansketchSparse, CansketchDense, CansketchTT)
    return {drm.__name__: {c.__name__: issubclass(drm, c) for c in caps} for drm in ALL_DRM}


def blocked_stream_sketch(tensor: Tensor, left_drm: CanSlice, right_drm: CanSlice,
                          left_rank_slices: List[Tuple[int, ...]],
                          right_rank_slices: List[Tuple[int, ...]]) -> SketchContainer:
    """Sketch block by block over DRM rank slices and assemble (reference :493-525)."""
    for drm in (left_drm, right_drm):
        if not isinstance(drm, CanSlice):
            raise ValueError(f"Blocked sketch not supported for DRM {type(drm).__name__}")
    blocks = _blocked_stream_sketch_components(tensor, left_drm, right_drm, left_rank_slices,
                                               right_rank_slices)
    return _assemble_blocked_stream_sketches(left_rank_slices, right_rank_slices, tensor.shape, blocks)
