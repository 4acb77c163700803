"""Fast path: TT (or sum-of-TT) input with tensor-train DRMs on both sides, streaming method.

One C call (``ttsk_tt_sketch``) runs both DRM chains, every Omega and every Psi on the device
and leaves the sketch in ONE packed buffer ``[Psi_0 .. Psi_{d-1}, Omega_0 .. Omega_{d-2}]`` --
the layout the multi-GPU partial-sketch sum reduces.  Numerically this is the same sequence of
contractions as ``TensorTrainDRM.sketch_tt`` + ``sketch_omega_tt`` / ``sketch_psi_tt``
(reference tensor_train_drm.py:71-88, tensor_train_sketch.py:8-35), minus the Python round trips.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import numpy as np

from . import _native as nat
from .device import DevArray
from .drm.tensor_train_drm import TensorTrainDRM
from .tensor import TensorSum, TensorTrain

_I64 = ctypes.c_int64
MAX_BATCH = 32   # tensors per batched pass (SK_MAXB in csrc/skinny.h); the library slices larger batches itself


class TTSketchPlan:
    """Argument block of ``ttsk_tt_sketch`` for fixed DRMs and a fixed input signature
    (mode sizes and TT ranks); ``run`` sketches any TT with that signature."""

    def __init__(self, shape, tt_rank, left_drm: TensorTrainDRM, right_drm: TensorTrainDRM):
        d = len(shape)
        self.d = d
        self.shape = tuple(int(x) for x in shape)
        self.tt_rank = tuple(int(x) for x in tt_rank)
        arr = lambda v: (_I64 * len(v))(*[int(x) for x in v])
        self.n = arr(self.shape)
        self.s = arr((1,) + self.tt_rank + (1,))
        self.lt = arr((1,) + tuple(left_drm.true_rank))
        self.rt = arr((1,) + tuple(right_drm.true_rank))
        self.l_lo, self.l_hi = arr(left_drm.rank_min), arr(left_drm.rank_max)
        self.r_lo, self.r_hi = arr(right_drm.rank_min), arr(right_drm.rank_max)
        self._keep = (left_drm.dev_cores(), right_drm.dev_cores())
        for k, c in enumerate(self._keep[0]):
            want = (self.lt[k], self.shape[k], self.lt[k + 1])
            if tuple(c.shape) != want:
                raise ValueError(f"left DRM core {k} has shape {c.shape}, expected {want}")
        for k, c in enumerate(self._keep[1]):
            want = (self.rt[k], self.shape[d - 1 - k], self.rt[k + 1])
            if tuple(c.shape) != want:
                raise ValueError(f"right DRM core {k} has shape {c.shape}, expected {want}")
        P = ctypes.c_void_p
        self.DL = (P * (d - 1))(*[c.contiguous().ptr for c in self._keep[0]])
        self.DR = (P * (d - 1))(*[c.contiguous().ptr for c in self._keep[1]])
        self.left_rank = tuple(left_drm.rank)
        self.right_rank = tuple(right_drm.rank[::-1])
        self.size = int(nat.lib().ttsk_tt_sketch_size(d, self.n, self.l_lo, self.l_hi, self.r_lo, self.r_hi))
        self._layout = None

    def new_buffer(self) -> DevArray:
        return DevArray.empty((self.size,))

    def core_pointers(self, tt: TensorTrain):
        cores = tt.dev_cores()
        if tuple(tt.shape) != self.shape or tuple(tt.rank) != self.tt_rank:
            raise ValueError(f"TT of shape {tt.shape} / rank {tt.rank} does not fit the plan")
        keep = [c.contiguous() for c in cores]
        return (ctypes.c_void_p * self.d)(*[c.ptr for c in keep]), keep

    def run(self, X_ptrs, out: DevArray, accumulate: bool = False, stream: int = 0):
        nat.call("ttsk_tt_sketch", self.d, self.n, self.s, self.lt, self.l_lo, self.l_hi, self.rt,
                 self.r_lo, self.r_hi, X_ptrs, self.DL, self.DR, ctypes.c_void_p(out.ptr),
                 1 if accumulate else 0, stream)

    def run_batch(self, X_ptrs, nb: int, out: DevArray, out_stride: int, accumulate: bool = False, stream: int = 0):
        """``nb`` tensors of the plan's signature in one pass: ``X_ptrs`` holds nb * d core pointers
        (tensor-major), sketch ``b`` lands at ``out[b * out_stride:]``."""
        nat.call("ttsk_tt_sketch_batch", nb, self.d, self.n, self.s, self.lt, self.l_lo, self.l_hi, self.rt,
                 self.r_lo, self.r_hi, X_ptrs, self.DL, self.DR, ctypes.c_void_p(out.ptr),
                 _I64(out_stride), 1 if accumulate else 0, stream)

    def run_sum(self, X_ptrs, nb: int, out: DevArray, accumulate: bool = False, stream: int = 0):
        """The sketch of the SUM of ``nb`` tensors of the plan's signature as one packed sketch at ``out``
        (``ttsk_tt_sketch_sum``: chains per tensor, Psi / Omega contracted over (tensor, rank) at once)."""
        nat.call("ttsk_tt_sketch_sum", nb, self.d, self.n, self.s, self.lt, self.l_lo, self.l_hi, self.rt,
                 self.r_lo, self.r_hi, X_ptrs, self.DL, self.DR, ctypes.c_void_p(out.ptr),
                 1 if accumulate else 0, stream)

    def views(self, out: DevArray) -> Tuple[List[DevArray], List[DevArray]]:
        """Psi / Omega arrays as views into the packed (contiguous) buffer."""
        d = self.d
        if self._layout is None:
            lr, rr = (1,) + self.left_rank, self.right_rank + (1,)
            shapes = [(lr[mu], self.shape[mu], rr[mu]) for mu in range(d)] + \
                     [(self.left_rank[mu], self.right_rank[mu]) for mu in range(d - 1)]
            off, lay = 0, []
            for shp in shapes:
                st, acc = [], 1
                for n in reversed(shp):
                    st.append(acc)
                    acc *= n
                lay.append((off, shp, tuple(reversed(st))))
                off += acc
            self._layout = lay
        if out.strides != (1,):
            out = out.contiguous()
        arrs = [DevArray(out.buf, out.offset + off, shp, st) for off, shp, st in self._layout]
        return arrs[:d], arrs[d:]


def try_stream_sketch(tensor, left_drm, right_drm, method) -> Optional[Tuple[list, list]]:
    """(Psi, Omega) device arrays through the one-call path, or None if it does not apply."""
    from .sketch_dispatch import SketchMethod
    if method != SketchMethod.streaming:
        return None
    if type(left_drm) is not TensorTrainDRM or type(right_drm) is not TensorTrainDRM:
        return None
    if left_drm.transpose or not right_drm.transpose:
        return None
    terms = tensor.tensors if isinstance(tensor, TensorSum) else [tensor]
    if not terms or not all(type(t) is TensorTrain for t in terms):
        return None
    d = len(tensor.shape)
    if d < 2 or len(left_drm.cores) != d - 1 or len(right_drm.cores) != d - 1:
        return None
    if tuple(left_drm.shape) != tuple(tensor.shape) or tuple(right_drm.shape) != tuple(tensor.shape):
        raise ValueError(f"Shape {left_drm.shape} of DRM doesn't match tensor's shape {tensor.shape}")
    # Terms of one signature (mode sizes, TT ranks) go through the device together: every chain product is one
    # launch over all of them and Psi / Omega come out already summed (ttsk_tt_sketch_sum) -- the TensorSum loop
    # of sketch_dispatch.py:85-139 without its per-term sketches.
    groups = {}
    for tt in terms:
        groups.setdefault((tuple(tt.shape), tuple(tt.rank)), []).append(tt)
    out, plan0, first = None, None, True
    for (shape, rank), tts in groups.items():
        plan = TTSketchPlan(shape, rank, left_drm, right_drm)
        plan0 = plan0 or plan
        if out is None:
            out = DevArray.empty((plan.size,))
        keep, flat = [], []
        for tt in tts:
            ptrs, k = plan.core_pointers(tt)
            keep.append(k)
            flat += [ptrs[i] for i in range(plan.d)]
        plan.run_sum((ctypes.c_void_p * len(flat))(*flat), len(tts), out, accumulate=not first)
        first = False
    return plan0.views(out)


def _carve(shapes) -> List[DevArray]:
    """contiguous arrays of the given shapes as views of one allocation (16-byte aligned pieces)"""
    sizes = [int(np.prod(sh)) for sh in shapes]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += n + (n & 1)
    buf = DevArray.empty((max(tot, 1),))
    out = []
    for sh, off in zip(shapes, offs):
        st, acc = [], 1
        for n in reversed(sh):
            st.append(acc)
            acc *= int(n)
        out.append(DevArray(buf.buf, off, sh, tuple(reversed(st))))
    return out


def try_orth_sketch(tensor, left_drm, right_drm, method) -> Optional[Tuple[list, list]]:
    """``orthogonal`` / ``hmt`` sketch of ONE tensor train with tensor-train DRMs through ``ttsk_tt_orth_sketch``:
    (cores, Omega) as device arrays, or None if the one-call path does not apply.  The verdicts of its fast
    factorisations are deferred -- the caller reads ``ttsk_deferred_status`` (reference sketch_dispatch.py:160-193)."""
    from .sketch_dispatch import SketchMethod
    if method not in (SketchMethod.orthogonal, SketchMethod.hmt) or type(tensor) is not TensorTrain:
        return None
    orth = method == SketchMethod.orthogonal
    if type(right_drm) is not TensorTrainDRM or not right_drm.transpose:
        return None
    if orth and (type(left_drm) is not TensorTrainDRM or left_drm.transpose):
        return None
    d = len(tensor.shape)
    drms = [right_drm] + ([left_drm] if orth else [])
    if d < 2 or any(len(m.cores) != d - 1 or tuple(m.shape) != tuple(tensor.shape) for m in drms):
        return None
    if any(tuple(m.rank_min) != (0,) * (d - 1) or tuple(m.rank_max) != tuple(m.true_rank) for m in drms):
        return None                                   # a rank slice of a blocked sketch: the general path
    arr = lambda v: (_I64 * len(v))(*[int(x) for x in v])
    P = ctypes.c_void_p
    n, s = arr(tensor.shape), arr((1,) + tuple(tensor.rank) + (1,))
    rt = arr((1,) + tuple(right_drm.true_rank))
    keep = [[c.contiguous() for c in tensor.dev_cores()], [c.contiguous() for c in right_drm.dev_cores()]]
    X = (P * d)(*[c.ptr for c in keep[0]])
    DR = (P * (d - 1))(*[c.ptr for c in keep[1]])
    right_rank = tuple(right_drm.rank[::-1])
    if orth:
        lt = arr((1,) + tuple(left_drm.true_rank))
        keep.append([c.contiguous() for c in left_drm.dev_cores()])
        DL = (P * (d - 1))(*[c.ptr for c in keep[2]])
        out_rank = tuple(left_drm.rank)
        om_shapes = [(out_rank[mu], right_rank[mu]) for mu in range(d - 1)]
    else:
        lt, DL, om_shapes = None, None, []
        out_rank = right_rank
    kr = (1,) + out_rank + (1,)
    # every output of the call in ONE allocation (a dozen pool round trips cost the host more than the device idles for)
    arrs = _carve([(kr[mu], tensor.shape[mu], kr[mu + 1]) for mu in range(d)] + om_shapes)
    cores, Omega = arrs[:d], arrs[d:]
    om = (P * (d - 1))(*[o.ptr for o in Omega]) if orth else None
    try:
        nat.call("ttsk_tt_orth_sketch", d, n, s, lt, rt, X, DL, DR, (P * d)(*[c.ptr for c in cores]), om, 0)
    except nat.TtskUnsupported:
        return None
    return cores, Omega


def try_orth_sketch_batch(tensors, left_drm, right_drm, method):
    """``try_orth_sketch`` for tensor trains of ONE signature through ``ttsk_tt_orth_sketch_batch`` (the sketches run as
    concurrent chains on the library's streams): a list of (cores, Omega) per tensor and an int32 device array of verdicts
    (1 = repeat that tensor on the robust path), or None if the batch path does not apply."""
    from .sketch_dispatch import SketchMethod
    if method not in (SketchMethod.orthogonal, SketchMethod.hmt) or not tensors:
        return None
    if any(type(t) is not TensorTrain for t in tensors):
        return None
    first = tensors[0]
    if any(tuple(t.shape) != tuple(first.shape) or tuple(t.rank) != tuple(first.rank) for t in tensors):
        return None
    orth = method == SketchMethod.orthogonal
    if type(right_drm) is not TensorTrainDRM or not right_drm.transpose:
        return None
    if orth and (type(left_drm) is not TensorTrainDRM or left_drm.transpose):
        return None
    d = len(first.shape)
    drms = [right_drm] + ([left_drm] if orth else [])
    if d < 2 or any(len(m.cores) != d - 1 or tuple(m.shape) != tuple(first.shape) for m in drms):
        return None
    if any(tuple(m.rank_min) != (0,) * (d - 1) or tuple(m.rank_max) != tuple(m.true_rank) for m in drms):
        return None
    B = len(tensors)
    arr = lambda v: (_I64 * len(v))(*[int(x) for x in v])
    P = ctypes.c_void_p
    n, s = arr(first.shape), arr((1,) + tuple(first.rank) + (1,))
    rt = arr((1,) + tuple(right_drm.true_rank))
    keep = [[c.contiguous() for t in tensors for c in t.dev_cores()], [c.contiguous() for c in right_drm.dev_cores()]]
    X = (P * (B * d))(*[c.ptr for c in keep[0]])
    DR = (P * (d - 1))(*[c.ptr for c in keep[1]])
    right_rank = tuple(right_drm.rank[::-1])
    if orth:
        lt = arr((1,) + tuple(left_drm.true_rank))
        keep.append([c.contiguous() for c in left_drm.dev_cores()])
        DL = (P * (d - 1))(*[c.ptr for c in keep[2]])
        out_rank = tuple(left_drm.rank)
        om_shapes = [(out_rank[mu], right_rank[mu]) for mu in range(d - 1)]
    else:
        lt, DL, om_shapes = None, None, []
        out_rank = right_rank
    kr = (1,) + out_rank + (1,)
    per = [(kr[mu], first.shape[mu], kr[mu + 1]) for mu in range(d)] + om_shapes
    arrs = _carve(per * B)
    outs = [(arrs[b * len(per):b * len(per) + d], arrs[b * len(per) + d:(b + 1) * len(per)]) for b in range(B)]
    cores = (P * (B * d))(*[c.ptr for o in outs for c in o[0]])
    om = (P * (B * (d - 1)))(*[c.ptr for o in outs for c in o[1]]) if orth else None
    status = DevArray.zeros(((B + 1) // 2,), dtype=np.int64)         # B int32 verdicts
    try:
        nat.call("ttsk_tt_orth_sketch_batch", B, d, n, s, lt, rt, X, DL, DR, cores, om, P(status.ptr), 0)
    except nat.TtskUnsupported:
        return None
    return outs, status
