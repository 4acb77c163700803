"""Host-side rank logic and the pseudo-inverse products of the sketch path.

Counterpart of the on-path parts of the reference's ``tt_sketch/utils.py``:
``process_tt_rank`` / ``trim_ranks`` (:121-175, host integer logic),
``matricize`` / ``dematricize`` (:63-95), ``right_mul_pinv`` / ``left_mul_pinv``
(:98-109, here: device Jacobi-SVD pinv + MFMA GEMM) and ``random_normal``
(:223-227, here: the device counter-based generator).
"""
from __future__ import annotations

import ctypes
from typing import Generator, List, Sequence, Tuple, Union

import numpy as np
import numpy.typing as npt

from . import _native as nat
from .device import DevArray, as_dev, contract

ArrayList = List[npt.NDArray[np.float64]]
ArrayGenerator = Generator[npt.NDArray[np.float64], None, None]
TTRank = Union[int, Tuple[int, ...]]


def trim_ranks(dims: Tuple[int, ...], ranks: Tuple[int, ...]) -> Tuple[int, ...]:
    """Largest TT ranks not exceeding ``ranks`` that a tensor of shape ``dims`` can
    attain (reference utils.py:121-156): bound every bond by the row/column count of
    its unfolding, then relax neighbouring bonds until r_{k+1} <= r_k * n_k and
    r_k <= n_k * r_{k+1} hold everywhere (at most 100 sweeps, as the reference)."""
    d = len(dims)
    left = np.cumprod([int(n) for n in dims], dtype=object)
    right = np.cumprod([int(n) for n in dims[::-1]], dtype=object)[::-1]
    r = [1] + [min(int(ranks[k]), int(left[k]), int(right[k + 1])) for k in range(d - 1)] + [1]
    for _ in range(100):
        dirty = False
        for k, n in enumerate(dims):
            if r[k + 1] > r[k] * n:
                r[k + 1] = r[k] * n
                dirty = True
            if r[k] > n * r[k + 1]:
                r[k] = n * r[k + 1]
                dirty = True
        if not dirty:
            break
    return tuple(int(x) for x in r[1:-1])


def process_tt_rank(rank: TTRank, shape: Tuple[int, ...], trim: bool) -> Tuple[int, ...]:
    """int -> constant tuple, length check (ValueError), optional trimming
    (reference utils.py:159-175)."""
    try:
        out = tuple(rank)  # type: ignore[arg-type]
    except TypeError:
        out = (rank,) * (len(shape) - 1)  # type: ignore[assignment]
    if len(out) != len(shape) - 1:
        raise ValueError(f"TT-rank {out} doesn't have right number of elements")
    return trim_ranks(shape, out) if trim else out


def matricize(A: npt.NDArray, mode: Union[int, Sequence[int]], mat_shape: bool = False):
    """Bring ``mode`` to the front and flatten the rest (reference utils.py:63-83)."""
    lead = (mode,) if isinstance(mode, (int, np.integer)) else tuple(mode)
    rest = tuple(a for a in range(A.ndim) if a not in lead)
    B = np.transpose(A, lead + rest)
    head = B.shape[:len(lead)]
    if mat_shape:
        head = (int(np.prod(head, dtype=np.int64)),)
    return B.reshape(head + (int(np.prod(B.shape[len(lead):], dtype=np.int64)),))


def dematricize(A, mode, shape):
    """Inverse of ``matricize(A, mode)`` for an integer mode (reference utils.py:86-95)."""
    rest = [n for a, n in enumerate(shape) if a != mode]
    B = A.reshape([A.shape[0]] + rest)
    axes = list(range(1, len(shape)))
    axes.insert(mode, 0)
    return np.transpose(B, axes)


def pinv_dev(Omega, rcond=None, stream=0) -> DevArray:
    """pinv(Omega) on the device with LAPACK gelsd's truncation (sigma <= rcond*sigma_max
    dropped, rcond = eps when None) -- the factorisation behind utils.py:98-109."""
    Om = as_dev(Omega, stream).contiguous(stream)
    l, r = Om.shape
    P = DevArray.empty((r, l), stream=stream)
    nat.call("ttsk_pinv", ctypes.c_void_p(Om.ptr), l, r, -1.0 if rcond is None else float(rcond),
             ctypes.c_void_p(P.ptr), None, stream)
    return P


def pinv_dev_many(Omegas, rcond=None, streams=None):
    """pinv of several independent matrices, one library stream each: all fast-path attempts are queued
    before the host waits for the first verdict (``ttsk_pinv_begin`` / ``ttsk_pinv_end``)."""
    streams = list(streams) if streams is not None else list(range(min(len(Omegas), nat.NUM_STREAMS)))
    rc = -1.0 if rcond is None else float(rcond)
    out = [None] * len(Omegas)
    for lo in range(0, len(Omegas), len(streams)):           # one outstanding begin per stream
        group = []
        for k, st in zip(range(lo, min(lo + len(streams), len(Omegas))), streams):
            Om = as_dev(Omegas[k], st).contiguous(st)
            P = DevArray.empty((Om.shape[1], Om.shape[0]), stream=st)
            nat.call("ttsk_pinv_begin", ctypes.c_void_p(Om.ptr), Om.shape[0], Om.shape[1], rc, ctypes.c_void_p(P.ptr), st)
            group.append((k, st, Om, P))
        for k, st, Om, P in group:
            nat.call("ttsk_pinv_end", ctypes.c_void_p(Om.ptr), Om.shape[0], Om.shape[1], rc, ctypes.c_void_p(P.ptr), None, st)
            out[k] = P
    return out


def _like_input(result: DevArray, *inputs):
    return result if any(isinstance(x, DevArray) for x in inputs) else result.get()


def refine_right(C: DevArray, A: DevArray, B: DevArray, P: DevArray, stream=0) -> DevArray:
    """One step of iterative refinement of ``C ~ A @ pinv(B)`` with ``P ~ pinv(B)``: ``C + (A - C B) P``.  The reference
    solves with scipy's lstsq (utils.py:98-102); a product with an explicitly formed pseudo-inverse loses
    kappa(B) eps in directions the least-squares solve keeps clean (TT-GMRES iterates: 3e-9 against 6e-14 on the
    assembled tensor) -- the residual, computed in full precision, brings it back."""
    from .device import axpby
    R = axpby(contract("ij,jk->ik", C, B, stream=stream), A, 1.0, -1.0, stream=stream)          # A - C B
    return axpby(C.copy(stream) if not C.is_contiguous() else C, contract("ij,jk->ik", R, P, stream=stream), 1.0, 1.0, stream=stream)


def refine_left(C: DevArray, A: DevArray, B: DevArray, P: DevArray, stream=0) -> DevArray:
    """the same for ``C ~ pinv(A) @ B``: ``C + P (B - A C)``"""
    from .device import axpby
    R = axpby(contract("ij,jk->ik", A, C, stream=stream), B, 1.0, -1.0, stream=stream)          # B - A C
    return axpby(C.copy(stream) if not C.is_contiguous() else C, contract("ij,jk->ik", P, R, stream=stream), 1.0, 1.0, stream=stream)


def right_mul_pinv(A, B, cond=None):
    """``A @ pinv(B)`` (reference utils.py:98-102)."""
    Ad, Bd = as_dev(A).contiguous(), as_dev(B).contiguous()
    P = pinv_dev(Bd, cond)
    out = refine_right(contract("ij,jk->ik", Ad, P), Ad, Bd, P)
    return _like_input(out, A, B)


def left_mul_pinv(A, B, cond=None):
    """``pinv(A) @ B`` (reference utils.py:105-109)."""
    Ad, Bd = as_dev(A).contiguous(), as_dev(B).contiguous()
    P = pinv_dev(Ad, cond)
    out = refine_left(contract("ij,jk->ik", P, Bd), Ad, Bd, P)
    return _like_input(out, A, B)


def random_normal_dev(shape, seed=None, scale: float = 1.0, stream=0) -> DevArray:
    """N(0, scale^2) array generated on the device (hash -> ndtri, keyed by seed and
    element index).  Stands in for utils.py:178-227, whose stream is host dependent."""
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    out = DevArray.empty(shape, stream=stream)
    nat.call("ttsk_fill_normal", ctypes.c_void_p(out.ptr), ctypes.c_size_t(out.size),
             ctypes.c_uint64(int(seed) % 2**64), float(scale), stream)
    return out


def random_normal_dev_many(shapes, seeds, scales, stream=0):
    """Several ``random_normal_dev`` arrays from ONE launch (``ttsk_fill_normal_many``); array i holds exactly
    what ``random_normal_dev(shapes[i], seeds[i], scales[i])`` would."""
    outs = [DevArray.empty(tuple(sh), stream=stream) for sh in shapes]
    k = len(outs)
    if k:
        nat.call("ttsk_fill_normal_many", k, (ctypes.c_void_p * k)(*[o.ptr for o in outs]),
                 (ctypes.c_size_t * k)(*[o.size for o in outs]), (ctypes.c_uint64 * k)(*[int(s) % 2**64 for s in seeds]),
                 (ctypes.c_double * k)(*[float(c) for c in scales]), stream)
    return outs


def random_normal(shape, seed=None) -> npt.NDArray[np.float64]:
    return random_normal_dev(shape, seed).get()


# ---------------------------------------------------------------------------
# Test-tensor generators and the oblique projector of the reference's utils.py (:20-60, :112-118): what its
# scripts and experiments build their inputs from.  Host data (they ARE the inputs); the projector's
# pseudo-inverse and products run on the device.
def hilbert_tensor(n_dims: int, size: int) -> npt.NDArray:
    """X[i_1, ..., i_d] = 1 / (i_1 + ... + i_d + 1)."""
    total = np.zeros((size,) * n_dims)
    for axis in range(n_dims):
        total = total + np.arange(size).reshape([size if a == axis else 1 for a in range(n_dims)])
    return 1.0 / (total + 1.0)


def sqrt_tensor(shape: Tuple[int, ...], a=-0.2, b=2) -> npt.NDArray:
    """sqrt(|t_1 + ... + t_d|) on the grid t_mu = linspace(a, b, n_mu), scaled to Frobenius norm 1.  As in the
    reference (whose ``np.meshgrid`` default is 'xy' indexing) the result has its first two modes swapped:
    shape (n_2, n_1, n_3, ...)."""
    shape = tuple(shape)
    if len(shape) >= 2:
        shape = (shape[1], shape[0]) + shape[2:]
    d = len(shape)
    total = np.zeros(tuple(shape))
    for axis, n in enumerate(shape):
        total = total + np.linspace(a, b, n).reshape([n if k == axis else 1 for k in range(d)])
    X = np.sqrt(np.abs(total))
    return X / np.linalg.norm(X)


def power_decay_tensor(shape: Tuple[int, ...], pow: float = 2.0, seed=None) -> npt.NDArray:
    """A Gaussian tensor whose mode-mu unfoldings get their singular values, scaled to sigma_1 = 1, multiplied by
    k^(-pow) -- one mode after the other, as in the reference."""
    sub = np.random.SeedSequence(seed).generate_state(1)[0]
    A = random_normal(tuple(shape), seed=int(sub))
    for mode in range(A.ndim):
        M = matricize(A, mode)
        U, S, Vt = np.linalg.svd(M, full_matrices=False)
        S = (S / S[0]) / np.arange(1, S.size + 1) ** pow
        A = dematricize((U * S) @ Vt, mode, A.shape)
    return A


def projector(X: npt.NDArray, Y=None) -> npt.NDArray:
    """The oblique projector P_{X,Y} = X (Y^T X)^+ Y^T (Y = X: the orthogonal projector onto range(X))."""
    Xd = as_dev(X)
    Yd = Xd if Y is None else as_dev(Y)
    core = pinv_dev(contract("ki,kj->ij", Yd, Xd))
    out = contract("ij,kj->ik", contract("ij,jk->ik", Xd, core), Yd)
    return _like_input(out, X, X if Y is None else Y)
