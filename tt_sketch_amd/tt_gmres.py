"""TT-GMRES on sums of tensor trains with sketched rounding -- the caller of the streaming sketch
(reference ``tt_gmres.py``).  Every Krylov vector, operator product and rounded
sum stays in HBM: an ``MPO`` applied to a train is one batched device product per core, the sum of
trains goes through the batched sketch pass (``tt_fused``), ``to_tt`` assembles on the device, and
the exact roundings use ``TensorTrain.round_dev``.  Only scalars (inner products, norms, the small
Hessenberg least-squares problem) come back to the host.

Interface as the reference's: ``TTLinearMap``, ``MPO``, ``TTPrecond``, ``TTLinearMapSum``,
``round_tt_sum``, ``tt_sum_gmres`` with the same arguments, history keys and error conditions.
"""
from __future__ import annotations

import abc
import logging
from collections import defaultdict
from math import ceil
from time import perf_counter
from typing import Any, Dict, List, Literal, Optional, Tuple, Union

import numpy as np
import numpy.typing as npt
import scipy.linalg

from .device import DevArray, as_dev, axpby, contract
from .sketch import orthogonal_sketch, stream_sketch
from .tensor import Tensor, TensorSum, TensorTrain, _host
from .utils import ArrayList, TTRank, process_tt_rank


class TTLinearMap(abc.ABC):
    """Linear map between tensor-train spaces (reference tt_gmres.py:30-38)."""

    in_shape: Tuple[int, ...]
    out_shape: Tuple[int, ...]

    @abc.abstractmethod
    def __call__(self, other: TensorTrain) -> TensorTrain:
        ...


class MPO(Tensor, TTLinearMap):
    """Matrix product operator, cores ``(rank[mu-1], in_shape[mu], out_shape[mu], rank[mu])``
    (reference tt_gmres.py:41-136)."""

    def __init__(self, cores: ArrayList) -> None:
        self.cores = cores
        self.in_shape = tuple(int(C.shape[1]) for C in cores)
        self.out_shape = tuple(int(C.shape[2]) for C in cores)
        self.rank = tuple(int(C.shape[0]) for C in cores[1:])
        self.shape = tuple(a * b for a, b in zip(self.in_shape, self.out_shape))
        self._dev = None
        self._dev_key = None

    def dev_cores(self) -> List[DevArray]:
        key = self._dev_key
        if self._dev is None or key is None or len(key) != len(self.cores) or any(a is not b for a, b in zip(key, self.cores)):
            self._dev = [as_dev(c).contiguous() for c in self.cores]
            self._dev_key = tuple(self.cores)
        return self._dev

    prepare_device = dev_cores

    @property
    def size(self) -> int:
        return int(sum(C.size for C in self.cores))

    @property
    def T(self) -> "MPO":
        """Transpose of the linear map (in <-> out per mode), not the mode reversal of other tensors
        (reference tt_gmres.py:65-70)."""
        return self.__class__([C.transpose(0, 2, 1, 3) if isinstance(C, DevArray)
                               else np.transpose(C, (0, 2, 1, 3)) for C in self.cores])

    def to_tt(self) -> TensorTrain:
        return TensorTrain([_host(C).reshape(C.shape[0], C.shape[1] * C.shape[2], C.shape[3])
                            for C in self.cores])

    def to_numpy(self) -> npt.NDArray:
        """Dense ``(in_0, out_0, ..., in_{d-1}, out_{d-1})`` array (reference tt_gmres.py:79-88)."""
        acc = _host(self.cores[0])
        acc = acc.reshape(acc.shape[1:])
        for C in self.cores[1:]:
            acc = np.tensordot(acc, _host(C), axes=(acc.ndim - 1, 0))
        return acc.reshape(acc.shape[:-1])

    def __call__(self, other: TensorTrain) -> TensorTrain:
        """Core-wise product ``sum_j M[i,j,k,l] C[a,j,b] -> (i a, k, l b)`` (reference
        tt_gmres.py:90-101), on the device: per operator-rank index ``i`` one product batched over
        the train's left rank ``a``, written straight into its place of the new core."""
        if tuple(other.shape) != self.in_shape:
            raise ValueError(f"MPO maps shape {self.in_shape}, got a tensor of shape {other.shape}")
        cores = []
        for M, C in zip(self.dev_cores(), other.dev_cores()):
            ri, _, no, rl = M.shape
            ra, _, rb = C.shape
            new = DevArray.empty((ri, ra, no, rl, rb))
            for i in range(ri):
                contract("jkl,ajb->aklb", M[i], C, out=new[i])
            cores.append(new.reshape(ri * ra, no, rl * rb))
        return TensorTrain(cores)

    @classmethod
    def random(cls, rank: TTRank, in_shape: Tuple[int, ...], out_shape: Tuple[int, ...]) -> "MPO":
        """Random operator with symmetrised cores of norm sqrt(n_in n_out) each (reference
        tt_gmres.py:103-121; legacy global NumPy generator as there)."""
        both = tuple(a * b for a, b in zip(in_shape, out_shape))
        rk = (1,) + tuple(process_tt_rank(rank, both, trim=True)) + (1,)
        cores = []
        for k, (a, b) in enumerate(zip(in_shape, out_shape)):
            C = np.random.normal(size=(rk[k], a, b, rk[k + 1]))
            C += C.transpose(0, 2, 1, 3).reshape(C.shape)
            cores.append(C * (np.sqrt(a * b) / np.linalg.norm(C)))
        return cls(cores)

    @classmethod
    def eye(cls, shape) -> "MPO":
        return cls([np.eye(n).reshape(1, n, n, 1) for n in shape])

    def __mul__(self, other: float) -> "MPO":
        cores = list(self.cores)
        cores[0] = _host(cores[0]) * other
        return self.__class__(cores)


class TTPrecond(TTLinearMap):
    """Multiplies one mode by the inverse of a matrix, through its QR factors (reference
    tt_gmres.py:137-168).  ``Q^T`` and ``R^{-1}`` are uploaded once; applying them is two small
    device products on the one affected core."""

    def __init__(self, A, shape, mode=0):
        self.A = np.asarray(A, dtype=np.float64)
        self.Q, self.R = np.linalg.qr(self.A)
        self.mode = mode
        self.in_shape = shape
        self.out_shape = shape
        self._dev = None

    def _factors(self):
        if self._dev is None:
            Rinv = scipy.linalg.solve_triangular(self.R, np.eye(self.R.shape[0]))
            self._dev = (as_dev(self.A), as_dev(np.ascontiguousarray(self.Q.T)), as_dev(Rinv))
        return self._dev

    def _apply(self, other: TensorTrain, mats) -> TensorTrain:
        cores = list(other.dev_cores())
        C = cores[self.mode]
        for mat in mats:
            C = contract("kj,ajb->akb", mat, C)
        cores[self.mode] = C
        return TensorTrain(cores)

    def backward_call(self, other: TensorTrain) -> TensorTrain:
        _, Qt, Rinv = self._factors()
        return self._apply(other, (Qt, Rinv))

    def forward_call(self, other: TensorTrain) -> TensorTrain:
        return self._apply(other, (self._factors()[0],))

    __call__ = backward_call


class TTLinearMapSum:
    """A list of ``TTLinearMap`` applied to a train (or to each term of a sum of trains); the result
    is the lazy sum of all products (reference tt_gmres.py:171-207)."""

    def __init__(self, linear_maps: List[TTLinearMap]) -> None:
        if len(linear_maps) == 0:
            raise ValueError("linear_maps cannot be empty")
        self.linear_maps = linear_maps
        self.in_shape = linear_maps[0].in_shape
        self.out_shape = linear_maps[0].out_shape
        for lm in linear_maps[1:]:
            if lm.in_shape != self.in_shape:
                raise ValueError("in_shape mismatch")
            if lm.out_shape != self.out_shape:
                raise ValueError("out_shape mismatch")

    def __call__(self, input_tensor: Union[TensorTrain, TensorSum]) -> TensorSum:
        terms = [input_tensor] if isinstance(input_tensor, TensorTrain) else input_tensor.tensors
        return TensorSum([lm(t) for lm in self.linear_maps for t in terms])


ROUNDING_MODE = Literal["exact", "pairwise", "sketch", "orth_sketch", None]


def round_tt_sum(tt_sum: TensorSum, max_rank: TTRank, eps: Optional[float] = None,
                 method: ROUNDING_MODE = "sketch", oversample_factor: float = 2) -> TensorTrain:
    """Round a sum of trains to ``max_rank`` (reference tt_gmres.py:258-305): ``"exact"`` = direct
    sum then TT-SVD, ``"pairwise"`` = add and round term by term, ``"sketch"`` = streaming sketch
    with right rank ``ceil(oversample_factor * left)``, ``"orth_sketch"`` = orthogonal sketch,
    ``None`` = no rounding.  The two SVD variants run ``add`` / ``round_dev`` on the device."""
    if method == "exact":
        terms = [t.to_device() for t in tt_sum.tensors]
        tt = terms[0]
        for t in terms[1:]:
            tt = tt.add(t)
        return tt.round_dev(eps, max_rank)
    if method == "pairwise":
        terms = [t.to_device() for t in tt_sum.tensors]
        tt = terms[0]
        for t in terms[1:]:
            tt = tt.add(t).round_dev(eps=eps, max_rank=max_rank)
        return tt
    if method in ("sketch", "orth_sketch"):
        left_rank = process_tt_rank(max_rank, tt_sum.shape, trim=True)
        right_rank = tuple(ceil(r * oversample_factor) for r in left_rank)
        if method == "sketch":
            return stream_sketch(tt_sum, left_rank=left_rank, right_rank=right_rank).to_tt()
        return orthogonal_sketch(tt_sum, left_rank=left_rank, right_rank=right_rank)
    if method is None:
        return tt_sum  # type: ignore[return-value]
    raise ValueError(f"Unknown rounding method: {method}")


def tt_sum_gmres(A: TTLinearMapSum, b: TensorTrain, max_rank: TTRank,
                 precond: Optional[TTPrecond] = None, final_round_rank: Optional[TTRank] = None,
                 x0: Optional[TensorTrain] = None, tolerance: float = 1e-6, maxiter: int = 100,
                 symmetric: bool = False, rounding_method: ROUNDING_MODE = "pairwise",
                 rounding_method_final: Optional[ROUNDING_MODE] = None, save_basis: bool = False,
                 verbose: bool = False) -> Tuple[TensorTrain, Dict[str, List]]:
    """GMRES for a ``TTLinearMapSum`` (reference tt_gmres.py:308-432; Dolgov, arXiv:1206.5512, with
    the rounding after each operator product and each Gram-Schmidt step done by ``round_tt_sum``).
    Returns the rounded solution and the history dictionary with the reference's keys."""
    if final_round_rank is None:
        final_round_rank = max_rank
    if rounding_method_final is None:
        rounding_method_final = rounding_method
    if A.out_shape != b.shape:
        raise ValueError("Output shape of linear map doesn't match RHS")
    if x0 is not None and x0.shape != A.in_shape:
        raise ValueError("Input shape of liner map doesn't match initial value")
    if A.out_shape != A.in_shape:
        raise ValueError("TT-GMRES only works for automorphisms")

    max_rank = process_tt_rank(max_rank, A.in_shape, trim=True)
    if x0 is None:
        x0 = TensorTrain.zero(shape=A.in_shape, rank=1)
    b, x0 = b.to_device(), x0.to_device()

    def operator(x: TensorTrain) -> TensorSum:
        out = A(x)
        if precond is not None:
            out = TensorSum([precond(t) for t in out.tensors])
        return out

    # norms of explicit (rounded) trains: the Gram chain is exact to rounding there and saves a QR sweep
    def norm(x: TensorTrain) -> float:
        return x.gram_norm() if x.resident() else x.norm()

    rhs = precond(b) if precond is not None else b
    b_norm = norm(b)
    t_start = perf_counter()
    res = round_tt_sum(rhs - operator(x0), max_rank=max_rank, method=rounding_method)
    res_norm = norm(res)
    beta = res_norm
    basis: List[TensorTrain] = [res / beta]
    H = np.zeros((maxiter + 1, maxiter))

    history: Dict[str, Any] = defaultdict(list)
    history["w_norm"].append(norm(basis[-1]))
    history["rank"].append(res.rank)
    history["residual_norm"].append(res_norm / b_norm)
    history["step_time"].append(perf_counter() - t_start)

    y = np.zeros(0)
    j = -1
    for j in range(maxiter):
        t_step = perf_counter()
        delta = tolerance / (res_norm / beta)
        if verbose:
            logging.info(f"Iteration {j + 1}/{maxiter}, residual norm: {res_norm / b_norm:.4e}")
        w = round_tt_sum(operator(basis[-1]), eps=delta, max_rank=max_rank, method=rounding_method)

        lo = max(0, j - 2) if symmetric else 0
        for i in range(lo, j + 1):
            H[i, j] = w.dot(basis[i])
        # Gram-Schmidt against the (recent) basis, rounded again
        w = round_tt_sum(w - TensorSum(basis[lo:j + 1]) * H[lo:j + 1, j], eps=delta,
                         max_rank=max_rank, method=rounding_method)
        H[j + 1, j] = norm(w)
        basis.append(w / H[j + 1, j])
        history["step_time"].append(perf_counter() - t_step)

        e1 = np.zeros(j + 2)
        e1[0] = beta
        y, (res_norm,), _, _ = np.linalg.lstsq(H[:j + 2, :j + 1], e1, rcond=None)
        history["step_time_with_res_norm"].append(perf_counter() - t_step)
        history["residual_norm"].append(np.sqrt(res_norm) / b_norm)
        history["rank"].append(w.rank)
        history["w_norm"].append(H[j + 1, j])
        history["delta"].append(delta)
        if res_norm / b_norm < tolerance:
            break

    y = y[:j + 1]
    basis = basis[:j + 1]
    t_final = perf_counter()
    result = round_tt_sum(x0 + TensorSum(basis) * y, eps=None, max_rank=final_round_rank,
                          method=rounding_method_final)
    history["final_round_time"] = perf_counter() - t_final
    history["total_time"] = perf_counter() - t_start
    if save_basis:
        history["H_matrix"] = H
        history["nu_list"] = basis
        history["y"] = y
    return result, history
