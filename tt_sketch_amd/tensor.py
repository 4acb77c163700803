"""Tensor types of the sketch API (host-side data model + HBM residency).

Mirrors the public surface of the reference's ``tt_sketch/tensor.py`` (classes,
constructor signatures, ``.T``, ``shape``/``rank``/``cores``...), because these are
the input and output types of ``stream_sketch`` / ``orthogonal_sketch`` /
``hmt_sketch``.  What is new here is residency: every tensor lazily uploads its
payload to HBM once (``dev_*`` accessors) and ``.T`` re-uses that upload through
strided views, so a right sketch (drm_base.py:122-145 in the reference transposes
the tensor on every call) costs no copy and no PCIe traffic.

The arithmetic helpers (``error``, ``norm``, ``dot``, ``round``, ...) are off the hot
path and stay plain NumPy on the host.

Residency contract: payload arrays are treated as immutable once a tensor has been
sketched; replace a core (``tt[i] = new``) rather than writing into it, or call
``invalidate_device()``.
"""
from __future__ import annotations

import abc
from functools import cached_property
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import numpy.typing as npt

from .device import DevArray, as_dev, axpby, contract, copy_into
from .utils import ArrayList, TTRank, process_tt_rank, random_normal


def _host(a) -> np.ndarray:
    return a.get() if isinstance(a, DevArray) else np.asarray(a)


def _same_objects(key, items) -> bool:
    """The cached device copies belong to exactly these payload objects (identity, not id(): the id of
    a collected array can be handed to a new one)."""
    return key is not None and len(key) == len(items) and all(a is b for a, b in zip(key, items))


class Tensor(abc.ABC):
    """Base class: shape, transpose, conversion and the lazy-sum arithmetic."""

    shape: Tuple[int, ...]

    @property
    @abc.abstractmethod
    def T(self):
        """Tensor with the order of the modes reversed."""

    @property
    @abc.abstractmethod
    def size(self) -> int:
        """Number of stored floating point values."""

    @abc.abstractmethod
    def to_numpy(self) -> npt.NDArray[np.float64]:
        """Dense ndarray of the same shape."""

    def prepare_device(self) -> None:
        """Upload the payload to HBM (once); views such as ``.T`` then share it."""

    def invalidate_device(self) -> None:
        for name in ("_dev", "_dev_key"):
            if hasattr(self, name):
                setattr(self, name, None)

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def dense(self) -> "DenseTensor":
        return DenseTensor(self.to_numpy())

    # -- error / norms (host NumPy; reference tensor.py:53-88)
    def error(self, other, relative: bool = False, rmse: bool = False, fast: bool = False) -> float:
        if isinstance(other, np.ndarray):
            other = DenseTensor(other)
        ref_norm = other.norm()
        if fast:
            mine = self.norm()
            tot = mine**2 + ref_norm**2
            err = np.sqrt(tot) * np.sqrt(abs(1 - 2 * self.dot(other) / tot))
        else:
            err = np.linalg.norm(self.to_numpy() - other.to_numpy())
        if relative:
            if ref_norm == 0:
                return np.inf
            err /= ref_norm
        if rmse:
            err /= np.sqrt(np.prod(self.shape))
        return err

    def dot(self, other, reverse: bool = False) -> float:
        if isinstance(other, TensorSum):
            return other.dot(self)
        if not reverse:
            return other.dot(self, reverse=True)
        return float(np.dot(self.to_numpy().ravel(), other.to_numpy().ravel()))

    def norm(self) -> float:
        return float(np.sqrt(abs(self.dot(self))))

    def __matmul__(self, other) -> float:
        return self.dot(other)

    # -- lazy sums (reference tensor.py:100-123)
    def __add__(self, other) -> "TensorSum":
        mine = self.tensors if isinstance(self, TensorSum) else [self]
        theirs = other.tensors if isinstance(other, TensorSum) else [other]
        return TensorSum(list(mine) + list(theirs))

    @abc.abstractmethod
    def __mul__(self, other: float):
        """Scalar multiple."""

    def __rmul__(self, other: float):
        return self.__mul__(other)

    def __truediv__(self, other: float):
        return self.__mul__(1 / other)

    def __neg__(self):
        return self * -1

    def __sub__(self, other):
        return self + (-other)


# --------------------------------------------------------------------------- dense
class DenseTensor(Tensor):
    """Full ndarray (reference tensor.py:140-182)."""

    def __init__(self, data, _dev: Optional[DevArray] = None) -> None:
        self.data = data                  # ndarray, or a DevArray for a tensor built on the device
        self.shape = tuple(data.shape)
        self._dev = _dev
        self._dev_src = data if _dev is not None else None

    def dev_data(self) -> DevArray:
        if self._dev is None or self._dev_src is not self.data:    # `.data` was replaced: upload again
            self._dev = as_dev(self.data)
            self._dev_src = self.data
        return self._dev

    prepare_device = dev_data

    @property
    def T(self) -> "DenseTensor":
        axes = tuple(reversed(range(len(self.shape))))
        if isinstance(self.data, DevArray):
            return DenseTensor(self.data.transpose(axes))
        cur = self._dev is not None and self._dev_src is self.data
        return DenseTensor(np.transpose(self.data, axes), self._dev.transpose(axes) if cur else None)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    def to_numpy(self):
        return _host(self.data)

    def to_sparse(self) -> "SparseTensor":
        X = self.to_numpy()
        idx = np.indices(X.shape).reshape(X.ndim, -1)
        return SparseTensor(X.shape, idx, X.reshape(-1))

    def __mul__(self, other: float) -> "DenseTensor":
        return DenseTensor(self.to_numpy() * other)

    @classmethod
    def random(cls, shape: Tuple[int, ...]) -> "DenseTensor":
        return cls(random_normal(shape))

    def __repr__(self) -> str:
        return f"<Dense tensor of shape {self.shape} at {hex(id(self))}>"


# --------------------------------------------------------------------------- sparse
class SparseTensor(Tensor):
    """COO tensor: ``indices`` (d, nnz) int64, ``entries`` (nnz,) (reference tensor.py:185-291).

    On the device the index matrix is uploaded once; ``.T`` only reverses the order in
    which its rows are addressed (``dev_row_order``)."""

    def __init__(self, shape, indices, entries, _dev=None, _order=None) -> None:
        self.shape = tuple(int(n) for n in shape)
        if isinstance(indices, tuple):
            indices = np.stack(indices)
        self.indices = indices
        self.entries = entries
        self._dev = _dev
        self._order = tuple(range(len(self.shape))) if _order is None else tuple(_order)

    def _upload(self):
        if self._dev is None:
            idx = np.ascontiguousarray(np.asarray(self.indices), dtype=np.int64)
            # the device kernels address cores, Psi slices and sample tables by these values: out of range
            # is a memory fault there, so it is an error here (the reference would raise IndexError in its slicing)
            if idx.size and (idx.min() < 0 or (idx.max(axis=1) >= np.asarray(self.shape, dtype=np.int64)).any()):
                raise IndexError(f"SparseTensor: an index lies outside the shape {self.shape}")
            # rows are stored in the order of the *first* upload; later views permute
            inv = np.argsort(self._order)
            self._dev = (DevArray.from_host(idx[inv], dtype=np.int64),
                         DevArray.from_host(np.asarray(self.entries, dtype=np.float64)), {})
        return self._dev

    def prepare_device(self) -> None:
        self._upload()

    def dev_indices(self) -> DevArray:
        """(d, nnz) int64 buffer; logical row i lives at physical row dev_row_order[i]."""
        return self._upload()[0]

    def dev_entries(self) -> DevArray:
        return self._upload()[1]

    def dev_mode_perm(self, mu: int) -> DevArray:
        """Permutation that visits the nonzeros in order of their (logical) mode-``mu`` index; sorted
        once per tensor and mode on the device and shared with ``.T`` views."""
        import ctypes
        from . import _native as nat
        idx = self._upload()[0]
        phys = self._order[mu]
        cache = self._dev[2]
        if phys not in cache:
            perm = DevArray.empty((self.nnz,), dtype=np.int64)
            nat.call("ttsk_sparse_sort_mode", ctypes.c_void_p(idx.ptr + phys * self.nnz * 8),
                     ctypes.c_size_t(self.nnz), int(self.shape[mu]), ctypes.c_void_p(perm.ptr), 0)
            cache[phys] = perm
        return cache[phys]

    @property
    def dev_row_order(self) -> Tuple[int, ...]:
        return self._order

    @property
    def T(self) -> "SparseTensor":
        return SparseTensor(self.shape[::-1], self.indices[::-1], self.entries, self._dev,
                            self._order[::-1])

    @property
    def size(self) -> int:
        return self.nnz * (self.ndim + 1)

    @property
    def nnz(self) -> int:
        return len(self.entries)

    def split(self, n_summands: int) -> "TensorSum":
        """Contiguous nnz shards as a TensorSum (reference tensor.py:215-234)."""
        step = self.nnz // n_summands
        parts: List[Tensor] = []
        for i in range(n_summands):
            hi = (i + 1) * step if i < n_summands - 1 else self.nnz
            sl = slice(i * step, hi)
            parts.append(SparseTensor(self.shape, tuple(row[sl] for row in self.indices),
                                      self.entries[sl]))
        return TensorSum(parts)

    def to_numpy(self):
        X = np.zeros(self.shape)
        X[tuple(self.indices)] = self.entries
        return X

    def norm(self) -> float:
        return float(np.linalg.norm(self.entries))

    def dot(self, other, reverse=False) -> float:
        if hasattr(other, "gather"):
            return float(np.dot(other.gather(self.indices), self.entries))
        return super().dot(other, reverse=reverse)

    @classmethod
    def random(cls, shape, nnz: int, seed: Optional[int] = None) -> "SparseTensor":
        rng = np.random.default_rng(seed)
        flat = rng.choice(int(np.prod(shape)), size=nnz, replace=False)
        return cls(shape, np.stack(np.unravel_index(flat, shape)), rng.standard_normal(nnz))

    def __mul__(self, other: float) -> "SparseTensor":
        return SparseTensor(self.shape, self.indices, self.entries * other)

    def gather(self, indices) -> npt.NDArray[np.float64]:
        keys = np.ravel_multi_index(tuple(indices), self.shape)
        table = self.dict
        return np.array([table.get(int(k), 0.0) for k in keys])

    @cached_property
    def dict(self) -> Dict[int, float]:
        keys = np.ravel_multi_index(tuple(self.indices), self.shape)
        return {int(k): float(v) for k, v in zip(keys, self.entries)}

    def __repr__(self) -> str:
        return (f"<Sparse tensor of shape {self.shape} with {self.nnz} non-zero"
                f" entries at {hex(id(self))}>")


# --------------------------------------------------------------------------- TT
class TensorTrain(Tensor):
    """Tensor train with cores ``(r_{k-1}, n_k, r_k)`` (reference tensor.py:294-609)."""

    def __init__(self, cores: ArrayList, _dev=None) -> None:
        self.cores = cores
        self.shape = tuple(int(C.shape[1]) for C in cores)
        self.rank = tuple(int(C.shape[0]) for C in cores[1:])
        self._dev = _dev
        self._dev_key = None if _dev is None else tuple(cores)      # the objects, compared with `is`

    def dev_cores(self) -> List[DevArray]:
        if self._dev is None or not _same_objects(self._dev_key, self.cores):
            self._dev = [as_dev(c) for c in self.cores]
            self._dev_key = tuple(self.cores)
        return self._dev

    prepare_device = dev_cores

    @property
    def T(self) -> "TensorTrain":
        flip = (2, 1, 0)
        cores = [c.transpose(flip) if isinstance(c, DevArray) else np.transpose(c, flip)
                 for c in self.cores[::-1]]
        dev = None
        if self._dev is not None and _same_objects(self._dev_key, self.cores):
            dev = [c.transpose(flip) for c in self._dev[::-1]]
        return TensorTrain(cores, dev)

    def to_numpy(self):
        acc = _host(self.cores[0])
        acc = acc.reshape(acc.shape[1:])
        for C in self.cores[1:]:
            acc = np.tensordot(acc, _host(C), axes=(acc.ndim - 1, 0))
        return acc.reshape(acc.shape[:-1])

    @classmethod
    def random(cls, shape, rank: TTRank, seed: Optional[int] = None, orthog: bool = False,
               trim: Optional[bool] = None, norm_goal: str = "norm-1") -> "TensorTrain":
        """Gaussian cores; ``norm-1`` scales by 1/sqrt(r1*n), ``norm-preserve`` by 1/sqrt(r1)
        (reference tensor.py:323-378).  Uses a single NumPy Generator stream per core."""
        if trim is None:
            trim = bool(orthog)
        if orthog and not trim:
            raise ValueError("Trimming must be enabled if orthogonalization is enabled.")
        rk = (1,) + tuple(process_tt_rank(rank, shape, trim=trim)) + (1,)
        seeds = np.random.SeedSequence(seed).generate_state(len(shape))
        cores = []
        for k, n in enumerate(shape):
            r1, r2 = rk[k], rk[k + 1]
            M = np.random.default_rng(seeds[k]).standard_normal((r1 * n, r2))
            if orthog and k < len(shape) - 1:
                M, _ = np.linalg.qr(M, mode="reduced")
            elif norm_goal == "norm-1":
                M /= np.sqrt(r1 * n)
            elif norm_goal == "norm-preserve":
                M /= np.sqrt(r1)
            else:
                raise ValueError(f"Unknown norm goal: {norm_goal}")
            cores.append(M.reshape(r1, n, r2))
        return cls(cores)

    @classmethod
    def zero(cls, shape, rank: TTRank) -> "TensorTrain":
        rk = (1,) + process_tt_rank(rank, shape, trim=False) + (1,)
        return cls([np.zeros((rk[k], n, rk[k + 1])) for k, n in enumerate(shape)])

    def partial_dense(self, dir: str = "lr") -> ArrayList:
        """Dense partial products X_0...X_mu as matrices (reference tensor.py:390-406)."""
        cs = [_host(c) for c in self.cores]
        if dir == "lr":
            out = [cs[0].reshape(-1, cs[0].shape[-1])]
            for c in cs[1:-1]:
                nxt = np.tensordot(out[-1], c, axes=(1, 0))
                out.append(nxt.reshape(-1, nxt.shape[-1]))
        elif dir == "rl":
            out = [cs[-1].reshape(cs[-1].shape[0], -1)]
            for c in cs[-2:0:-1]:
                nxt = np.tensordot(c, out[-1], axes=(2, 0))
                out.append(nxt.reshape(nxt.shape[0], -1))
        else:
            raise ValueError(dir)
        return out

    def __getitem__(self, k: int):
        return self.cores[k]

    def __setitem__(self, k: int, data) -> None:
        self.cores[k] = data
        self.invalidate_device()

    def gather(self, idx) -> npt.NDArray:
        """Entries at the given multi-indices (rows of ``idx`` are modes)."""
        idx = np.stack(idx) if not isinstance(idx, np.ndarray) else idx
        cs = [_host(c) for c in self.cores]
        acc = cs[0][0][idx[0]]                       # (N, r1)
        for k in range(1, self.ndim):
            sl = cs[k][:, idx[k], :]                 # (r, N, r')
            acc = np.einsum("nr,rns->ns", acc, sl)
        return acc.reshape(-1)

    def orthogonalize(self) -> "TensorTrain":
        """Left-orthogonalising QR sweep (reference tensor.py:559-572)."""
        out, carry = [], None
        for k, C in enumerate(self.cores):
            C = _host(C)
            if carry is not None:
                C = np.tensordot(carry, C, axes=(1, 0))
            if k < self.ndim - 1:
                Q, carry = np.linalg.qr(C.reshape(-1, C.shape[2]))
                out.append(Q.reshape(C.shape[0], C.shape[1], -1))
            else:
                out.append(C)
        return TensorTrain(out)

    def norm(self) -> float:
        if self.resident():
            return float(np.linalg.norm(self.orthogonalize_dev().cores[-1].get()))
        return float(np.linalg.norm(self.orthogonalize().cores[-1]))

    def gram_norm(self) -> float:
        """sqrt(<x, x>) by the Gram chain (2 d small device products, no QR sweep).  Accurate to a few ulp
        for a train that is not itself a difference of nearly equal terms -- a direct sum ``a + (-b)`` with
        a ~ b cancels inside the chain and is only good to sqrt(eps) ||a||; ``norm`` (QR sweep, reference
        tensor.py:442-444) has no such limit and is what ``error`` uses."""
        return float(np.sqrt(max(self.dot(self), 0.0)))

    def resident(self) -> bool:
        """True if the cores live in HBM only (as ``to_tt`` / ``round_dev`` / an MPO product leave
        them); arithmetic on such a train stays on the device."""
        return all(isinstance(c, DevArray) for c in self.cores)

    def to_device(self) -> "TensorTrain":
        """The same train with device-resident cores (uploads host cores once)."""
        return self if self.resident() else TensorTrain(list(self.dev_cores()))

    # ---- device versions (the step after to_tt) -------------------------
    def orthogonalize_dev(self) -> "TensorTrain":
        """Left-orthogonalising QR sweep on the device: thin QR by ``ttsk_qr_thin`` (CholeskyQR2
        with LAPACK's signs, Householder fallback), R recovered as Q^T M by the long-K kernel.
        Same result as ``orthogonalize`` (reference tensor.py:559-572) up to rounding."""
        import ctypes
        from . import _native as nat
        from .device import contract
        cores = self.dev_cores()
        out, carry = [], None
        for k, C in enumerate(cores):
            if carry is not None:
                C = contract("ij,jkl->ikl", carry, C)
            if k < self.ndim - 1:
                r1, n, r2 = C.shape
                M = C.contiguous().reshape(r1 * n, r2)
                m = r1 * n
                # wide unfolding (m < r2): Q (m x m) from the leading square block, R = Q^T M is m x r2
                Q = M.copy() if m >= r2 else M[:, :m].contiguous()
                q = min(m, r2)
                nat.call("ttsk_qr_thin", ctypes.c_void_p(Q.ptr), m, q, 0)
                carry = contract("ai,aj->ij", Q, M)
                nat.call("ttsk_triu", ctypes.c_void_p(carry.ptr), q, r2, 0)
                out.append(Q.reshape(r1, n, q))
            else:
                out.append(C.contiguous())
        return TensorTrain(out)

    def round_dev(self, eps: Optional[float] = None, max_rank: Optional[TTRank] = None,
                  orthogonalized: bool = False) -> "TensorTrain":
        """TT-SVD rounding on the device, the algorithm of ``round`` (reference tensor.py:446-484):
        the SVD of each wide unfolding M (r x n r') goes through the thin QR of M^T and a Jacobi SVD
        of the r x r factor (``ttsk_svd_small``).  Cores stay on the device; they equal ``round``'s
        up to the sign gauge of the singular vectors (the represented tensor is the same)."""
        import ctypes
        from . import _native as nat
        tt = self if orthogonalized else self.orthogonalize_dev()
        eps = 0 if eps is None else eps
        cap = process_tt_rank(tt.rank if max_rank is None else max_rank, tt.shape, trim=True)
        cores = tt.dev_cores()
        out, carry = [], None
        for k in range(tt.ndim - 1, -1, -1):
            C = cores[k]
            if carry is not None:
                C = contract("ijk,kl->ijl", C, carry)
            if k == 0:
                out.append(C)
                continue
            r1, n, r2 = C.shape
            C = C.contiguous()
            k2 = n * r2
            if k2 < r1:
                # tall unfolding M (r1 x k2): one-sided Jacobi over the ROWS of M (columns of M^T padded
                # to r1 x r1), so rows that are exactly zero -- a zero summand -- give exactly zero
                # singular values as they do in LAPACK.  M^T = US Vt  =>  M = Vt^T diag(S) (US/S)^T.
                A = DevArray.zeros((r1, r1))
                copy_into(A[:k2], C.reshape(r1, k2).T)
                Qc = None
            else:
                Mt = C.reshape(r1, k2).T.contiguous()                # (n r2, r1), tall
                Qc = Mt.copy()
                nat.call("ttsk_qr_thin", ctypes.c_void_p(Qc.ptr), k2, r1, 0)
                Rc = contract("ai,aj->ij", Qc, Mt)                   # (r1, r1) upper triangular
                nat.call("ttsk_triu", ctypes.c_void_p(Rc.ptr), r1, r1, 0)
                A = Rc.T.contiguous()                                # M = Rc^T Qc^T
            if r1 > 1024:
                raise ValueError(f"round_dev: TT rank {r1} > 1024 is beyond the one-workgroup SVD; use round()")
            US, S, Vt = DevArray.empty((r1, r1)), DevArray.empty((r1,)), DevArray.empty((r1, r1))
            nat.call("ttsk_svd_small", ctypes.c_void_p(A.ptr), r1, r1, ctypes.c_void_p(US.ptr),
                     ctypes.c_void_p(S.ptr), ctypes.c_void_p(Vt.ptr), 0)
            sv = S.get()
            r = max(1, min(int(np.sum(sv > sv[0] * eps)), cap[k - 1], k2))
            if Qc is None:
                eye = DevArray.from_host(np.eye(r))
                inv = np.divide(1.0, sv[:r], out=np.zeros(r), where=sv[:r] > 0)
                carry = contract("ka,kb->ab", Vt[:r], eye, k_scale=S[:r].contiguous())
                V = contract("ka,ck->ac", eye, US[:k2, :r], k_scale=DevArray.from_host(inv))
            else:
                carry = US[:, :r]
                V = contract("ab,cb->ac", Vt[:r], Qc)
            out.append(V.reshape(r, n, r2))
        return TensorTrain(out[::-1])

    def round(self, eps: Optional[float] = None, max_rank: Optional[TTRank] = None,
              orthogonalized: bool = False) -> "TensorTrain":
        """TT-SVD rounding (reference tensor.py:446-484); host LAPACK (``round_dev`` is the device
        version)."""
        tt = self if orthogonalized else self.orthogonalize()
        eps = 0 if eps is None else eps
        cap = process_tt_rank(tt.rank if max_rank is None else max_rank, tt.shape, trim=True)
        out, carry = [], None
        for k in range(tt.ndim - 1, -1, -1):
            C = _host(tt.cores[k])
            if carry is not None:
                C = np.tensordot(C, carry, axes=(2, 0))
            if k > 0:
                U, S, Vt = np.linalg.svd(C.reshape(C.shape[0], -1))
                r = max(1, min(int(np.sum(S > S[0] * eps)), cap[k - 1]))
                carry = U[:, :r] * S[:r]
                out.append(Vt[:r].reshape(r, C.shape[1], C.shape[2]))
            else:
                out.append(C)
        return TensorTrain(out[::-1])

    def svdvals(self) -> List[npt.NDArray]:
        tt = self.orthogonalize()
        vals, carry = [], None
        for k in range(tt.ndim - 1, -1, -1):
            C = tt.cores[k]
            if carry is not None:
                C = np.tensordot(C, carry, axes=(2, 0))
            M = C.reshape(C.shape[0], -1) if k > 0 else C.reshape(-1, C.shape[2])
            U, S, _ = np.linalg.svd(M)
            carry = U * S[:U.shape[1]] if k > 0 else None
            vals.append(S)
        return vals[::-1]

    def __mul__(self, other: float) -> "TensorTrain":
        if self.resident():
            last = self.cores[-1].copy()
            axpby(last, last, float(other), 0.0)
            return TensorTrain(list(self.cores[:-1]) + [last])
        cores = [np.array(_host(c)) for c in self.cores]
        cores[-1] = cores[-1] * other
        return TensorTrain(cores)

    __rmul__ = __mul__

    @property
    def size(self) -> int:
        return int(sum(c.size for c in self.cores))

    def add(self, other: "TensorTrain") -> "TensorTrain":
        """Direct-sum addition of two TTs (reference tensor.py:503-525)."""
        if self.resident() and other.resident():
            out, d = [], self.ndim
            for k, (a, b) in enumerate(zip(self.cores, other.cores)):
                ra1, n, ra2 = a.shape
                rb1, _, rb2 = b.shape
                r1 = 1 if k == 0 else ra1 + rb1
                r2 = 1 if k == d - 1 else ra2 + rb2
                blk = DevArray.zeros((r1, n, r2))
                copy_into(blk[:ra1, :, :ra2], a)
                copy_into(blk[r1 - rb1:, :, r2 - rb2:], b)
                out.append(blk)
            return TensorTrain(out)
        A = [_host(c) for c in self.cores]
        B = [_host(c) for c in other.cores]
        out = [np.concatenate((A[0], B[0]), axis=2)]
        for a, b in zip(A[1:-1], B[1:-1]):
            blk = np.zeros((a.shape[0] + b.shape[0], a.shape[1], a.shape[2] + b.shape[2]))
            blk[:a.shape[0], :, :a.shape[2]] = a
            blk[a.shape[0]:, :, a.shape[2]:] = b
            out.append(blk)
        out.append(np.concatenate((A[-1], B[-1]), axis=0))
        return TensorTrain(out)

    def dot(self, other, reverse=False) -> float:
        if isinstance(other, TensorTrain) and self.resident() and other.resident():
            acc = None
            for a, b in zip(self.cores, other.cores):
                t = a.reshape(a.shape[1], a.shape[2]) if acc is None else contract("ij,ika->jka", acc, a)
                b = b.reshape(b.shape[1], b.shape[2]) if acc is None else b
                acc = contract("ka,kb->ab", t, b) if acc is None else contract("jka,jkb->ab", t, b)
            return float(acc.get().sum())
        if isinstance(other, TensorTrain):
            acc = np.ones((1, 1))
            for a, b in zip(self.cores, other.cores):
                acc = np.einsum("ij,ika,jkb->ab", acc, _host(a), _host(b), optimize=True)
            return float(acc.sum())
        return super().dot(other, reverse=reverse)

    def error(self, other, relative: bool = False, rmse: bool = False, fast: bool = False) -> float:
        if hasattr(other, "to_tt"):
            other = other.to_tt()
        if isinstance(other, TensorTrain):
            err = self.add(-other).norm()
            if relative:
                ref = other.norm()
                if ref == 0:
                    return np.inf
                err /= ref
            if rmse:
                err /= np.sqrt(np.prod(self.shape))
            return err
        return super().error(other, relative=relative, rmse=rmse, fast=fast)

    def __repr__(self) -> str:
        return f"<Tensor train of shape {self.shape} with rank {self.rank} at {hex(id(self))}>"


# --------------------------------------------------------------------------- sums
class TensorSum(Tensor):
    """Lazy sum of tensors of one shape (reference tensor.py:612-671)."""

    def __init__(self, tensors: List[Tensor], shape=None) -> None:
        self.tensors = tensors
        self.shape = tuple(tensors[0].shape) if shape is None else tuple(shape)

    @property
    def size(self) -> int:
        return sum(t.size for t in self.tensors)

    @property
    def T(self) -> "TensorSum":
        return TensorSum([t.T for t in self.tensors])

    def to_numpy(self):
        acc = np.zeros(self.shape)
        for t in self.tensors:
            acc += t.to_numpy()
        return acc

    def __add__(self, other) -> "TensorSum":
        extra = other.tensors if isinstance(other, TensorSum) else [other]
        return TensorSum(self.tensors + list(extra))

    def __iadd__(self, other) -> "TensorSum":
        if isinstance(other, TensorSum):
            self.tensors.extend(other.tensors)
        else:
            self.tensors.append(other)
        return self

    def prepare_device(self) -> None:
        for t in self.tensors:
            t.prepare_device()

    @property
    def num_summands(self) -> int:
        return len(self.tensors)

    def __mul__(self, other: Union[float, Iterable[float]]) -> "TensorSum":
        try:
            coeffs = list(other)  # type: ignore[arg-type]
        except TypeError:
            return TensorSum([t * other for t in self.tensors])
        if len(coeffs) != len(self.tensors):
            raise ValueError("one coefficient per summand expected")
        return TensorSum([t * c for t, c in zip(self.tensors, coeffs)])

    def dot(self, other, reverse=False) -> float:
        return sum(t.dot(other, reverse) for t in self.tensors)

    def __repr__(self) -> str:
        return f"<Sum of {self.num_summands} tensors of shape {self.shape} at {hex(id(self))}>"


# --------------------------------------------------------------------------- CP
class CPTensor(Tensor):
    """CP format, factor matrices ``(n_k, R)`` (reference tensor.py:674-743)."""

    def __init__(self, cores: ArrayList, _dev=None) -> None:
        self.cores = cores
        self.rank = int(cores[0].shape[1])
        self.shape = tuple(int(C.shape[0]) for C in cores)
        self._dev = _dev
        self._dev_key = None if _dev is None else tuple(cores)      # the objects, compared with `is`

    def dev_cores(self) -> List[DevArray]:
        if self._dev is None or not _same_objects(self._dev_key, self.cores):
            self._dev = [as_dev(c) for c in self.cores]
            self._dev_key = tuple(self.cores)
        return self._dev

    prepare_device = dev_cores

    @property
    def size(self) -> int:
        return int(sum(C.size for C in self.cores))

    @property
    def T(self) -> "CPTensor":
        dev = None
        if self._dev is not None and _same_objects(self._dev_key, self.cores):
            dev = self._dev[::-1]
        return CPTensor(self.cores[::-1], dev)

    def to_numpy(self):
        acc = _host(self.cores[0])
        for C in self.cores[1:]:
            acc = acc[..., None, :] * _host(C)
        return acc.sum(axis=-1)

    @classmethod
    def random(cls, shape, rank: int, seed: Optional[int] = None) -> "CPTensor":
        seeds = np.random.SeedSequence(seed).generate_state(len(shape))
        return cls([np.random.default_rng(s).standard_normal((n, rank)) / np.sqrt(n)
                    for s, n in zip(seeds, shape)])

    def __getitem__(self, k: int):
        return self.cores[k]

    def __setitem__(self, k: int, data) -> None:
        self.cores[k] = data
        self.invalidate_device()

    def gather(self, idx) -> npt.NDArray:
        acc = 1.0
        for C, rows in zip(self.cores, idx):
            acc = acc * _host(C)[rows]
        return np.sum(acc, axis=1)

    def __mul__(self, other: float) -> "CPTensor":
        cores = list(self.cores)
        cores[0] = _host(cores[0]) * other
        return CPTensor(cores)

    def __repr__(self) -> str:
        return f"<CP tensor of shape {self.shape} and rank {self.rank} at {hex(id(self))}>"


# --------------------------------------------------------------------------- Tucker
class TuckerTensor(Tensor):
    """Tucker format: core ``(s_1..s_d)`` and factors ``(s_k, n_k)`` (reference tensor.py:746-816)."""

    def __init__(self, factors: ArrayList, core, _dev=None) -> None:
        self.core = core
        self.factors = factors
        self.shape = tuple(int(U.shape[1]) for U in factors)
        self.rank = tuple(int(U.shape[0]) for U in factors)
        self._dev = _dev

    def dev_parts(self) -> Tuple[List[DevArray], DevArray]:
        if self._dev is None:
            self._dev = ([as_dev(U) for U in self.factors], as_dev(self.core))
        return self._dev

    prepare_device = dev_parts

    @property
    def T(self) -> "TuckerTensor":
        axes = tuple(reversed(range(len(self.shape))))
        dev = None if self._dev is None else (self._dev[0][::-1], self._dev[1].transpose(axes))
        return TuckerTensor(self.factors[::-1], np.transpose(self.core, axes), dev)

    @property
    def size(self) -> int:
        return int(self.core.size + sum(U.size for U in self.factors))

    def to_numpy(self):
        acc = _host(self.core)
        for k, U in enumerate(self.factors):
            acc = np.moveaxis(np.tensordot(acc, _host(U), axes=(k, 0)), -1, k)
        return acc

    def __mul__(self, other: float) -> "TuckerTensor":
        return TuckerTensor(self.factors, _host(self.core) * other)

    @classmethod
    def random(cls, shape, rank, seed: Optional[int] = None) -> "TuckerTensor":
        try:
            rk = tuple(rank)
        except TypeError:
            rk = (rank,) * len(shape)
        rk = tuple(min(r, n) for r, n in zip(rk, shape))
        seq = np.random.SeedSequence(seed)
        core = np.random.default_rng(seq.generate_state(1)[0]).standard_normal(rk)
        factors = []
        for r, n, s in zip(rk, shape, seq.generate_state(len(shape))):
            U = np.random.default_rng(s).standard_normal((r, n))
            factors.append(np.linalg.qr(U.T)[0].T)
        return cls(factors, core)

    def __repr__(self) -> str:
        return f"<Tucker tensor of shape {self.shape} and rank {self.rank} at {hex(id(self))}>"
