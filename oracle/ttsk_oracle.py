"""CPU oracle for the streaming TT-sketch hot path -- TEST INFRASTRUCTURE ONLY.

A NumPy restatement of the reference algorithm (RikVoorhaar/tt-sketch v1.1),
written as plain functions on plain arrays.  It is the checker for the HIP
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``tt_sketch_amd/`` does.

Parity status: PINNED.  Every function below is compared against outputs of
the reference itself (imported in the build container, see
``tests/golden/make_golden.py``) on the committed fixtures in
``tests/golden/*.npz`` by ``tests/test_oracle_golden.py``.

Data conventions (all fp64, C-contiguous unless noted)
  TT tensor      list of d cores, core mu of shape (s_{mu-1}, n_mu, s_mu)
  CP tensor      list of d factor matrices (n_mu, R)
  Tucker tensor  (factors, core): factors[mu] of shape (s_mu, n_mu), core (s_0..s_{d-1})
  dense tensor   ndarray of shape (n_0..n_{d-1})
  sparse tensor  (shape, indices (d, nnz) int64, entries (nnz,))
  sum            list of (kind, data) pairs

File:line citations refer to /root/reference/.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """ctypes handle on the C restatement of the hash sampler (hash_sampler.c)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libttsk_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
            )
        lib = ctypes.CDLL(path)
        lib.ttsk_oracle_ndtri.restype = ctypes.c_double
        lib.ttsk_oracle_ndtri.argtypes = [ctypes.c_double]
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# hash sampler (fast_lazy_gaussian.pyx)
# --------------------------------------------------------------------------
def hash_u64(vals: np.ndarray) -> np.ndarray:
    """fast_lazy_gaussian.pyx:13-37; returns a hashed copy."""
    out = np.ascontiguousarray(vals, dtype=np.uint64).copy()
    _lib().ttsk_oracle_hash_u64(_p(out), ctypes.c_size_t(out.size))
    return out


def _prep_idx(indices, shape):
    idx = np.ascontiguousarray(np.asarray(indices).astype(np.uint64))
    shp = np.ascontiguousarray(np.asarray(shape, dtype=np.uint64))
    return idx, shp, idx.shape[0], idx.shape[1]


def inds_to_rand_double(indices, shape, rank_min, rank_max, seed) -> np.ndarray:
    """fast_lazy_gaussian.pyx:52-105; (N, rank) doubles in [2^-511, 2)."""
    idx, shp, m, N = _prep_idx(indices, shape)
    out = np.empty((N, rank_max - rank_min))
    flat = np.empty(N, dtype=np.uint64)
    _lib().ttsk_oracle_inds_to_rand_double(
        _p(idx), _p(shp), ctypes.c_int(m), ctypes.c_size_t(N),
        ctypes.c_int(int(rank_min)), ctypes.c_int(int(rank_max)),
        ctypes.c_uint64(int(seed) % 2**63), _p(flat), _p(out))
    return out


def ndtri(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _lib().ttsk_oracle_ndtri_array(_p(x), _p(out), ctypes.c_size_t(x.size))
    return out


def inds_to_normal(indices, shape, rank_min, rank_max, seed) -> np.ndarray:
    """fast_lazy_gaussian.pyx:183-202; (N, rank) standard normals."""
    idx, shp, m, N = _prep_idx(indices, shape)
    out = np.empty((N, rank_max - rank_min))
    flat = np.empty(N, dtype=np.uint64)
    _lib().ttsk_oracle_inds_to_normal(
        _p(idx), _p(shp), ctypes.c_int(m), ctypes.c_size_t(N),
        ctypes.c_int(int(rank_min)), ctypes.c_int(int(rank_max)),
        ctypes.c_uint64(int(seed) % 2**63), _p(flat), _p(out))
    return out


def inds_to_sparse_sign(indices, shape, rank, rank_min, rank_max, nnz_per_row,
                        seed) -> np.ndarray:
    """fast_lazy_gaussian.pyx:156-180; (N, rank_max-rank_min) int16."""
    idx, shp, m, N = _prep_idx(indices, shape)
    full = np.zeros((N, int(rank)), dtype=np.int16)
    flat = np.empty(N, dtype=np.uint64)
    u = np.empty((N, int(nnz_per_row)))
    _lib().ttsk_oracle_inds_to_sparse_sign(
        _p(idx), _p(shp), ctypes.c_int(m), ctypes.c_size_t(N),
        ctypes.c_int(int(rank)), ctypes.c_int(int(nnz_per_row)),
        ctypes.c_uint64(int(seed) % 2**63), _p(flat), _p(u), _p(full))
    return full[:, int(rank_min):int(rank_max)]


# --------------------------------------------------------------------------
# tensors: transposition (Tensor.T of each kind)
# --------------------------------------------------------------------------
def transpose(kind: str, data):
    if kind == "tt":  # tensor.py:311-313
        return [np.transpose(c, (2, 1, 0)) for c in data[::-1]]
    if kind == "cp":  # tensor.py:692-694
        return data[::-1]
    if kind == "tucker":  # tensor.py:761-765
        factors, core = data
        return (factors[::-1], np.transpose(core))
    if kind == "dense":  # tensor.py:162-165
        return np.transpose(data)
    if kind == "sparse":  # tensor.py:201-204
        shape, idx, ent = data
        return (tuple(shape[::-1]), idx[::-1], ent)
    if kind == "sum":  # tensor.py:629-631
        return [(k, transpose(k, t)) for k, t in data]
    raise ValueError(kind)


def shape_of(kind: str, data) -> Tuple[int, ...]:
    if kind == "tt":
        return tuple(c.shape[1] for c in data)
    if kind == "cp":
        return tuple(c.shape[0] for c in data)
    if kind == "tucker":
        return tuple(u.shape[1] for u in data[0])
    if kind == "dense":
        return tuple(data.shape)
    if kind == "sparse":
        return tuple(data[0])
    if kind == "sum":
        return shape_of(*data[0])
    raise ValueError(kind)


# --------------------------------------------------------------------------
# DRM descriptors.  Rank bookkeeping is kept in *sketch order* exactly like
# drm_base.py:24-63: for transpose=True the tuples are already reversed.
# --------------------------------------------------------------------------
@dataclass
class TTDrm:
    """TensorTrainDRM (tensor_train_drm.py:23-58) with given cores."""
    cores: List[np.ndarray]  # d-1 cores in sketch order
    shape: Tuple[int, ...]   # shape of the tensor (user order)
    transpose: bool
    rank_min: Tuple[int, ...] = None
    rank_max: Tuple[int, ...] = None

    def __post_init__(self):
        if self.rank_min is None:
            self.rank_min = (0,) * len(self.cores)
        if self.rank_max is None:
            self.rank_max = tuple(c.shape[2] for c in self.cores)

    @property
    def rank(self):
        return tuple(b - a for a, b in zip(self.rank_min, self.rank_max))


@dataclass
class DenseDrm:
    """DenseGaussianDRM (dense_gaussian_drm.py:17-57) with given matrices
    (already row-sliced to [rank_min:rank_max])."""
    mats: List[np.ndarray]  # mats[mu] of shape (rank_mu, prod n_{<=mu}) sketch order
    shape: Tuple[int, ...]
    transpose: bool

    @property
    def rank(self):
        return tuple(m.shape[0] for m in self.mats)


@dataclass
class HashGaussDrm:
    """SparseGaussianDRM (sparse_gaussian_drm.py:11-44)."""
    seed: int
    shape: Tuple[int, ...]
    transpose: bool
    rank_min: Tuple[int, ...]
    rank_max: Tuple[int, ...]

    @property
    def rank(self):
        return tuple(b - a for a, b in zip(self.rank_min, self.rank_max))


@dataclass
class HashSignDrm:
    """SparseSignDRM (sparse_sign_drm.py:11-51)."""
    seed: int
    shape: Tuple[int, ...]
    transpose: bool
    true_rank: Tuple[int, ...]
    rank_min: Tuple[int, ...]
    rank_max: Tuple[int, ...]
    nnz: Tuple[int, ...] = None

    def __post_init__(self):
        if self.nnz is None:
            self.nnz = self.true_rank

    @property
    def rank(self):
        return tuple(b - a for a, b in zip(self.rank_min, self.rank_max))


# --------------------------------------------------------------------------
# left-to-right partial contractions of a DRM with a tensor (DRM.sketch_*)
# --------------------------------------------------------------------------
def _chain_tt_ttdrm(cores, drm: TTDrm):
    """tensor_train_drm.py:71-88."""
    out = []
    acc = None
    for mu, D in enumerate(drm.cores):
        X = cores[mu]
        if mu == 0:
            acc = np.einsum("ijk,ijl->kl", X, D)
        else:
            acc = np.einsum("ij,ikl,jkm->lm", acc, X, D, optimize="optimal")
        out.append(acc[:, drm.rank_min[mu]:drm.rank_max[mu]])
    return out


def _chain_cp_ttdrm(factors, drm: TTDrm):
    """tensor_train_drm.py:90-107."""
    out = []
    acc = None
    for mu, D in enumerate(drm.cores):
        V = factors[mu]
        if mu == 0:
            acc = np.einsum("ij,lik->jk", V, D)
        else:
            acc = np.einsum("ij,ki,jkl->il", acc, V, D, optimize="optimal")
        out.append(acc[:, drm.rank_min[mu]:drm.rank_max[mu]])
    return out


def _chain_sparse_ttdrm(sp, drm: TTDrm):
    """tensor_train_drm.py:60-69; yields (rank, nnz)."""
    _, idx, _ = sp
    out = []
    acc = None
    for mu, D in enumerate(drm.cores):
        g = D[:, idx[mu], :]
        if mu == 0:
            acc = g.reshape(g.shape[1:])
        else:
            acc = np.einsum("ijk,ji->jk", g, acc)
        out.append(acc[:, drm.rank_min[mu]:drm.rank_max[mu]].T)
    return out


def _chain_dense_ttdrm(drm: TTDrm):
    """tensor_train_drm.py:109-122; yields (rho_mu, prod n_{<=mu})."""
    out = []
    acc = drm.cores[0].reshape(-1, drm.cores[0].shape[-1])
    out.append(acc.T)
    for D in drm.cores[1:]:
        acc = np.einsum("ij,jkl->ikl", acc, D)
        acc = acc.reshape(-1, acc.shape[-1])
        out.append(acc.T)
    return out


def _chain_tucker_ttdrm(tk, drm: TTDrm):
    """tensor_train_drm.py:124-145."""
    factors, _ = tk
    rank = drm.rank
    out = []
    acc = np.einsum("ijk,jl->ilk", drm.cores[0], factors[0].T)
    acc = acc.reshape(factors[0].shape[0], rank[0])
    out.append(acc)
    for mu in range(1, len(drm.cores)):
        red = np.einsum("jkl,km->jml", drm.cores[mu], factors[mu].T)
        acc = np.einsum("ij,jml->iml", acc, red)
        acc = acc.reshape(-1, acc.shape[-1])
        out.append(acc)
    return out


def _partial_dense_lr(cores):
    """tensor.py:390-397."""
    parts = [cores[0].reshape(-1, cores[0].shape[-1])]
    for c in cores[1:-1]:
        nxt = np.einsum("ij,jkl->ikl", parts[-1], c)
        parts.append(nxt.reshape(-1, nxt.shape[-1]))
    return parts


def drm_contractions(kind: str, data, drm) -> list:
    """``list(drm.sketch_<kind>(tensor))`` including the ``handle_transpose``
    wrapper (drm_base.py:122-145): transposed DRMs see the transposed tensor
    and the list is reversed.  For kind == "sum" returns a list (over mu) of
    tuples (over summands) as sketch_dispatch.py:142-147."""
    if kind == "sum":
        per = [drm_contractions(k, t, drm) for k, t in data]
        d = len(drm.shape)
        return [tuple(p[mu] for p in per) for mu in range(d - 1)]
    if tuple(drm.shape) != shape_of(kind, data):
        raise ValueError("shape mismatch")
    t = transpose(kind, data) if drm.transpose else data
    if isinstance(drm, TTDrm):
        if kind == "tt":
            out = _chain_tt_ttdrm(t, drm)
        elif kind == "cp":
            out = _chain_cp_ttdrm(t, drm)
        elif kind == "sparse":
            out = _chain_sparse_ttdrm(t, drm)
        elif kind == "dense":
            out = _chain_dense_ttdrm(drm)
        elif kind == "tucker":
            out = _chain_tucker_ttdrm(t, drm)
        else:
            raise ValueError(kind)
    elif isinstance(drm, DenseDrm):
        if kind == "dense":  # dense_gaussian_drm.py:77-80
            out = list(drm.mats)
        elif kind == "tt":  # dense_gaussian_drm.py:68-75
            out = [(m @ p).T for m, p in zip(drm.mats, _partial_dense_lr(t))]
        elif kind == "sparse":  # dense_gaussian_drm.py:59-66 (C-order ravel)
            shp, idx, _ = t
            out = []
            for mu in range(len(shp) - 1):
                flat = np.ravel_multi_index(tuple(idx[:mu + 1]), shp[:mu + 1])
                out.append(drm.mats[mu][:, flat])
        else:
            raise ValueError(kind)
    elif isinstance(drm, HashGaussDrm):
        if kind != "sparse":
            raise ValueError(kind)
        shp, idx, _ = t
        out = []
        for mu in range(len(shp) - 1):  # sparse_gaussian_drm.py:29-44
            s = (mu + int(drm.seed)) % 2**63
            out.append(inds_to_normal(idx[:mu + 1], shp[:mu + 1],
                                      drm.rank_min[mu], drm.rank_max[mu], s).T)
    elif isinstance(drm, HashSignDrm):
        if kind != "sparse":
            raise ValueError(kind)
        shp, idx, _ = t
        out = []
        for mu in range(len(shp) - 1):  # sparse_sign_drm.py:34-51
            s = (mu + int(drm.seed)) % 2**63
            out.append(inds_to_sparse_sign(idx[:mu + 1], shp[:mu + 1],
                                           drm.true_rank[mu], drm.rank_min[mu],
                                           drm.rank_max[mu], drm.nnz[mu], s).T)
    else:
        raise ValueError(type(drm))
    return out[::-1] if drm.transpose else out


# --------------------------------------------------------------------------
# Omega / Psi per tensor kind (sketching_methods/*.py)
# --------------------------------------------------------------------------
def _unfold(A, k):
    """utils.py:63-83 with mode=range(k), mat_shape=True (C-order)."""
    rows = int(np.prod(A.shape[:k], dtype=np.int64))
    return A.reshape(rows, -1)


def omega(kind, data, left, right, mu, omega_shape):
    if kind in ("tt", "cp"):  # tensor_train_sketch.py:8-11, cp_sketch.py:6-9
        return left.T @ right
    if kind == "dense":  # dense_sketch.py:7-16
        return left @ _unfold(data, mu + 1) @ right.T
    if kind == "sparse":  # sparse_sketch.py:39-46
        return (left * data[2]) @ right.T
    if kind == "tucker":  # tucker_sketch.py:9-19
        return left.T @ _unfold(data[1], mu + 1) @ right
    if kind == "sum":  # sketch_dispatch.py:85-105
        acc = np.zeros(omega_shape)
        for (k, t), l, r in zip(data, left, right):
            acc += omega(k, t, l, r, mu, omega_shape)
        return acc
    raise ValueError(kind)


def psi(kind, data, left, right, mu, psi_shape):
    if kind == "tt":  # tensor_train_sketch.py:14-35
        X = data[mu]
        if left is None:
            return np.einsum("ijk,kl->ijl", X, right)
        if right is None:
            return np.einsum("ij,jkl->ikl", left.T, X)
        return np.einsum("ij,jkl,lm->ikm", left.T, X, right, optimize="optimal")
    if kind == "cp":  # cp_sketch.py:12-36
        V = data[mu]
        if left is None:
            return np.einsum("ji,il->jl", V, right)[None]
        if right is None:
            return np.einsum("il,kl->ik", left.T, V)[:, :, None]
        return np.einsum("ij,kj,jm->ikm", left.T, V, right, optimize="optimal")
    if kind == "dense":  # dense_sketch.py:19-52
        d = data.ndim
        if left is None:
            return (_unfold(data, 1) @ right.T)[None]
        if right is None:
            return (left @ _unfold(data, d - 1))[:, :, None]
        X3 = data.reshape(int(np.prod(data.shape[:mu], dtype=np.int64)),
                          data.shape[mu], -1)
        return np.einsum("ij,jkl,ml->ikm", left, X3, right, optimize="optimal")
    if kind == "sparse":  # sparse_sketch.py:8-36,49-69
        shp, idx, ent = data
        out = np.zeros(psi_shape)
        d = len(shp)
        for j in range(psi_shape[1]):
            mask = idx[mu] == j
            if mu == 0:
                out[:, j, :] = (ent[mask] @ right[:, mask].T).reshape(1, -1)
            elif mu == d - 1:
                out[:, j, :] = (left[:, mask] @ ent[mask]).reshape(-1, 1)
            else:
                out[:, j, :] = (left[:, mask] * ent[mask]) @ right[:, mask].T
        return out
    if kind == "tucker":  # tucker_sketch.py:22-46
        factors, core = data
        ld = left.shape[0] if left is not None else 1
        rd = right.shape[0] if right is not None else 1
        C3 = core.reshape(ld, factors[mu].shape[0], rd)
        if left is None:
            P = np.einsum("ijk,kl->ijl", C3, right)
        elif right is None:
            P = np.einsum("ij,jkl->ikl", left.T, C3)
        else:
            P = np.einsum("ij,jkl,lm->ikm", left.T, C3, right, optimize="optimal")
        return np.einsum("ijk,jl->ilk", P, factors[mu])
    if kind == "sum":  # sketch_dispatch.py:111-136
        acc = np.zeros(psi_shape)
        n = len(data)
        left = (None,) * n if left is None else left
        right = (None,) * n if right is None else right
        for (k, t), l, r in zip(data, left, right):
            acc += psi(k, t, l, r, mu, psi_shape)
        return acc
    raise ValueError(kind)


# --------------------------------------------------------------------------
# solves (utils.py:98-109, sketch_dispatch.py:160-174)
# --------------------------------------------------------------------------
def right_mul_pinv(A, B):
    """A @ pinv(B) via gelsd with cond=eps (utils.py:98-102)."""
    return scipy.linalg.lstsq(B.T, A.T, cond=None)[0].T


def left_mul_pinv(A, B):
    """pinv(A) @ B (utils.py:105-109)."""
    return scipy.linalg.lstsq(A, B, cond=None)[0]


def orth_step(Psi, Omega):
    """sketch_dispatch.py:160-174."""
    r1, n, r2 = Psi.shape
    final = r2 if Omega is None else Omega.shape[0]
    M = Psi.reshape(r1 * n, r2)
    if Omega is not None:
        M = right_mul_pinv(M, Omega)
    Q, _ = scipy.linalg.qr(M, mode="economic")
    return Q.reshape(r1, n, final)


# --------------------------------------------------------------------------
# general_sketch (sketch_dispatch.py:202-275) and assembly (sketch.py:400-443)
# --------------------------------------------------------------------------
def general_sketch(kind, data, left_drm, right_drm, method="streaming",
                   return_contractions=False):
    shape = shape_of(kind, data)
    d = len(shape)
    if method != "hmt":
        if left_drm is None:
            raise ValueError("left_drm required")
        left_c = drm_contractions(kind, data, left_drm)
    else:
        left_c = None
    right_c = drm_contractions(kind, data, right_drm)
    right_rank = tuple(right_drm.rank[::-1])               # user order
    left_rank = left_drm.rank if left_drm is not None else right_rank

    Omegas = []
    if method != "hmt":
        for mu in range(d - 1):
            Omegas.append(omega(kind, data, left_c[mu], right_c[mu], mu,
                                (left_rank[mu], right_rank[mu])))

    Psis = []
    orth_cores: List[np.ndarray] = []
    orth_left = []
    for mu in range(d):
        if mu > 0:
            if method in ("hmt", "orthogonal"):
                # OrthogTTDRM (sketch_dispatch.py:177-193): a left TT-DRM whose
                # cores are the orthogonalised Psi cores produced so far.
                orth_cores.append(Psis[-1])
                odrm = TTDrm(list(orth_cores), shape, False,
                             rank_min=(0,) * len(orth_cores),
                             rank_max=tuple(left_rank[:len(orth_cores)]))
                lsk = _orth_contraction(kind, data, odrm, mu - 1)
                orth_left.append(lsk)
            else:
                lsk = left_c[mu - 1]
            r1 = left_rank[mu - 1]
        else:
            lsk, r1 = None, 1
        if mu < d - 1:
            rsk, r2 = right_c[mu], right_rank[mu]
        else:
            rsk, r2 = None, 1
        P = psi(kind, data, lsk, rsk, mu, (r1, shape[mu], r2))
        if mu < d - 1:
            if method == "orthogonal":
                P = orth_step(P, Omegas[mu])
            elif method == "hmt":
                P = orth_step(P, None)
        Psis.append(P)
    if return_contractions:
        return Psis, Omegas, left_c, right_c
    return Psis, Omegas


def _orth_contraction(kind, data, odrm: TTDrm, mu):
    """mu-th item of the lazily advanced generator of OrthogTTDRM."""
    if kind == "sum":
        return tuple(_orth_contraction(k, t, odrm, mu) for k, t in data)
    if kind == "tt":
        return _chain_tt_ttdrm(data, odrm)[mu]
    if kind == "cp":
        return _chain_cp_ttdrm(data, odrm)[mu]
    if kind == "sparse":
        return _chain_sparse_ttdrm(data, odrm)[mu]
    if kind == "dense":
        return _chain_dense_ttdrm(odrm)[mu]
    if kind == "tucker":
        return _chain_tucker_ttdrm(data, odrm)[mu]
    raise ValueError(kind)


def assemble(Psis, Omegas, direction="auto"):
    """sketch.py:400-443."""
    left_rank = tuple(P.shape[0] for P in Psis[1:])
    right_rank = tuple(P.shape[2] for P in Psis[:-1])
    if direction == "auto":
        bigger = np.all(np.array(left_rank) > np.array(right_rank))
        direction = "left" if bigger else "right"
    cores = []
    if direction == "right":
        for P, Om in zip(Psis[:-1], Omegas):
            r1, n, r2 = P.shape
            cores.append(right_mul_pinv(P.reshape(r1 * n, r2), Om)
                         .reshape(r1, n, Om.shape[0]))
        cores.append(Psis[-1])
    elif direction == "left":
        cores.append(Psis[0])
        for P, Om in zip(Psis[1:], Omegas):
            r1, n, r2 = P.shape
            cores.append(left_mul_pinv(Om, P.reshape(r1, n * r2))
                         .reshape(Om.shape[1], n, r2))
    else:
        raise ValueError(direction)
    return cores


# --------------------------------------------------------------------------
# helpers used by tests and the bench (not part of the reference path)
# --------------------------------------------------------------------------
def tt_svd(X, rank=None):
    """tt_svd.py:10-49: left-to-right sweep, SVD of the (r n) x rest unfolding, r = max(min(cols of U, cap), 1);
    caps trimmed as utils.py:121-175 (process_tt_rank, trim=True)."""
    shape = X.shape
    d = len(shape)
    if rank is None:
        rank = (int(np.prod(shape, dtype=np.int64)),) * (d - 1)
    rank = tuple(rank) if not np.isscalar(rank) else (int(rank),) * (d - 1)
    cap = list(rank)
    for _ in range(100):                      # utils.py:121-158: sweep until the ranks are feasible
        old = list(cap)
        for i in range(d - 1):
            lo = shape[i] * (cap[i - 1] if i > 0 else 1)
            hi = shape[i + 1] * (cap[i + 1] if i < d - 2 else 1)
            cap[i] = min(cap[i], lo, hi)
        if cap == old:
            break
    cores, rest, r_prev = [], np.asarray(X, dtype=np.float64).reshape(1, -1), 1
    for k in range(d - 1):
        M = rest.reshape(r_prev * shape[k], -1)
        U, S, Vt = np.linalg.svd(M, full_matrices=False)
        r = max(min(U.shape[1], cap[k]), 1)
        cores.append(U[:, :r].reshape(r_prev, shape[k], r))
        rest = S[:r, None] * Vt[:r]
        r_prev = r
    cores.append(rest.reshape(r_prev, shape[-1], 1))
    return cores


def tt_to_numpy(cores):
    """tensor.py:315-321."""
    acc = cores[0].reshape(cores[0].shape[1:])
    for c in cores[1:]:
        acc = np.einsum("...j,jkl->...kl", acc, c)
    return acc.reshape(acc.shape[:-1])


def random_tt(shape, rank, rng, scale="norm-1"):
    """Synthetic TT cores N(0,1)/sqrt(r1*n) (SURVEY.md 8d inputs)."""
    d = len(shape)
    rk = (1,) + tuple(rank if not np.isscalar(rank) else (rank,) * (d - 1)) + (1,)
    cores = []
    for i in range(d):
        c = rng.standard_normal((rk[i], shape[i], rk[i + 1]))
        c /= np.sqrt(rk[i] * shape[i]) if scale == "norm-1" else np.sqrt(rk[i])
        cores.append(c)
    return cores


def random_tt_drm(shape, rank, transpose, rng) -> TTDrm:
    """DRM cores distributed as tensor_train_drm.py:52-56 / tensor.py:370-371
    (N(0,1)/sqrt(r1), last core dropped); single-stream rng, see SURVEY 8c."""
    d = len(shape)
    shp = tuple(shape[::-1]) if transpose else tuple(shape)
    rk = tuple(rank if not np.isscalar(rank) else (rank,) * (d - 1))
    if transpose:
        rk = rk[::-1]
    full = random_tt(shp, rk, rng, scale="norm-preserve")
    return TTDrm(full[:-1], tuple(shape), transpose)
