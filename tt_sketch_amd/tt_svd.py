"""Classical TT-SVD of a dense array on the device (reference ``tt_sketch/tt_svd.py:10-49``).

Left-to-right sweep over the C-order unfoldings ``M_mu`` (``r_{mu-1} n_mu`` x rest) of the resident tensor.
The reference takes a LAPACK SVD of every (usually very wide) unfolding; here

    M^T = Q R            thin QR of the tall transpose (``ttsk_qr_thin``: CholeskyQR2 / Householder)
    R^T = U S V^T        one-sided Jacobi SVD of the small square factor (``ttsk_svd_small``: one workgroup up
                         to m = 1024, all compute units with a barrier per round up to 8192)
    core_mu   = U[:, :r]                       r = max(min(#columns, rank cap), 1), reference :23, :35
    remainder = S_r V_r^T Q^T  (= U_r^T M)     carried to the next mode, reference :29-31, :38-42

so the tensor is read twice per mode (transpose copy + product) and the factorisation work is on m x m
matrices, m = r_{mu-1} n_mu.  Singular vectors are determined up to sign (and arbitrarily inside a zero or
repeated singular value), so results equal the reference's as tensors with identical TT ranks
(tests/golden/tt_svd_cases.npz: runs of the reference; `oracle.ttsk_oracle.tt_svd` is the NumPy restatement
the tests use as checker).  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _native as nat
from .device import DevArray, contract, copy_into
from .tensor import Tensor, TensorTrain
from .utils import TTRank, process_tt_rank

_SVD_MAX = 8192          # ttsk_svd_small: one workgroup up to 1024 columns, the whole chip (svd_grid.hip) beyond


def _inv_singular(sv: np.ndarray, m: int) -> np.ndarray:
    """1 / sigma for the directions above the numerical rank, 0 below: there U = (U S) / S would be noise over
    noise.  LAPACK completes U with arbitrary orthonormal vectors in that case; here those columns of the core
    are zero (they meet rows of the remainder that are zero to rounding either way, the tensor is the same)."""
    floor = (sv[0] if sv.size else 0.0) * 64 * np.finfo(np.float64).eps * np.sqrt(max(m, 1))
    return np.divide(1.0, sv, out=np.zeros(sv.size), where=sv > floor)


def _svd_wide(M: DevArray, cap: int):
    """(U_r (m, r), remainder (r, cols)) of M (m, cols) with r = max(min(min(m, cols), cap), 1)."""
    m, cols = M.shape
    P = ctypes.c_void_p
    if cols >= m:
        if m > _SVD_MAX:
            raise ValueError(f"tt_svd: unfolding with {m} rows is beyond the Jacobi SVD (<= {_SVD_MAX}); "
                             "lower the rank cap of the previous mode")
        Q = DevArray.empty((cols, m))
        copy_into(Q, M.T)                                       # M^T, tall
        nat.call("ttsk_qr_thin", P(Q.ptr), cols, m, 0)
        R = contract("ai,ja->ij", Q, M)                         # Q^T M^T = R (m, m), upper triangular
        nat.call("ttsk_triu", P(R.ptr), m, m, 0)
        A = R.T.contiguous()                                    # M = R^T Q^T
        US, S, Vt = DevArray.empty((m, m)), DevArray.empty((m,)), DevArray.empty((m, m))
        nat.call("ttsk_svd_small", P(A.ptr), m, m, P(US.ptr), P(S.ptr), P(Vt.ptr), 0)
        r = max(min(m, cap), 1)
        sv = S.get()[:r]
        inv = DevArray.from_host(_inv_singular(sv, m))
        eye = DevArray.from_host(np.eye(r))
        U = contract("ik,kj->ij", US[:, :r], eye, k_scale=inv)              # U_r = (U S)_r S_r^{-1}
        SVt = contract("ik,kj->ij", eye, Vt[:r], k_scale=S[:r].contiguous())  # S_r V_r^T
        rest = contract("ab,cb->ac", SVt, Q)                    # (r, cols)
        return U, rest
    # tall unfolding (the last modes of a sweep with generous caps): M = U S V^T directly
    if cols > _SVD_MAX:
        raise ValueError(f"tt_svd: tall unfolding with {cols} columns is beyond the Jacobi SVD (<= {_SVD_MAX})")
    A = M.contiguous()
    US, S, Vt = DevArray.empty((m, cols)), DevArray.empty((cols,)), DevArray.empty((cols, cols))
    nat.call("ttsk_svd_small", P(A.ptr), m, cols, P(US.ptr), P(S.ptr), P(Vt.ptr), 0)
    r = max(min(cols, cap), 1)
    sv = S.get()[:r]
    inv = DevArray.from_host(_inv_singular(sv, m))
    eye = DevArray.from_host(np.eye(r))
    U = contract("ik,kj->ij", US[:, :r], eye, k_scale=inv)
    rest = contract("ik,kj->ij", eye, Vt[:r], k_scale=S[:r].contiguous())
    return U, rest


def tt_svd(tensor: Tensor, rank: Optional[TTRank] = None) -> TensorTrain:
    """TT-SVD of ``tensor`` (any type with ``dense()``; a ``DenseTensor`` stays resident) in a left-to-right
    sweep; cores stay in HBM (``np.asarray(core)`` copies out)."""
    shape = tuple(int(n) for n in tensor.shape)
    d = len(shape)
    if rank is None:
        rank = (int(np.prod(shape, dtype=np.int64)),) * (d - 1)
    cap = process_tt_rank(rank, shape, trim=True)
    dense = tensor if hasattr(tensor, "dev_data") else tensor.dense()
    rest = dense.dev_data().reshape(1, -1)
    cores, r_prev = [], 1
    for k in range(d - 1):
        M = rest.reshape(r_prev * shape[k], -1)
        U, rest = _svd_wide(M, cap[k])
        r = U.shape[1]
        cores.append(U.reshape(r_prev, shape[k], r))
        r_prev = r
    cores.append(rest.reshape(r_prev, shape[-1], 1).contiguous())
    return TensorTrain(cores)
