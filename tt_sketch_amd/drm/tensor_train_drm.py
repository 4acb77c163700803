"""Tensor-train DRM: sketches with partial contractions of a fixed random TT.

API of the reference's ``tt_sketch/drm/tensor_train_drm.py:23-145`` (constructor incl. the
``cores=`` injection kwarg, ``sketch_tt / sketch_cp / sketch_sparse / sketch_dense /
sketch_tucker``).  Cores are sampled on the device (hash -> ndtri generator keyed by
``seed`` and the core number; N(0,1)/sqrt(r_{mu-1}) as tensor.py:370-371 "norm-preserve",
last core never materialised) and every chain step is an MFMA contraction in HBM.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple, Union

import numpy as np

from .. import _native as nat
from ..device import DevArray, as_dev, contract
from ..drm_base import CanSlice, handle_transpose
from ..sketching_methods.abstract_methods import (CansketchCP, CansketchDense, CansketchSparse,
                                                  CansketchTT, CanSketchTucker)
from ..utils import random_normal_dev_many


class TensorTrainDRM(CansketchSparse, CansketchTT, CansketchCP, CanSlice, CansketchDense,
                     CanSketchTucker):
    cores: list

    def __init__(self, rank: Union[Tuple[int, ...], int], shape: Tuple[int, ...], transpose: bool,
                 seed: Optional[int] = None, **kwargs) -> None:
        super().__init__(rank, shape, transpose, seed=seed, **kwargs)
        if "cores" in kwargs:
            self.cores = kwargs["cores"]
        else:
            walk = self.shape[::-1] if transpose else self.shape
            rk = (1,) + tuple(self.true_rank)
            k = len(walk) - 1
            self.cores = random_normal_dev_many([(rk[mu], walk[mu], rk[mu + 1]) for mu in range(k)],
                                                [(self.seed << 20) + mu for mu in range(k)],
                                                [1.0 / np.sqrt(rk[mu]) for mu in range(k)])
        self._dev: List[DevArray] = []
        self._dev_src: list = []          # the host objects the device copies were made from

    def dev_cores(self) -> List[DevArray]:
        """Device copies of ``self.cores`` (the list may grow: OrthogTTDRM appends)."""
        for k, c in enumerate(self.cores):
            if k < len(self._dev) and self._dev_src[k] is c:
                continue
            d = as_dev(c)
            if k < len(self._dev):
                self._dev[k], self._dev_src[k] = d, c
            else:
                self._dev.append(d)
                self._dev_src.append(c)
        return self._dev

    def _core(self, mu) -> DevArray:
        return self.dev_cores()[mu]

    def _cut(self, mu, mat: DevArray) -> DevArray:
        return mat[:, self.rank_min[mu]:self.rank_max[mu]]

    # ------------------------------------------------------------------ TT input
    @handle_transpose
    def sketch_tt(self, tensor):
        """L_mu[l,m] = sum L_{mu-1}[i,j] X_mu[i,k,l] D_mu[j,k,m] (reference :71-88)."""
        Xs = tensor.dev_cores()
        L = None
        for mu in range(len(self.shape) - 1):
            X, D = Xs[mu], self._core(mu)
            if mu == 0:
                L = contract("ijk,ijl->kl", X, D)
            elif X.strides[2] > X.strides[0]:       # view of a transposed core
                T = contract("ij,ikl->jlk", L, X)
                L = contract("jlk,jkm->lm", T, D)
            else:
                T = contract("ij,ikl->jkl", L, X)
                L = contract("jkl,jkm->lm", T, D)
            yield self._cut(mu, L)

    # ------------------------------------------------------------------ CP input
    @handle_transpose
    def sketch_cp(self, tensor):
        """L_mu[i,l] = sum_{j,k} L_{mu-1}[i,j] V_mu[k,i] D_mu[j,k,l] (reference :90-107)."""
        Vs = tensor.dev_cores()
        L = None
        for mu in range(len(self.shape) - 1):
            V, D = Vs[mu], self._core(mu)
            if mu == 0:
                L = contract("ij,ik->jk", V, D[0])
            else:
                W = contract("ij,jkl->ikl", L, D)
                L = contract("ki,ikl->il", V, W)
            yield self._cut(mu, L)

    # ------------------------------------------------------------------ sparse input
    @handle_transpose
    def sketch_sparse(self, tensor):
        """v_e <- v_e D_mu[:, idx_mu[e], :] per nonzero (reference :60-69); yields (rank, nnz)."""
        idx = tensor.dev_indices()
        order = tensor.dev_row_order
        N = tensor.nnz
        v = None
        for mu, _ in enumerate(self.cores):     # the list may grow while we walk it (OrthogTTDRM)
            D = self._core(mu).contiguous()
            rho, n, rhop = D.shape
            out = DevArray.empty((N, rhop))
            row_ptr = idx.ptr + order[mu] * N * 8
            nat.call("ttsk_sparse_ttdrm_step", None if v is None else ctypes.c_void_p(v.ptr), rho,
                     ctypes.c_void_p(D.ptr), n, rhop, ctypes.c_void_p(row_ptr), ctypes.c_size_t(N),
                     ctypes.c_void_p(out.ptr), 0)
            v = out
            yield self._cut(mu, v).T

    # ------------------------------------------------------------------ dense input
    @handle_transpose
    def sketch_dense(self, tensor):
        """The DRM as dense matrices (rho_mu, prod n_{<=mu}) (reference :109-122).

        Handed out as ``ChainedUnfolding`` recipes (P_k = P_{k-1} x D_k): the dense Omega / Psi routines
        contract the tensor one mode at a time and never form the (rho x n^k) matrices; anything else that
        touches one gets the matrix (``as_dev`` / ``np.asarray`` / ``.get()`` materialise it)."""
        from ..sketching_methods.dense_sketch import ChainedUnfolding
        P = None
        for mu in range(len(self.shape) - 1):
            P = ChainedUnfolding(P, self._core(mu))
            yield P

    # ------------------------------------------------------------------ Tucker input
    @handle_transpose
    def sketch_tucker(self, tensor):
        """DRM against the Tucker factors (reference :124-145); yields (prod s_{<=mu}, rho_mu)."""
        Us, _ = tensor.dev_parts()
        P = contract("jk,lj->lk", self._core(0)[0], Us[0])
        yield P
        for mu in range(1, len(self.shape) - 1):
            red = contract("jkl,mk->jml", self._core(mu), Us[mu])
            P = contract("ij,jml->iml", P, red)
            P = P.reshape(-1, P.shape[-1])
            yield P
