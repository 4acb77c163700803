"""Dense Gaussian DRM: one explicit (rank, prod n_{<=mu}) Gaussian matrix per unfolding.

API of the reference's ``tt_sketch/drm/dense_gaussian_drm.py:17-80``.  Matrices are filled on
the device, row-major, keyed by (seed, mu, element) so that the leading rows are unchanged
when the rank grows (CanIncreaseRank) and a ``rank_min:rank_max`` slice is a row view.
For parity tests ``sketching_mats`` may be overwritten after construction.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple, Union

import numpy as np

from .. import _native as nat
from ..device import DevArray, as_dev, contract
from ..drm_base import CanIncreaseRank, handle_transpose
from ..sketching_methods.abstract_methods import CansketchDense, CansketchSparse, CansketchTT
from ..utils import random_normal_dev


class DenseGaussianDRM(CansketchTT, CansketchSparse, CansketchDense, CanIncreaseRank):
    sketching_mats: list

    def __init__(self, rank: Union[Tuple[int, ...], int], shape: Tuple[int, ...], transpose: bool,
                 seed: Optional[int] = None, **kwargs) -> None:
        super().__init__(rank, shape, transpose, seed=seed, **kwargs)
        walk = self.shape[::-1] if transpose else self.shape
        self.sketching_mats = []
        cols = 1
        for mu, (r, n) in enumerate(zip(self.true_rank, walk[:-1])):
            cols *= n
            full = random_normal_dev((r, cols), seed=(self.seed << 20) + 0x5A5A0000 + mu)
            self.sketching_mats.append(full[self.rank_min[mu]:self.rank_max[mu]])

    def _mat(self, mu) -> DevArray:
        m = self.sketching_mats[mu]
        if not isinstance(m, DevArray):
            m = self.sketching_mats[mu] = as_dev(m)
        return m

    @handle_transpose
    def sketch_dense(self, tensor):
        """reference :77-80."""
        for mu in range(len(self.sketching_mats)):
            yield self._mat(mu)

    @handle_transpose
    def sketch_tt(self, tensor):
        """(mat_mu @ X_{<=mu}).T with X_{<=mu} the dense partial product (reference :68-75)."""
        Xs = tensor.dev_cores()
        P = Xs[0].reshape(-1, Xs[0].shape[-1])
        for mu in range(len(self.sketching_mats)):
            if mu > 0:
                P = contract("ij,jkl->ikl", P, Xs[mu])
                P = P.reshape(-1, P.shape[-1])
            yield contract("ri,is->sr", self._mat(mu), P)

    @handle_transpose
    def sketch_sparse(self, tensor):
        """Columns of mat_mu at the C-order ravelled leading indices (reference :59-66)."""
        idx = tensor.dev_indices()
        order = tensor.dev_row_order
        N = tensor.nnz
        for mu in range(len(tensor.shape) - 1):
            mat = self._mat(mu).contiguous()
            rank, cols = mat.shape
            out = DevArray.empty((N, rank))
            m = mu + 1
            nat.call("ttsk_sparse_densedrm_gather", ctypes.c_void_p(mat.ptr), rank, cols,
                     ctypes.c_void_p(idx.ptr), N, (ctypes.c_int * m)(*order[:m]),
                     (ctypes.c_int64 * m)(*tensor.shape[:m]), m, ctypes.c_size_t(N),
                     ctypes.c_void_p(out.ptr), 0)
            yield out.T
