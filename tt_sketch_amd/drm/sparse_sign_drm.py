"""Sparse sign DRM: every DRM row has a fixed number of +-1 entries at hashed positions.

API of the reference's ``tt_sketch/drm/sparse_sign_drm.py:11-51`` (``num_non_zero_per_row``
defaults to the full rank); sampler in csrc/sampler.hip (``fast_lazy_gaussian.pyx:121-180``).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple, Union

from .. import _native as nat
from ..device import DevArray
from ..drm_base import CanSlice, handle_transpose
from ..sketching_methods.abstract_methods import CansketchSparse


class SparseSignDRM(CansketchSparse, CanSlice):
    def __init__(self, rank: Union[Tuple[int, ...], int], shape: Tuple[int, ...], transpose: bool,
                 seed: Optional[int] = None, num_non_zero_per_row: Optional[Tuple[int, ...]] = None,
                 **kwargs) -> None:
        super().__init__(rank, shape, transpose, seed=seed, **kwargs)
        self.nnz = self.true_rank if num_non_zero_per_row is None else num_non_zero_per_row

    @handle_transpose
    def sketch_sparse(self, tensor):
        """yields (rank, nnz) matrices of 0/+-1 as fp64 (reference :34-51)."""
        idx = tensor.dev_indices()
        order = tensor.dev_row_order
        N = tensor.nnz
        for mu in range(len(tensor.shape) - 1):
            m = mu + 1
            lo, hi = self.rank_min[mu], self.rank_max[mu]
            out = DevArray.empty((N, hi - lo))
            nat.call("ttsk_sparse_sign_dev", ctypes.c_void_p(idx.ptr), N,
                     (ctypes.c_int * m)(*order[:m]), (ctypes.c_uint64 * m)(*tensor.shape[:m]), m,
                     ctypes.c_size_t(N), int(self.true_rank[mu]), lo, hi, int(self.nnz[mu]),
                     ctypes.c_uint64((mu + int(self.seed)) % 2**63), ctypes.c_void_p(out.ptr), 0)
            yield out.T
