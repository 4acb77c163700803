"""'Sparse' Gaussian DRM: entries of a dense Gaussian DRM computed on demand from a hash of
(flat index, column, seed) -- only at the nonzero positions of a sparse tensor.

API of the reference's ``tt_sketch/drm/sparse_gaussian_drm.py:11-44``; the sampler itself is
the device restatement of ``fast_lazy_gaussian.pyx`` (csrc/sampler.hip).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple, Union

from .. import _native as nat
from ..device import DevArray
from ..drm_base import CanIncreaseRank, handle_transpose
from ..sketching_methods.abstract_methods import CansketchSparse


class SparseGaussianDRM(CansketchSparse, CanIncreaseRank):
    def __init__(self, rank: Union[Tuple[int, ...], int], shape: Tuple[int, ...], transpose: bool,
                 seed: Optional[int] = None, **kwargs) -> None:
        super().__init__(rank, shape, transpose, seed=seed, **kwargs)

    @handle_transpose
    def sketch_sparse(self, tensor):
        """G[e,k] = ndtri(u(hash(flat_e + hash(k) + seed_mu))), seed_mu = (mu+seed) mod 2^63;
        yields (rank, nnz) (reference :29-44)."""
        idx = tensor.dev_indices()
        order = tensor.dev_row_order
        N = tensor.nnz
        for mu in range(len(tensor.shape) - 1):
            m = mu + 1
            lo, hi = self.rank_min[mu], self.rank_max[mu]
            out = DevArray.empty((N, hi - lo))
            nat.call("ttsk_sparse_normal_dev", ctypes.c_void_p(idx.ptr), N,
                     (ctypes.c_int * m)(*order[:m]), (ctypes.c_uint64 * m)(*tensor.shape[:m]), m,
                     ctypes.c_size_t(N), lo, hi, ctypes.c_uint64((mu + int(self.seed)) % 2**63),
                     ctypes.c_void_p(out.ptr), 0)
            yield out.T
