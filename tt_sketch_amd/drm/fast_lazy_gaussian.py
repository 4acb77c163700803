"""Drop-in for the reference's Cython module ``tt_sketch/drm/fast_lazy_gaussian.pyx``: the
same four callables (``hash_int_c``, ``_inds_to_rand_double``, ``inds_to_normal``,
``inds_to_sparse_sign``; pyx:14,53,156,183), host arrays in and out, computed on the GPU."""
import ctypes

import numpy as np

from .. import _native as nat


def hash_int_c(vals) -> None:
    """In-place 64-bit mix of a uint64 array (pyx:13-37)."""
    arr = np.asarray(vals)
    if arr.dtype != np.uint64 or not arr.flags.c_contiguous:
        raise TypeError("hash_int_c needs a C-contiguous uint64 array")
    nat.call("ttsk_hash_u64", ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(arr.size))


def _prep(indices, shape):
    idx = np.ascontiguousarray(np.asarray(indices).astype(np.uint64))
    shp = np.ascontiguousarray(np.asarray(shape, dtype=np.uint64))
    if idx.ndim != 2 or idx.shape[0] != shp.size:
        raise ValueError("indices must be (len(shape), N)")
    return idx, shp


def _inds_to_rand_double(indices, shape, rank_min, rank_max, seed):
    """Flat (N*rank,) doubles in [2^-511, 2) (pyx:52-105)."""
    idx, shp = _prep(indices, shape)
    N = idx.shape[1]
    out = np.empty(N * (int(rank_max) - int(rank_min)))
    nat.call("ttsk_inds_to_rand_double", ctypes.c_void_p(idx.ctypes.data),
             ctypes.c_void_p(shp.ctypes.data), idx.shape[0], ctypes.c_size_t(N), int(rank_min),
             int(rank_max), ctypes.c_uint64(int(seed) % 2**64), ctypes.c_void_p(out.ctypes.data))
    return out


def inds_to_normal(indices, shape, rank_min, rank_max, seed):
    """(N, rank_max-rank_min) standard normals (pyx:183-202)."""
    idx, shp = _prep(indices, shape)
    N = idx.shape[1]
    out = np.empty((N, int(rank_max) - int(rank_min)))
    nat.call("ttsk_inds_to_normal", ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(shp.ctypes.data),
             idx.shape[0], ctypes.c_size_t(N), int(rank_min), int(rank_max),
             ctypes.c_uint64(int(seed) % 2**63), ctypes.c_void_p(out.ctypes.data))
    return out


def inds_to_sparse_sign(indices, shape, rank, rank_min, rank_max, non_zero_per_row, seed):
    """(N, rank_max-rank_min) int16 rows of a sparse sign matrix (pyx:156-180)."""
    idx, shp = _prep(indices, shape)
    N = idx.shape[1]
    out = np.zeros((N, int(rank_max) - int(rank_min)), dtype=np.int16)
    nat.call("ttsk_inds_to_sparse_sign", ctypes.c_void_p(idx.ctypes.data),
             ctypes.c_void_p(shp.ctypes.data), idx.shape[0], ctypes.c_size_t(N), int(rank), int(rank_min),
             int(rank_max), int(non_zero_per_row), ctypes.c_uint64(int(seed) % 2**63),
             ctypes.c_void_p(out.ctypes.data))
    return out
