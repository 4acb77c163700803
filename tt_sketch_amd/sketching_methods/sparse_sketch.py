"""Omega / Psi for sparse (COO) inputs (reference ``sparse_sketch.py:8-69``).

Omega is a skinny GEMM over the nonzeros with the entries folded in as a scale along the
contracted index; Psi is a scatter of rank-one updates into the slices selected by the mode
index, done by a run-reducing kernel (one atomic per slice change instead of the reference's
O(n_mu * nnz) boolean masks)."""
import ctypes

from .. import _native as nat
from ..device import DevArray, as_dev, contract


def sketch_omega_sparse(left_sketch, right_sketch, *, tensor, **kwargs):
    """Omega = (L * entries) R^T with L (l, nnz), R (r, nnz)."""
    return contract("ie,je->ij", as_dev(left_sketch), as_dev(right_sketch),
                    k_scale=tensor.dev_entries())


def sketch_psi_sparse(left_sketch, right_sketch, *, tensor, mu: int, psi_shape, **kwargs):
    """Psi[:, j, :] = sum_{e: idx_mu[e]=j} entries[e] L[:,e] R[:,e]^T."""
    l, n, r = (int(x) for x in psi_shape)
    N = tensor.nnz
    out = DevArray.zeros((l, n, r))
    if left_sketch is None and right_sketch is None:
        raise ValueError("sketch_psi_sparse needs at least one side")
    Lv = None if left_sketch is None else as_dev(left_sketch).T.contiguous()
    Rv = None if right_sketch is None else as_dev(right_sketch).T.contiguous()
    idx = tensor.dev_indices()
    row_ptr = idx.ptr + tensor.dev_row_order[mu] * N * 8
    nat.call("ttsk_sparse_psi", ctypes.c_void_p(tensor.dev_entries().ptr), ctypes.c_void_p(row_ptr),
             ctypes.c_size_t(N), None if Lv is None else ctypes.c_void_p(Lv.ptr), l,
             None if Rv is None else ctypes.c_void_p(Rv.ptr), r, n, ctypes.c_void_p(out.ptr), 0)
    return out
