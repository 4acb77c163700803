"""Omega / Psi for sparse (COO) inputs (reference ``sparse_sketch.py:8-69``).

Omega is a skinny GEMM over the nonzeros with the entries folded in as a scale along the
contracted index; Psi is a scatter of rank-one updates into the slices selected by the mode
index, done by a run-reducing kernel (one atomic per slice change instead of the reference's
O(n_mu * nnz) boolean masks)."""
import ctypes

from .. import _native as nat
from ..device import DevArray, as_dev, contract


def _psi_call(tensor, Lv, l, Rv, r, n, idx_ptr, perm, out):
    nat.call("ttsk_sparse_psi", ctypes.c_void_p(tensor.dev_entries().ptr), idx_ptr,
             None if perm is None else ctypes.c_void_p(perm.ptr), ctypes.c_size_t(tensor.nnz),
             None if Lv is None else ctypes.c_void_p(Lv.ptr), l, None if Rv is None else ctypes.c_void_p(Rv.ptr), r,
             n, ctypes.c_void_p(out.ptr), 0)


def sketch_omega_sparse(left_sketch, right_sketch, *, tensor, **kwargs):
    """Omega = (L * entries) R^T with L (l, nnz), R (r, nnz): the Psi kernel with a single slice."""
    Lv, Rv = as_dev(left_sketch).T.contiguous(), as_dev(right_sketch).T.contiguous()
    l, r = Lv.shape[1], Rv.shape[1]
    out = DevArray.zeros((l, 1, r))
    _psi_call(tensor, Lv, l, Rv, r, 1, None, None, out)
    return out.reshape(l, r)


def sketch_psi_sparse(left_sketch, right_sketch, *, tensor, mu: int, psi_shape, **kwargs):
    """Psi[:, j, :] = sum_{e: idx_mu[e]=j} entries[e] L[:,e] R[:,e]^T, nonzeros visited in mode-index
    order (permutation cached on the tensor) so that each slice is one register-resident run."""
    l, n, r = (int(x) for x in psi_shape)
    N = tensor.nnz
    out = DevArray.zeros((l, n, r))
    if left_sketch is None and right_sketch is None:
        raise ValueError("sketch_psi_sparse needs at least one side")
    Lv = None if left_sketch is None else as_dev(left_sketch).T.contiguous()
    Rv = None if right_sketch is None else as_dev(right_sketch).T.contiguous()
    idx = tensor.dev_indices()
    row_ptr = ctypes.c_void_p(idx.ptr + tensor.dev_row_order[mu] * N * 8)
    _psi_call(tensor, Lv, l, Rv, r, n, row_ptr, tensor.dev_mode_perm(mu), out)
    return out
