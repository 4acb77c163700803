"""Omega / Psi for dense inputs (reference ``dense_sketch.py:7-52``): DRM matrices applied to
C-order unfoldings of the tensor, which are zero-copy reshapes of the resident array.

``Omega_mu = (A_mu X^{<mu+1>}) B_mu^T`` and ``Psi_{mu+1} = (A_mu X^{<mu+1>}) x B_{mu+1}^T`` start with
the same product -- the one full pass over the tensor.  The reference recomputes it; here it is kept
for the duration of one ``general_sketch`` (5 instead of 9 passes over X for d = 5).
"""
import numpy as np

from ..device import as_dev, contract

_shared = {}     # key -> (A, X, A X^{<mu+1>}); cleared by general_sketch_device


def clear_shared() -> None:
    _shared.clear()


def _unfold(X, k):
    return X.reshape(int(np.prod(X.shape[:k], dtype=np.int64)), -1)


def _left_product(A, X, mu):
    """A X^{<mu+1>}, (l, prod n_{>mu}); A is the left sketch of modes 0..mu."""
    key = (id(A.buf), A.offset, A.shape, A.strides, id(X.buf), X.offset, mu)
    hit = _shared.get(key)
    if hit is None:
        hit = _shared[key] = (A, X, contract("ip,pq->iq", A, _unfold(X, mu + 1)))
    return hit[2]


def sketch_omega_dense(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Omega_mu = A_mu X^{<mu+1>} B_mu^T."""
    X = tensor.dev_data()
    A, B = as_dev(left_sketch), as_dev(right_sketch)
    if A.shape[0] <= B.shape[0]:
        return contract("iq,jq->ij", _left_product(A, X, mu), B)
    return contract("ip,pj->ij", A, contract("pq,jq->pj", _unfold(X, mu + 1), B))


def sketch_psi_dense(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    X = tensor.dev_data()
    d = X.ndim
    if left_sketch is None:
        return contract("kq,mq->km", _unfold(X, 1), as_dev(right_sketch))[None]
    if right_sketch is None:
        return contract("ip,pk->ik", as_dev(left_sketch), _unfold(X, d - 1))[:, :, None]
    A, B = as_dev(left_sketch), as_dev(right_sketch)
    J = int(np.prod(X.shape[:mu], dtype=np.int64))
    X3 = X.reshape(J, X.shape[mu], -1)
    K, Lr = X3.shape[1], X3.shape[2]
    l, r = A.shape[0], B.shape[0]
    key = (id(A.buf), A.offset, A.shape, A.strides, id(X.buf), X.offset, mu - 1)
    if key in _shared or l * J * K * Lr + l * K * Lr * r <= J * K * Lr * r + l * J * K * r:
        T = _left_product(A, X, mu - 1).reshape(l, K, Lr)       # shared with Omega_{mu-1}
        return contract("ikl,ml->ikm", T, B)
    return contract("ij,jkm->ikm", A, contract("jkl,ml->jkm", X3, B))
