"""Omega / Psi for dense inputs (reference ``dense_sketch.py:7-52``): DRM matrices applied to
C-order unfoldings of the tensor, which are zero-copy reshapes of the resident array."""
import numpy as np

from ..device import as_dev, contract


def _unfold(X, k):
    return X.reshape(int(np.prod(X.shape[:k], dtype=np.int64)), -1)


def sketch_omega_dense(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Omega_mu = A_mu X^{<mu+1>} B_mu^T."""
    Xm = _unfold(tensor.dev_data(), mu + 1)
    A, B = as_dev(left_sketch), as_dev(right_sketch)
    if A.shape[0] <= B.shape[0]:
        return contract("iq,jq->ij", contract("ip,pq->iq", A, Xm), B)
    return contract("ip,pj->ij", A, contract("pq,jq->pj", Xm, B))


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
    if l * J * K * Lr + l * K * Lr * r <= J * K * Lr * r + l * J * K * r:
        return contract("ikl,ml->ikm", contract("ij,jkl->ikl", A, X3), B)
    return contract("ij,jkm->ikm", A, contract("jkl,ml->jkm", X3, B))
