"""Omega / Psi for tensor-train inputs (reference ``tensor_train_sketch.py:8-35``)."""
from ..device import as_dev, contract


def sketch_omega_tt(left_sketch, right_sketch, **kwargs):
    """Omega_mu = L_mu^T R_mu."""
    return contract("ji,jk->ik", as_dev(left_sketch), as_dev(right_sketch))


def sketch_psi_tt(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Psi_mu = L_{mu-1}^T X_mu R_mu; the cheaper of the two association orders."""
    X = tensor.dev_cores()[mu]
    if left_sketch is None:
        return contract("ijk,kl->ijl", X, as_dev(right_sketch))
    if right_sketch is None:
        return contract("ji,jkl->ikl", as_dev(left_sketch), X)
    L, R = as_dev(left_sketch), as_dev(right_sketch)
    s, n, sp = X.shape
    l, r = L.shape[1], R.shape[1]
    if l * s * n * sp + l * n * sp * r <= s * n * sp * r + l * s * n * r:
        return contract("ikl,lm->ikm", contract("ji,jkl->ikl", L, X), R)
    return contract("ji,jkm->ikm", L, contract("jkl,lm->jkm", X, R))
