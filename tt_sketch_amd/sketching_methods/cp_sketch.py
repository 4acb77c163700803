"""Omega / Psi for CP inputs (reference ``cp_sketch.py:6-36``)."""
from ..device import as_dev, contract


def sketch_omega_cp(left_sketch, right_sketch, **kwargs):
    return contract("ji,jk->ik", as_dev(left_sketch), as_dev(right_sketch))


def sketch_psi_cp(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Psi[i,k,m] = sum_j L[j,i] V_mu[k,j] R[j,m] (one rank-1 slab per CP term)."""
    V = tensor.dev_cores()[mu]
    if left_sketch is None:
        return contract("ji,il->jl", V, as_dev(right_sketch))[None]
    if right_sketch is None:
        return contract("li,kl->ik", as_dev(left_sketch), V)[:, :, None]
    W = contract("kj,jm->jkm", V, as_dev(right_sketch))
    return contract("ji,jkm->ikm", as_dev(left_sketch), W)
