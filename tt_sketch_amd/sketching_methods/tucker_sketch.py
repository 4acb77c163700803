"""Omega / Psi for Tucker inputs (reference ``tucker_sketch.py:9-46``)."""
import numpy as np

from ..device import as_dev, contract


def sketch_omega_tucker(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Omega_mu = L^T core^{<mu+1>} R."""
    _, core = tensor.dev_parts()
    Cm = core.reshape(int(np.prod(core.shape[:mu + 1], dtype=np.int64)), -1)
    return contract("ib,bj->ij", contract("ai,ab->ib", as_dev(left_sketch), Cm), as_dev(right_sketch))


def sketch_psi_tucker(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """(L^T x_1 core x_3 R) followed by the mode product with the factor U_mu."""
    Us, core = tensor.dev_parts()
    ld = 1 if left_sketch is None else as_dev(left_sketch).shape[0]
    rd = 1 if right_sketch is None else as_dev(right_sketch).shape[0]
    C3 = core.reshape(ld, tensor.rank[mu], rd)
    if left_sketch is None:
        P = contract("ijk,kl->ijl", C3, as_dev(right_sketch))
    elif right_sketch is None:
        P = contract("ji,jkl->ikl", as_dev(left_sketch), C3)
    else:
        P = contract("ikl,lm->ikm", contract("ji,jkl->ikl", as_dev(left_sketch), C3),
                     as_dev(right_sketch))
    return contract("ijk,jl->ilk", P, Us[mu])
