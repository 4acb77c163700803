"""Size-independent checker for the DENSE sketch path: when the dense tensor is the full form of a
low-rank TT, every Omega / Psi of the reference's dense path (sketching_methods/dense_sketch.py:7-52)
factors through small matrices, so the expected sketch of an 8.6 GB tensor costs a few MFLOP.

    X^{<mu+1>} = P_mu Q_mu,   P_mu (prod n_{<=mu}, s) / Q_mu (s, prod n_{>mu}) the TT's partial products
    Omega_mu   = (A_mu P_mu) (Q_mu B_mu^T)
    Psi_mu     = (A_{mu-1} P_{mu-1}) x G_mu x (Q_mu B_mu^T)

A_mu / B_mu are the DRM's dense matrices as `DRM.sketch_dense` yields them.  For a TensorTrainDRM on the
right, B_mu's columns are C-ordered over the REVERSED modes (d-1, ..., mu+1) while the unfolding's are
C-ordered over (mu+1, ..., d-1) -- the reference's quirk.  With equal mode sizes the
t-th digit of both orders is the same number, so B_mu pairs tensor mode mu+1+t with DRM core t and
`Q_mu B_mu^T` is a forward chain (`_rb_ttdrm`).  `tests/test_oracle_golden.py` checks these formulas against
the oracle's dense path (itself pinned to the reference) at small sizes.
"""
import numpy as np


def left_chain(cores, dcores):
    """LA_mu = A_mu P_mu (l_mu, s_{mu+1}), mu = 0..d-2 (the TT-path left contraction, transposed)."""
    out = []
    L = np.einsum("jl,ja->la", dcores[0][0], cores[0][0])
    out.append(L)
    for mu in range(1, len(dcores)):
        L = np.einsum("la,ljm,ajb->mb", L, dcores[mu], cores[mu], optimize=True)
        out.append(L)
    return out


def _rb_ttdrm(cores, ecores, mu):
    """Q_mu B_mu^T (s_{mu+1}, r_mu) for a right TensorTrainDRM with cores `ecores` in ITS walking order
    (mode d-1 first) under the quirk: tensor mode mu+1+t meets DRM core t (equal mode sizes)."""
    d = len(cores)
    M = np.einsum("ajg,je->age", cores[mu + 1], ecores[0][0])
    for t in range(1, d - 1 - mu):
        M = np.einsum("age,gjh,ejf->ahf", M, cores[mu + 1 + t], ecores[t], optimize=True)
    return M[:, 0, :]


def dense_sketch_of_tt_ttdrm(cores, lcores, ecores):
    """(Psi list, Omega list) the dense path gives for X = full(cores) with TensorTrainDRMs (left cores
    `lcores`, right cores `ecores` in walking order); all modes must have the same size."""
    d = len(cores)
    assert len({c.shape[1] for c in cores}) == 1, "the digit argument needs equal mode sizes"
    LA = left_chain(cores, lcores)
    RB = [_rb_ttdrm(cores, ecores, mu) for mu in range(d - 1)]
    return _assemble(cores, LA, RB)


def dense_sketch_of_tt_matrices(cores, A, B):
    """The same for explicit DRM matrices (DenseGaussianDRM): A[mu] (l, prod n_{<=mu}),
    B[mu] (r, prod n_{>mu}) as `sketch_dense` yields them in user order."""
    d = len(cores)
    P = cores[0].reshape(cores[0].shape[1], -1)
    LA = [A[0] @ P]
    for mu in range(1, d - 1):
        P = np.einsum("ia,ajb->ijb", P, cores[mu]).reshape(-1, cores[mu].shape[2])
        LA.append(A[mu] @ P)
    Q = cores[d - 1].reshape(cores[d - 1].shape[0], -1)
    RB = [None] * (d - 1)
    RB[d - 2] = Q @ B[d - 2].T
    for mu in range(d - 3, -1, -1):
        Q = np.einsum("ajb,bc->ajc", cores[mu + 1], Q).reshape(cores[mu + 1].shape[0], -1)
        RB[mu] = Q @ B[mu].T
    return _assemble(cores, LA, RB)


def _assemble(cores, LA, RB):
    d = len(cores)
    Om = [LA[mu] @ RB[mu] for mu in range(d - 1)]
    Psi = [np.einsum("ika,ar->ikr", cores[0], RB[0])]
    for mu in range(1, d - 1):
        Psi.append(np.einsum("la,akb,br->lkr", LA[mu - 1], cores[mu], RB[mu], optimize=True))
    Psi.append(np.einsum("la,akb->lkb", LA[d - 2], cores[d - 1]))
    return Psi, Om
