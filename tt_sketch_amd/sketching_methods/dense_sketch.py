"""Omega / Psi for dense inputs (reference ``dense_sketch.py:7-52``): DRM matrices applied to
C-order unfoldings of the tensor, which are zero-copy reshapes of the resident array.

Generic DRM matrices (``DenseGaussianDRM``, user plug-ins):
``Omega_mu = (A_mu X^{<mu+1>}) B_mu^T`` and ``Psi_{mu+1} = (A_mu X^{<mu+1>}) x B_{mu+1}^T`` start with
the same product -- the one full pass over the tensor.  The reference recomputes it; here it is kept
for the duration of one ``general_sketch`` (5 instead of 9 passes over X for d = 5).

``TensorTrainDRM`` hands out recipes (``ChainedUnfolding``) instead of matrices.  Left: A_mu = A_{mu-1} x D_mu,
so ``Z_mu = A_mu X^{<mu+1>}`` follows from ``Z_{mu-1}`` by contracting one mode with one core -- only ``Z_0``
reads the tensor, the (rho x n^mu) matrices are never formed -- and
``Omega_mu[p, m] = sum_{i,k} D_mu[i, k, p] Psi_mu[i, k, m]`` needs no pass at all.  Right: see
``_right_product``.  Z_0 and Psi_0 -- the two products that read X -- come from ONE pass where ``ttsk_dense_first_pass``
covers the shape (``_first_pass``), from two otherwise; Z_1 and Psi_1 read Z_0 (rho/n of X) -- once through the same kernel
where Z_0 is large, twice otherwise; no
matrix larger than n^{d-2} x rho exists; same numbers as the reference up to the order of summation
(tests: golden fixtures at 1e-12, C2 at full size).
"""
import os

import numpy as np

from ..device import DevArray, as_dev, contract

_shared = {}     # key -> (objects the key's ids refer to ..., product); cleared by general_sketch_device
_perm_cache = {}  # (buffer of a DRM matrix A_3, view) -> (that buffer, its transposed copy for ttsk_dense_left_pass)


def clear_shared() -> None:
    _shared.clear()


class ChainedUnfolding:
    """The sketching matrix (rho_mu x n_0...n_mu) of a TensorTrainDRM after mu + 1 modes of the (for a right
    DRM: transposed) tensor, as a recipe: the previous recipe and the core D_mu (rho_{mu-1}, n_mu, rho_mu)
    (reference tensor_train_drm.py:109-122 forms the matrix).  ``materialise`` / ``get`` / ``np.asarray``
    give the matrix the reference yields."""

    def __init__(self, prev, core: DevArray):
        self.prev, self.core = prev, core
        self.depth = 0 if prev is None else prev.depth + 1
        self.shape = (core.shape[2], core.shape[1] * (1 if prev is None else prev.shape[1]))
        self._P = None          # (n_0...n_mu, rho_mu), contiguous

    def _rows(self) -> DevArray:
        if self._P is None:
            if self.prev is None:
                self._P = self.core.reshape(-1, self.core.shape[-1])
            else:
                P = contract("ij,jkl->ikl", self.prev._rows(), self.core)
                self._P = P.reshape(-1, P.shape[-1])
        return self._P

    def materialise(self) -> DevArray:
        return self._rows().T

    def get(self) -> np.ndarray:
        return self.materialise().get()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)


def _unfold(X, k):
    return X.reshape(int(np.prod(X.shape[:k], dtype=np.int64)), -1)


def _norm(x):
    """None and recipes as they are; anything else as a device matrix."""
    return x if x is None or isinstance(x, ChainedUnfolding) else as_dev(x)


def _mat(x) -> DevArray:
    return x.materialise() if isinstance(x, ChainedUnfolding) else x


def _ident(x):
    """What identifies a sketching matrix for the lifetime of one general_sketch."""
    if x is None or isinstance(x, ChainedUnfolding):
        return id(x)
    return (id(x.buf), x.offset, x.shape, x.strides)


def prepare_left(tensor, left_contractions) -> bool:
    """DRM MATRICES on the left (DenseGaussianDRM, plug-ins), order >= 5: the first four left products Z_mu = A_mu X^{<mu+1>}
    from ONE read of the tensor (``ttsk_dense_left_pass``) instead of one pass each; they land in ``_shared`` under the
    keys ``_left_product`` looks them up by.  Called by ``general_sketch`` before the Omega loop."""
    import ctypes
    from .. import _native as nat
    if os.environ.get("TTSK_DENSE_LEFT_PASS", "1") == "0":
        return False
    X = tensor.dev_data()
    A = [_norm(x) for x in left_contractions]
    if X.ndim < 5 or len(A) < 4 or any(isinstance(a, ChainedUnfolding) or a is None for a in A[:4]) or not X.is_contiguous():
        return False
    n0, n1, n2, n3 = (int(v) for v in X.shape[:4])
    n4 = int(np.prod(X.shape[4:], dtype=np.int64))
    l = int(A[0].shape[0])
    C = n3 * n4
    if any(int(A[mu].shape[0]) != l for mu in range(4)) or l > 20 or n0 % 4 or C % 512 or n4 not in (64, 128, 256, 512):
        return False
    if [tuple(A[mu].shape) for mu in range(4)] != [(l, n0), (l, n0 * n1), (l, n0 * n1 * n2), (l, n0 * n1 * n2 * n3)]:
        return False
    A0 = A[0].contiguous()
    # the transposed copies the pass reads (i0 fastest).  That of A_3 is as large as A_3 itself (2.7 GB at C2): kept with the
    # matrix it was made from for as long as that matrix lives (a DRM reused over many sketches pays for it once)
    A1t = A[1].reshape(l, n0, n1).transpose(0, 2, 1).contiguous()
    A2t = A[2].reshape(l, n0, n1, n2).transpose(0, 3, 2, 1).contiguous()
    key = (id(A[3].buf), A[3].offset, tuple(A[3].shape), tuple(A[3].strides))
    hit = _perm_cache.get(key)
    if hit is None or hit[0] is not A[3].buf:
        if len(_perm_cache) >= 4:
            _perm_cache.clear()
        hit = _perm_cache[key] = (A[3].buf, A[3].reshape(l, n0, n1, n2, n3).transpose(0, 3, 2, 4, 1).contiguous())
    A3 = hit[1]
    Z0, Z1 = DevArray.empty((l, n1 * n2 * C)), DevArray.empty((l, n2 * C))
    Z2, E3 = DevArray.empty((l, C)), DevArray.empty((l, C))
    V = ctypes.c_void_p
    try:
        nat.call("ttsk_dense_left_pass", V(X.ptr), n0, n1, n2, C, n4, l, V(A0.ptr), V(A1t.ptr), V(A2t.ptr), V(A3.ptr), V(Z0.ptr),
                 V(Z1.ptr), V(Z2.ptr), V(E3.ptr), 0)
    except nat.TtskUnsupported:
        return False
    Z3 = contract("aij,i->aj", E3.reshape(l, n3, n4), DevArray.from_host(np.ones(n3)))
    for mu, Z in enumerate((Z0, Z1, Z2, Z3)):
        _shared[("left", _ident(A[mu]), id(X.buf), X.offset, mu)] = (A[mu], X, Z)
    return True


def _left_product(A, X, mu):
    """A X^{<mu+1>}, (l, prod n_{>mu}); A is the left sketch of modes 0..mu."""
    key = ("left", _ident(A), id(X.buf), X.offset, mu)
    hit = _shared.get(key)
    if hit is None:
        if not isinstance(A, ChainedUnfolding):
            Z = contract("ip,pq->iq", A, _unfold(X, mu + 1))
        elif A.prev is None:
            Z = contract("kp,kq->pq", A.core[0], _unfold(X, 1))             # the one pass over X
        else:
            rho, n, _ = A.core.shape
            Zp = _left_product(A.prev, X, mu - 1).reshape(rho, n, -1)
            step = max(1, 128 // n)
            if Zp.shape[2] >= 8192 and n <= 128 and rho > step:
                # contracted extent rho*n in slices of <= 128: the streaming small-K kernel with an
                # accumulating output (3.5 TB/s) instead of the generic tiles (1.1 TB/s at rho*n = 1280)
                Z = None
                for j0 in range(0, rho, step):
                    Z = contract("jkp,jkq->pq", A.core[j0:j0 + step], Zp[j0:j0 + step], out=Z,
                                 accumulate=Z is not None)
            else:
                Z = contract("jkp,jkq->pq", A.core, Zp)
        hit = _shared[key] = (A, X, Z)
    return hit[2]


def _right_product(S: DevArray, B) -> DevArray:
    """S B^T for a (rows, cols) view S of the tensor or of a left product and the right sketch B (r, cols).

    The reference pairs B's columns with S's position by position, and a TensorTrainDRM's matrix is
    B[m, (q, t)] = sum_p P[q, p] D[p, t, m] with P the matrix of one core less (tensor_train_drm.py:109-122
    on the transposed tensor), t the LAST mode of S's columns.  So
    S B^T = (sum_q S[b, q, t] P[q, p]) x D: one streamed pass with an (n^{k-1} x rho) matrix instead of
    the (n^k x r) one -- for Psi_0 of C2 84 MB next to the tensor instead of 5.4 GB, and no product to
    form it."""
    if isinstance(B, ChainedUnfolding) and B.prev is not None:
        rho, t, r = B.core.shape
        rows, cols = S.shape
        q = cols // t
        if q * r > rows * rho:
            U = contract("qp,bqt->bpt", B.prev._rows(), S.reshape(rows, q, t))
            return contract("bpt,ptm->bm", U, B.core)
    return contract("bq,mq->bm", S, _mat(B))


# beyond mode 0 the fused pair only pays on a large left product (C2: 2.7 GB); TTSK_DENSE_PAIR_MIN_BYTES for A/B runs
_PAIR_MIN_BYTES = int(os.environ.get("TTSK_DENSE_PAIR_MIN_BYTES", 1 << 28))


def _first_pass(A, B, X, mu: int = 0) -> bool:
    """Z_mu and Psi_mu -- the two products that read the tensor (mu = 0) or the previous left product Z_{mu-1} (mu > 0) --
    from ONE read of it (``ttsk_dense_first_pass``, csrc/dense_pass.hip) when both DRMs are tensor trains and the shape is in
    the kernel's cover.  Both land in ``_shared`` under the keys ``_left_product`` / ``_psi_chained`` look them up by."""
    import ctypes
    from .. import _native as nat
    if not (isinstance(A, ChainedUnfolding) and A.depth == mu and isinstance(B, ChainedUnfolding) and B.prev is not None):
        return False
    if os.environ.get("TTSK_DENSE_ONE_PASS", "1") == "0":
        return False
    if mu == 0:
        S, rows = X, int(X.shape[0])
        if X.ndim < 3 or not X.is_contiguous():
            return False
    else:
        # rows (rho_{mu-1}, n_mu) of the previous left product: the same two sums one level down
        if ("left", _ident(A.prev), id(X.buf), X.offset, mu - 1) not in _shared:
            return False
        S = _left_product(A.prev, X, mu - 1)
        rows = int(A.core.shape[0]) * int(A.core.shape[1])
        if S.size * 8 < _PAIR_MIN_BYTES or not S.is_contiguous() or S.size % rows:
            return False
    # (B's columns meet the tensor's position by position -- _right_product -- so the last chained core's mode size is
    # what splits the columns, whatever the tensor's own last mode is)
    T = int(B.core.shape[1])
    cols = S.size // rows
    if A.core.shape[0] * A.core.shape[1] != rows or cols % T:
        return False
    Q = cols // T
    ll, rho = int(A.core.shape[2]), int(B.core.shape[0])
    if rows % 32 or T % 16 or Q % 8 or ll > 32 or rho > 64 or Q * T >= 1 << 27:
        return False
    P = B.prev._rows().contiguous()
    if tuple(P.shape) != (Q, rho):
        return False
    C = A.core.reshape(rows, ll).contiguous()
    Z, U = DevArray.empty((ll, Q * T)), DevArray.empty((rows, rho, T))
    V = ctypes.c_void_p
    try:
        nat.call("ttsk_dense_first_pass", V(S.ptr), rows, Q, T, V(C.ptr), ll, V(P.ptr), rho, V(Z.ptr), V(U.ptr), 0)
    except nat.TtskUnsupported:
        return False
    _shared[("left", _ident(A), id(X.buf), X.offset, mu)] = (A, X, Z)
    Psi = contract("bpt,ptm->bm", U, B.core)
    Psi = Psi[None] if mu == 0 else Psi.reshape(int(A.core.shape[0]), int(A.core.shape[1]), -1)
    _shared[("psi", _ident(A.prev), _ident(B), id(X.buf), X.offset, mu)] = (A.prev, B, X, Psi)
    return True


def _psi_chained(Aprev, B, X, mu, keep=True):
    """Psi_mu (rho_{mu-1}, n_mu, r_mu) for a chained left sketch; computed once per (left, right, mu)."""
    key = ("psi", _ident(Aprev), _ident(B), id(X.buf), X.offset, mu)
    hit = _shared.get(key) if keep else _shared.pop(key, None)
    if hit is not None:
        return hit[-1]
    if Aprev is None:
        P = (_psi0_generic(X, B) if (B is not None and not isinstance(B, ChainedUnfolding)) else _right_product(_unfold(X, 1), B))[None]
    else:
        Z = _left_product(Aprev, X, mu - 1)
        if B is None:
            P = Z[:, :, None]
        else:
            rho, n = Z.shape[0], X.shape[mu]
            P = _right_product(Z.reshape(rho * n, -1), B).reshape(rho, n, -1)
    if keep:
        _shared[key] = (Aprev, B, X, P)
    return P


def _psi0_generic(X, B) -> DevArray:
    """X^{<1>} B_0^T (n_0 x r) for a generic right matrix, shared between Psi_0 and Omega_0."""
    key = ("psi0", _ident(B), id(X.buf), X.offset)
    hit = _shared.get(key)
    if hit is None:
        hit = _shared[key] = (B, X, _right_product(_unfold(X, 1), B))
    return hit[2]


def sketch_omega_dense(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    """Omega_mu = A_mu X^{<mu+1>} B_mu^T."""
    X = tensor.dev_data()
    A, B = _norm(left_sketch), _norm(right_sketch)
    if isinstance(A, ChainedUnfolding) and A.depth == mu:
        if ("psi", _ident(A.prev), _ident(B), id(X.buf), X.offset, mu) not in _shared:
            _first_pass(A, B, X, mu)
        Psi = _psi_chained(A.prev, B, X, mu)
        return contract("ikp,ikm->pm", A.core, Psi)
    A = _mat(A)
    if mu == 0 and B is not None and not isinstance(B, ChainedUnfolding):
        # Omega_0 = A_0 X^{<1>} B_0^T = A_0 Psi_0 (Psi_0 = X^{<1>} B_0^T, tensor_train... dense_sketch.py:29-36 with no left
        # sketch): Psi_0 has to read the tensor and B_0 anyway; Omega_0 from it is an (l x n_0) x (n_0 x r) product instead of
        # a pass over the first left product (2.7 GB at C2) and B_0 (5.4 GB) once more
        return contract("ip,pj->ij", A, _psi0_generic(X, B))
    if A.shape[0] <= B.shape[0]:
        return _right_product(_left_product(A, X, mu), B)
    return contract("ip,pj->ij", A, _right_product(_unfold(X, mu + 1), B))


def sketch_psi_dense(left_sketch, right_sketch, *, tensor, mu: int, **kwargs):
    X = tensor.dev_data()
    d = X.ndim
    A, B = _norm(left_sketch), _norm(right_sketch)
    if A is None or (isinstance(A, ChainedUnfolding) and A.depth == mu - 1):
        # (Psi_0 is shared with Omega_0 of a chained left sketch, which asked for it first)
        return _psi_chained(A, B, X, mu, keep=False)
    A = _mat(A)
    if B is None:
        # Psi_{d-1} = A_{d-2} X^{<d-1>} IS the left product Omega_{d-2} was formed from: shared, not a further pass over X
        return _left_product(A, X, d - 2)[:, :, None]
    J = int(np.prod(X.shape[:mu], dtype=np.int64))
    K, Lr = X.shape[mu], int(np.prod(X.shape[mu + 1:], dtype=np.int64))
    l, r = A.shape[0], B.shape[0]
    key = ("left", _ident(A), id(X.buf), X.offset, mu - 1)
    if key in _shared or l * J * K * Lr + l * K * Lr * r <= J * K * Lr * r + l * J * K * r:
        T = _left_product(A, X, mu - 1)                          # shared with Omega_{mu-1}
        return _right_product(T.reshape(l * K, Lr), B).reshape(l, K, r)
    return contract("ij,jkm->ikm", A, _right_product(X.reshape(J * K, Lr), B).reshape(J, K, r))
