"""CPU restatement of the reference's TT-GMRES building blocks on plain lists of NumPy cores --
TEST INFRASTRUCTURE ONLY (imported by tests/ alone; the product never routes through it).

Follows /root/reference/tt_sketch/tt_gmres.py (MPO.__call__ :90-101, TTPrecond :137-168,
round_tt_sum "exact"/"pairwise" :274-285, tt_sum_gmres :308-432) and tensor.py (add :503-525,
round :446-484, orthogonalize :559-572, dot :527-540).  PINNED against tests/golden/gmres_case.npz,
which tests/golden/make_golden.py produced by running the reference itself.
"""
import numpy as np
import scipy.linalg


def mpo_apply(mpo, tt):
    out = []
    for M, C in zip(mpo, tt):
        P = np.einsum("ijkl,ajb->iaklb", M, C)
        out.append(P.reshape(P.shape[0] * P.shape[1], P.shape[2], -1))
    return out


def precond_apply(A, tt, mode, backward=True):
    out = [c.copy() for c in tt]
    C = out[mode]
    mat = C.transpose(1, 0, 2).reshape(C.shape[1], -1)
    if backward:
        Q, R = np.linalg.qr(A)
        mat = scipy.linalg.solve_triangular(R, Q.T @ mat)
    else:
        mat = A @ mat
    out[mode] = mat.reshape(C.shape[1], C.shape[0], C.shape[2]).transpose(1, 0, 2)
    return out


def tt_scale(tt, c):
    return list(tt[:-1]) + [tt[-1] * c]


def tt_add(a, b):
    d = len(a)
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        if k == 0:
            out.append(np.concatenate((x, y), axis=2))
        elif k == d - 1:
            out.append(np.concatenate((x, y), axis=0))
        else:
            blk = np.zeros((x.shape[0] + y.shape[0], x.shape[1], x.shape[2] + y.shape[2]))
            blk[:x.shape[0], :, :x.shape[2]] = x
            blk[x.shape[0]:, :, x.shape[2]:] = y
            out.append(blk)
    return out


def tt_orth(tt):
    out, carry = [], None
    for k, C in enumerate(tt):
        if carry is not None:
            C = np.einsum("ij,jkl->ikl", carry, C)
        if k < len(tt) - 1:
            Q, carry = np.linalg.qr(C.reshape(-1, C.shape[2]))
            out.append(Q.reshape(C.shape[0], C.shape[1], -1))
        else:
            out.append(C)
    return out


def tt_norm(tt):
    return float(np.linalg.norm(tt_orth(tt)[-1]))


def tt_dot(a, b):
    acc = np.ones((1, 1))
    for x, y in zip(a, b):
        acc = np.einsum("ij,ika,jkb->ab", acc, x, y, optimize=True)
    return float(acc.sum())


def trim_rank(rank, shape):
    """Feasible TT ranks <= rank (reference utils.py:121-146)."""
    rank = list(rank)
    for _ in range(100):
        old = list(rank)
        full = [1] + rank + [1]
        for k in range(len(rank)):
            rank[k] = min(rank[k], full[k] * shape[k], shape[k + 1] * full[k + 2])
            full = [1] + rank + [1]
        if old == rank:
            break
    return tuple(rank)


def tt_round(tt, eps=None, max_rank=None):
    tt = tt_orth(tt)
    shape = tuple(c.shape[1] for c in tt)
    eps = 0 if eps is None else eps
    cur = tuple(c.shape[0] for c in tt[1:])
    if max_rank is None:
        cap = cur
    elif isinstance(max_rank, (int, np.integer)):
        cap = (int(max_rank),) * (len(tt) - 1)
    else:
        cap = tuple(max_rank)
    cap = trim_rank(cap, shape)
    out, carry = [], None
    for k in range(len(tt) - 1, -1, -1):
        C = tt[k]
        if carry is not None:
            C = np.einsum("ijk,kl->ijl", C, carry)
        if k == 0:
            out.append(C)
            continue
        U, S, Vt = np.linalg.svd(C.reshape(C.shape[0], -1), full_matrices=False)
        r = max(1, min(int(np.sum(S > S[0] * eps)), cap[k - 1]))
        carry = U[:, :r] * S[:r]
        out.append(Vt[:r].reshape(r, C.shape[1], C.shape[2]))
    return out[::-1]


def round_sum(terms, max_rank, eps=None, method="exact"):
    if method == "exact":
        tt = terms[0]
        for t in terms[1:]:
            tt = tt_add(tt, t)
        return tt_round(tt, eps, max_rank)
    if method == "pairwise":
        tt = terms[0]
        for t in terms[1:]:
            tt = tt_round(tt_add(tt, t), eps, max_rank)
        return tt
    raise ValueError(method)


def gmres(maps, b, max_rank, precond=None, tolerance=1e-6, maxiter=100, method="pairwise"):
    """maps: list of MPO core lists; precond: (matrix, mode) or None.  Returns (x cores, history)."""
    shape = tuple(c.shape[1] for c in b)
    max_rank = trim_rank((max_rank,) * (len(shape) - 1) if np.isscalar(max_rank) else max_rank, shape)
    x0 = [np.zeros((1, n, 1)) for n in shape]

    def pre(t):
        return t if precond is None else precond_apply(precond[0], t, precond[1])

    def operator(x):
        return [pre(mpo_apply(m, x)) for m in maps]

    b_norm = tt_norm(b)
    res = round_sum([pre(b)] + [tt_scale(t, -1) for t in operator(x0)], max_rank, None, method)
    res_norm = beta = tt_norm(res)
    basis = [tt_scale(res, 1 / beta)]
    H = np.zeros((maxiter + 1, maxiter))
    hist = {"residual_norm": [res_norm / b_norm], "rank": [tuple(c.shape[0] for c in res[1:])],
            "w_norm": [tt_norm(basis[-1])]}
    for j in range(maxiter):
        delta = tolerance / (res_norm / beta)
        w = round_sum(operator(basis[-1]), max_rank, delta, method)
        for i in range(j + 1):
            H[i, j] = tt_dot(w, basis[i])
        w = round_sum([w] + [tt_scale(v, -H[i, j]) for i, v in enumerate(basis[:j + 1])], max_rank,
                      delta, method)
        H[j + 1, j] = tt_norm(w)
        basis.append(tt_scale(w, 1 / H[j + 1, j]))
        e1 = np.zeros(j + 2)
        e1[0] = beta
        y, (res_norm,), _, _ = np.linalg.lstsq(H[:j + 2, :j + 1], e1, rcond=None)
        hist["residual_norm"].append(np.sqrt(res_norm) / b_norm)
        hist["rank"].append(tuple(c.shape[0] for c in w[1:]))
        hist["w_norm"].append(H[j + 1, j])
        if res_norm / b_norm < tolerance:
            break
    y = y[:j + 1]
    x = round_sum([x0] + [tt_scale(v, c) for v, c in zip(basis[:j + 1], y)], max_rank, None, method)
    hist["H_matrix"], hist["y"] = H, y
    return x, hist
