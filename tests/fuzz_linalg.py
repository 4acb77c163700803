"""Randomised sweep of the small dense solves against NumPy / SciPy, a tool and not part of the collected
suite (`python tests/fuzz_linalg.py SEED SECONDS` on a GPU box): ttsk_pinv (normal-equations path, Jacobi
path, rank-deficient input), ttsk_qr_thin (CholeskyQR2 and Householder, LAPACK signs), ttsk_svd_small."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg

from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray

nat.call("ttsk_init", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
P = ctypes.c_void_p
bad = n_cases = 0
t0 = time.time()


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def report(what, *info):
    global bad
    bad += 1
    print("MISMATCH", what, *info, flush=True)


while time.time() - t0 < budget:
    n_cases += 1
    pick = rng.integers(0, 3)
    if pick == 0:                                   # pinv
        l, r = int(rng.integers(1, 140)), int(rng.integers(1, 140))
        if rng.random() < 0.15:
            l, r = int(rng.integers(1, 12)), int(rng.integers(200, 700))
        A = rng.standard_normal((l, r))
        mode = rng.integers(0, 3)
        if mode == 1 and min(l, r) > 2:             # exactly rank deficient
            k = int(rng.integers(1, min(l, r)))
            A = rng.standard_normal((l, k)) @ rng.standard_normal((k, r))
        elif mode == 2 and min(l, r) > 1:           # graded singular values down to 1e-6
            U, _, Vt = np.linalg.svd(A, full_matrices=False)
            A = (U * np.logspace(0, -6, min(l, r))) @ Vt
        d, out = DevArray.from_host(A), DevArray.empty((r, l))
        rank = ctypes.c_int(-1)
        nat.call("ttsk_pinv", P(d.ptr), l, r, -1.0, P(out.ptr), ctypes.byref(rank), 0)
        got = out.get()
        if mode == 1 and min(l, r) > 2:
            want = np.linalg.pinv(A, rcond=1e-10)
            tol = 1e-7
        else:
            want = np.linalg.pinv(A, rcond=2.3e-16)
            tol = 1e-9 * np.linalg.cond(A) if mode == 2 else 1e-8
        if not np.all(np.isfinite(got)) or rel(got, want) > max(tol, 1e-9):
            report("pinv", l, r, mode, rel(got, want), rank.value)
    elif pick == 1:                                 # thin QR with LAPACK signs
        n = int(rng.integers(1, 140))
        m = n + int(rng.integers(0, 3000)) if rng.random() < 0.8 else n
        A = rng.standard_normal((m, n))
        if rng.random() < 0.2 and n > 2:            # rank deficient: Householder path
            A[:, -1] = A[:, 0]
        d = DevArray.from_host(A)
        nat.call("ttsk_qr_thin", P(d.ptr), m, n, 0)
        Q = d.get()
        orth = np.linalg.norm(Q.T @ Q - np.eye(n))
        R = Q.T @ A
        low = np.linalg.norm(np.tril(R, -1)) / max(np.linalg.norm(R), 1e-300)
        Qref = scipy.linalg.qr(A, mode="economic")[0]
        if orth > 1e-11 * max(n, 1) or low > 1e-11:
            report("qr", m, n, orth, low)
        elif np.linalg.matrix_rank(A) == n and rel(Q, Qref) > 1e-8 * max(np.linalg.cond(A), 1):
            report("qr signs", m, n, rel(Q, Qref))
    else:                                           # small SVD
        n = int(rng.integers(1, 160))
        m = n + int(rng.integers(0, 200))
        A = rng.standard_normal((m, n))
        if rng.random() < 0.3 and n > 3:
            A[:, n // 2:] = 0.0                     # structurally zero columns stay exactly zero
        d = DevArray.from_host(A)
        US, S, Vt = DevArray.empty((m, n)), DevArray.empty((n,)), DevArray.empty((n, n))
        nat.call("ttsk_svd_small", P(d.ptr), m, n, P(US.ptr), P(S.ptr), P(Vt.ptr), 0)
        us, s, vt = US.get(), S.get(), Vt.get()
        sref = np.linalg.svd(A, compute_uv=False)
        if rel(us @ vt, A) > 1e-12 or np.linalg.norm(vt @ vt.T - np.eye(n)) > 1e-11 * n or \
                np.max(np.abs(s - sref)) > 1e-12 * sref[0] or np.any(np.diff(s) > 0) or \
                np.sum(s == 0) != np.sum(np.linalg.norm(A, axis=0) == 0):
            report("svd", m, n, rel(us @ vt, A), np.max(np.abs(s - sref)))
print(f"{n_cases} cases, {bad} bad", flush=True)
