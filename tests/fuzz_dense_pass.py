"""Randomised sweep of ttsk_dense_first_pass (csrc/dense_pass.hip) against numpy's einsum of the same two sums -- a tool, not part
of the collected suite: `python tests/fuzz_dense_pass.py SEED SECONDS` on a GPU box.  Random first extents (multiples of 32 up to
320), middle extents (multiples of 8), last modes (multiples of 16), left ranks 1..32, right ranks 1..64 (odd ones included):
every accumulator shape, both block forms (two kinds of workgroups on whole blocks of 64; blocks of 32), partial Z over blocks."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray, as_dev, sync

nat.call("ttsk_init", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0, cases, worst = time.time(), 0, 0.0
V = ctypes.c_void_p
while time.time() - t0 < budget:
    n0 = 32 * int(rng.integers(1, 11))
    Q = 8 * int(rng.integers(1, 40)) if rng.random() < 0.8 else 8 * int(rng.integers(200, 600))
    T = 16 * int(rng.integers(1, 9))
    ll, r = int(rng.integers(1, 33)), int(rng.integers(1, 65))
    if n0 * Q * T > 3e7:
        continue
    X, C, P = rng.standard_normal((n0, Q, T)), rng.standard_normal((n0, ll)), rng.standard_normal((Q, r))
    Xd, Cd, Pd = as_dev(X), as_dev(C), as_dev(P)
    Z, U = DevArray.empty((ll, Q, T)), DevArray.empty((n0, r, T))
    nat.call("ttsk_dense_first_pass", V(Xd.ptr), n0, Q, T, V(Cd.ptr), ll, V(Pd.ptr), r, V(Z.ptr), V(U.ptr), 0)
    sync()
    zr, ur = np.einsum("bp,bqt->pqt", C, X), np.einsum("qp,bqt->bpt", P, X)
    err = max(np.abs(Z.get() - zr).max() / np.abs(zr).max(), np.abs(U.get() - ur).max() / np.abs(ur).max())
    cases += 1
    worst = max(worst, err)
    if not err <= 1e-12:
        print("FAIL", dict(n0=n0, Q=Q, T=T, ll=ll, r=r, err=err), flush=True)
print(f"fuzz_dense_pass: {cases} cases, worst relative difference {worst:.2e}")
