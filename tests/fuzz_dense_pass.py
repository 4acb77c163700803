"""Randomised shapes for ttsk_dense_first_pass against numpy's einsum (not collected by pytest: run on a GPU box as
``python tests/fuzz_dense_pass.py [seconds] [seed]``).  Every shape in the kernel's cover: first mode 32 / 64 / a multiple of 64, last mode a
multiple of 16, middle extent a multiple of 8 (one tile up to hundreds per q range, ragged ranges), ranks 1..20 / 2..40."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, as_dev, sync
    nat.call("ttsk_init", 0)
    rng = np.random.default_rng(seed)
    V = ctypes.c_void_p
    t_end, cases, worst = time.time() + seconds, 0, 0.0
    while time.time() < t_end:
        n0 = int(rng.choice([32, 64, 64, 128, 192, 320]))            # beyond 64: blocks of 64 with partial Z
        T = 16 * int(rng.integers(1, 7))
        Q = 8 * int(rng.integers(1, 41)) if rng.random() < 0.8 else 8 * int(rng.integers(100, 1200))
        ll, r = int(rng.integers(1, 21)), 2 * int(rng.integers(1, 21))
        X = rng.standard_normal((n0, Q, T))
        C = rng.standard_normal((n0, ll))
        P = rng.standard_normal((Q, r))
        Xd, Cd, Pd = as_dev(X), as_dev(C), as_dev(P)
        Z, U = DevArray.empty((ll, Q, T)), DevArray.empty((n0, r, T))
        nat.call("ttsk_dense_first_pass", V(Xd.ptr), n0, Q, T, V(Cd.ptr), ll, V(Pd.ptr), r, V(Z.ptr), V(U.ptr), 0)
        sync()
        wz, wu = np.einsum("bp,bqt->pqt", C, X), np.einsum("qp,bqt->bpt", P, X)
        ez = float(np.max(np.abs(Z.get() - wz)) / np.max(np.abs(wz)))
        eu = float(np.max(np.abs(U.get() - wu)) / np.max(np.abs(wu)))
        worst = max(worst, ez, eu)
        if not (ez < 1e-13 and eu < 1e-13):
            print(f"FAIL n0={n0} Q={Q} T={T} ll={ll} r={r}: Z {ez:.2e} U {eu:.2e}")
            sys.exit(1)
        cases += 1
    print(f"fuzz_dense_pass: {cases} cases, worst relative deviation {worst:.2e}")


if __name__ == "__main__":
    main()
