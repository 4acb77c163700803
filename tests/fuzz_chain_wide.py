"""Randomised sweep of the chunked fused chain step (ttsk_chain_step_wide, csrc/chain_wide.h) against the two einsums it
replaces -- a tool, not part of the collected suite: `python tests/fuzz_chain_wide.py SEED SECONDS` on a GPU box.
Shapes are drawn over everything the kernel's plan distinguishes: DRM ranks 1..160 in and out independently (odd
ranks, strips, every chunk count), TT ranks 1..176 on both bonds (one or two row tiles per wave, several tensors per
workgroup for few rows), both stride patterns, with and without the T side output, few and many slices, batches."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray, sync

nat.call("ttsk_init", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
P = ctypes.c_void_p


def draw(hi, small_bias=0.3):
    """sizes with extra weight on the small and the boundary values"""
    if rng.random() < small_bias:
        return int(rng.integers(1, min(hi, 24) + 1))
    return int(rng.integers(1, hi + 1))


t0, cases, skipped, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    A, A2 = draw(160), draw(160)
    K1, J = draw(176), draw(176)
    n = int(rng.integers(1, 40))
    nb = int(rng.integers(1, 10))
    if nb * n * (K1 * J + A * A2) > 6e6:      # keep the einsum reference quick
        continue
    right, wt = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    W = [rng.standard_normal((K1, A)) for _ in range(nb)]
    E = rng.standard_normal((A, n, A2))
    if right:
        X = [rng.standard_normal((J, n, K1)) for _ in range(nb)]
        strides = (n * K1, K1, 1)
        T = [np.einsum("ca,jkc->akj", w, x) for w, x in zip(W, X)]
    else:
        X = [rng.standard_normal((K1, n, J)) for _ in range(nb)]
        strides = (1, J, n * J)
        T = [np.einsum("ca,ckj->akj", w, x) for w, x in zip(W, X)]
    want = [np.einsum("akj,akb->jb", t, E) for t in T]
    dW, dX, dE = [DevArray.from_host(w) for w in W], [DevArray.from_host(x) for x in X], DevArray.from_host(E)
    dO = [DevArray.zeros((J, A2)) for _ in range(nb)]
    dT = [DevArray.zeros((A, n, J)) for _ in range(nb)] if wt else None
    arr = lambda xs: (P * nb)(*[x.ptr for x in xs])
    try:
        nat.call("ttsk_chain_step_wide", nb, n, K1, A, A2, J, arr(dW), A, arr(dX), strides[0], strides[1], strides[2],
                 X[0].size, P(dE.ptr), arr(dT) if wt else None, arr(dO), 0)
    except nat.TtskUnsupported:
        skipped += 1
        continue
    sync()
    cases += 1
    for b in range(nb):
        e = np.linalg.norm(dO[b].get() - want[b]) / max(np.linalg.norm(want[b]), 1e-300)
        if wt:
            e = max(e, np.linalg.norm(dT[b].get() - T[b]) / max(np.linalg.norm(T[b]), 1e-300))
        worst = max(worst, e)
        if not e <= 1e-12:
            print("FAIL", dict(nb=nb, n=n, K1=K1, A=A, A2=A2, J=J, right=right, wt=wt, b=b, err=e), flush=True)
print(f"fuzz_chain_wide: {cases} cases ({skipped} outside the kernel's cover), worst relative error {worst:.2e}")
