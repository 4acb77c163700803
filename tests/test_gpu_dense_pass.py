"""ttsk_dense_first_pass: the two products of a dense-tensor sketch with tensor-train DRMs that read the tensor
(reference dense_sketch.py:7-52 with the matrices of tensor_train_drm.py:109-122), from one read of it.

The kernel against numpy's einsum of the same two sums (fp64, 1e-13 of the largest entry: same products, another
order of summation), then through the public API against the oracle's dense path -- which forms the DRM matrices
and the unfoldings as the reference does -- and against the two-pass device path.
"""
import ctypes

import numpy as np
import pytest

from oracle import ttsk_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


# (n0, Q, T, ll, r): one tile; several tiles per q range (ragged); every t range count; ranks below / at the tile edges
SHAPES = [(32, 8, 16, 20, 40), (64, 8, 16, 20, 40), (64, 24, 32, 20, 40), (32, 4096, 16, 20, 40), (64, 8000, 32, 20, 40),
          (32, 40, 48, 7, 22), (64, 16, 64, 16, 32), (32, 72, 128, 17, 34), (64, 264, 64, 3, 2), (64, 8, 16, 1, 40),
          (32, 16, 256, 20, 38), (64, 16, 16, 20, 40), (32, 2056, 16, 9, 12),
          # first mode beyond 64: blocks of 64 with partial Z (what the second pair of a sketch runs on)
          (128, 16, 16, 20, 40), (192, 40, 32, 7, 22), (1280, 64, 64, 20, 40), (640, 512, 16, 20, 40)]


@pytest.mark.parametrize("n0,Q,T,ll,r", SHAPES)
def test_first_pass_against_einsum(tsa, n0, Q, T, ll, r):
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, as_dev, sync
    rng = np.random.default_rng(n0 + Q + T + ll + r)
    X = rng.standard_normal((n0, Q, T))
    C = rng.standard_normal((n0, ll))
    P = rng.standard_normal((Q, r))
    Xd, Cd, Pd = as_dev(X), as_dev(C), as_dev(P)
    Z, U = DevArray.empty((ll, Q, T)), DevArray.empty((n0, r, T))
    V = ctypes.c_void_p
    nat.call("ttsk_dense_first_pass", V(Xd.ptr), n0, Q, T, V(Cd.ptr), ll, V(Pd.ptr), r, V(Z.ptr), V(U.ptr), 0)
    sync()
    assert _rel(Z.get(), np.einsum("bp,bqt->pqt", C, X)) < 1e-13
    assert _rel(U.get(), np.einsum("qp,bqt->bpt", P, X)) < 1e-13


@pytest.mark.parametrize("bad", ["n0", "T", "Q", "ll", "r", "odd r", "partials"])
def test_first_pass_declines_outside_its_cover(tsa, bad):
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    n0, Q, T, ll, r = 32, 16, 16, 20, 40
    if bad == "n0": n0 = 96
    if bad == "T": T = 24
    if bad == "Q": Q = 10
    if bad == "ll": ll = 21
    if bad == "r": r = 42
    if bad == "odd r": r = 39
    if bad == "partials": n0, Q, T = 64 * 40, 8 * 1024, 128       # 40 blocks x 20 x 2^20 doubles of partial Z: 6.7 GB
    if bad == "partials":          # (declined before anything is touched: no 21 GB operand for this)
        X = C = P = Z = U = DevArray.empty((64,))
    else:
        X, C, P = DevArray.empty((n0, Q, T)), DevArray.empty((n0, ll)), DevArray.empty((Q, r))
        Z, U = DevArray.empty((ll, Q, T)), DevArray.empty((n0, r, T))
    V = ctypes.c_void_p
    with pytest.raises(nat.TtskUnsupported):
        nat.call("ttsk_dense_first_pass", V(X.ptr), n0, Q, T, V(C.ptr), ll, V(P.ptr), r, V(Z.ptr), V(U.ptr), 0)


@pytest.mark.parametrize("shape,l,r,second", [((32, 16, 4, 6), 5, 8, False), ((64, 32, 8), 20, 40, False),
                                               ((32, 16, 5, 8, 3), 6, 10, False), ((32, 48, 6, 4), (4, 7, 9), (12, 10, 6), False),
                                               ((64, 16, 16, 16), 20, 40, True), ((32, 16, 16, 8), 4, 8, True),
                                               ((32, 16, 16, 8, 2), (8, 5, 3, 2), (6, 10, 12, 14), True)])
def test_dense_sketch_with_the_first_pass_vs_oracle(tsa, monkeypatch, shape, l, r, second):
    """general_sketch of a DenseTensor with TensorTrainDRMs whose first two modes put it on the one-pass kernel (the right
    DRM's matrix meets the tensor's columns position by position, so its outermost core carries the size of mode 1): every
    Psi and Omega against the oracle's dense path (1e-11), and against the device path with the kernel switched off."""
    from tt_sketch_amd.sketching_methods import dense_sketch
    d = len(shape)
    rng = np.random.default_rng(5)
    X = rng.standard_normal(shape)
    ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
    left = tsa.TensorTrainDRM(l, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=rd.cores)
    calls = []
    orig = dense_sketch._first_pass
    monkeypatch.setattr(dense_sketch, "_first_pass", lambda *a: calls.append(orig(*a)) or calls[-1])
    monkeypatch.setattr(dense_sketch, "_PAIR_MIN_BYTES", 0)          # the second pair through the kernel too, where it fits
    sk = tsa.general_sketch(tsa.DenseTensor(X), left, right, tsa.SketchMethod.streaming)
    assert calls[0] is True                                  # the kernel took the pass over the tensor
    assert calls[1] is second                                # ... and Z_1 / Psi_1 over Z_0 where the shape allows
    oP, oO = orc.general_sketch("dense", X, ld, rd, "streaming")
    for got, want in zip(list(sk.Psi_cores) + list(sk.Omega_mats), list(oP) + list(oO)):
        assert np.asarray(got).shape == np.asarray(want).shape and _rel(got, want) < 1e-11
    monkeypatch.setattr(dense_sketch, "_first_pass", lambda *a: False)
    two = tsa.general_sketch(tsa.DenseTensor(X), left, right, tsa.SketchMethod.streaming)
    for got, want in zip(list(sk.Psi_cores) + list(sk.Omega_mats), list(two.Psi_cores) + list(two.Omega_mats)):
        assert _rel(got, want) < 1e-12
