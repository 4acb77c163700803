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
          (128, 16, 16, 20, 40), (192, 40, 32, 7, 22), (1280, 64, 64, 20, 40), (640, 512, 16, 20, 40),
          # beyond ranks 20 / 40: 3 and 4 column tiles, 2 row tiles, odd right ranks (padded copy of P),
          # first modes that are multiples of 32 only; every pairing of the accumulator shapes
          (64, 24, 32, 21, 42), (32, 4096, 16, 21, 42), (64, 40, 48, 32, 64), (32, 72, 128, 25, 48), (64, 264, 64, 20, 41),
          (32, 16, 16, 32, 40), (64, 16, 32, 24, 39), (96, 40, 32, 7, 22), (96, 24, 16, 30, 57), (160, 64, 64, 21, 64),
          (64, 8, 16, 20, 49), (32, 8, 16, 1, 1), (64, 2056, 16, 29, 33),
          # one side tiny beside a wide other one (found by tests/fuzz_dense_pass.py: the half a kind owns reached beyond the rank)
          (256, 152, 80, 2, 51), (128, 8, 32, 3, 61), (64, 16, 16, 30, 2), (64, 24, 32, 1, 64), (128, 16, 16, 32, 1)]


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


@pytest.mark.parametrize("bad", ["n0", "T", "Q", "ll", "r", "partials"])
def test_first_pass_declines_outside_its_cover(tsa, bad):
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    n0, Q, T, ll, r = 32, 16, 16, 20, 40
    if bad == "n0": n0 = 48
    if bad == "T": T = 24
    if bad == "Q": Q = 10
    if bad == "ll": ll = 33
    if bad == "r": r = 65
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
                                               ((32, 16, 16, 8, 2), (8, 5, 3, 2), (6, 10, 12, 14), True),
                                               # ranks beyond 20 / 40, an odd right rank
                                               ((64, 32, 8), 21, 42, False), ((64, 16, 16, 16), 32, 64, True),
                                               ((32, 16, 16, 8), 24, 39, True), ((32, 16, 5, 8, 3), (30, 21, 9, 3), (57, 33, 20, 8), False)])
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


# ---------------------------------------------------------------------------------------------------------------------------
# ttsk_dense_left_pass: DRM MATRICES on the left (DenseGaussianDRM): Z_0 .. Z_3 from one read of the tensor
LEFT_SHAPES = [
    # (n0, n1, n2, n3, n4, l)
    (8, 3, 8, 8, 64, 20),        # i2 dealt over the XCDs, one column tile, strip rows 16..19
    (4, 2, 5, 16, 64, 20),       # n2 not a multiple of 8 (plain block map), two column tiles
    (12, 5, 3, 4, 128, 7),       # no strip, n4 = 128: four i3 per column tile
    (16, 2, 2, 2, 512, 16),      # n4 = 512: one i3 per tile, two tiles
    (8, 4, 8, 2, 256, 17),
    (64, 2, 8, 8, 64, 20),       # the C2 first mode
]


@pytest.mark.parametrize("n0,n1,n2,n3,n4,l", LEFT_SHAPES)
def test_left_pass_against_einsum(tsa, n0, n1, n2, n3, n4, l):
    """the four left products of dense_sketch.py:15-16 / :40-51 with the matrices of dense_gaussian_drm.py:77-80"""
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, as_dev, sync
    rng = np.random.default_rng(n0 + n1 + n2 + n3 + n4 + l)
    X = rng.standard_normal((n0, n1, n2, n3, n4))
    A = [rng.standard_normal((l, int(np.prod(X.shape[:mu + 1])))) for mu in range(4)]
    C = n3 * n4
    Xd = as_dev(X)
    A0 = as_dev(A[0])
    A3 = as_dev(np.ascontiguousarray(A[3].reshape(l, n0, n1, n2, n3).transpose(0, 3, 2, 4, 1)))
    A1t = as_dev(np.ascontiguousarray(A[1].reshape(l, n0, n1).transpose(0, 2, 1)))
    A2t = as_dev(np.ascontiguousarray(A[2].reshape(l, n0, n1, n2).transpose(0, 3, 2, 1)))
    Z0, Z1, Z2, E3 = (DevArray.empty(s) for s in ((l, n1 * n2 * C), (l, n2 * C), (l, C), (l, C)))
    V = ctypes.c_void_p
    nat.call("ttsk_dense_left_pass", V(Xd.ptr), n0, n1, n2, C, n4, l, V(A0.ptr), V(A1t.ptr), V(A2t.ptr), V(A3.ptr), V(Z0.ptr),
             V(Z1.ptr), V(Z2.ptr), V(E3.ptr), 0)
    sync()
    want = [A[mu] @ X.reshape(A[mu].shape[1], -1) for mu in range(4)]
    assert _rel(Z0.get(), want[0]) < 1e-13
    assert _rel(Z1.get(), want[1]) < 1e-13
    assert _rel(Z2.get(), want[2]) < 1e-13
    assert _rel(E3.get().reshape(l, n3, n4).sum(axis=1), want[3]) < 1e-13


def test_left_pass_declines_outside_its_cover(tsa):
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    V = ctypes.c_void_p
    d = DevArray.zeros((64,))
    for (n0, n1, n2, C, n4, l) in [(6, 2, 2, 512, 64, 8), (8, 2, 2, 500, 50, 8), (8, 2, 2, 512, 32, 8), (8, 2, 2, 512, 64, 21)]:
        with pytest.raises(nat.TtskUnsupported):
            nat.call("ttsk_dense_left_pass", V(d.ptr), n0, n1, n2, C, n4, l, V(d.ptr), V(d.ptr), V(d.ptr), V(d.ptr), V(d.ptr),
                     V(d.ptr), V(d.ptr), V(d.ptr), 0)


@pytest.mark.parametrize("shape,l,r", [((8, 3, 8, 8, 64), 6, 9), ((4, 4, 3, 16, 64), 20, 22), ((8, 2, 2, 8, 8, 8), 5, 7)])
def test_dense_gaussian_sketch_with_the_left_pass_vs_oracle(tsa, monkeypatch, shape, l, r):
    """general_sketch of a dense tensor with DenseGaussianDRMs of order 5 / 6 (the last modes merged behind the fourth): the
    fused left pass and the mode-by-mode path both against the oracle (dense_sketch.py:7-52 as the reference writes it),
    incl. Omega_0 = A_0 Psi_0 and Psi_{d-1} from the shared left product."""
    from tt_sketch_amd.sketching_methods import dense_sketch
    rng = np.random.default_rng(sum(shape) + l)
    d = len(shape)
    X = rng.standard_normal(shape)
    left = tsa.DenseGaussianDRM(l, shape, False, seed=5)
    right = tsa.DenseGaussianDRM(r, shape, True, seed=6)
    ld = orc.DenseDrm([np.asarray(m) for m in left.sketching_mats], shape, False)
    rd = orc.DenseDrm([np.asarray(m) for m in right.sketching_mats], shape, True)
    oP, oO = orc.general_sketch("dense", X, ld, rd, "streaming")
    hits = []
    real = dense_sketch.prepare_left
    monkeypatch.setattr(dense_sketch, "prepare_left", lambda *a, **k: hits.append(real(*a, **k)) or hits[-1])
    for on in ("1", "0"):
        monkeypatch.setenv("TTSK_DENSE_LEFT_PASS", on)
        sk = tsa.general_sketch(tsa.DenseTensor(X), left, right, tsa.SketchMethod.streaming)
        for a, b in zip(sk.Psi_cores + sk.Omega_mats, oP + oO):
            assert a.shape == b.shape and np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(b)
    assert hits == [True, False]


@pytest.mark.parametrize("rows,N,K,pad", [(64, 40, 8192, 0), (20, 40, 4096, 0), (130, 7, 12288, 64), (1, 48, 4160, 0), (200, 33, 6400, 2),
                                          (64, 16, 4096, 0), (70, 36, 65536, 0)])
def test_rows_against_a_matrix_long_k(tsa, rows, N, K, pad):
    """C = S B^T with both operands contiguous along a long contracted index (the right-hand products of a dense sketch with
    DRM matrices, dense_sketch.py:15-16): rows_longk_kernel behind ttsk_gemm / contract, against numpy; row blocks of 64 with a
    short last block, 1 .. 48 matrix rows (tiles + 4-wide strips, odd counts), operands that are row views of wider arrays."""
    from tt_sketch_amd.device import as_dev, contract, sync
    rng = np.random.default_rng(rows + N + K)
    S = rng.standard_normal((rows, K + pad))
    B = rng.standard_normal((N, K + pad))
    Sd, Bd = as_dev(S)[:, :K], as_dev(B)[:, :K]
    C = contract("bq,mq->bm", Sd, Bd)
    sync()
    want = S[:, :K] @ B[:, :K].T
    assert _rel(C.get(), want) < 1e-13
