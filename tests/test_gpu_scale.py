"""GPU parity at the BASELINE.json configuration sizes (or the largest size the CPU oracle finishes
in seconds) plus size-independent properties at full size.  C1..C5 as in SURVEY.md section 8."""
import time

import numpy as np
import pytest

from oracle import ttsk_oracle as orc
from tests.golden_io import rel

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


def _check(sk, oP, oO, tol=TOL):
    for a, b in zip(sk.Psi_cores + sk.Omega_mats, oP + oO):
        assert rel(a, b) < tol, (a.shape, rel(a, b))


def test_c1_tt_dense_gaussian_plumbing(tsa):
    """configs[0]: d=4 n=20 rank-5 TT, DenseGaussianDRM l=7 r=9."""
    from tests.gpu_build import make_drm, make_tensor
    rng = np.random.default_rng(1)
    shape = (20,) * 4
    cores = orc.random_tt(shape, 5, rng)
    def dense_drm(rank, transpose):
        shp = shape[::-1] if transpose else shape
        mats, cols = [], 1
        for n in shp[:-1]:
            cols *= n
            mats.append(rng.standard_normal((rank, cols)))
        return orc.DenseDrm(mats, shape, transpose)
    ld, rd = dense_drm(7, False), dense_drm(9, True)
    sk = tsa.general_sketch(make_tensor("tt", cores), make_drm(ld), make_drm(rd), tsa.SketchMethod.streaming)
    _check(sk, *orc.general_sketch("tt", cores, ld, rd, "streaming"))
    stt = tsa.SketchedTensorTrain(sk, make_drm(ld), make_drm(rd))
    assert stt.to_tt().error(tsa.TensorTrain(cores), relative=True) < 1e-9


def test_c2_dense_scaled(tsa):
    """configs[1] scaled to what the oracle does in seconds: dense d=5 n=24, TT-DRM l=20 r=25,
    and DenseGaussianDRM on d=4 (the quirk of dense_sketch.py right matrices is pinned by the golden set)."""
    from tests.gpu_build import make_drm, make_tensor
    rng = np.random.default_rng(2)
    shape = (24,) * 5
    X = rng.standard_normal(shape)
    ld, rd = orc.random_tt_drm(shape, 20, False, rng), orc.random_tt_drm(shape, 25, True, rng)
    t0 = time.perf_counter()
    sk = tsa.general_sketch(make_tensor("dense", X), make_drm(ld), make_drm(rd), tsa.SketchMethod.streaming)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    oP, oO = orc.general_sketch("dense", X, ld, rd, "streaming")
    t_cpu = time.perf_counter() - t0
    _check(sk, oP, oO)
    print(f"\n[C2 scaled n=24] gpu (incl. H2D of {X.nbytes / 1e6:.0f} MB) {t_gpu * 1e3:.1f} ms, oracle {t_cpu * 1e3:.0f} ms")


def test_c3_full_size_vs_oracle(tsa):
    """configs[2] at full size: TT d=6 n=200 s=100, TT-DRM l=50 r=100 through the public API."""
    rng = np.random.default_rng(3)
    shape = (200,) * 6
    cores = orc.random_tt(shape, 100, rng)
    ld, rd = orc.random_tt_drm(shape, 50, False, rng), orc.random_tt_drm(shape, 100, True, rng)
    left = tsa.TensorTrainDRM(50, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(100, shape, True, seed=2, cores=rd.cores)
    stt = tsa.stream_sketch(tsa.TensorTrain(cores), (50,) * 5, (100,) * 5, left_drm=left, right_drm=right)
    _check(stt.sketch_, *orc.general_sketch("tt", cores, ld, rd, "streaming"))


def test_c4_sparse_scaled_and_full_size_properties(tsa):
    """configs[3]: FROSTT-style COO d=5 shape (200,150,100,120,300), SparseGaussianDRM l=10 r=15.
    nnz=2e5 against the oracle; nnz=4e6 through linearity over nnz shards and a Psi checksum."""
    shape = (200, 150, 100, 120, 300)
    rng = np.random.default_rng(4)

    def make(nnz):
        idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
        return idx, rng.standard_normal(nnz)
    idx, val = make(200_000)
    T = tsa.SparseTensor(shape, idx, val)
    ld = tsa.SparseGaussianDRM(10, shape, False, seed=3)
    rd = tsa.SparseGaussianDRM(15, shape, True, seed=4)
    sk = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    old = orc.HashGaussDrm(3, shape, False, (0,) * 4, (10,) * 4)
    ord_ = orc.HashGaussDrm(4, shape, True, (0,) * 4, (15,) * 4)
    oP, oO = orc.general_sketch("sparse", (shape, idx, val), old, ord_, "streaming")
    _check(sk, oP, oO, tol=1e-11)      # |ndtri| differs by a few ulp between device and libm
    # full-ish size: sketch(whole) == sketch(shard A) + sketch(shard B)
    idx, val = make(4_000_000)
    T = tsa.SparseTensor(shape, idx, val)
    t0 = time.perf_counter()
    whole = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    t_gpu = time.perf_counter() - t0
    parts = T.split(2)
    a = tsa.general_sketch(parts.tensors[0], ld, rd, tsa.SketchMethod.streaming)
    b = tsa.general_sketch(parts.tensors[1], ld, rd, tsa.SketchMethod.streaming)
    _check(whole, (a + b).Psi_cores, (a + b).Omega_mats, tol=1e-11)
    # sum_j Psi_mu[:, j, :] equals the Omega-like product without the mode split
    print(f"\n[C4 nnz=4e6] gpu sketch {t_gpu * 1e3:.0f} ms (incl. H2D of indices)")


def test_c5_tensor_sum_full_size(tsa):
    """configs[4]: TensorSum of 32 rank-20 TTs, d=6 n=128, shared TT-DRMs l=50 r=100, full size."""
    rng = np.random.default_rng(5)
    shape = (128,) * 6
    terms = [orc.random_tt(shape, 20, rng) for _ in range(32)]
    coef = np.logspace(0, -10, 32)
    for c, t in zip(coef, terms):
        t[-1] *= c
    ld, rd = orc.random_tt_drm(shape, 50, False, rng), orc.random_tt_drm(shape, 100, True, rng)
    left = tsa.TensorTrainDRM(50, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(100, shape, True, seed=2, cores=rd.cores)
    S = tsa.TensorSum([tsa.TensorTrain(t) for t in terms])
    S.prepare_device()
    t0 = time.perf_counter()
    stt = tsa.stream_sketch(S, (50,) * 5, (100,) * 5, left_drm=left, right_drm=right)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    oP, oO = orc.general_sketch("sum", [("tt", t) for t in terms], ld, rd, "streaming")
    t_cpu = time.perf_counter() - t0
    _check(stt.sketch_, oP, oO)
    print(f"\n[C5] gpu {t_gpu * 1e3:.1f} ms (incl. D2H of the sketch), oracle {t_cpu * 1e3:.0f} ms")
