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


@pytest.mark.parametrize("l,r", [(55, 110), (55, 58), (145, 290), (25, 50), (5, 10), (95, 98)])
def test_ref150_published_shape_vs_oracle(tsa, l, r):
    """The reference's own timing benchmark (scripts/plot_timings.py:28-36,94-124): shape 100^5, TT-rank 150 (trimmed
    to (100, 150, 150, 100)), left rank l (trimmed), right rank r = 2 l or l + 3 (not trimmed, sketch.py:188-209).
    TT ranks beyond 128, odd DRM ranks, DRM ranks that change between modes: the chunked fused chain step
    (csrc/chain_wide.h) and the chunked Psi product where they apply, the two-launch kernels elsewhere."""
    from tt_sketch_amd.utils import process_tt_rank
    rng = np.random.default_rng(150 + l)
    shape = (100,) * 5
    cores = orc.random_tt(shape, (100, 150, 150, 100), rng)
    lrank = process_tt_rank(l, shape, trim=True)
    ld, rd = orc.random_tt_drm(shape, lrank, False, rng), orc.random_tt_drm(shape, r, True, rng)
    left = tsa.TensorTrainDRM(lrank, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=rd.cores)
    stt = tsa.stream_sketch(tsa.TensorTrain(cores), lrank, (r,) * 4, left_drm=left, right_drm=right)
    _check(stt.sketch_, *orc.general_sketch("tt", cores, ld, rd, "streaming"))


def test_ref150_batched_pass_vs_oracle(tsa):
    """Four rank-150 TTs through one batched pass (ttsk_tt_sketch_batch) at l=55 / r=110: every tensor against the oracle."""
    import ctypes
    from tt_sketch_amd.device import DevArray, sync
    from tt_sketch_amd.tt_fused import TTSketchPlan
    rng = np.random.default_rng(151)
    shape, tt_rank, B = (100,) * 5, (100, 150, 150, 100), 4
    tts = [orc.random_tt(shape, tt_rank, rng) for _ in range(B)]
    ld, rd = orc.random_tt_drm(shape, (55,) * 4, False, rng), orc.random_tt_drm(shape, 110, True, rng)
    left = tsa.TensorTrainDRM(55, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(110, shape, True, seed=2, cores=rd.cores)
    plan = TTSketchPlan(shape, tt_rank, left, right)
    stride = plan.size + (plan.size & 1)
    out = DevArray.empty((B * stride,))
    keep, flat = [], []
    for t in tts:
        p1, k1 = plan.core_pointers(tsa.TensorTrain(t))
        keep.append(k1)
        flat += [p1[i] for i in range(plan.d)]
    plan.run_batch((ctypes.c_void_p * len(flat))(*flat), B, out, stride)
    sync()
    got = out.get()
    for b in range(B):
        oP, oO = orc.general_sketch("tt", tts[b], ld, rd, "streaming")
        want = np.concatenate([a.ravel() for a in oP + oO])
        assert rel(got[b * stride:b * stride + plan.size], want) < TOL, b


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


# ---------------------------------------------------------------------------------------------------
# BASELINE sizes on the HIP path with a real check: the kernel variants that
# only run at these sizes (64-bit origins of the streamed kernel, rebased long-K row tiles) were timed
# in round 1 but never compared with anything.
# ---------------------------------------------------------------------------------------------------
def _dense_of_tt_on_device(tsa, cores):
    """full(cores) built in HBM (never on the host): 64^5 doubles = 8.6 GB at C2."""
    from tt_sketch_amd.device import DevArray, contract
    P = DevArray.from_host(cores[0].reshape(cores[0].shape[1], -1))
    for c in cores[1:]:
        P = contract("ia,ajb->ijb", P, DevArray.from_host(c))
        P = P.reshape(-1, P.shape[-1])
    return P.reshape(tuple(c.shape[1] for c in cores))


def _check_lists(got, want, tol, what):
    for k, (a, b) in enumerate(zip(got, want)):
        a = np.asarray(a)
        assert a.shape == b.shape, (what, k, a.shape, b.shape)
        assert rel(a, b) < tol, (what, k, a.shape, rel(a, b))


def test_c2_dense_full_size(tsa):
    """configs[1] at FULL size: dense d=5 n=64 (8.59 GB), TensorTrainDRM l=20 r=40.  X is the full form
    of a rank-3 TT, built on the device; the expected sketch of the dense path (incl. the reversed-mode
    quirk of the right matrices) then factors through small matrices (tests/structured.py, pinned to the
    oracle's dense path by test_oracle_golden.py).  Plus linearity over the two half slabs of mode 0."""
    from tests import structured as st
    from tt_sketch_amd.device import DevArray, sync
    from tt_sketch_amd import _native as nat
    import ctypes
    rng = np.random.default_rng(2)
    d, n, s, l, r = 5, 64, 3, 20, 40
    shape = (n,) * d
    cores = [c * 8.0 for c in orc.random_tt(shape, s, rng)]           # entries of X of order 1
    ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
    left = tsa.TensorTrainDRM(l, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=rd.cores)
    X = _dense_of_tt_on_device(tsa, cores)
    assert X.size * 8 == 8 * 64 ** 5 and X.is_contiguous()
    # spot check of the device-built tensor against the TT
    probe = X[17, 5, 63, 0].get()                                       # (64,) fibre of the last mode
    want = (cores[0][0, 17] @ cores[1][:, 5] @ cores[2][:, 63] @ cores[3][:, 0]) @ cores[4][:, :, 0]
    assert probe.shape == (64,) and rel(probe, want) < 1e-13
    T = tsa.DenseTensor(X)
    t0 = time.perf_counter()
    sk = tsa.general_sketch(T, left, right, tsa.SketchMethod.streaming)
    sync()
    t_gpu = time.perf_counter() - t0
    sP, sO = st.dense_sketch_of_tt_ttdrm(cores, ld.cores, rd.cores)
    _check_lists(sk.Psi_cores + sk.Omega_mats, sP + sO, 1e-11, "C2 TT-DRM")
    # linearity: sketch(X) = sketch(X with the upper half of mode 0 zeroed) + sketch(the rest)
    half = X.size // 2
    parts = []
    for lo in (0, half):
        Y = X.copy()
        nat.call("ttsk_memset", ctypes.c_void_p(Y.ptr + 8 * (half - lo)), 0, ctypes.c_size_t(8 * half), 0)
        parts.append(tsa.general_sketch(tsa.DenseTensor(Y), left, right, tsa.SketchMethod.streaming))
        del Y
    tot = parts[0] + parts[1]
    _check_lists(tot.Psi_cores + tot.Omega_mats, sk.Psi_cores + sk.Omega_mats, 1e-11, "C2 linearity")
    print(f"\n[C2 full size, TT-DRM] dense sketch of 8.59 GB: {t_gpu * 1e3:.1f} ms (first call)")


def test_c2_dense_full_size_gaussian_drm(tsa):
    """The same tensor with device-sampled DenseGaussianDRMs l=20 r=40 (8.2 GB of Gaussian matrices: the
    k-contiguous generic-tile variants of the long-K kernel).  The matrices are read back and the
    expected sketch is formed on the host through the TT's partial products."""
    from tests import structured as st
    from tt_sketch_amd.device import sync
    rng = np.random.default_rng(12)
    d, n, s, l, r = 5, 64, 3, 20, 40
    shape = (n,) * d
    cores = [c * 8.0 for c in orc.random_tt(shape, s, rng)]
    X = _dense_of_tt_on_device(tsa, cores)
    left = tsa.DenseGaussianDRM(l, shape, False, seed=41)
    right = tsa.DenseGaussianDRM(r, shape, True, seed=42)
    sk = tsa.general_sketch(tsa.DenseTensor(X), left, right, tsa.SketchMethod.streaming)
    sync()
    A = [np.asarray(m) for m in left.sketching_mats]                 # (l, n^{mu+1})
    B = [np.asarray(m) for m in right.sketching_mats][::-1]          # user order: B[mu] (r, n^{d-1-mu})
    assert A[3].shape == (l, n ** 4) and B[0].shape == (r, n ** 4)
    sP, sO = st.dense_sketch_of_tt_matrices(cores, A, B)
    _check_lists(sk.Psi_cores + sk.Omega_mats, sP + sO, 1e-11, "C2 dense-Gaussian DRM")


def test_c4_sparse_full_size(tsa):
    """configs[3] at FULL size: nnz = 10^7, shape (200,150,100,120,300), SparseGaussianDRM l=10 r=15.
    sketch(whole) == sketch(first 10^5 nonzeros) + sketch(the rest); the 10^5 part against the oracle."""
    shape = (200, 150, 100, 120, 300)
    rng = np.random.default_rng(44)
    nnz, small = 10_000_000, 100_000
    idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
    val = rng.standard_normal(nnz)
    ld = tsa.SparseGaussianDRM(10, shape, False, seed=3)
    rd = tsa.SparseGaussianDRM(15, shape, True, seed=4)
    T = tsa.SparseTensor(shape, idx, val)
    t0 = time.perf_counter()
    whole = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    wP, wO = whole.Psi_cores, whole.Omega_mats
    t_gpu = time.perf_counter() - t0
    a = tsa.general_sketch(tsa.SparseTensor(shape, idx[:, :small], val[:small]), ld, rd, tsa.SketchMethod.streaming)
    b = tsa.general_sketch(tsa.SparseTensor(shape, idx[:, small:], val[small:]), ld, rd, tsa.SketchMethod.streaming)
    tot = a + b
    _check_lists(tot.Psi_cores + tot.Omega_mats, wP + wO, 1e-11, "C4 shards")
    old = orc.HashGaussDrm(3, shape, False, (0,) * 4, (10,) * 4)
    ord_ = orc.HashGaussDrm(4, shape, True, (0,) * 4, (15,) * 4)
    oP, oO = orc.general_sketch("sparse", (shape, idx[:, :small], val[:small]), old, ord_, "streaming")
    _check_lists(a.Psi_cores + a.Omega_mats, oP + oO, 1e-11, "C4 10^5 part vs oracle")
    # a checksum the domain offers: sum_j Psi_mu[:, j, :] = (L_{mu-1} * entries) R_mu^T needs no mode split;
    # for mu = d-1 and mu = 0 that is the sum over all nonzeros of the other side's panel
    assert abs(wP[-1].sum() - sum(p.sum() for p in (a.Psi_cores[-1], b.Psi_cores[-1]))) < 1e-9 * np.abs(wP[-1]).sum()
    print(f"\n[C4 full size] nnz=1e7 sketch {t_gpu * 1e3:.0f} ms (incl. H2D of 480 MB and D2H of the sketch)")


def test_c3_batch16_every_tensor_vs_oracle(tsa):
    """The bench's own pass -- 16 TTs of the C3 signature in ONE ttsk_tt_sketch_batch -- with every one of
    the 16 sketches compared with the oracle (bench.py checks tensors 0 and 15 only)."""
    import ctypes
    from tt_sketch_amd import tt_fused
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(33)
    shape, nb = (200,) * 6, 16
    ld, rd = orc.random_tt_drm(shape, 50, False, rng), orc.random_tt_drm(shape, 100, True, rng)
    left = tsa.TensorTrainDRM(50, shape, False, seed=1, cores=ld.cores)
    right = tsa.TensorTrainDRM(100, shape, True, seed=2, cores=rd.cores)
    tts = [orc.random_tt(shape, 100, rng) for _ in range(nb)]
    dev = [tsa.TensorTrain(c) for c in tts]
    plan = tt_fused.TTSketchPlan(dev[0].shape, dev[0].rank, left, right)
    keep, flat = [], []
    for t in dev:
        ptrs, k = plan.core_pointers(t)
        keep.append(k)
        flat += [ptrs[i] for i in range(plan.d)]
    stride = plan.size + (plan.size & 1)
    out = DevArray.zeros((nb * stride,))
    plan.run_batch((ctypes.c_void_p * len(flat))(*flat), nb, out, stride)
    worst = 0.0
    for b, cores in enumerate(tts):
        oP, oO = orc.general_sketch("tt", cores, ld, rd, "streaming")
        Psi, Om = plan.views(out[b * stride:b * stride + plan.size])
        for a, c in zip(Psi + Om, oP + oO):
            e = rel(a.get(), c)
            worst = max(worst, e)
            assert e < TOL, (b, a.shape, e)
    print(f"\n[C3 batch 16] worst relative error over 16 x 11 arrays: {worst:.2e}")
