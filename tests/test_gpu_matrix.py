"""The reference's property tests over the full DRM-pair matrix, run on the HIP path.

Same case matrix as the reference's tests/test_sketching_matrix.py (every ordered pair of DRM types that can
sketch the tensor kind x order x rank x {streaming, orthogonal, hmt}; :269-306 sparse, :462-510 TT, :547-595
CP, :523-544 Tucker, :338-363 dense) and the same properties (:208-254 exact recovery below 1e-9, same seed
-> same tensor, next seed -> another one; :137-187 blocked == whole for sliceable pairs; :41-130 rank increase
keeps the leading block and slicing undoes it exactly).  Every sketch goes through the C ABI.
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


def _drm_types(kind):
    import tt_sketch_amd as t
    from tt_sketch_amd.sketching_methods import abstract_methods as am
    mixin = {"sparse": am.CansketchSparse, "tt": am.CansketchTT, "cp": am.CansketchCP}[kind]
    return [c.__name__ for c in t.ALL_DRM if issubclass(c, mixin)]


def _pairs(kind):
    names = _drm_types(kind)
    return ["|".join(p) for p in itertools.permutations(names, 2)] + ["|".join((n, n)) for n in names]


METHODS = ["streaming", "orthogonal", "hmt"]


def _sketch(tsa, method, T, left_rank, right_rank, seed, L, R):
    if method == "hmt":
        return tsa.hmt_sketch(T, right_rank, seed, R)
    f = tsa.stream_sketch if method == "streaming" else tsa.orthogonal_sketch
    return f(T, left_rank, right_rank, seed=seed, left_drm_type=L, right_drm_type=R)


def _recovery(tsa, X, T, left_rank, right_rank, seed, L, R, method):
    a = _sketch(tsa, method, T, left_rank, right_rank, seed, L, R)
    assert a.error(X) < 1e-9
    b = _sketch(tsa, method, T, left_rank, right_rank, seed, L, R)
    assert b.error(a) < 1e-9                      # same seed: same tensor (up to the order of summation)
    c = _sketch(tsa, method, T, left_rank, right_rank, seed + 1, L, R)
    assert not np.all(a.to_numpy() == c.to_numpy())


def _blocked(tsa, T, seed, L, R):
    k = len(T.shape) - 1
    stt, ld, rd = tsa.stream_sketch(T, (9,) * k, (8,) * k, seed=seed, left_drm_type=L, right_drm_type=R, return_drm=True)
    cuts_l = [[(v,) * k for v in c] for c in ((0, 3, 6, 9), (0, 3, 9), (0, 9))]
    cuts_r = [[(v,) * k for v in c] for c in ((0, 4, 6, 8), (0, 6, 8), (0, 8))]
    for ls in cuts_l:
        for rs in cuts_r:
            blk = tsa.blocked_stream_sketch(T, ld, rd, ls, rs)
            for a, b in zip(blk.Psi_cores + blk.Omega_mats, stt.Psi_cores + stt.Omega_mats):
                assert np.allclose(a, b)


def _rank_increase(tsa, X, T, left_rank, right_rank, seed, L, R):
    shape, d = T.shape, len(T.shape)
    ld = L(left_rank, transpose=False, shape=X.shape, seed=seed)
    rd = R(right_rank, transpose=True, shape=X.shape, seed=(seed + 7919 * d) % 2**32)
    s1 = tsa.stream_sketch(T, left_rank, right_rank, left_drm=ld, right_drm=rd, seed=seed)
    new_l, new_r = tuple(r + 2 for r in left_rank), tuple(r + 3 for r in right_rank)
    s3 = s1.increase_rank(T, new_l, new_r)
    ld2, rd2 = ld.increase_rank(new_l), rd.increase_rank(new_r)
    s2 = tsa.stream_sketch(T, new_l, new_r, left_drm=ld2, right_drm=rd2, seed=seed)
    l1, r1 = (1,) + tuple(left_rank), tuple(right_rank) + (1,)
    nl1, nr1 = (1,) + new_l, new_r + (1,)
    for other in (s2, s3):
        for i, (a, b) in enumerate(zip(s1.Psi_cores, other.Psi_cores)):
            assert b.shape == (nl1[i], shape[i], nr1[i])
            assert np.allclose(a, b[:l1[i], :, :r1[i]])
        for i, z in enumerate(other.Omega_mats):
            assert z.shape == (new_l[i], new_r[i])
    s4 = tsa.stream_sketch(T, left_rank, right_rank, left_drm=ld2.slice(None, left_rank),
                           right_drm=rd2.slice(None, right_rank), seed=seed)
    for a, b in zip(s1.Psi_cores + s1.Omega_mats, s4.Psi_cores + s4.Omega_mats):
        # exact `==` in the reference (one BLAS, one order); the sparse Psi flushes run sums with fp64 atomics
        # on the device, so the last bit may differ there
        assert np.allclose(a, b, rtol=1e-13, atol=0)


def _case(tsa, kind, n_dims, rank, pair, method, first_mode):
    from tt_sketch_amd.drm_base import CanIncreaseRank, CanSlice
    seed = 180
    shape = tuple(range(first_mode, first_mode + n_dims))
    left_rank = tuple(range(rank, rank + n_dims - 1))
    right_rank = tuple(range(rank + 1, rank + n_dims))
    by_name = {c.__name__: c for c in tsa.ALL_DRM}
    L, R = (by_name[n] for n in pair.split("|"))
    if kind == "cp":
        T = tsa.CPTensor.random(shape, rank, seed=seed)
        X = T.to_numpy()
    else:
        T = tsa.TensorTrain.random(shape, rank, seed=seed)
        X = T.to_numpy()
        if kind == "sparse":
            T = T.dense().to_sparse()
    _recovery(tsa, X, T, left_rank, right_rank, seed, L, R, method)
    if method != "streaming":
        return           # the block / rank properties do not depend on the method: once per (pair, order, rank)
    if issubclass(L, CanSlice) and issubclass(R, CanSlice):
        _blocked(tsa, T, seed, L, R)
    if issubclass(L, CanIncreaseRank) and issubclass(R, CanIncreaseRank):
        _rank_increase(tsa, X, T, left_rank, right_rank, seed, L, R)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("rank", [2, 5])
@pytest.mark.parametrize("n_dims", [2, 3])
@pytest.mark.parametrize("pair", _pairs("sparse"))
def test_exact_recovery_sparse(tsa, n_dims, rank, pair, method):
    _case(tsa, "sparse", n_dims, rank, pair, method, 9)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("rank", [2, 3])
@pytest.mark.parametrize("n_dims", [2, 3])
@pytest.mark.parametrize("pair", _pairs("tt"))
def test_exact_recovery_tt(tsa, n_dims, rank, pair, method):
    _case(tsa, "tt", n_dims, rank, pair, method, 10)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("rank", [2, 3])
@pytest.mark.parametrize("n_dims", [2, 3])
@pytest.mark.parametrize("pair", _pairs("cp"))
def test_exact_recovery_cp(tsa, n_dims, rank, pair, method):
    _case(tsa, "cp", n_dims, rank, pair, method, 10)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("rank", [2, 3])
@pytest.mark.parametrize("n_dims", [2, 3])
def test_exact_recovery_tucker(tsa, n_dims, rank, method):
    seed = 180
    shape = tuple(range(10, 10 + n_dims))
    T = tsa.TuckerTensor.random(shape, rank, seed=seed)
    _recovery(tsa, T.to_numpy(), T, tuple(range(rank, rank + n_dims - 1)), tuple(range(rank + 1, rank + n_dims)), seed,
              tsa.TensorTrainDRM, tsa.TensorTrainDRM, method)


@pytest.mark.parametrize("n_dims", [2, 3, 4])
def test_sketch_dense(tsa, n_dims):
    """reference :338-363: a dense rank-5 tensor through DenseGaussianDRMs, streaming and orthogonal."""
    shape = tuple(range(5, 5 + n_dims))
    X = tsa.TensorTrain.random(shape, 5, seed=179).to_numpy()
    T = tsa.DenseTensor(X)
    left_rank = tuple(range(5, 5 + n_dims - 1))
    right_rank = tuple(range(6, 6 + n_dims - 1))
    for f in (tsa.stream_sketch, tsa.orthogonal_sketch):
        out = f(T, left_rank, right_rank, seed=179, left_drm_type=tsa.DenseGaussianDRM, right_drm_type=tsa.DenseGaussianDRM)
        assert out.error(X) < 1e-8
