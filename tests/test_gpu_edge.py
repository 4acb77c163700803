"""Edge shapes through the public API on the HIP path: d = 2, a mode of size 1, rank 1, sketch ranks beyond what
the unfoldings allow (trimmed), sums of one / of mixed kinds, a sparse tensor with one and with no nonzero, an
out-of-range index, d = 7, the classical TT-SVD at d = 2 and with a unit mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


def _recovers(tsa, T, X, l, r, tol=1e-12):
    out = tsa.stream_sketch(T, l, r, seed=3)
    assert out.error(X) < tol * max(1.0, np.linalg.norm(X)), out.error(X)


def test_two_modes(tsa):
    X = tsa.TensorTrain.random((7, 9), 3, seed=1)
    D = X.to_numpy()
    for T in (X, tsa.DenseTensor(D), X.dense().to_sparse()):
        _recovers(tsa, T, D, 3, 4)
    assert tsa.tt_svd(tsa.DenseTensor(D), 3).error(D) < 1e-13


def test_a_mode_of_size_one(tsa):
    Y = tsa.TensorTrain.random((5, 1, 6, 4), 2, seed=2)
    D = Y.to_numpy()
    for T in (Y, tsa.DenseTensor(D), Y.dense().to_sparse()):
        _recovers(tsa, T, D, 2, 3)
    assert tsa.orthogonal_sketch(Y, 2, 3, seed=3).error(D) < 1e-12
    assert tsa.hmt_sketch(Y, 3, seed=3).error(D) < 1e-12
    assert tsa.tt_svd(tsa.DenseTensor(D), 2).error(D) < 1e-13


def test_rank_one_and_oversized_sketch_ranks(tsa):
    Z = tsa.TensorTrain.random((4, 5, 6), 1, seed=3)
    D = Z.to_numpy()
    _recovers(tsa, Z, D, 1, 2)
    _recovers(tsa, Z, D, 50, 60)                      # trimmed to what the unfoldings allow
    for T in (tsa.CPTensor.random((4, 5, 6), 1, seed=6), tsa.TuckerTensor.random((4, 5, 6), 1, seed=6)):
        _recovers(tsa, T, T.to_numpy(), 1, 2)


def test_sums(tsa):
    Z = tsa.TensorTrain.random((4, 5, 6), 1, seed=3)
    W = tsa.TensorTrain.random((4, 5, 6), 2, seed=4)
    _recovers(tsa, tsa.TensorSum([Z]), Z.to_numpy(), 2, 3)
    S = tsa.TensorSum([Z, W.dense().to_sparse(), tsa.DenseTensor(W.to_numpy())])
    _recovers(tsa, S, Z.to_numpy() + 2 * W.to_numpy(), 4, 5)
    stt = tsa.stream_sketch(W, 3, 4, seed=1)
    assert (stt + Z).error(W.to_numpy() + Z.to_numpy()) < 1e-12


def test_sparse_with_one_and_with_no_nonzero(tsa):
    sp1 = tsa.SparseTensor((4, 5, 6), np.array([[1], [2], [3]]), np.array([2.5]))
    D1 = np.zeros((4, 5, 6))
    D1[1, 2, 3] = 2.5
    _recovers(tsa, sp1, D1, 2, 3)
    sp0 = tsa.SparseTensor((4, 5, 6), np.zeros((3, 0), dtype=np.int64), np.zeros(0))
    assert np.linalg.norm(tsa.stream_sketch(sp0, 2, 3, seed=1).to_numpy()) == 0.0
    with pytest.raises(IndexError):
        tsa.stream_sketch(tsa.SparseTensor((4, 5, 6), np.array([[4], [0], [0]]), np.array([1.0])), 2, 3, seed=1)


def test_seven_modes(tsa):
    V = tsa.TensorTrain.random((3,) * 7, 2, seed=5)
    D = V.to_numpy()
    _recovers(tsa, V, D, 2, 3)
    _recovers(tsa, tsa.DenseTensor(D), D, 2, 3)
