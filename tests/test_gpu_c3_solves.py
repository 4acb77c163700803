"""C3-size parity for the solves and the sequential variants.

orthogonal_sketch / hmt_sketch / stream_sketch(...).to_tt() (both assembly directions) at the
BASELINE configs[2] size -- d=6 n=200 TT-rank 100, TensorTrainDRM l=50 r=100, injected DRM cores --
against the oracle's restatement of sketch.py:44-151,400-443 and sketch_dispatch.py:160-193.
What only happens at this size: CholeskyQR2 on 10 000 x 50 / 20 000 x 100 unfoldings, pinv-apply on
10 000 x 100, the Cholesky gate and (rank-deficient case) the Jacobi fallback with its queued verdict.

Tensors of 200^6 entries are compared in TT form: ||A - B|| through a QR sweep over the direct sum
(accurate to rounding; the dot-product form loses half the digits).
"""
import numpy as np
import pytest

from oracle import ttsk_oracle as orc

pytestmark = pytest.mark.gpu

SHAPE, S_IN, L, R = (200,) * 6, 100, 50, 100
TENSOR_TOL = 1e-10         # tensor-level bar for cores that went through a pseudo-inverse / QR: SURVEY section 8c's bar for full-rank
                           # Omega (measured since the refinement step of the assembly: 5e-14 median, 4e-11 max)
ORTH_TOL = 1e-11           # || Q^T Q - I ||_F of every left unfolding


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


def tt_norm(cores):
    """|| TT || by a left-to-right QR sweep (no squaring)."""
    carry = np.ones((1, 1))
    for c in cores:
        r1, n, r2 = c.shape
        m = np.tensordot(carry, c, axes=(1, 0)).reshape(-1, r2)
        carry = np.linalg.qr(m, mode="r")
    return float(np.linalg.norm(carry))


def tt_rel_diff(a, b):
    """|| A - B || / || B || for two TTs given as core lists."""
    d = len(a)
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        x, y = np.asarray(x), np.asarray(y)
        r1 = 1 if k == 0 else x.shape[0] + y.shape[0]
        r2 = 1 if k == d - 1 else x.shape[2] + y.shape[2]
        blk = np.zeros((r1, x.shape[1], r2))
        blk[:x.shape[0], :, :x.shape[2]] = x
        blk[r1 - y.shape[0]:, :, r2 - y.shape[2]:] = -y if k == 0 else y
        out.append(blk)
    return tt_norm(out) / tt_norm([np.asarray(c) for c in b])


@pytest.fixture(scope="module")
def c3(tsa):
    rng = np.random.default_rng(33)
    cores = orc.random_tt(SHAPE, S_IN, rng)
    ld, rd = orc.random_tt_drm(SHAPE, L, False, rng), orc.random_tt_drm(SHAPE, R, True, rng)
    return cores, ld, rd


def _drms(tsa, ld, rd):
    left = tsa.TensorTrainDRM(L, SHAPE, False, seed=1, cores=[np.array(c) for c in ld.cores])
    right = tsa.TensorTrainDRM(R, SHAPE, True, seed=2, cores=[np.array(c) for c in rd.cores])
    return left, right


def _orthonormal(cores):
    worst = 0.0
    for c in cores[:-1]:
        q = np.asarray(c).reshape(-1, c.shape[2])
        worst = max(worst, float(np.linalg.norm(q.T @ q - np.eye(q.shape[1]))))
    return worst


def test_c3_orthogonal_sketch_vs_oracle(tsa, c3):
    """sketch.py:81-151 + sketch_dispatch.py:160-193 at full size."""
    cores, ld, rd = c3
    left, right = _drms(tsa, ld, rd)
    tt = tsa.orthogonal_sketch(tsa.TensorTrain(cores), (L,) * 5, (R,) * 5, left_drm=left, right_drm=right)
    got = [np.asarray(c) for c in tt.cores]
    want, _ = orc.general_sketch("tt", cores, ld, rd, "orthogonal")
    assert [c.shape for c in got] == [c.shape for c in want]
    assert _orthonormal(got) < ORTH_TOL
    assert tt_rel_diff(got, want) < TENSOR_TOL


def test_c3_hmt_sketch_vs_oracle(tsa, c3):
    """sketch.py:44-78 at full size (one-sided: QR of (r n) x r unfoldings, no Omega)."""
    cores, _, rd = c3
    right = tsa.TensorTrainDRM(R, SHAPE, True, seed=2, cores=[np.array(c) for c in rd.cores])
    tt = tsa.hmt_sketch(tsa.TensorTrain(cores), (R,) * 5, drm=right)
    got = [np.asarray(c) for c in tt.cores]
    want, _ = orc.general_sketch("tt", cores, None, rd, "hmt")
    assert [c.shape for c in got] == [c.shape for c in want]
    assert _orthonormal(got) < ORTH_TOL
    assert tt_rel_diff(got, want) < TENSOR_TOL


@pytest.mark.parametrize("direction", ["right", "left"])
def test_c3_to_tt_vs_oracle(tsa, c3, direction):
    """sketch.py:400-443 at full size, both assembly directions (gelsd in the oracle)."""
    from tt_sketch_amd.sketch import assemble_sketched_tt
    cores, ld, rd = c3
    left, right = _drms(tsa, ld, rd)
    stt = tsa.stream_sketch(tsa.TensorTrain(cores), (L,) * 5, (R,) * 5, left_drm=left, right_drm=right)
    oP, oO = orc.general_sketch("tt", cores, ld, rd, "streaming")
    want = orc.assemble(oP, oO, direction)
    for dev in (False, True):
        got = [np.asarray(c) for c in assemble_sketched_tt(stt.sketch_, direction=direction, device=dev)]
        assert [c.shape for c in got] == [c.shape for c in want]
        assert tt_rel_diff(got, want) < TENSOR_TOL
    if direction == "right":            # "auto" picks it for l < r: the public call
        got = [np.asarray(c) for c in stt.to_tt().cores]
        assert tt_rel_diff(got, want) < TENSOR_TOL


def test_c3_rank_deficient_omega(tsa):
    """Input TT-rank 30 < l = 50: every Omega (50 x 100) has rank 30, the Cholesky attempts are rejected
    and the queued Jacobi fallback (gelsd's eps truncation) decides.  Exact recovery (< 1e-9, as
    tests/test_sketching_matrix.py:229) for to_tt and orthogonal_sketch, and agreement with the oracle."""
    rng = np.random.default_rng(34)
    cores = orc.random_tt(SHAPE, 30, rng)
    ld, rd = orc.random_tt_drm(SHAPE, L, False, rng), orc.random_tt_drm(SHAPE, R, True, rng)
    left, right = _drms(tsa, ld, rd)
    X = tsa.TensorTrain(cores)
    stt = tsa.stream_sketch(X, (L,) * 5, (R,) * 5, left_drm=left, right_drm=right)
    got = [np.asarray(c) for c in stt.to_tt().cores]
    assert tt_rel_diff(got, cores) < 1e-9
    oP, oO = orc.general_sketch("tt", cores, ld, rd, "streaming")
    # rank-deficient Omega: both sides truncate at their own noise floor -- the exact-recovery bar, not the full-rank one
    assert tt_rel_diff(got, orc.assemble(oP, oO, "right")) < 1e-9
    tt = tsa.orthogonal_sketch(X, (L,) * 5, (R,) * 5, left_drm=left, right_drm=right)
    got = [np.asarray(c) for c in tt.cores]
    assert tt_rel_diff(got, cores) < 1e-9
    # Q of a rank-deficient unfolding: LAPACK completes it to an orthonormal basis; so must the device
    assert _orthonormal(got) < ORTH_TOL
