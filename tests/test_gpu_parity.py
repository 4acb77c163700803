"""GPU parity: the HIP path (through the C ABI) against the golden fixtures produced by the
reference and against the CPU oracle on the same inputs.  Mirrors the reference's
tests/test_sketching_matrix.py (exact recovery, linearity, blocked == unblocked, rank increase,
left/right assembly) and tests/test_fast_lazy_gaussian.py.

Tolerances: contraction outputs ||d||_F <= 1e-12 ||ref||_F; hash integers and
sparse signs bit-exact; hash Gaussians <= 8 ulp (device log/sqrt vs libm in the tail branch of
ndtri, where x0 - x1 amplifies a 1-2 ulp difference of log); exact recovery < 1e-9.
"""
import json
import os

import numpy as np
import pytest

from oracle import ttsk_oracle as orc
from tests.golden_io import GOLDEN, Cases, rel

pytestmark = pytest.mark.gpu

CASES = Cases()
TOL = 1e-12
ULP_BAR = 8      # hash Gaussians, device vs oracle / reference fixtures: the ONE stated bar (measured max 6)


@pytest.fixture(scope="module")
def tsa():
    import tt_sketch_amd
    from tt_sketch_amd import _native
    _native.call("ttsk_init", 0)
    return tt_sketch_amd


# ------------------------------------------------------------------ contraction engine
def test_contract_against_einsum(tsa):
    from tt_sketch_amd.device import DevArray, contract
    rng = np.random.default_rng(0)
    specs = [("ij,jk->ik", (37, 53), (53, 29)), ("ji,jk->ik", (70, 33), (70, 65)),
             ("ij,ikl->jkl", (23, 11), (23, 9, 17)), ("jkl,jkm->lm", (11, 9, 17), (11, 9, 13)),
             ("ijk,ijl->kl", (1, 19, 7), (1, 19, 5)), ("ki,ikl->il", (9, 6), (6, 9, 4)),
             ("kj,jm->jkm", (8, 5), (5, 7)), ("ij,jkl->ikl", (300, 70), (70, 3, 5)),
             ("ie,je->ij", (6, 5000), (9, 5000)), ("ijk,jl->ilk", (4, 6, 5), (6, 8)),
             # every kernel family x odd / even extents x split-K (M or N <= 128: skinny families)
             ("ie,je->ij", (7, 5040), (5, 5040)), ("ie,je->ij", (8, 5041), (5, 5041)),
             ("ij,jk->ik", (100, 100), (100, 20000)), ("ij,jk->ik", (333, 100), (100, 101)),
             ("ij,jk->ik", (130, 257), (257, 131)), ("ji,jk->ik", (4001, 50), (4001, 51)),
             ("ij,jk->ik", (17, 33), (33, 4099)), ("ij,kj->ik", (4099, 35), (18, 35)),
             ("jlk,jkm->lm", (11, 13, 45), (11, 45, 7)), ("jkl,jkm->lm", (50, 40, 30), (50, 40, 20))]
    for spec, sa, sb in specs:
        A, B = rng.standard_normal(sa), rng.standard_normal(sb)
        got = contract(spec, DevArray.from_host(A), DevArray.from_host(B)).get()
        assert rel(got, np.einsum(spec, A, B)) < 1e-13, spec
    # strided views: transposed core, column slice, accumulate
    X = rng.standard_normal((12, 9, 14))
    L = rng.standard_normal((14, 20))
    Xt = DevArray.from_host(X).transpose(2, 1, 0)
    Ld = DevArray.from_host(L)[:, 3:11]
    got = contract("ij,ikl->jlk", Ld, Xt).get()
    assert rel(got, np.einsum("ij,ikl->jlk", L[:, 3:11], X.transpose(2, 1, 0))) < 1e-13
    out = DevArray.from_host(np.ones((8, 9, 12)))
    contract("ij,ikl->jkl", Ld, Xt, out=out, accumulate=True, alpha=0.5)
    assert rel(out.get(), 1 + 0.5 * np.einsum("ij,ikl->jkl", L[:, 3:11], X.transpose(2, 1, 0))) < 1e-13
    scale = rng.standard_normal(5000)
    A, B = rng.standard_normal((6, 5000)), rng.standard_normal((9, 5000))
    got = contract("ie,je->ij", DevArray.from_host(A), DevArray.from_host(B),
                   k_scale=DevArray.from_host(scale)).get()
    assert rel(got, (A * scale) @ B.T) < 1e-13


def test_chain_kernels_against_einsum(tsa):
    """The barrier-free chain kernels (csrc/skinny.h): streamed x small in both orientations, odd
    extents, K tails (K % 4, K % 20), batch index joining the streamed index, strided views,
    alpha / accumulate; long-K with one- and two-level contraction index, chunk tails, odd tile
    counts and the operand swap.  tensor_train_drm.py:79-141 / tensor_train_sketch.py:21-35."""
    from tt_sketch_amd.device import DevArray, contract
    rng = np.random.default_rng(7)
    specs = [
        # streamed x small: small operand first / second, k-fast and m-fast streams
        ("pq,np->qn", (100, 100), (20000, 100)), ("pq,pn->qn", (100, 50), (100, 20000)),
        ("mk,kc->mc", (10000, 100), (100, 100)), ("pq,np->qn", (37, 53), (5001, 37)),
        ("pq,np->qn", (1, 3), (2500, 1)), ("pq,np->qn", (50, 128), (3000, 50)),
        ("pq,np->qn", (127, 17), (2049, 127)), ("mk,kc->mc", (2100, 13), (13, 20)),
        ("pq,pn->qn", (22, 81), (22, 4100)), ("mk,kc->mc", (40000, 64), (64, 96)),
        # batch joins the streamed index (right-chain GEMM1: b = k, n = p'')
        ("pq,nkp->qkn", (100, 100), (100, 200, 100)), ("pq,nkp->qkn", (30, 21), (50, 70, 30)),
        # long-K, both operands contiguous along their output index
        ("qkp,qkm->pm", (100, 200, 100), (100, 200, 100)), ("kp,kq->pq", (10000, 100), (10000, 50)),
        ("kp,kq->pq", (4100, 2), (4100, 128)), ("kp,kq->pq", (9999, 34), (9999, 66)),
        ("qkp,qkm->pm", (7, 601, 18), (7, 601, 122)), ("qkp,qkm->pm", (300, 14, 100), (300, 14, 30)),
        ("kp,kq->pq", (5000, 128), (5000, 128)), ("kp,kq->pq", (4097, 16), (4097, 16)),
        # long-K with the longer side cut into row tiles of 128 (dense unfoldings: 20 x K times K x 4096)
        ("kp,kq->pq", (6000, 300), (6000, 20)), ("kp,kq->pq", (4100, 18), (4100, 1000)),
        ("kp,kq->pq", (70000, 130), (70000, 64)), ("kp,kq->pq", (4096, 2), (4096, 4098)),
        # long-K with k-contiguous ("generic") operands on either or both sides, odd extents
        ("pk,kq->pq", (20, 50000), (50000, 64)), ("pk,qk->pq", (64, 40000), (40, 40000)),
        ("kp,qk->pq", (9000, 300), (21, 9000)), ("pk,qk->pq", (37, 5003), (131, 5003)),
        ("kp,kq->pq", (5001, 31), (5001, 29)),
    ]
    for spec, sa, sb in specs:
        A, B = rng.standard_normal(sa), rng.standard_normal(sb)
        got = contract(spec, DevArray.from_host(A), DevArray.from_host(B)).get()
        assert rel(got, np.einsum(spec, A, B)) < 1e-13, spec
    # strided small operand (rank slice), alpha and accumulate through the streamed kernel
    R = rng.standard_normal((100, 120))
    T = rng.standard_normal((6000, 100))
    Rd = DevArray.from_host(R)[:, 10:110]
    out = DevArray.from_host(np.full((6000, 100), 2.0))
    contract("mk,kc->mc", DevArray.from_host(T), Rd, out=out, accumulate=True, alpha=-0.5)
    assert rel(out.get(), 2.0 - 0.5 * T @ R[:, 10:110]) < 1e-13
    # long-K with a genuinely two-level contraction index (k is a slice of a longer axis, so the
    # (q, k) walk wraps at every q) and chunk boundaries that fall inside a q
    Tbig, Ebig = rng.standard_normal((9, 700, 34)), rng.standard_normal((9, 650, 50))
    got = contract("qkp,qkm->pm", DevArray.from_host(Tbig)[:, 3:603, :], DevArray.from_host(Ebig)[:, 10:610, :]).get()
    assert rel(got, np.einsum("qkp,qkm->pm", Tbig[:, 3:603, :], Ebig[:, 10:610, :])) < 1e-13
    # long-K with accumulate / alpha applied by the slab reduction
    A, B = rng.standard_normal((8192, 60)), rng.standard_normal((8192, 40))
    out = DevArray.from_host(np.ones((60, 40)))
    contract("kp,kq->pq", DevArray.from_host(A), DevArray.from_host(B), out=out, accumulate=True, alpha=0.25)
    assert rel(out.get(), 1 + 0.25 * A.T @ B) < 1e-13


# ------------------------------------------------------------------ hash sampler
def test_sampler_golden(tsa):
    from tt_sketch_amd.drm import fast_lazy_gaussian as flg
    z = np.load(os.path.join(GOLDEN, "hash_sampler.npz"))
    h = z["hash_in"].copy()
    flg.hash_int_c(h)
    assert np.array_equal(h, z["hash_out"])
    meta = json.loads(str(z["meta"]))
    worst = 0.0
    for ci, c in enumerate(meta):
        idx = z[f"s{ci}_idx"]
        if idx.shape[1] > 0:
            rd = flg._inds_to_rand_double(idx.astype(np.uint64), np.array(c["shape"], dtype=np.uint64),
                                          c["rank_min"], c["rank_max"], np.uint64(c["seed"]))
            assert np.array_equal(rd.view(np.uint64), z[f"s{ci}_rand_double"].reshape(-1).view(np.uint64))
        nm = flg.inds_to_normal(idx, c["shape"], c["rank_min"], c["rank_max"], c["seed"])
        want = z[f"s{ci}_normal"]
        assert nm.shape == want.shape
        if want.size:
            ulp = np.abs(nm - want) / np.spacing(np.abs(want))
            worst = max(worst, float(ulp.max()))
        for nnz in (1, 2, c["true_rank"]):
            sg = flg.inds_to_sparse_sign(idx, c["shape"], c["true_rank"], c["rank_min"], c["rank_max"],
                                         nnz, c["seed"])
            assert sg.dtype == np.int16 and np.array_equal(sg, z[f"s{ci}_sign_nnz{nnz}"])
    assert worst <= ULP_BAR, f"hash Gaussians differ by {worst} ulp"


def test_sampler_properties(tsa):
    """reference tests/test_fast_lazy_gaussian.py: input untouched, permutation equivariance,
    rank-extension prefix property, normality."""
    from tt_sketch_amd.drm import fast_lazy_gaussian as flg
    rng = np.random.default_rng(5)
    shape = (50, 60, 70)
    N = 20000
    idx = np.stack([rng.integers(0, n, N) for n in shape])
    keep = idx.copy()
    a = flg.inds_to_normal(idx, shape, 0, 4, 11)
    assert np.array_equal(idx, keep)
    perm = rng.permutation(N)
    assert np.array_equal(flg.inds_to_normal(idx[:, perm], shape, 0, 4, 11), a[perm])
    b = flg.inds_to_normal(idx, shape, 0, 9, 11)
    assert np.array_equal(b[:, :4], a)
    assert np.array_equal(flg.inds_to_normal(idx, shape, 2, 9, 11), b[:, 2:])
    srt = np.sort(b.ravel())
    from scipy.special import ndtr
    assert np.max(np.abs(ndtr(srt) - (np.arange(srt.size) + 0.5) / srt.size)) < 0.01
    # every width of the row-tile kernel (incl. 26..32: tile + tail queue beyond 64 KB of LDS) and the wide
    # kernel (> 32) against the oracle at the one stated bar (ULP_BAR), ragged last tile
    sub = idx[:, :1003]
    for w in (1, 7, 25, 26, 32, 33, 40):
        got = flg.inds_to_normal(sub, shape, 3, 3 + w, 99)
        want = orc.inds_to_normal(sub, shape, 3, 3 + w, 99)
        assert got.shape == want.shape == (1003, w)
        assert np.max(np.abs(got - want) / np.spacing(np.abs(want))) <= ULP_BAR
    s = flg.inds_to_sparse_sign(idx, shape, 12, 0, 12, 3, 7)
    assert set(np.unique(s)) <= {-1, 0, 1} and np.all(np.sum(s != 0, axis=1) == 3)
    assert np.array_equal(s, orc.inds_to_sparse_sign(idx, shape, 12, 0, 12, 3, 7))
    want = orc.inds_to_normal(idx, shape, 0, 9, 11)
    assert np.max(np.abs(b - want) / np.spacing(np.abs(want))) <= ULP_BAR
    assert np.mean(b == want) > 0.5


def test_device_normal_sampler(tsa):
    """A3 / A9: the counter-based N(0,1) fill behind every default-seeded TensorTrainDRM /
    DenseGaussianDRM (ttsk_fill_normal; reference tensor.py:358-371 via utils.py:178-227,
    dense_gaussian_drm.py:44-57).  The reference's stream is host dependent, so the
    generator is validated statistically and by injecting ITS samples into the oracle."""
    from scipy.special import ndtr
    from tt_sketch_amd.utils import random_normal_dev
    n = 1_000_003                                    # odd length: the ragged tail of the fill kernel
    x = random_normal_dev((n,), seed=1234).get()
    assert np.all(np.isfinite(x))
    assert abs(x.mean()) < 5 / np.sqrt(n) and abs(x.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(np.mean(x ** 3)) < 5 * np.sqrt(15 / n) and abs(np.mean(x ** 4) - 3) < 5 * np.sqrt(96 / n)
    srt = np.sort(x)
    ks = np.max(np.abs(ndtr(srt) - (np.arange(n) + 0.5) / n))
    assert ks < 1.95 / np.sqrt(n), ks                # Kolmogorov-Smirnov at alpha = 0.001
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5 / np.sqrt(n)      # neighbours uncorrelated
    # a pure function of (seed, element index): prefix stable, reproducible, scale is a factor
    assert np.array_equal(random_normal_dev((5000,), seed=1234).get(), x[:5000])
    assert np.array_equal(random_normal_dev((n,), seed=1234).get(), x)
    assert np.allclose(random_normal_dev((5000,), seed=1234, scale=0.25).get(), 0.25 * x[:5000], rtol=1e-15, atol=0)
    y = random_normal_dev((n,), seed=1235).get()
    assert not np.any(y == x) or np.mean(y == x) < 1e-5
    assert abs(np.corrcoef(x, y)[0, 1]) < 5 / np.sqrt(n)               # streams of different seeds independent


def test_default_seeded_drms(tsa):
    """TensorTrainDRM(seed=...) samples d - 1 cores N(0, 1/r_{mu-1}) (norm-preserve, tensor.py:370-371; the
    last core is never built, tensor_train_drm.py:56), reproducibly per seed; DenseGaussianDRM samples
    N(0,1) rows whose leading block survives a rank increase.  Feeding the device-sampled data to the
    ORACLE reproduces the device sketch."""
    shape, l, r = (40, 30, 50, 20, 35), 24, 31
    left = tsa.TensorTrainDRM(l, shape, False, seed=5)
    right = tsa.TensorTrainDRM(r, shape, True, seed=6)
    assert len(left.cores) == len(right.cores) == len(shape) - 1
    for drm, walk, rk in ((left, shape, l), (right, shape[::-1], r)):
        for mu, c in enumerate(drm.cores):
            h = np.asarray(c)
            r1 = 1 if mu == 0 else rk
            assert h.shape == (r1, walk[mu], rk)
            m = h.size
            assert abs(h.var() * r1 - 1) < 6 * np.sqrt(2 / m), (mu, h.var() * r1)
            assert abs(h.mean()) < 6 / np.sqrt(m * r1)
    again = tsa.TensorTrainDRM(l, shape, False, seed=5)
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(left.cores, again.cores))
    other = tsa.TensorTrainDRM(l, shape, False, seed=7)
    assert not np.array_equal(np.asarray(left.cores[1]), np.asarray(other.cores[1]))
    # norm preservation in expectation: E ||L_0||_F^2 = ||X_0||_F^2 for the first chain step
    rng = np.random.default_rng(0)
    cores = orc.random_tt(shape, 6, rng)
    tt = tsa.TensorTrain(cores)
    stt = tsa.stream_sketch(tt, (l,) * 4, (r,) * 4, left_drm=left, right_drm=right)
    old = orc.TTDrm([np.asarray(c) for c in left.cores], shape, False)
    ord_ = orc.TTDrm([np.asarray(c) for c in right.cores], shape, True)
    oP, oO = orc.general_sketch("tt", cores, old, ord_, "streaming")
    for a, b in zip(stt.Psi_cores + stt.Omega_mats, oP + oO):
        assert rel(a, b) < TOL
    ratios = []
    for seed in range(40):
        d0 = np.asarray(tsa.TensorTrainDRM(200, (30, 8), False, seed=100 + seed).cores[0])   # (1, 30, 200)
        v = rng.standard_normal(30)
        ratios.append(np.sum((v @ d0[0]) ** 2) / 200 / np.sum(v ** 2))
    assert abs(np.mean(ratios) - 1) < 0.1          # each ratio has std sqrt(2/200) = 0.1
    # DenseGaussianDRM: N(0,1), rank_min:rank_max rows are views of the same sample, increase_rank keeps the block
    dg = tsa.DenseGaussianDRM((5, 6, 7), (9, 8, 7, 6), False, seed=3)
    mats = [np.asarray(m) for m in dg.sketching_mats]
    assert [m.shape for m in mats] == [(5, 9), (6, 72), (7, 504)]
    big = tsa.DenseGaussianDRM(3000, (10, 12), False, seed=3)
    g = np.asarray(big.sketching_mats[0])
    assert abs(g.var() - 1) < 6 * np.sqrt(2 / g.size) and abs(g.mean()) < 6 / np.sqrt(g.size)
    grown = dg.increase_rank((8, 9, 9))
    for a, b in zip(mats, grown.sketching_mats):
        assert np.array_equal(np.asarray(b)[:a.shape[0]], a)
    sl = dg.slice((1, 2, 3), (4, 5, 6))
    for a, b, lo, hi in zip(mats, sl.sketching_mats, (1, 2, 3), (4, 5, 6)):
        assert np.array_equal(np.asarray(b), a[lo:hi])
    # dense-DRM sketch of a TT with the device-sampled matrices injected into the oracle
    dshape = (9, 8, 7, 6)
    dcores = orc.random_tt(dshape, 3, rng)
    dl = tsa.DenseGaussianDRM((5, 6, 5), dshape, False, seed=8)
    dr = tsa.DenseGaussianDRM((7, 8, 7), dshape, True, seed=9)
    sk = tsa.general_sketch(tsa.TensorTrain(dcores), dl, dr, tsa.SketchMethod.streaming)
    ol = orc.DenseDrm([np.asarray(m) for m in dl.sketching_mats], dshape, False)
    orr = orc.DenseDrm([np.asarray(m) for m in dr.sketching_mats], dshape, True)
    oP, oO = orc.general_sketch("tt", dcores, ol, orr, "streaming")
    for a, b in zip(sk.Psi_cores + sk.Omega_mats, oP + oO):
        assert rel(a, b) < TOL


# ------------------------------------------------------------------ golden sketch cases
@pytest.mark.parametrize("name", CASES.names())
def test_contractions_golden(tsa, name):
    from tests.gpu_build import make_drm, make_tensor
    kind, data = CASES.tensor(name)
    tensor = make_tensor(kind, data)
    from tt_sketch_amd.sketch_dispatch import get_sketch_method
    for side in ("left", "right"):
        drm = make_drm(CASES.drm(name, side))
        got = list(get_sketch_method(tensor, drm)(tensor))
        if kind == "sum":
            for s in range(len(data)):
                want = CASES.out(name, f"{side}_contractions_s{s}")
                for g, w in zip(got, want):
                    assert rel(np.asarray(g[s]), w) < TOL, (side, s)
        else:
            want = CASES.out(name, f"{side}_contractions")
            assert len(got) == len(want)
            for mu, (g, w) in enumerate(zip(got, want)):
                assert rel(np.asarray(g), w) < TOL, (side, mu)


@pytest.mark.parametrize("name", CASES.names())
def test_sketch_golden(tsa, name):
    from tests.gpu_build import make_drm, make_tensor
    from tt_sketch_amd.sketch import assemble_sketched_tt
    kind, data = CASES.tensor(name)
    tensor = make_tensor(kind, data)
    m = CASES.meta[name]
    for method in m["methods"]:
        left = None if method == "hmt" else make_drm(CASES.drm(name, "left"))
        right = make_drm(CASES.drm(name, "right"))
        sk = tsa.general_sketch(tensor, left, right, tsa.SketchMethod(method))
        wantP, wantO = CASES.out(name, f"{method}/Psi"), CASES.out(name, f"{method}/Omega")
        assert len(sk.Psi_cores) == len(wantP) and len(sk.Omega_mats) == len(wantO)
        for g, w in zip(sk.Omega_mats, wantO):
            assert rel(g, w) < TOL
        if method == "streaming":
            for g, w in zip(sk.Psi_cores, wantP):
                assert rel(g, w) < TOL
            if not m.get("sliced"):
                # pinv: Jacobi SVD vs gelsd; compare at the tensor level (gauge/threshold free)
                for direction in ("left", "right"):
                    C = assemble_sketched_tt(sk, direction=direction)
                    wantC = CASES.out(name, f"{method}/C_{direction}")
                    assert [c.shape for c in C] == [c.shape for c in wantC]
                    a, b = orc.tt_to_numpy(C), orc.tt_to_numpy(wantC)
                    assert rel(a, b) < 1e-8
        else:
            # QR gauge: compare the represented tensor and orthonormality
            a, b = orc.tt_to_numpy(sk.Psi_cores), orc.tt_to_numpy(wantP)
            assert rel(a, b) < 1e-8
            for P in sk.Psi_cores[:-1]:
                Q = P.reshape(-1, P.shape[2])
                assert np.linalg.norm(Q.T @ Q - np.eye(Q.shape[1])) < 1e-12


# ------------------------------------------------------------------ solves
def test_pinv_and_qr(tsa):
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    from tt_sketch_amd.utils import left_mul_pinv, right_mul_pinv
    rng = np.random.default_rng(3)
    for l, r in [(5, 9), (9, 5), (50, 100), (7, 7), (1, 4), (130, 70)]:
        Om = rng.standard_normal((l, r))
        A = rng.standard_normal((40, r))
        assert rel(right_mul_pinv(A, Om), orc.right_mul_pinv(A, Om)) < 1e-10
        B = rng.standard_normal((l, 33))
        assert rel(left_mul_pinv(Om, B), orc.left_mul_pinv(Om, B)) < 1e-10
    # rank deficient Omega: truncation like gelsd (cond = eps)
    Om = rng.standard_normal((8, 3)) @ rng.standard_normal((3, 12))
    A = rng.standard_normal((20, 12))
    assert rel(right_mul_pinv(A, Om), orc.right_mul_pinv(A, Om)) < 1e-8
    for m, n in [(30, 7), (1000, 50), (64, 64), (513, 33), (5, 1)]:
        M = rng.standard_normal((m, n))
        d = DevArray.from_host(M)
        nat.call("ttsk_qr_thin", ctypes.c_void_p(d.ptr), m, n, 0)
        Q = d.get()
        import scipy.linalg
        Qref, _ = scipy.linalg.qr(M, mode="economic")
        assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12 * n
        assert rel(Q, Qref) < 1e-10, (m, n)   # same Householder sign convention as LAPACK
    # rank-deficient input: CholeskyQR2 is rejected and the Householder kernels take over
    M = rng.standard_normal((400, 3)) @ rng.standard_normal((3, 12))
    d = DevArray.from_host(M)
    nat.call("ttsk_qr_thin", ctypes.c_void_p(d.ptr), 400, 12, 0)
    Q = d.get()
    assert np.linalg.norm(Q.T @ Q - np.eye(12)) < 1e-11
    assert rel(Q @ (Q.T @ M), M) < 1e-10
    # ill-conditioned but full rank (kappa ~ 1e9 > the CholeskyQR2 gate): still LAPACK's Q
    U, _ = np.linalg.qr(rng.standard_normal((300, 8)))
    V, _ = np.linalg.qr(rng.standard_normal((8, 8)))
    M = (U * np.logspace(0, -9, 8)) @ V.T
    d = DevArray.from_host(M)
    nat.call("ttsk_qr_thin", ctypes.c_void_p(d.ptr), 300, 8, 0)
    Q = d.get()
    assert np.linalg.norm(Q.T @ Q - np.eye(8)) < 1e-11
    assert rel(Q @ (Q.T @ M), M) < 1e-6
    # moderately ill-conditioned Omega (kappa ~ 1e4): normal equations rejected, Jacobi SVD result
    Om = (np.linalg.qr(rng.standard_normal((20, 20)))[0] * np.logspace(0, -4, 20)) @ rng.standard_normal((20, 35))
    A = rng.standard_normal((15, 35))
    assert rel(right_mul_pinv(A, Om), orc.right_mul_pinv(A, Om)) < 1e-8


# ------------------------------------------------------------------ API-level properties
@pytest.mark.parametrize("method", ["streaming", "orthogonal", "hmt"])
@pytest.mark.parametrize("kind", ["tt", "cp", "tucker", "dense", "sparse"])
def test_exact_recovery_defaults(tsa, kind, method):
    """reference test_sketching_matrix.py:661-693: every tensor kind x method with default DRMs."""
    n_dims, rank, seed = 3, 3, 180
    shape = tuple(range(9, 9 + n_dims))
    X_tt = tsa.TensorTrain.random(shape, rank, seed=seed)
    X = X_tt.to_numpy()
    left_rank = tuple(range(rank, rank + n_dims - 1))
    right_rank = tuple(range(rank + 1, rank + n_dims))
    if kind == "tt":
        T = X_tt
    elif kind == "cp":
        T = tsa.CPTensor.random(shape, rank, seed=seed)
        X = T.to_numpy()
    elif kind == "tucker":
        T = tsa.TuckerTensor.random(shape, rank, seed=seed)
        X = T.to_numpy()
    elif kind == "dense":
        T = tsa.DenseTensor(X)
    else:
        T = tsa.DenseTensor(X).to_sparse()
    if method == "hmt":
        out = tsa.hmt_sketch(T, right_rank, seed=seed)
        out2 = tsa.hmt_sketch(T, right_rank, seed=seed + 1)
    elif method == "orthogonal":
        out = tsa.orthogonal_sketch(T, left_rank, right_rank, seed=seed)
        out2 = tsa.orthogonal_sketch(T, left_rank, right_rank, seed=seed + 1)
    else:
        out = tsa.stream_sketch(T, left_rank, right_rank, seed=seed)
        out2 = tsa.stream_sketch(T, left_rank, right_rank, seed=seed + 1)
    assert out.error(X) < 1e-9
    assert not np.all(out.to_numpy() == out2.to_numpy())


def test_linearity_streaming_update_and_blocks(tsa):
    """reference :410-449 (linearity over TensorSum, stt + X) and :137-187 (blocked)."""
    seed, rank, n_dims = 179, 4, 4
    shape = tuple(range(7, 7 + n_dims))
    X1 = tsa.TensorTrain.random(shape, rank, seed=1)
    left_rank = tuple(range(rank, rank + n_dims - 1))
    right_rank = tuple(range(rank + 1, rank + n_dims))
    sp = X1.dense().to_sparse()
    s16, s2 = sp.split(16), sp.split(2)
    a = tsa.stream_sketch(s16, left_rank, right_rank, seed)
    b = tsa.stream_sketch(sp, left_rank, right_rank, seed)
    c = tsa.stream_sketch(s2, left_rank, right_rank, seed)
    for Y1, Y2, Y3 in zip(a.Psi_cores + a.Omega_mats, b.Psi_cores + b.Omega_mats,
                          c.Psi_cores + c.Omega_mats):
        assert np.allclose(Y1, Y2) and np.allclose(Y1, Y3)
    X2 = tsa.TensorTrain.random(shape, rank, seed=2)
    big_l, big_r = tuple(r + 10 for r in left_rank), tuple(r + 10 for r in right_rank)
    stt = tsa.stream_sketch(sp + X2, big_l, big_r, seed)
    full = X1.to_numpy() + X2.to_numpy()
    assert stt.to_tt().error(full) < 1e-8
    stt6 = tsa.stream_sketch(sp, big_l, big_r, seed) + X2
    assert stt6.error(full) < 1e-8
    # blocked == unblocked with sliceable DRMs
    for drm_type in (tsa.SparseGaussianDRM, tsa.TensorTrainDRM):
        ld = drm_type(big_l, shape, False, seed=seed)
        rd = drm_type(big_r, shape, True, seed=seed + 1)
        whole = tsa.general_sketch(sp, ld, rd, tsa.SketchMethod.streaming)
        cut_l = [(0,) * 3, tuple(r // 2 for r in big_l), big_l]
        cut_r = [(0,) * 3, tuple(r // 3 for r in big_r), big_r]
        blocked = tsa.blocked_stream_sketch(sp, ld, rd, cut_l, cut_r)
        for Y1, Y2 in zip(whole.Psi_cores + whole.Omega_mats, blocked.Psi_cores + blocked.Omega_mats):
            assert np.allclose(Y1, Y2, rtol=1e-9, atol=1e-11), drm_type.__name__


def test_rank_increase(tsa):
    """reference :41-130 with the hash DRM (the only CanIncreaseRank DRMs on sparse input)."""
    shape, rank, seed = (9, 10, 11), 3, 180
    X = tsa.TensorTrain.random(shape, rank, seed=seed).dense().to_sparse()
    left_rank, right_rank = (3, 4), (4, 5)
    ld = tsa.SparseGaussianDRM(left_rank, shape, False, seed=seed)
    rd = tsa.SparseGaussianDRM(right_rank, shape, True, seed=seed + 7)
    s1 = tsa.stream_sketch(X, left_rank, right_rank, left_drm=ld, right_drm=rd)
    nl, nr = (5, 6), (7, 8)
    s3 = s1.increase_rank(X, nl, nr)
    s2 = tsa.stream_sketch(X, nl, nr, left_drm=ld.increase_rank(nl), right_drm=rd.increase_rank(nr))
    lp, rp = (1,) + left_rank, right_rank + (1,)
    for other in (s2, s3):
        for i, (Y1, Y2) in enumerate(zip(s1.Psi_cores, other.Psi_cores)):
            assert np.allclose(Y1, Y2[:lp[i], :, :rp[i]])
        assert [Z.shape for Z in other.Omega_mats] == [(a, b) for a, b in zip(nl, nr)]
    s4 = tsa.stream_sketch(X, left_rank, right_rank, left_drm=ld.increase_rank(nl).slice(None, left_rank),
                           right_drm=rd.increase_rank(nr).slice(None, right_rank))
    for Y1, Y2 in zip(s1.Psi_cores + s1.Omega_mats, s4.Psi_cores + s4.Omega_mats):
        assert np.allclose(Y1, Y2, rtol=1e-12, atol=1e-13)


def test_massive_oversample_and_errors(tsa):
    """reference :309-335 (left/right assembly agree, resulting ranks) and the ValueErrors."""
    X = tsa.TensorTrain.random((5, 6, 7, 8), 5, seed=4)
    for lr, rr in ((100, 90), (90, 100)):
        stt = tsa.stream_sketch(X, lr, rr, seed=180)
        lt, rt = tsa.TensorTrain(stt.C_cores("left")), tsa.TensorTrain(stt.C_cores("right"))
        assert np.allclose(lt.to_numpy(), rt.to_numpy())
        assert lt.rank == stt.right_rank and rt.rank == stt.left_rank
    with pytest.raises(ValueError):
        tsa.stream_sketch(X, (3, 5, 3), (4, 4, 4))
    with pytest.raises(ValueError):
        tsa.orthogonal_sketch(X, 5, 4)
    with pytest.raises(ValueError):
        tsa.stream_sketch(X, (3, 4), (5, 6))
    with pytest.raises(ValueError):
        bad = tsa.TensorTrainDRM(3, (5, 6, 7, 9), False, seed=1)
        list(bad.sketch_tt(X))
    with pytest.raises(ValueError):
        tsa.stream_sketch(X, (3, 3, 3), (4, 4, 4), left_drm=tsa.TensorTrainDRM(4, X.shape, False, seed=1))


def test_one_call_tt_path_matches_generic_and_oracle(tsa):
    """ttsk_tt_sketch (one C call) == generator protocol path == oracle, incl. rank slices
    and TensorSum accumulation."""
    from tt_sketch_amd import tt_fused
    from tt_sketch_amd.sketch_dispatch import general_sketch_device
    from tests.gpu_build import make_drm, make_tensor
    rng = np.random.default_rng(11)
    shape, s = (9, 12, 7, 10, 8), (3, 6, 5, 4)
    lr, rr = (4, 7, 6, 5), (6, 9, 8, 7)
    terms = [orc.random_tt(shape, s, rng) for _ in range(3)]
    ld = orc.random_tt_drm(shape, lr, False, rng)
    rd = orc.random_tt_drm(shape, rr, True, rng)
    ld.rank_min, ld.rank_max = (1, 0, 2, 0), (4, 6, 6, 5)
    rd.rank_min, rd.rank_max = (0, 3, 1, 2), (5, 8, 9, 6)       # walking order of the right DRM
    for data in (("tt", terms[0]), ("sum", [("tt", t) for t in terms])):
        T = make_tensor(*data)
        L, R = make_drm(ld), make_drm(rd)
        fused = tt_fused.try_stream_sketch(T, L, R, tsa.SketchMethod.streaming)
        assert fused is not None
        gen = general_sketch_device(T, L, R, tsa.SketchMethod.streaming)
        oP, oO = orc.general_sketch(data[0], data[1], ld, rd, "streaming")
        for a, b, c in zip(fused[0] + fused[1], gen[0] + gen[1], oP + oO):
            assert rel(a.get(), c) < TOL and rel(b.get(), c) < TOL


def test_rccl_single_rank(tsa):
    """The RCCL path of the partial-sketch sum with a communicator of one rank: all-reduce and
    reduce leave the buffer unchanged and the calls succeed on a library stream.  Runs in a child
    process under a time limit (communicator set-up loads RCCL's kernels, ~6 s, and must not be
    able to stall the suite)."""
    import subprocess
    import sys
    code = (
        "import ctypes, numpy as np\n"
        "from tt_sketch_amd import _native as nat\n"
        "from tt_sketch_amd.device import DevArray, sync\n"
        "from tt_sketch_amd.distributed import RcclComm\n"
        "nat.call('ttsk_init', 0)\n"
        "x = np.random.default_rng(9).standard_normal(100003)\n"
        "buf = DevArray.from_host(x)\n"
        "comm = RcclComm(0, 1, lambda b: b)\n"
        "comm.allreduce_sum(buf, stream=2)\n"
        "comm.reduce_sum(buf, root=0, stream=2)\n"
        "sync()\n"
        "assert np.array_equal(buf.get(), x)\n"
        "comm.close()\n"
        "print('rccl single rank ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired as exc:      # a hang is a failure (the child has been killed and reaped by subprocess.run)
        pytest.fail(f"RCCL communicator set-up did not return within 150 s; stderr {(exc.stderr or b'')[-2000:]!r}")
    assert res.returncode == 0 and "rccl single rank ok" in res.stdout, res.stderr[-2000:]


def test_sum_slices(tsa):
    """ttsk_sum_slices == the `+=` loop over partial sketches (sketch_dispatch.py:141-147)."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(5)
    nb, n, stride = 5, 10000, 10006
    src = rng.standard_normal(nb * stride)
    dst0 = rng.standard_normal(n)
    want = sum(src[b * stride:b * stride + n] for b in range(nb))
    for acc in (0, 1):
        d, s_ = DevArray.from_host(dst0), DevArray.from_host(src)
        nat.call("ttsk_sum_slices", ctypes.c_void_p(d.ptr), ctypes.c_void_p(s_.ptr), nb, ctypes.c_size_t(stride),
                 ctypes.c_size_t(n), acc, 0)
        assert rel(d.get(), want + acc * dst0) < 1e-15
    # odd length / odd stride / a base that is only 8-byte aligned: the scalar variant
    for n2, stride2, shift in ((n - 1, stride, 0), (n, stride - 1, 0), (n - 2, stride, 1)):
        d, s_ = DevArray.from_host(dst0), DevArray.from_host(src)
        nat.call("ttsk_sum_slices", ctypes.c_void_p(d.ptr), ctypes.c_void_p(s_.ptr + 8 * shift), nb,
                 ctypes.c_size_t(stride2), ctypes.c_size_t(n2), 1, 0)
        want2 = dst0[:n2] + sum(src[shift + b * stride2:shift + b * stride2 + n2] for b in range(nb))
        got = d.get()
        assert rel(got[:n2], want2) < 1e-15 and np.array_equal(got[n2:], dst0[n2:])


def test_batched_one_call_path(tsa):
    """ttsk_tt_sketch_batch: nb tensors of one signature in one pass give, tensor by tensor, the
    sketch of the single-tensor call and of the oracle (rank slices included; nb > 32 is sliced)."""
    import ctypes
    from tt_sketch_amd import tt_fused
    from tests.gpu_build import make_drm, make_tensor
    rng = np.random.default_rng(23)
    for shape, ranks, lr, rr, nb in [((9, 12, 7, 10, 8), (3, 6, 5, 4), (4, 7, 6, 5), (6, 9, 8, 7), 3),
                                     ((64, 48, 64), (40, 36), (20, 24), (30, 28), 2),
                                     # large enough for the chain kernels: even ranks (long-K kernel) and odd
                                     # ranks (streamed kernel with tails, generic split-K for GEMM2)
                                     ((150, 160, 150, 140), (30, 32, 28), (26, 24, 22), (28, 30, 34), 3),
                                     ((150, 151, 149), (31, 29), (25, 27), (33, 35), 2),
                                     ((5, 4, 6), (3, 2), (2, 3), (4, 3), 35)]:
        ld, rd = orc.random_tt_drm(shape, lr, False, rng), orc.random_tt_drm(shape, rr, True, rng)
        if len(shape) == 5:
            ld.rank_min, ld.rank_max = (1, 0, 2, 0), (4, 6, 6, 5)
            rd.rank_min, rd.rank_max = (0, 3, 1, 2), (5, 8, 9, 6)       # walking order of the right DRM
        L, R = make_drm(ld), make_drm(rd)
        tts = [orc.random_tt(shape, ranks, rng) for _ in range(nb)]
        dev = [make_tensor("tt", c) for c in tts]
        plan = tt_fused.TTSketchPlan(dev[0].shape, dev[0].rank, L, R)
        keep, flat = [], []
        for t in dev:
            ptrs, k = plan.core_pointers(t)
            keep.append(k)
            flat += [ptrs[i] for i in range(plan.d)]
        X = (ctypes.c_void_p * len(flat))(*flat)
        stride = plan.size + 5
        from tt_sketch_amd.device import DevArray
        out = DevArray.zeros((nb * stride,))
        plan.run_batch(X, nb, out, stride)
        for b, cores in enumerate(tts):
            oP, oO = orc.general_sketch("tt", cores, ld, rd, "streaming")
            Psi, Om = plan.views(out[b * stride:b * stride + plan.size])
            for a, c in zip(Psi + Om, oP + oO):
                assert rel(a.get(), c) < TOL


def test_sketch_of_a_sum_in_one_call(tsa):
    """ttsk_tt_sketch_sum: the sketch of a sum of nb TTs of one signature == the sum of the oracle's sketches
    (sketch_dispatch.py:85-139), for odd and even sizes (merged and two-level contracted index), rank slices,
    nb > 32 (sliced by the library), accumulate, and shapes large enough for the chain kernels (the interleaved
    intermediate of the two-launch step) with and without the fused step kernel's cover."""
    import ctypes
    from tt_sketch_amd import tt_fused
    from tt_sketch_amd.device import DevArray
    from tests.gpu_build import make_drm, make_tensor
    rng = np.random.default_rng(29)
    for shape, ranks, lr, rr, nb in [((9, 12, 7, 10, 8), (3, 6, 5, 4), (4, 7, 6, 5), (6, 9, 8, 7), 3),
                                     ((8, 6, 10, 4), (2, 4, 6), (4, 6, 2), (6, 8, 4), 5),
                                     ((5, 4, 6), (3, 2), (2, 3), (4, 3), 35),
                                     ((150, 160, 150, 140), (30, 32, 28), (26, 24, 22), (28, 30, 34), 3),
                                     ((150, 151, 149), (31, 29), (25, 27), (33, 35), 2),
                                     ((128, 128, 128, 128), (20, 20, 20), (50, 50, 50), (100, 100, 100), 6),
                                     ((64, 64, 64, 64), (100, 100, 100), (50, 50, 50), (100, 100, 100), 4),
                                     # TT ranks beyond 48: Psi of the sum in one launch, accumulators kept over the terms (stream_small_sum_kernel)
                                     ((70, 66, 68, 40), (52, 57, 49), (26, 28, 30), (54, 58, 70), 3),
                                     ((40, 150, 30), (150, 60), (20, 40), (110, 45), 5)]:
        ld, rd = orc.random_tt_drm(shape, lr, False, rng), orc.random_tt_drm(shape, rr, True, rng)
        if len(shape) == 5:
            ld.rank_min, ld.rank_max = (1, 0, 2, 0), (4, 6, 6, 5)
            rd.rank_min, rd.rank_max = (0, 3, 1, 2), (5, 8, 9, 6)       # walking order of the right DRM
        L, R = make_drm(ld), make_drm(rd)
        tts = [orc.random_tt(shape, ranks, rng) for _ in range(nb)]
        dev = [make_tensor("tt", c) for c in tts]
        plan = tt_fused.TTSketchPlan(dev[0].shape, dev[0].rank, L, R)
        keep, flat = [], []
        for t in dev:
            ptrs, k = plan.core_pointers(t)
            keep.append(k)
            flat += [ptrs[i] for i in range(plan.d)]
        X = (ctypes.c_void_p * len(flat))(*flat)
        want = None
        for cores in tts:
            oP, oO = orc.general_sketch("tt", cores, ld, rd, "streaming")
            want = [a + b for a, b in zip(want, oP + oO)] if want else list(oP + oO)
        out = DevArray.empty((plan.size,))
        plan.run_sum(X, nb, out)
        Psi, Om = plan.views(out)
        for a, c in zip(Psi + Om, want):
            assert a.shape == c.shape and rel(a.get(), c) < 1e-12, (shape, nb, rel(a.get(), c))
        plan.run_sum(X, nb, out, accumulate=True)              # out += the same sketch
        for a, c in zip(Psi + Om, want):
            assert rel(a.get(), 2 * c) < 1e-12, (shape, nb, "accumulate")
        # ... and through the public API
        sk = tsa.general_sketch(tsa.TensorSum(dev), L, R, tsa.SketchMethod.streaming)
        for a, c in zip(sk.Psi_cores + sk.Omega_mats, want):
            assert rel(np.asarray(a), c) < 1e-12


def test_round_on_device(tsa):
    """TensorTrain.round_dev / orthogonalize_dev == the host versions (reference tensor.py:446-484,
    :559-572) as tensors; orthogonality and rank rules identical; ttsk_svd_small reconstructs."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(17)
    A = rng.standard_normal((40, 40))
    d = DevArray.from_host(A)
    US, S, Vt = DevArray.empty((40, 40)), DevArray.empty((40,)), DevArray.empty((40, 40))
    nat.call("ttsk_svd_small", ctypes.c_void_p(d.ptr), 40, 40, ctypes.c_void_p(US.ptr), ctypes.c_void_p(S.ptr),
             ctypes.c_void_p(Vt.ptr), 0)
    assert rel(US.get() @ Vt.get(), A) < 1e-12
    assert rel(S.get(), np.linalg.svd(A, compute_uv=False)) < 1e-12
    assert np.linalg.norm(Vt.get() @ Vt.get().T - np.eye(40)) < 1e-12
    for shape, ranks, kw in [((9, 12, 7, 10), (5, 8, 6), dict(max_rank=4)),
                             ((30, 40, 30, 20, 30), (20, 35, 35, 18), dict(max_rank=12)),
                             ((30, 40, 30, 20, 30), (20, 35, 35, 18), dict(eps=0.05)),
                             ((16, 16, 16), (8, 8), dict()),
                             ((3, 4, 3, 5), (9, 20, 11), dict(max_rank=5)),      # infeasible (wide / tall) ranks
                             ((3, 4, 3, 5), (9, 20, 11), dict(eps=1e-3))]:
        tt = tsa.TensorTrain(orc.random_tt(shape, ranks, rng))
        o = tt.orthogonalize_dev()
        assert rel(o.to_numpy(), tt.to_numpy()) < 1e-12
        assert o.resident() and tt.round_dev(**kw).resident()
        for C in o.cores[:-1]:
            Q = C.get().reshape(-1, C.shape[2])
            assert np.linalg.norm(Q.T @ Q - np.eye(Q.shape[1])) < 1e-11
        want = tt.round(**kw)
        got = tt.round_dev(**kw)
        assert got.rank == want.rank
        assert rel(got.to_numpy(), want.to_numpy()) < 1e-10


# ------------------------------------------------------------------ TT-GMRES
def _gmres_problem(tsa):
    from tt_sketch_amd.tt_gmres import MPO, TTLinearMapSum, TTPrecond
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmres_case.npz"))
    shape = tuple(int(n) for n in z["shape"])
    d = len(shape)
    maps = [[z[f"map{m}_core{k}"] for k in range(d)] for m in range(3)]
    b = tsa.TensorTrain([z[f"b_core{k}"] for k in range(d)])
    A = TTLinearMapSum([MPO(list(cores)) for cores in maps])
    return z, shape, maps, b, A, TTPrecond(z["precond"], shape, mode=1)


def test_mpo_and_precond_on_device(tsa):
    """MPO.__call__ / TTPrecond against the reference's outputs (tt_gmres.py:90-101, :137-168) and
    the reference's own test_mpo_contract property."""
    from tt_sketch_amd.tt_gmres import MPO
    z, shape, maps, b, A, pre = _gmres_problem(tsa)
    assert rel(MPO(maps[2])(b).to_numpy(), z["mpo_apply"]) < 1e-13
    assert rel(pre.backward_call(b).to_numpy(), z["precond_backward"]) < 1e-12
    assert rel(pre.forward_call(b).to_numpy(), z["precond_forward"]) < 1e-13
    np.random.seed(5)
    mpo = MPO.random(3, (4, 5, 6), (4, 5, 6))
    dense = mpo.to_numpy()
    assert np.linalg.norm(dense - dense.transpose(1, 0, 3, 2, 5, 4)) < 1e-12
    tt = tsa.TensorTrain.random((4, 5, 6), 2, seed=1)
    assert np.linalg.norm(np.einsum("ijk,iajbkc", tt.to_numpy(), dense) - mpo(tt).to_numpy()) < 1e-12
    assert rel(mpo.T(tt).to_numpy(), np.einsum("ijk,aibjck", tt.to_numpy(), dense)) < 1e-12
    with pytest.raises(ValueError):
        mpo(tsa.TensorTrain.random((4, 5, 7), 2, seed=1))


def test_resident_tt_arithmetic(tsa):
    rng = np.random.default_rng(3)
    a = tsa.TensorTrain(orc.random_tt((5, 6, 7, 4), (3, 4, 2), rng))
    b = tsa.TensorTrain(orc.random_tt((5, 6, 7, 4), (2, 5, 3), rng))
    ad, bd = a.to_device(), b.to_device()
    assert ad.resident() and not a.resident()
    assert rel(ad.add(bd).to_numpy(), a.to_numpy() + b.to_numpy()) < 1e-14
    assert ad.add(bd).rank == a.add(b).rank
    assert abs(ad.dot(bd) - a.dot(b)) < 1e-13 * a.norm() * b.norm()
    assert abs(ad.norm() - a.norm()) < 1e-13 * a.norm()
    assert abs(ad.gram_norm() - a.norm()) < 1e-13 * a.norm()
    near = tsa.TensorTrain([c.copy() for c in a.cores])
    near.cores[-1] = near.cores[-1] * (1 + 1e-11)
    assert abs(ad.error(near.to_device(), relative=True) - 1e-11) < 1e-13       # no sqrt(eps) floor
    assert rel((ad * -2.5).to_numpy(), -2.5 * a.to_numpy()) < 1e-15
    assert rel((ad - bd * 0.5).to_numpy(), a.to_numpy() - 0.5 * b.to_numpy()) < 1e-14
    assert abs(ad.error(bd) - a.error(b)) < 1e-12 * a.norm()


@pytest.mark.parametrize("method", ["exact", "pairwise"])
@pytest.mark.parametrize("use_pre", [0, 1])
def test_gmres_matches_reference_run(tsa, method, use_pre):
    """tt_sum_gmres with the deterministic roundings reproduces the reference's run (golden) and
    the oracle's: residual history, ranks, Hessenberg matrix and solution."""
    from oracle import tt_gmres_oracle as g
    from tt_sketch_amd.tt_gmres import tt_sum_gmres
    z, shape, maps, b, A, pre = _gmres_problem(tsa)
    x, hist = tt_sum_gmres(A, b, max_rank=6, precond=pre if use_pre else None, tolerance=1e-8, maxiter=8,
                           rounding_method=method, save_basis=True)
    key = f"{method}_{use_pre}"
    assert x.resident()
    assert np.allclose(hist["residual_norm"], z[key + "_residual_norm"], rtol=1e-6)
    assert np.array_equal(np.array(hist["rank"]), z[key + "_rank"])
    assert np.allclose(hist["H_matrix"], z[key + "_H"], rtol=1e-5, atol=1e-8)
    assert rel(x.to_numpy(), z[key + "_x"]) < 1e-7
    xo, ho = g.gmres(maps, [z[f"b_core{k}"] for k in range(len(shape))], 6,
                     precond=(z["precond"], 1) if use_pre else None, tolerance=1e-8, maxiter=8, method=method)
    assert rel(x.to_numpy(), orc.tt_to_numpy(xo)) < 1e-7
    for k in ("w_norm", "delta", "step_time", "step_time_with_res_norm", "final_round_time", "total_time"):
        assert k in hist


@pytest.mark.parametrize("method", ["sketch", "orth_sketch"])
def test_gmres_sketched_rounding(tsa, method):
    """Sketched rounding at a rank that represents every iterate exactly: the run does not depend on
    the random DRMs (no stream parity) and must reproduce the reference's -- including its
    stopping rule on the squared residual (tt_gmres.py:407-416) -- and approach the dense solution."""
    from tt_sketch_amd.tt_gmres import tt_sum_gmres
    z, shape, maps, b, A, pre = _gmres_problem(tsa)
    x, hist = tt_sum_gmres(A, b, max_rank=30, tolerance=1e-9, maxiter=25, rounding_method=method)
    assert np.allclose(hist["residual_norm"], z[method + "_full_residual_norm"], rtol=1e-4)
    assert np.array_equal(np.array(hist["rank"]), z[method + "_full_rank"])
    assert rel(x.to_numpy(), z[method + "_full_x"]) < 1e-7
    N = int(np.prod(shape))
    dense = sum(np.einsum("aibjckdl->abcdijkl", m.to_numpy()).reshape(N, N) for m in A.linear_maps)
    x_true = np.linalg.solve(dense.T, b.to_numpy().ravel()).reshape(shape)
    assert rel(x.to_numpy(), x_true) < 10 * hist["residual_norm"][-1]
    # truncating sketch: still converges, ranks capped
    x, hist = tt_sum_gmres(A, b, max_rank=10, tolerance=1e-6, maxiter=6, rounding_method=method)
    assert hist["residual_norm"][-1] < 2e-2 and max(x.rank) <= 10
    with pytest.raises(ValueError):
        tt_sum_gmres(A, tsa.TensorTrain.random((6, 5, 4, 4), 2, seed=0), max_rank=4)
    with pytest.raises(ValueError):
        tt_sum_gmres(A, b, max_rank=4, rounding_method="nope", maxiter=1)


@pytest.mark.parametrize("N,n,l,r", [(1000, 7, 10, 15), (5000, 3, 20, 30), (777, 1, 16, 16), (4096, 11, 1, 9),
                                     (300, 2, 33, 5), (64, 40, 4, 4)])
def test_sparse_psi_kernels_against_numpy(tsa, N, n, l, r):
    """ttsk_sparse_psi (sparse_sketch.py:8-36): the MFMA form (sorted, long slices, ranks <= 32), its
    single-slice form with per-wave partial blocks, and the scatter kernel (short slices / larger rank)
    against a NumPy scatter-add; including a missing left factor (first mode)."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(N + n)
    idx = rng.integers(0, n, N).astype(np.int64)
    val = rng.standard_normal(N)
    Lv, Rv = rng.standard_normal((N, l)), rng.standard_normal((N, r))
    perm = np.argsort(idx, kind="stable").astype(np.int64)
    d_idx, d_val, d_perm = DevArray.from_host(idx), DevArray.from_host(val), DevArray.from_host(perm)
    d_L, d_R = DevArray.from_host(Lv), DevArray.from_host(Rv)
    for use_L in (True, False):
        ll = l if use_L else 1
        want = np.zeros((ll, n, r))
        np.add.at(want, (slice(None), idx, slice(None)),
                  (val[:, None, None] * (Lv[:, :, None] if use_L else 1.0) * Rv[:, None, :]).transpose(1, 0, 2))
        out = DevArray.zeros((ll, n, r))
        nat.call("ttsk_sparse_psi", ctypes.c_void_p(d_val.ptr), ctypes.c_void_p(d_idx.ptr) if n > 1 else None,
                 ctypes.c_void_p(d_perm.ptr) if n > 1 else None, ctypes.c_size_t(N),
                 ctypes.c_void_p(d_L.ptr) if use_L else None, ll, ctypes.c_void_p(d_R.ptr), r, n,
                 ctypes.c_void_p(out.ptr), 0)
        assert rel(out.get(), want) < 1e-13


def test_c_abi_argument_errors(tsa):
    """Error convention of the boundary: bad arguments come back as TTSK_ERR_ARG ->
    ValueError with a message, never as a fault; the library stays usable afterwards."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    A = DevArray.from_host(np.eye(4))
    out = DevArray.empty((4, 4))
    S = DevArray.empty((4,))
    P = ctypes.c_void_p
    with pytest.raises(ValueError):          # m < n
        nat.call("ttsk_svd_small", P(A.ptr), 2, 4, P(out.ptr), P(S.ptr), P(out.ptr), 0)
    with pytest.raises(ValueError):          # NULL output
        nat.call("ttsk_svd_small", P(A.ptr), 4, 4, None, P(S.ptr), P(out.ptr), 0)
    with pytest.raises(ValueError):          # bad shape
        nat.call("ttsk_pinv_begin", P(A.ptr), 0, 4, -1.0, P(out.ptr), 0)
    with pytest.raises(ValueError):          # NULL input
        nat.call("ttsk_pinv", None, 4, 4, -1.0, P(out.ptr), None, 0)
    with pytest.raises(ValueError):          # thin QR needs m >= n
        nat.call("ttsk_qr_thin", P(A.ptr), 2, 4, 0)
    with pytest.raises(ValueError):
        nat.call("ttsk_triu", None, 4, 4, 0)
    with pytest.raises(ValueError):          # l + r beyond the staging buffer of the scatter kernel / NULL values
        nat.call("ttsk_sparse_psi", None, None, None, ctypes.c_size_t(4), None, 1, None, 1, 1, P(out.ptr), 0)
    with pytest.raises((ValueError, RuntimeError)):   # stream index out of range
        nat.call("ttsk_triu", P(A.ptr), 4, 4, 999)
    # still alive
    nat.call("ttsk_pinv", P(A.ptr), 4, 4, -1.0, P(out.ptr), None, 0)
    assert rel(out.get(), np.eye(4)) < 1e-14


def test_bench_collective_path_with_one_rank():
    """bench.py's multi-GPU step (local sum of the rank's sketches, all-reduce on the dedicated stream, stream
    waits around the reused buffers) rehearsed with a communicator of one rank; child process, time limit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TTSK_BENCH_FORCE_COMM="1")
    try:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu", "--steps", "6", "--warmup", "2",
                              "--batch", "4", "--no-extra"], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as exc:
        pytest.fail(f"bench.py with the collective path forced did not finish in 240 s; stderr {(exc.stderr or b'')[-2000:]!r}")
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["tts_per_step"] == 4


@pytest.mark.parametrize("case", [
    # (nb, n, K1, A, A2, J, right-chain strides?, T written?)
    (16, 200, 100, 100, 100, 100, True, False),      # C3 right chain: 6 tiles + 1 strip, 7 waves
    (16, 200, 100, 50, 50, 100, False, True),        # C3 left chain: 3 tiles + 1 strip (2 of 4 columns valid), T stored
    (3, 37, 100, 100, 100, 100, True, False),        # few slices, unequal ranges
    (2, 50, 97, 100, 100, 83, True, True),           # odd K1 and J: masked fragments, partial last wave
    (5, 64, 64, 52, 50, 33, False, True),            # ranks in one structure (3 tiles + strip), A != A2
    (1, 30, 100, 98, 100, 100, False, False),        # even / odd padding inside the strip
    (4, 40, 52, 36, 36, 40, False, True),            # 2 tiles + strip; 13 k-blocks: the run ends inside (ring padded to 15)
    (3, 33, 23, 16, 16, 20, True, False),            # one tile, K1 = 23: partial last k-block
    (2, 25, 60, 112, 112, 112, True, True),          # 7 full tiles, 7 waves (154 KB of LDS), runs of 5 k-blocks
    (2, 30, 101, 70, 70, 64, False, True),           # 26 k-blocks; 4 tiles + 2 strips (6 of 8 columns valid)
    (6, 30, 100, 90, 90, 100, True, False),          # remainder 10 > 8: a zero-padded sixth tile
    (2, 20, 64, 22, 22, 30, True, True),             # 1 tile + 2 strips
    (32, 24, 20, 50, 50, 20, False, True),           # C5-like: TT rank 20 (5 k-blocks), batch of 32
])
def test_fused_chain_step_against_einsum(tsa, case):
    """ttsk_chain_step (csrc/chain_fused.h): one step of TensorTrainDRM.sketch_tt
    (tensor_train_drm.py:81-87) with the intermediate on chip == the two einsums, and the optional T."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, sync
    nb, n, K1, A, A2, J, right, wt = case
    rng = np.random.default_rng(hash(case) % 2**32)
    W = [rng.standard_normal((K1, A)) for _ in range(nb)]
    E = rng.standard_normal((A, n, A2))
    if right:      # X[j][k][c]
        X = [rng.standard_normal((J, n, K1)) for _ in range(nb)]
        strides = (n * K1, K1, 1)
        want_T = [np.einsum("ca,jkc->akj", w, x) for w, x in zip(W, X)]
    else:          # X[c][k][j]
        X = [rng.standard_normal((K1, n, J)) for _ in range(nb)]
        strides = (1, J, n * J)
        want_T = [np.einsum("ca,ckj->akj", w, x) for w, x in zip(W, X)]
    want = [np.einsum("akj,akb->jb", t, E) for t in want_T]
    dW, dX = [DevArray.from_host(w) for w in W], [DevArray.from_host(x) for x in X]
    dE = DevArray.from_host(E)
    dO = [DevArray.zeros((J, A2)) for _ in range(nb)]
    dT = [DevArray.zeros((A, n, J)) for _ in range(nb)] if wt else None
    P = ctypes.c_void_p
    arr = lambda xs: (P * nb)(*[x.ptr for x in xs])
    nat.call("ttsk_chain_step", nb, n, K1, A, A2, J, arr(dW), A, arr(dX), strides[0], strides[1], strides[2],
             X[0].size, P(dE.ptr), arr(dT) if wt else None, arr(dO), 0)
    sync()
    for b in range(nb):
        assert rel(dO[b].get(), want[b]) < TOL, (b, rel(dO[b].get(), want[b]))
        if wt:
            assert rel(dT[b].get(), want_T[b]) < TOL, (b, rel(dT[b].get(), want_T[b]))


@pytest.mark.parametrize("shape", [(2000, 100, 50), (3000, 60, None), (4000, 290, 145), (2500, 148, 145), (1800, 200, None),
                                   (300, 256, 256), (700, 130, 129),
                                   (100, 290, 100), (100, 100, None), (130, 120, 110), (64, 64, None)])   # nearly square: one-workgroup Householder
def test_orth_step_one_call_and_ranks_beyond_128(tsa, shape):
    """ttsk_orth_step (sketch_dispatch.py:160-174 in one call, verdicts deferred) == scipy lstsq + qr incl. LAPACK's
    column signs; ranks 129..256 through the 2 x 2 block Cholesky (rank 145 / 290 of scripts/plot_timings.py)."""
    import ctypes
    import scipy.linalg
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    m, r2, l = shape
    rng = np.random.default_rng(m + r2)
    Psi = rng.standard_normal((m, r2))
    Om = None if l is None else rng.standard_normal((l, r2))
    k = r2 if l is None else l
    dP = DevArray.from_host(Psi)
    dO = None if Om is None else DevArray.from_host(Om)
    dQ = DevArray.empty((m, k))
    P = ctypes.c_void_p
    nat.call("ttsk_orth_step", P(dP.ptr), m, r2, None if dO is None else P(dO.ptr), k, P(dQ.ptr), 0)
    flag = ctypes.c_int(7)
    nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
    assert flag.value == 0
    M = Psi if Om is None else orc.right_mul_pinv(Psi, Om)
    want, _ = scipy.linalg.qr(M, mode="economic")
    got = dQ.get()
    assert np.linalg.norm(got.T @ got - np.eye(k)) < 1e-12
    assert rel(got, want) < 1e-10            # same Q as dgeqrf / dorgqr, signs included
    # the stand-alone entry points at the same sizes (pinv beyond 128 columns, QR beyond 128 columns)
    A = M.copy()
    dA = DevArray.from_host(A)
    nat.call("ttsk_qr_thin", P(dA.ptr), m, k, 0)
    assert rel(dA.get(), want) < 1e-10
    if Om is not None:
        from tt_sketch_amd.utils import right_mul_pinv
        assert rel(right_mul_pinv(Psi[:64], Om), orc.right_mul_pinv(Psi[:64], Om)) < 1e-10


def test_fast_solves_on_moderately_ill_conditioned_input(tsa):
    """What the first mode of a sketch with trimmed ranks looks like (Omega_0 = L_0^T R_0 with a SQUARE Gaussian L_0:
    kappa ~ 1e3..1e4; Psi_0 Omega_0^+ square with kappa ~ 1e5..1e6): the normal-equations pseudo-inverse with its
    Newton-Schulz step stays accurate and accepted up to kappa = 3e4, the square QR runs on the gate-free Householder
    kernel -- the deferred flag stays clear and the result equals lstsq + LAPACK QR."""
    import ctypes
    import scipy.linalg
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    from tt_sketch_amd.utils import right_mul_pinv
    rng = np.random.default_rng(44)
    P = ctypes.c_void_p
    l, r2, m = 100, 290, 100
    U, _ = np.linalg.qr(rng.standard_normal((l, l)))
    V, _ = np.linalg.qr(rng.standard_normal((r2, l)))
    for kappa in (1e2, 3e3, 2e4):
        Om = (U * np.logspace(0, -np.log10(kappa), l)) @ V.T
        Psi = rng.standard_normal((m, r2))
        want = orc.right_mul_pinv(Psi, Om)
        assert rel(right_mul_pinv(Psi, Om), want) < 1e-9, kappa        # blocking entry point: normal equations or Jacobi, its own gate
        dP, dO, dQ = DevArray.from_host(Psi), DevArray.from_host(Om), DevArray.empty((m, l))
        nat.call("ttsk_orth_step", P(dP.ptr), m, r2, P(dO.ptr), l, P(dQ.ptr), 0)
        flag = ctypes.c_int(7)
        nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
        assert flag.value == 0, kappa
        Q, _ = scipy.linalg.qr(want, mode="economic")
        q = dQ.get()
        assert np.linalg.norm(q.T @ q - np.eye(l)) < 1e-12
        assert rel(q, Q) < 1e-7 * max(1.0, kappa / 1e2), (kappa, rel(q, Q))          # Q of an ill-conditioned matrix moves with kappa eps
    # beyond the gate: flagged, and the blocking entry point falls back to the Jacobi SVD
    Om = (U * np.logspace(0, -7, l)) @ V.T
    dO = DevArray.from_host(Om)
    nat.call("ttsk_orth_step", P(dP.ptr), m, r2, P(dO.ptr), l, P(dQ.ptr), 0)
    nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
    assert flag.value == 1
    assert rel(right_mul_pinv(Psi, Om), orc.right_mul_pinv(Psi, Om)) < 1e-6


def test_orth_step_deferred_flag_on_rank_deficient_input(tsa):
    """A rank-deficient Omega (or unfolding) must set the deferred flag -- and only then; the flag clears on read."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(9)
    P = ctypes.c_void_p
    m, r2, l = 1500, 80, 40
    Psi = rng.standard_normal((m, r2))
    Om_bad = rng.standard_normal((l, 10)) @ rng.standard_normal((10, r2))
    dP, dO, dQ = DevArray.from_host(Psi), DevArray.from_host(Om_bad), DevArray.empty((m, l))
    flag = ctypes.c_int(0)
    nat.call("ttsk_orth_step", P(dP.ptr), m, r2, P(dO.ptr), l, P(dQ.ptr), 0)
    nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
    assert flag.value == 1
    nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
    assert flag.value == 0
    Psi_bad = rng.standard_normal((m, 7)) @ rng.standard_normal((7, r2))          # hmt: unfolding of rank 7
    dP = DevArray.from_host(Psi_bad)
    dQ = DevArray.empty((m, r2))
    nat.call("ttsk_orth_step", P(dP.ptr), m, r2, None, r2, P(dQ.ptr), 0)
    nat.call("ttsk_deferred_status", 0, ctypes.byref(flag))
    assert flag.value == 1


@pytest.mark.parametrize("case", [
    # (shape, nnz, l, r, rank_min?) -- small shapes: every factor is a table; large: sampled in the pass; mixed
    ((7, 6, 5, 8), 300, 3, 5, False),
    ((40, 30, 20, 25, 35), 20000, 10, 15, False),
    ((200, 150, 100, 120, 300), 60000, 10, 15, False),
    ((9, 11), 60, 4, 16, False),
    ((300, 7, 250), 5000, 16, 13, True),
    ((5, 4, 3, 6, 2, 7), 4000, 2, 3, False),
])
def test_sparse_fused_passes_vs_oracle_and_generator_path(tsa, case, monkeypatch):
    """csrc/sparse_fused.hip (one pass per mode, DRM rows sampled / gathered where consumed, no atomics) == the oracle's
    restatement of sparse_gaussian_drm.py:29-44 + sparse_sketch.py:8-69, == the generator path of round 2, and
    bit-identical from run to run."""
    shape, nnz, l, r, sliced = case
    d = len(shape)
    rng = np.random.default_rng(nnz)
    idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
    val = rng.standard_normal(nnz)
    lo_l, lo_r = ((1,) * (d - 1), (2,) * (d - 1)) if sliced else ((0,) * (d - 1), (0,) * (d - 1))
    hi_l, hi_r = tuple(a + l for a in lo_l), tuple(a + r for a in lo_r)
    kw = lambda lo, hi: dict(rank_min=lo, rank_max=hi, true_rank=hi)
    mk = lambda: (tsa.SparseGaussianDRM(hi_l, shape, False, seed=3, **kw(lo_l, hi_l)),
                  tsa.SparseGaussianDRM(hi_r, shape, True, seed=4, **kw(lo_r, hi_r)))
    T = tsa.SparseTensor(shape, idx, val)
    ld, rd = mk()
    from tt_sketch_amd import sparse_fused
    assert sparse_fused.try_sparse_gauss_sketch(T, ld, rd, tsa.SketchMethod.streaming) is not None
    sk = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    oP, oO = orc.general_sketch("sparse", (shape, idx, val), orc.HashGaussDrm(3, shape, False, lo_l, hi_l),
                                orc.HashGaussDrm(4, shape, True, lo_r, hi_r), "streaming")
    got = sk.Psi_cores + sk.Omega_mats
    for a, b in zip(got, oP + oO):
        assert a.shape == b.shape and rel(a, b) < 1e-11, (a.shape, rel(a, b))
    again = tsa.general_sketch(tsa.SparseTensor(shape, idx, val), *mk(), tsa.SketchMethod.streaming)
    for a, b in zip(got, again.Psi_cores + again.Omega_mats):
        assert np.array_equal(a, b)                    # fixed summation order: bit-reproducible
    monkeypatch.setenv("TTSK_SPARSE_FUSED", "0")
    old = tsa.general_sketch(tsa.SparseTensor(shape, idx, val), *mk(), tsa.SketchMethod.streaming)
    for a, b in zip(got, old.Psi_cores + old.Omega_mats):
        assert rel(a, b) < 1e-11


@pytest.mark.parametrize("case", [
    # (shape, nnz, left (kind, true rank, lo, hi, non-zeros per row), right (...)); "g" = SparseGaussianDRM, "s" = SparseSignDRM
    ((40, 30, 20, 25, 35), 20000, ("g", 24, 0, 24, None), ("g", 24, 0, 24, None)),        # two 16-column tiles per factor
    ((200, 150, 100, 120, 300), 60000, ("g", 17, 0, 17, None), ("g", 32, 0, 32, None)),   # just outside the old cover; sampled in the pass
    ((200, 150, 100, 120, 300), 60000, ("s", 24, 0, 24, None), ("s", 24, 0, 24, 5)),      # sign rows made in the pass (deep modes) and tables
    ((7, 6, 5, 8), 300, ("s", 10, 0, 10, 3), ("s", 15, 0, 15, None)),                     # every sign factor a table
    ((300, 7, 250, 9), 9000, ("s", 16, 3, 11, 9), ("g", 13, 2, 15, None)),                # a slice of a sign row; mixed pair
    ((300, 7, 250, 9), 9000, ("g", 20, 1, 21, None), ("s", 32, 5, 30, 32)),               # mixed pair, wide
    ((9, 11), 60, ("s", 4, 0, 4, 1), ("s", 32, 0, 32, 7)),
    ((200, 150, 100, 120, 300), 40000, ("g", 10, 0, 10, None), ("g", 24, 0, 24, None)),   # one side inside 16 columns: tiles beyond it are skipped
    ((40, 30, 20, 25, 35), 9000, ("g", 28, 0, 28, None), ("s", 12, 0, 12, 4)),
])
def test_sparse_fused_wide_and_sign_factors_vs_oracle(tsa, case, monkeypatch):
    """the one-pass-per-mode sparse sketch beyond 16 columns per factor (2 x 2 matrix tiles) and with
    SparseSignDRM factors (sparse_sign_drm.py:34-51: gathered from a per-prefix table or made in the pass), in every pairing
    with SparseGaussianDRM -- against the oracle, the generator path, and itself (bit-reproducible)."""
    shape, nnz, lspec, rspec = case
    d = len(shape)
    rng = np.random.default_rng(nnz + 1)
    idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
    val = rng.standard_normal(nnz)

    def mk(spec, transpose, seed):
        kind, tr, lo, hi, nz = spec
        t = lambda v: (v,) * (d - 1)
        if kind == "g":
            return (tsa.SparseGaussianDRM(t(hi), shape, transpose, seed=seed, rank_min=t(lo), rank_max=t(hi), true_rank=t(hi)),
                    orc.HashGaussDrm(seed, shape, transpose, t(lo), t(hi)))
        return (tsa.SparseSignDRM(t(tr), shape, transpose, seed=seed, rank_min=t(lo), rank_max=t(hi), true_rank=t(tr),
                                  num_non_zero_per_row=None if nz is None else t(nz)),
                orc.HashSignDrm(seed, shape, transpose, t(tr), t(lo), t(hi), None if nz is None else t(nz)))

    (ld, old), (rd, ord_) = mk(lspec, False, 3), mk(rspec, True, 4)
    T = tsa.SparseTensor(shape, idx, val)
    from tt_sketch_amd import sparse_fused
    assert sparse_fused.try_sparse_gauss_sketch(T, ld, rd, tsa.SketchMethod.streaming) is not None
    sk = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    oP, oO = orc.general_sketch("sparse", (shape, idx, val), old, ord_, "streaming")
    got = sk.Psi_cores + sk.Omega_mats
    for a, b in zip(got, oP + oO):
        assert a.shape == b.shape and rel(a, b) < 1e-11, (a.shape, rel(a, b))
    again = tsa.general_sketch(tsa.SparseTensor(shape, idx, val), mk(lspec, False, 3)[0], mk(rspec, True, 4)[0],
                               tsa.SketchMethod.streaming)
    for a, b in zip(got, again.Psi_cores + again.Omega_mats):
        assert np.array_equal(a, b)
    monkeypatch.setenv("TTSK_SPARSE_FUSED", "0")
    gen = tsa.general_sketch(tsa.SparseTensor(shape, idx, val), mk(lspec, False, 3)[0], mk(rspec, True, 4)[0],
                             tsa.SketchMethod.streaming)
    for a, b in zip(got, gen.Psi_cores + gen.Omega_mats):
        assert rel(a, b) < 1e-11


def test_sparse_fused_samples_are_bit_identical_to_the_sampler(tsa):
    """One nonzero per first-mode slice with entry 1: Psi_0[0, j, :] IS the right DRM row of that nonzero -- the
    samples made inside the pass (table and in-pass kinds) against ttsk_inds_to_normal, bit for bit."""
    from tt_sketch_amd.drm.fast_lazy_gaussian import inds_to_normal
    for shape in [(50, 4, 3), (64, 300, 200, 100)]:
        d, n0 = len(shape), shape[0]
        rng = np.random.default_rng(n0)
        idx = np.stack([np.arange(n0)] + [rng.integers(0, n, n0) for n in shape[1:]]).astype(np.int64)
        T = tsa.SparseTensor(shape, idx, np.ones(n0))
        ld = tsa.SparseGaussianDRM(5, shape, False, seed=11)
        rd = tsa.SparseGaussianDRM(12, shape, True, seed=12)
        sk = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
        # R_0: the right DRM's matrix with d - 1 index rows of the transposed tensor, seed (d - 2 + seed) mod 2^63
        want = inds_to_normal(idx[::-1][:d - 1], shape[::-1][:d - 1], 0, 12, (d - 2 + rd.seed) % 2**63)
        assert np.array_equal(sk.Psi_cores[0][0], want)


def test_stream_sketch_batch_public_entry_point(tsa):
    """stream_sketch_batch == [stream_sketch(t, ...) with the same DRMs]: TTs of one signature through the batched
    pass (40 tensors: more than one slice of 32), a mixed list through the per-tensor fallback; argument policy of
    stream_sketch (rank direction error)."""
    rng = np.random.default_rng(77)
    shape = (9, 10, 11, 8)
    tts = [tsa.TensorTrain(orc.random_tt(shape, 5, rng)) for _ in range(40)]
    sks, ld, rd = tsa.stream_sketch_batch(tts, 4, 7, seed=5, return_drm=True)
    assert len(sks) == 40 and type(ld).__name__ == "TensorTrainDRM"
    for k in (0, 1, 31, 32, 39):
        one = tsa.stream_sketch(tts[k], ld.rank, tuple(rd.rank[::-1]), left_drm=ld, right_drm=rd)
        for a, b in zip(sks[k].Psi_cores + sks[k].Omega_mats, one.Psi_cores + one.Omega_mats):
            assert a.shape == b.shape and rel(a, b) < TOL
        assert sks[k].to_tt().error(one.to_tt(), relative=True) < 1e-9
    mixed = [tts[0], tsa.DenseTensor(tts[1].to_numpy())]
    ms = tsa.stream_sketch_batch(mixed, ld.rank, tuple(rd.rank[::-1]), left_drm=ld, right_drm=rd)    # explicit DRMs: ranks as tuples (sketch.py:119)
    for a, b in zip(ms[0].Psi_cores, sks[0].Psi_cores):
        assert rel(a, b) < 1e-11
    # (the dense path pairs the right DRM's modes differently -- dense_sketch.py, SURVEY A14 -- so it is compared with itself)
    alone = tsa.stream_sketch(mixed[1], ld.rank, tuple(rd.rank[::-1]), left_drm=ld, right_drm=rd)
    for a, b in zip(ms[1].Psi_cores + ms[1].Omega_mats, alone.Psi_cores + alone.Omega_mats):
        assert rel(a, b) < 1e-13
    with pytest.raises(ValueError):
        tsa.stream_sketch_batch(tts[:2], 4, 4)
    assert tsa.stream_sketch_batch([], 4, 7) == []


def _chain_step_case(entry, case, seed_salt=0):
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, sync
    nb, n, K1, A, A2, J, right, wt = case
    rng = np.random.default_rng((hash(case) + seed_salt) % 2**32)
    W = [rng.standard_normal((K1, A)) for _ in range(nb)]
    E = rng.standard_normal((A, n, A2))
    if right:      # X[j][k][c]
        X = [rng.standard_normal((J, n, K1)) for _ in range(nb)]
        strides = (n * K1, K1, 1)
        want_T = [np.einsum("ca,jkc->akj", w, x) for w, x in zip(W, X)]
    else:          # X[c][k][j]
        X = [rng.standard_normal((K1, n, J)) for _ in range(nb)]
        strides = (1, J, n * J)
        want_T = [np.einsum("ca,ckj->akj", w, x) for w, x in zip(W, X)]
    want = [np.einsum("akj,akb->jb", t, E) for t in want_T]
    dW, dX = [DevArray.from_host(w) for w in W], [DevArray.from_host(x) for x in X]
    dE = DevArray.from_host(E)
    dO = [DevArray.zeros((J, A2)) for _ in range(nb)]
    dT = [DevArray.zeros((A, n, J)) for _ in range(nb)] if wt else None
    P = ctypes.c_void_p
    arr = lambda xs: (P * nb)(*[x.ptr for x in xs])
    nat.call(entry, nb, n, K1, A, A2, J, arr(dW), A, arr(dX), strides[0], strides[1], strides[2],
             X[0].size, P(dE.ptr), arr(dT) if wt else None, arr(dO), 0)
    sync()
    for b in range(nb):
        assert rel(dO[b].get(), want[b]) < TOL, (b, rel(dO[b].get(), want[b]))
        if wt:
            assert rel(dT[b].get(), want_T[b]) < TOL, (b, rel(dT[b].get(), want_T[b]))


@pytest.mark.parametrize("case", [
    # (nb, n, K1, A, A2, J, right-chain strides?, T written?) -- the structures ttsk_chain_step does not cover
    (1, 100, 150, 110, 110, 150, True, False),       # ref150 right chain l=55 x2: K1 > 128, 10 row tiles (3+3+2+2), 2 chunks of 56
    (1, 100, 150, 55, 55, 150, False, True),         # ref150 left chain: odd ranks (8-byte-aligned E rows), T stored, one chunk
    (2, 100, 100, 110, 110, 150, True, False),       # first interior step: TT ranks 100 -> 150
    (1, 100, 150, 58, 58, 100, True, False),         # "+3" right chain, last interior step (J = 100)
    (4, 40, 150, 145, 108, 150, False, True),        # rank changes structure between the modes; 4 chunks; T stored over chunks
    (3, 33, 130, 25, 25, 130, False, True),          # small odd rank: strips only in the output? (1 tile + 9 -> 2 tiles)
    (2, 27, 150, 5, 5, 150, True, False),            # rank 5: output of two strips, chunk of 16 with 5 columns
    (2, 20, 64, 8, 8, 64, False, True),              # rank 8: two strips
    (2, 31, 100, 95, 95, 100, True, True),           # odd A2 with the last slice in range: the patched unit
    (1, 16, 160, 64, 64, 176, True, False),          # 11 row tiles (3+3+3+2), 64-column structure falls back to 48 + 16
    (2, 50, 100, 160, 160, 100, True, False),        # 10 output tiles, one tile per wave
    (3, 30, 120, 100, 160, 112, False, True),        # A != A2, 7 waves, chunks of 56 / 48
    (8, 24, 20, 100, 100, 20, True, False),          # C5 right chain: TT rank 20 against DRM rank 100
    (8, 24, 20, 50, 50, 20, False, True),            # C5 left chain
    (2, 9, 37, 33, 47, 29, False, True),             # everything odd
    (1, 3, 4, 4, 6, 5, True, True),                  # tiny
])
def test_wide_chain_step_against_einsum(tsa, case):
    """ttsk_chain_step_wide (csrc/chain_wide.h): the fused step with the DRM rank cut into chunks over workgroups
    and up to two row tiles per wave == the two einsums of tensor_train_drm.py:81-87, and the optional T."""
    _chain_step_case("ttsk_chain_step_wide", case)


def _chain_sum_case(case, seed_salt=0):
    """ttsk_chain_step_sum against the two einsums of tensor_train_drm.py:81-87, T in either layout"""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, sync
    nb, n, K1, A, A2, J, right, wt = case          # wt: 0 = no T, 1 = interleaved over the terms, 2 = per term
    rng = np.random.default_rng((hash(case) + seed_salt) % 2**32)
    W = [rng.standard_normal((K1, A)) for _ in range(nb)]
    E = rng.standard_normal((A, n, A2))
    if right:      # X[j][k][c]
        X = [rng.standard_normal((J, n, K1)) for _ in range(nb)]
        strides = (n * K1, K1, 1)
        want_T = [np.einsum("ca,jkc->akj", w, x) for w, x in zip(W, X)]
    else:          # X[c][k][j]
        X = [rng.standard_normal((K1, n, J)) for _ in range(nb)]
        strides = (1, J, n * J)
        want_T = [np.einsum("ca,ckj->akj", w, x) for w, x in zip(W, X)]
    want = [np.einsum("akj,akb->jb", t, E) for t in want_T]
    dW, dX = [DevArray.from_host(w) for w in W], [DevArray.from_host(x) for x in X]
    dE = DevArray.from_host(E)
    dO = [DevArray.zeros((J, A2)) for _ in range(nb)]
    P = ctypes.c_void_p
    arr = lambda xs: (P * nb)(*[x.ptr for x in xs])
    dT, t_b, t_ld = None, 0, 0
    if wt == 1:
        dT, t_b, t_ld = DevArray.zeros((A, n, nb, J)), J, nb * J
    elif wt == 2:
        dT, t_b, t_ld = DevArray.zeros((nb, A, n, J)), A * n * J, J
    nat.call("ttsk_chain_step_sum", nb, n, K1, A, A2, J, arr(dW), A, arr(dX), strides[0], strides[1], strides[2],
             X[0].size, P(dE.ptr), None if dT is None else P(dT.ptr), t_b, t_ld, 0 if dT is None else dT.size, arr(dO), 0)
    sync()
    for b in range(nb):
        assert rel(dO[b].get(), want[b]) < TOL, (b, rel(dO[b].get(), want[b]))
        if wt:
            got = dT.get()[:, :, b, :] if wt == 1 else dT.get()[b]
            assert rel(got, want_T[b]) < TOL, (b, rel(got, want_T[b]))


@pytest.mark.parametrize("case", [
    # (nb, n, K1, A, A2, J, right-chain strides?, T: 0 none / 1 interleaved over the terms / 2 per term)
    (32, 128, 20, 100, 100, 20, True, 0),            # C5 right chain: 8 groups of 4 terms x 32 slice ranges
    (32, 128, 20, 50, 50, 20, False, 1),             # C5 left chain, T in the layout the Psi of a sum reads
    (8, 24, 20, 100, 100, 20, True, 0),
    (8, 24, 20, 50, 50, 20, False, 2),               # T per term (a batch of separate sketches)
    (7, 19, 20, 100, 100, 20, True, 0),              # terms do not fill the last group
    (5, 13, 17, 100, 100, 19, True, 1),              # J, K1 not multiples of 4: zero-padded strips
    (6, 10, 20, 64, 48, 20, False, 1),               # no strip column
    (9, 11, 12, 30, 22, 9, False, 2),                # small ranks: more terms per workgroup
    (4, 7, 8, 16, 8, 4, True, 0),                    # output of two strips only
    (12, 40, 20, 112, 112, 20, True, 0),             # large ranks: two terms per workgroup
    (16, 33, 20, 100, 56, 16, True, 1),              # A != A2, two strips behind three tiles
    (4, 1, 20, 100, 100, 20, True, 0),               # one slice
])
def test_chain_step_sum_against_einsum(tsa, case):
    """ttsk_chain_step_sum (csrc/chain_sum.h): rows of several low-rank terms stacked into 16-row tiles that straddle term
    boundaries in the second product, the first product per term as 4-row strips == the two einsums of
    tensor_train_drm.py:81-87 per term, and the optional T in both layouts."""
    _chain_sum_case(case)


def test_wide_and_first_fused_kernel_on_shared_shapes(tsa):
    """Where both kernels apply they sum in different orders (chunks); results agree to rounding."""
    for case in [(4, 50, 100, 100, 100, 100, True, False), (2, 40, 64, 52, 50, 33, False, True)]:
        _chain_step_case("ttsk_chain_step", case, 1)
        _chain_step_case("ttsk_chain_step_wide", case, 1)


def test_blocked_sketch_and_rank_increase_match_reference_runs(tsa):
    """(f)3 on the HIP path against runs of the reference itself (tests/golden/blocked_cases.npz, generated by
    tests/golden/make_golden_blocked.py): `blocked_stream_sketch` (sketch.py:493-525) over DRM rank slices and
    `SketchedTensorTrain.increase_rank` (:303-353), hash DRMs rebuilt from the recorded seeds."""
    z = np.load(os.path.join(GOLDEN, "blocked_cases.npz"))
    meta = json.loads(str(z["meta"]))

    def lists(prefix):
        P = [z[f"{prefix}/Psi/{i}"] for i in range(sum(1 for k in z.files if k.startswith(f"{prefix}/Psi/")))]
        O = [z[f"{prefix}/Omega/{i}"] for i in range(sum(1 for k in z.files if k.startswith(f"{prefix}/Omega/")))]
        return P + O
    tol = 1e-11                      # hash Gaussians agree to ULP_BAR ulp, the sums to rounding
    for name, m in meta.items():
        shape = tuple(m["shape"])
        X = tsa.SparseTensor(shape, z[f"{name}/indices"], z[f"{name}/entries"])
        if m["kind"] == "blocked":
            cls = getattr(tsa, m["drm"])
            left = cls(tuple(m["left_rank"]), shape, False, seed=m["left_seed"])
            right = cls(tuple(m["right_rank"]), shape, True, seed=m["right_seed"])
            blk = tsa.blocked_stream_sketch(X, left, right, [tuple(s) for s in m["left_slices"]],
                                            [tuple(s) for s in m["right_slices"]])
            for tag in ("blocked", "whole"):
                for a, b in zip(blk.Psi_cores + blk.Omega_mats, lists(f"{name}/{tag}")):
                    assert a.shape == b.shape and rel(a, b) < tol, (name, tag, a.shape, rel(a, b))
        else:
            l0, r0 = tuple(m["left_rank"]), tuple(m["right_rank"])
            left = tsa.SparseGaussianDRM(l0, shape, False, seed=m["left_seed"])
            right = tsa.SparseGaussianDRM(r0, shape, True, seed=m["right_seed"])
            stt = tsa.stream_sketch(X, l0, r0, left_drm=left, right_drm=right)
            for a, b in zip(stt.Psi_cores + stt.Omega_mats, lists(f"{name}/before")):
                assert a.shape == b.shape and rel(a, b) < tol, (name, "before", rel(a, b))
            stt2 = stt.increase_rank(X, tuple(m["new_left_rank"]), tuple(m["new_right_rank"]))
            assert stt2.left_drm.seed == m["new_left_seed"] and stt2.right_drm.seed == m["new_right_seed"]
            for tag in ("after", "direct"):
                for a, b in zip(stt2.Psi_cores + stt2.Omega_mats, lists(f"{name}/{tag}")):
                    assert a.shape == b.shape and rel(a, b) < tol, (name, tag, a.shape, rel(a, b))


def test_tt_svd_on_device_matches_reference_runs(tsa):
    """(f)4: tt_svd (reference tt_svd.py:10-49) as a device sweep -- thin QR of the transposed unfolding + Jacobi
    SVD of the square factor -- against runs of the reference (tests/golden/tt_svd_cases.npz): identical TT
    ranks, equal as a tensor at 1e-10, left-orthogonal cores; plus the oracle on a resident low-rank input."""
    z = np.load(os.path.join(GOLDEN, "tt_svd_cases.npz"))
    meta = json.loads(str(z["meta"]))
    for name, m in meta.items():
        X = z[f"{name}/X"]
        ref = [z[f"{name}/core/{i}"] for i in range(len(m["shape"]))]
        tt = tsa.tt_svd(tsa.DenseTensor(X), m["rank"])
        assert list(tt.rank) == m["tt_rank"], (name, tt.rank)
        cores = [np.asarray(c) for c in tt.cores]
        assert [c.shape for c in cores] == [c.shape for c in ref]
        assert rel(orc.tt_to_numpy(cores), orc.tt_to_numpy(ref)) < 1e-10, (name, rel(orc.tt_to_numpy(cores), orc.tt_to_numpy(ref)))
        for c in cores[:-1]:
            # U factors: orthonormal columns -- for the directions that carry the tensor.  Beyond the numerical
            # rank of an unfolding (lowrank5: rank 3 under a cap of 7) LAPACK completes U arbitrarily; the device
            # sweep leaves those columns zero (tt_svd.py::_inv_singular), the tensor is the same.
            Q = c.reshape(-1, c.shape[2])
            G = Q.T @ Q
            live = np.abs(np.diag(G) - 1) < 1e-8
            assert live.sum() >= min(3, Q.shape[1]) and (name == "lowrank5" or live.all()), (name, np.diag(G))
            assert np.max(np.abs(G[np.ix_(live, live)] - np.eye(int(live.sum())))) < 1e-10, name
    # other tensor types go through dense(); a TT input of rank 3 is recovered exactly with a generous cap
    rng = np.random.default_rng(8)
    cores = orc.random_tt((6, 5, 7, 4), 3, rng)
    tt = tsa.tt_svd(tsa.TensorTrain(cores), rank=5)
    assert list(tt.rank) == [5, 5, 4] and tt.error(tsa.TensorTrain(cores), relative=True) < 1e-10
    want = orc.tt_svd(orc.tt_to_numpy(cores), 5)
    assert rel(tt.to_numpy(), orc.tt_to_numpy(want)) < 1e-10


def test_jacobi_svd_over_the_whole_chip(tsa):
    """ttsk_svd_small beyond one workgroup (n > 1024: svd_grid.hip, a grid barrier per Jacobi round) against
    LAPACK: singular values at 1e-12 of the largest, A = US Vt, orthogonal factors; full rank and rank 40."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray
    rng = np.random.default_rng(31)
    P = ctypes.c_void_p
    for m, n, rank in ((1300, 1100, None), (1100, 1100, 40), (2300, 1030, None)):
        A = rng.standard_normal((m, n))
        if rank:
            A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))
        dA = DevArray.from_host(A)
        US, S, Vt = DevArray.empty((m, n)), DevArray.empty((n,)), DevArray.empty((n, n))
        nat.call("ttsk_svd_small", P(dA.ptr), m, n, P(US.ptr), P(S.ptr), P(Vt.ptr), 0)
        us, s, vt = US.get(), S.get(), Vt.get()
        want = np.linalg.svd(A, compute_uv=False)
        assert np.all(np.diff(s) <= 0)
        assert np.max(np.abs(s - want)) < 1e-12 * want[0], (m, n, np.max(np.abs(s - want)) / want[0])
        assert rel(us @ vt, A) < 1e-12, (m, n, rel(us @ vt, A))
        assert np.max(np.abs(vt @ vt.T - np.eye(n))) < 1e-12
        k = rank or n
        U = us[:, :k] / s[:k]
        assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-10, (m, n)


def test_whole_chip_jacobi_with_another_stream_busy(tsa):
    """VERDICT r2 hygiene / ADVICE: the grid-barrier kernel (cooperative launch) while other library streams hold the
    chip -- a queue of long products on streams 3 and 5 issued right before; the SVD must come out right (it waits
    for what is queued, then owns the GPU) and the products too."""
    import ctypes
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, contract, sync
    rng = np.random.default_rng(33)
    P = ctypes.c_void_p
    m, n = 1200, 1040
    A = rng.standard_normal((m, n))
    dA = DevArray.from_host(A)
    US, S, Vt = DevArray.empty((m, n)), DevArray.empty((n,)), DevArray.empty((n, n))
    X, Y = DevArray.from_host(rng.standard_normal((3000, 2000))), DevArray.from_host(rng.standard_normal((2000, 3000)))
    sync()
    busy = [contract("ij,jk->ik", X, Y, stream=st) for st in (3, 5, 3, 5, 3, 5)]
    nat.call("ttsk_svd_small", P(dA.ptr), m, n, P(US.ptr), P(S.ptr), P(Vt.ptr), 0)
    more = [contract("ij,jk->ik", X, Y, stream=3)]
    sync()
    want = np.linalg.svd(A, compute_uv=False)
    assert np.max(np.abs(S.get() - want)) < 1e-12 * want[0]
    assert rel(US.get() @ Vt.get(), A) < 1e-12
    ref = X.get() @ Y.get()
    for b in busy + more:
        assert rel(b.get(), ref) < 1e-12


def test_tt_svd_with_an_unfolding_beyond_one_workgroup(tsa):
    """tt_svd where r n = 36 x 36 = 1296 rows meet 1920 columns: QR of the transpose, then the 1296 x 1296
    factor on the whole-chip Jacobi kernel.  Against the oracle's restatement of reference tt_svd.py:10-49."""
    rng = np.random.default_rng(32)
    shape = (36, 36, 48, 40)
    X = orc.tt_to_numpy(orc.random_tt(shape, 9, rng)) + 1e-3 * rng.standard_normal(shape)
    tt = tsa.tt_svd(tsa.DenseTensor(X), rank=(36, 40, 30))
    want = orc.tt_svd(X, (36, 40, 30))
    assert list(tt.rank) == [c.shape[2] for c in want[:-1]] == [36, 40, 30]
    assert rel(tt.to_numpy(), orc.tt_to_numpy(want)) < 1e-10
    for c in tt.cores[:-1]:
        Q = np.asarray(c).reshape(-1, c.shape[2])
        assert np.max(np.abs(Q.T @ Q - np.eye(Q.shape[1]))) < 1e-10
