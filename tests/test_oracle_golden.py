"""Pin the CPU oracle (oracle/) against the reference's own outputs.

The fixtures were produced by running the reference (tests/golden/make_golden.py);
this file runs on CPU only and is what makes later HIP-vs-oracle parity claims
meaningful.  Counterpart of the reference's tests/test_fast_lazy_gaussian.py and
tests/test_sketching_matrix.py, but with golden vectors, which the reference lacks.
"""
import json
import os

import numpy as np
import pytest

from oracle import ttsk_oracle as orc
from tests.golden_io import GOLDEN, Cases, rel

CASES = Cases()
TOL = 1e-12  # ||delta||_F / ||ref||_F for contraction outputs


# ----------------------------------------------------------------- sampler
def test_hash_kat():
    # SURVEY.md 8c known answers
    got = orc.hash_u64(np.array([0, 1, 2, 179], dtype=np.uint64))
    want = np.array([0x6DF1829614FFC95F, 0x926F33A0A71291FE,
                     0x47AA1C14B4F9AEC6, 0x23978B9BD59BD32F], dtype=np.uint64)
    assert np.array_equal(got, want)


def test_normal_kat():
    idx = np.array([[0, 3, 1], [0, 4, 2]])
    got = orc.inds_to_normal(idx, (4, 5), 0, 3, 7)
    want = np.array([[-1.863503535488479, 0.01769032355989694, 0.6249262953893071],
                     [1.219759099294262, 0.11242813561704242, 0.38161408642785044],
                     [0.1844089289453476, 0.5386399428076478, 1.5529022436027455]])
    assert np.array_equal(got, want)
    assert np.array_equal(orc.inds_to_normal(idx, (4, 5), 1, 3, 7), want[:, 1:])
    sign = orc.inds_to_sparse_sign(idx, (4, 5), 6, 0, 6, 2, 7)
    assert np.array_equal(sign, np.array([[1, 0, 0, -1, 0, 0], [0, 0, 0, 1, 0, 1],
                                          [0, 0, 0, 1, 1, 0]], dtype=np.int16))


def test_sampler_golden():
    z = np.load(os.path.join(GOLDEN, "hash_sampler.npz"))
    assert np.array_equal(orc.hash_u64(z["hash_in"]), z["hash_out"])
    meta = json.loads(str(z["meta"]))
    for ci, c in enumerate(meta):
        idx = z[f"s{ci}_idx"]
        if idx.shape[1] > 0:
            rd = orc.inds_to_rand_double(idx, c["shape"], c["rank_min"], c["rank_max"], c["seed"])
            assert np.array_equal(rd.view(np.uint64), z[f"s{ci}_rand_double"].view(np.uint64))
        nm = orc.inds_to_normal(idx, c["shape"], c["rank_min"], c["rank_max"], c["seed"])
        assert nm.shape == z[f"s{ci}_normal"].shape
        assert np.array_equal(nm, z[f"s{ci}_normal"]), f"case {ci}"
        for nnz in (1, 2, c["true_rank"]):
            sg = orc.inds_to_sparse_sign(idx, c["shape"], c["true_rank"], c["rank_min"],
                                         c["rank_max"], nnz, c["seed"])
            assert np.array_equal(sg, z[f"s{ci}_sign_nnz{nnz}"]), f"case {ci} nnz {nnz}"


def test_ndtri_golden():
    z = np.load(os.path.join(GOLDEN, "hash_sampler.npz"))
    got = orc.ndtri(z["ndtri_x"])
    want = z["ndtri_y"]
    assert np.array_equal(got, want), np.max(np.abs(got - want)[np.isfinite(want)])


# ----------------------------------------------------------------- sketches
@pytest.mark.parametrize("name", CASES.names())
def test_contractions_golden(name):
    kind, data = CASES.tensor(name)
    for side in ("left", "right"):
        got = orc.drm_contractions(kind, data, CASES.drm(name, side))
        if kind == "sum":
            for s in range(len(data)):
                want = CASES.out(name, f"{side}_contractions_s{s}")
                for g, w in zip(got, want):
                    assert rel(g[s], w) < TOL
        else:
            want = CASES.out(name, f"{side}_contractions")
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert rel(g, w) < TOL


@pytest.mark.parametrize("name", CASES.names())
def test_sketch_golden(name):
    kind, data = CASES.tensor(name)
    m = CASES.meta[name]
    for method in m["methods"]:
        left = None if method == "hmt" else CASES.drm(name, "left")
        Psis, Omegas = orc.general_sketch(kind, data, left, CASES.drm(name, "right"), method)
        wantP = CASES.out(name, f"{method}/Psi")
        wantO = CASES.out(name, f"{method}/Omega")
        assert len(Psis) == len(wantP) and len(Omegas) == len(wantO)
        for g, w in zip(Omegas, wantO):
            assert rel(g, w) < TOL
        if method == "streaming":
            for g, w in zip(Psis, wantP):
                assert rel(g, w) < TOL
            if not m.get("sliced"):
                for direction in ("left", "right"):
                    C = orc.assemble(Psis, Omegas, direction)
                    wantC = CASES.out(name, f"{method}/C_{direction}")
                    for g, w in zip(C, wantC):
                        assert rel(g, w) < 1e-9
        else:
            # same LAPACK calls -> same gauge; compare directly, loosely
            for g, w in zip(Psis, wantP):
                assert rel(g, w) < 1e-8


# ------------------------------------------------------------------ TT-GMRES
def _gmres_case():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmres_case.npz"))
    d = len(z["shape"])
    maps = [[z[f"map{m}_core{k}"] for k in range(d)] for m in range(3)]
    b = [z[f"b_core{k}"] for k in range(d)]
    return z, maps, b


def test_gmres_oracle_blocks_match_reference():
    from oracle import tt_gmres_oracle as g
    z, maps, b = _gmres_case()
    assert rel(orc.tt_to_numpy(g.mpo_apply(maps[2], b)), z["mpo_apply"]) < 1e-13
    assert rel(orc.tt_to_numpy(g.precond_apply(z["precond"], b, 1, True)), z["precond_backward"]) < 1e-13
    assert rel(orc.tt_to_numpy(g.precond_apply(z["precond"], b, 1, False)), z["precond_forward"]) < 1e-13


@pytest.mark.parametrize("method", ["exact", "pairwise"])
@pytest.mark.parametrize("use_pre", [0, 1])
def test_gmres_oracle_matches_reference_run(method, use_pre):
    from oracle import tt_gmres_oracle as g
    z, maps, b = _gmres_case()
    x, hist = g.gmres(maps, b, 6, precond=(z["precond"], 1) if use_pre else None, tolerance=1e-8,
                      maxiter=8, method=method)
    key = f"{method}_{use_pre}"
    assert np.allclose(hist["residual_norm"], z[key + "_residual_norm"], rtol=1e-7)
    assert np.array_equal(np.array(hist["rank"]), z[key + "_rank"])
    assert np.allclose(hist["H_matrix"], z[key + "_H"], rtol=1e-6, atol=1e-9)
    assert rel(orc.tt_to_numpy(x), z[key + "_x"]) < 1e-8


def test_structured_dense_formulas_equal_the_oracle_dense_path():
    """tests/structured.py (the checker of the full-size dense GPU test) against the oracle's dense path,
    incl. the reversed-mode quirk of the right TensorTrainDRM matrices."""
    from tests import structured as st
    rng = np.random.default_rng(11)
    for d, n, s, l, r in ((5, 4, 3, 5, 7), (4, 6, 2, 3, 4), (3, 5, 2, 4, 6)):
        shape = (n,) * d
        cores = orc.random_tt(shape, s, rng)
        X = orc.tt_to_numpy(cores)
        ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
        oP, oO = orc.general_sketch("dense", X, ld, rd, "streaming")
        sP, sO = st.dense_sketch_of_tt_ttdrm(cores, ld.cores, rd.cores)
        for a, b in zip(sP + sO, oP + oO):
            assert a.shape == b.shape and rel(a, b) < 1e-12, (d, a.shape, rel(a, b))
        # explicit matrices (DenseGaussianDRM): right matrices sampled on the reversed shape
        A, cols = [], 1
        for mu in range(d - 1):
            cols *= n
            A.append(rng.standard_normal((l, cols)))
        Bw, cols = [], 1
        for mu in range(d - 1):
            cols *= n
            Bw.append(rng.standard_normal((r, cols)))           # walking order: mats[j] covers modes d-1..d-1-j
        dl, dr = orc.DenseDrm(A, shape, False), orc.DenseDrm(Bw, shape, True)
        oP, oO = orc.general_sketch("dense", X, dl, dr, "streaming")
        sP, sO = st.dense_sketch_of_tt_matrices(cores, A, Bw[::-1])
        for a, b in zip(sP + sO, oP + oO):
            assert a.shape == b.shape and rel(a, b) < 1e-12, (d, a.shape, rel(a, b))


def _blocked_fixture():
    z = np.load(os.path.join(GOLDEN, "blocked_cases.npz"))
    return z, json.loads(str(z["meta"]))


def _fx_lists(z, prefix):
    P = [z[f"{prefix}/Psi/{i}"] for i in range(sum(1 for k in z.files if k.startswith(f"{prefix}/Psi/")))]
    O = [z[f"{prefix}/Omega/{i}"] for i in range(sum(1 for k in z.files if k.startswith(f"{prefix}/Omega/")))]
    return P, O


def test_blocked_sketch_and_rank_increase_against_reference_fixtures():
    """(f)3 pinned to the reference: `blocked_stream_sketch` (sketch.py:493-525) and
    `SketchedTensorTrain.increase_rank` (:303-353) as the reference computed them with hash DRMs
    (tests/golden/make_golden_blocked.py).  Here: the package's slicing / placement logic with the oracle as
    the per-block sketch (the GPU twin of this test runs the HIP path)."""
    import tt_sketch_amd as tsa
    from tt_sketch_amd.distributed import HostComm, blocked_stream_sketch_sharded
    from tests.test_distributed_cpu import _oracle_sketch
    z, meta = _blocked_fixture()
    alone = HostComm(0, 1, lambda b: b, lambda b: [b])
    for name, m in meta.items():
        shape = tuple(m["shape"])
        X = tsa.SparseTensor(shape, z[f"{name}/indices"], z[f"{name}/entries"])
        if m["kind"] == "blocked":
            cls = getattr(tsa, m["drm"])
            left = cls(tuple(m["left_rank"]), shape, False, seed=m["left_seed"])
            right = cls(tuple(m["right_rank"]), shape, True, seed=m["right_seed"])
            lsl, rsl = [tuple(s) for s in m["left_slices"]], [tuple(s) for s in m["right_slices"]]
            if m["drm"] != "SparseGaussianDRM":
                continue                      # the oracle stand-in below knows the Gaussian hash DRM only
            blk = blocked_stream_sketch_sharded(X, left, right, lsl, rsl, alone, sketch_fn=_oracle_sketch)
            for tag in ("blocked", "whole"):
                P, O = _fx_lists(z, f"{name}/{tag}")
                for a, b in zip(blk.Psi_cores + blk.Omega_mats, P + O):
                    assert a.shape == b.shape and rel(a, b) < 1e-12, (name, tag, a.shape, rel(a, b))
        else:
            # increase_rank = blocked sketch over [0, old rank, new rank] with block (0, 0) reused
            l0, r0 = tuple(m["left_rank"]), tuple(m["right_rank"])
            l1, r1 = tuple(m["new_left_rank"]), tuple(m["new_right_rank"])
            assert m["new_left_seed"] == m["left_seed"] and m["new_right_seed"] == m["right_seed"]
            left = tsa.SparseGaussianDRM(l1, shape, False, seed=m["left_seed"])
            right = tsa.SparseGaussianDRM(r1, shape, True, seed=m["right_seed"])
            zeros = (0,) * (len(shape) - 1)
            blk = blocked_stream_sketch_sharded(X, left, right, [zeros, l0, l1], [zeros, r0, r1], alone, sketch_fn=_oracle_sketch)
            for tag in ("after", "direct"):
                P, O = _fx_lists(z, f"{name}/{tag}")
                for a, b in zip(blk.Psi_cores + blk.Omega_mats, P + O):
                    assert a.shape == b.shape and rel(a, b) < 1e-12, (name, tag, a.shape, rel(a, b))
            P0, O0 = _fx_lists(z, f"{name}/before")       # the leading block is the old sketch
            for mu, b in enumerate(P0):
                a = blk.Psi_cores[mu][:b.shape[0], :, :b.shape[2]]
                assert rel(a, b) < 1e-12


def test_oracle_tt_svd_against_reference_runs():
    """oracle.tt_svd (checker of the device tt_svd) == the reference's tt_svd on the committed runs: same TT
    ranks, same tensor (singular vectors are sign-free, so cores are compared through the tensor)."""
    z = np.load(os.path.join(GOLDEN, "tt_svd_cases.npz"))
    meta = json.loads(str(z["meta"]))
    for name, m in meta.items():
        X = z[f"{name}/X"]
        ref = [z[f"{name}/core/{i}"] for i in range(len(m["shape"]))]
        got = orc.tt_svd(X, m["rank"])
        assert [c.shape for c in got] == [c.shape for c in ref], name
        assert [c.shape[2] for c in got[:-1]] == m["tt_rank"]
        assert rel(orc.tt_to_numpy(got), orc.tt_to_numpy(ref)) < 1e-12, name
        if name in ("full3", "lowrank5"):
            assert rel(orc.tt_to_numpy(got), X) < 1e-12          # exact when the caps do not bind
