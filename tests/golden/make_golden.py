"""Generate the golden fixtures in tests/golden/ by running the REAL reference.

Build container only (needs /root/reference + oracle/_ref, see
oracle/build_ref.sh).  The committed .npz files hold data only: inputs,
injected DRM data and the outputs the reference produced for them.

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

DRM data is injected: TensorTrainDRM via ``cores=``,
DenseGaussianDRM by overwriting ``sketching_mats``, hash DRMs by seed.  Under
NumPy 2 the hash-DRM wrappers need ``drm.seed`` to be an unsigned integer
for their ``np.mod(..., dtype=np.uint64)`` call to type-check; the value is
unchanged.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402

ref_loader.load()

from tt_sketch.drm import (DenseGaussianDRM, SparseGaussianDRM,  # noqa: E402
                           SparseSignDRM, TensorTrainDRM)
from tt_sketch.drm import fast_lazy_gaussian as flg  # noqa: E402
from tt_sketch.sketch import assemble_sketched_tt  # noqa: E402
from tt_sketch.sketch_dispatch import (SketchMethod, general_sketch,  # noqa: E402
                                       get_sketch_method)
from tt_sketch.tensor import (CPTensor, DenseTensor, SparseTensor,  # noqa: E402
                              TensorSum, TensorTrain, TuckerTensor)
import scipy.special  # noqa: E402


# ------------------------------------------------------------------ sampler
def sampler_fixture():
    rng = np.random.default_rng(1234)
    out = {}
    v = np.concatenate([np.arange(16, dtype=np.uint64),
                        rng.integers(0, 2**63, 48, dtype=np.uint64),
                        np.array([2**64 - 1, 2**63, 179], dtype=np.uint64)])
    h = v.copy()
    flg.hash_int_c(h)
    out["hash_in"], out["hash_out"] = v, h
    cases = [
        dict(shape=(4, 5), N=3, rank_min=0, rank_max=3, seed=7, fixed=[[0, 3, 1], [0, 4, 2]]),
        dict(shape=(4, 5), N=3, rank_min=1, rank_max=3, seed=7, fixed=[[0, 3, 1], [0, 4, 2]]),
        dict(shape=(9, 10, 11), N=200, rank_min=0, rank_max=6, seed=180),
        dict(shape=(9, 10, 11), N=200, rank_min=2, rank_max=5, seed=2**31 + 12345),
        dict(shape=(13,), N=13, rank_min=0, rank_max=4, seed=3),
        dict(shape=(200, 150, 100, 120), N=500, rank_min=0, rank_max=15, seed=4),
        # 32-bit running product wraps (fast_lazy_gaussian.pyx:65)
        dict(shape=(70000, 70000, 3), N=300, rank_min=0, rank_max=5, seed=99),
        dict(shape=(3, 5, 7, 2, 4), N=0, rank_min=0, rank_max=3, seed=5),
    ]
    meta = []
    for ci, c in enumerate(cases):
        if "fixed" in c:
            idx = np.array(c["fixed"], dtype=np.int64)
        else:
            idx = np.stack([rng.integers(0, n, c["N"]) for n in c["shape"]]).astype(np.int64)
            idx = idx.reshape(len(c["shape"]), c["N"])
        out[f"s{ci}_idx"] = idx
        seed = c["seed"]
        if idx.shape[1] > 0:
            rd = np.array(flg._inds_to_rand_double(
                idx.astype(np.uint64), np.array(c["shape"], dtype=np.uint64),
                c["rank_min"], c["rank_max"], np.uint64(seed)))
            out[f"s{ci}_rand_double"] = rd.reshape(idx.shape[1], -1)
        out[f"s{ci}_normal"] = flg.inds_to_normal(idx, c["shape"], c["rank_min"],
                                                  c["rank_max"], np.uint64(seed))
        true_rank = c["rank_max"] + 3
        for nnz in (1, 2, true_rank):
            out[f"s{ci}_sign_nnz{nnz}"] = flg.inds_to_sparse_sign(
                idx, c["shape"], true_rank, c["rank_min"], c["rank_max"], nnz,
                np.uint64(seed))
        meta.append({k: v for k, v in c.items() if k != "fixed"} | {"true_rank": true_rank})
    # ndtri (third-party: SciPy's cephes ndtri, fast_lazy_gaussian.pyx:49)
    x = np.concatenate([
        rng.random(4000), 2.0 ** -rng.integers(1, 53, 500) * rng.random(500),
        1 - 2.0 ** -rng.integers(1, 53, 500), np.array([0.5, 0.135, 0.1353352832366127,
        0.8646647167633873, 2.0**-52, 1 - 2.0**-53, 1e-300, 0.0])])
    out["ndtri_x"] = x
    out["ndtri_y"] = scipy.special.ndtri(x)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "hash_sampler.npz"), **out)
    print("hash_sampler.npz", len(out), "arrays")


# ------------------------------------------------------------------ sketches
def make_tensor(kind, shape, rng, rank=3):
    d = len(shape)
    if kind == "tt":
        rk = (1,) + (rank,) * (d - 1) + (1,)
        cores = [rng.standard_normal((rk[i], shape[i], rk[i + 1])) for i in range(d)]
        return TensorTrain(cores), {"cores": cores}
    if kind == "cp":
        cores = [rng.standard_normal((n, rank)) for n in shape]
        return CPTensor(cores), {"cores": cores}
    if kind == "tucker":
        rk = tuple(min(rank + i % 2, n) for i, n in enumerate(shape))
        factors = [np.linalg.qr(rng.standard_normal((n, r)))[0].T.copy()
                   for r, n in zip(rk, shape)]
        core = rng.standard_normal(rk)
        return TuckerTensor(factors, core), {"factors": factors, "core": core}
    if kind == "dense":
        data = rng.standard_normal(shape)
        return DenseTensor(data), {"data": data}
    if kind == "sparse":
        nnz = 60
        idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
        ent = rng.standard_normal(nnz)
        return SparseTensor(tuple(shape), idx, ent), {"indices": idx, "entries": ent}
    if kind == "sparse_lowrank":
        # dense low-rank tensor stored as COO (as tests/test_sketching_matrix.py:269-306)
        rk = (1,) + (rank,) * (d - 1) + (1,)
        cores = [rng.standard_normal((rk[i], shape[i], rk[i + 1])) for i in range(d)]
        sp = TensorTrain(cores).dense().to_sparse()
        return sp, {"indices": np.asarray(sp.indices, dtype=np.int64), "entries": sp.entries}
    raise ValueError(kind)


def make_drm(drm_kind, rank, shape, transpose, rng, seed, rank_min=None, rank_max=None,
             true_rank=None):
    """Returns (reference DRM object, dict of injected data)."""
    kw = {}
    if rank_min is not None:
        kw = dict(rank_min=rank_min, rank_max=rank_max, true_rank=true_rank)
    full = true_rank if true_rank is not None else rank
    d = len(shape)
    if drm_kind == "tt":
        shp = shape[::-1] if transpose else shape
        rk = tuple(full[::-1]) if transpose else tuple(full)
        rka = (1,) + rk
        cores = [rng.standard_normal((rka[i], shp[i], rka[i + 1])) / np.sqrt(rka[i])
                 for i in range(d - 1)]
        drm = TensorTrainDRM(rank, shape, transpose, seed=seed, cores=cores, **kw)
        info = {"cores": cores}
    elif drm_kind == "dense":
        drm = DenseGaussianDRM(rank, shape, transpose, seed=seed, **kw)
        mats = [rng.standard_normal(m.shape) for m in drm.sketching_mats]
        drm.sketching_mats = mats
        info = {"mats": mats}
    elif drm_kind == "hashgauss":
        drm = SparseGaussianDRM(rank, shape, transpose, seed=seed, **kw)
        drm.seed = np.uint64(int(drm.seed))
        info = {}
    elif drm_kind == "hashsign":
        drm = SparseSignDRM(rank, shape, transpose, seed=seed, **kw)
        drm.seed = np.uint64(int(drm.seed))
        info = {"nnz": np.array(drm.nnz)}
    else:
        raise ValueError(drm_kind)
    info.update(seed=np.array(int(drm.seed)), rank_min=np.array(drm.rank_min),
                rank_max=np.array(drm.rank_max), true_rank=np.array(drm.true_rank),
                transpose=np.array(transpose))
    return drm, info


def put(out, prefix, d):
    for k, v in d.items():
        if isinstance(v, (list, tuple)) and len(v) and isinstance(v[0], np.ndarray):
            for i, a in enumerate(v):
                out[f"{prefix}/{k}/{i}"] = np.ascontiguousarray(a)
        else:
            out[f"{prefix}/{k}"] = np.asarray(v)


def sketch_fixture():
    rng = np.random.default_rng(20240)
    cases = []
    # (name, tensor kind(s), shape, left drm, right drm, left rank, right rank)
    S4 = (5, 6, 7, 4)
    S3 = (9, 10, 11)
    cases += [("tt_ttdrm", ["tt"], S4, "tt", "tt", (4, 5, 3), (5, 8, 4)),
              ("tt_densedrm", ["tt"], S3, "dense", "dense", (4, 5), (6, 7)),
              ("tt_mixdrm", ["tt"], S3, "tt", "dense", (4, 5), (6, 7)),
              ("tt_leftbig", ["tt"], S4, "tt", "tt", (7, 9, 8), (5, 6, 4)),
              ("cp_ttdrm", ["cp"], S4, "tt", "tt", (4, 5, 3), (5, 8, 4)),
              ("tucker_ttdrm", ["tucker"], S4, "tt", "tt", (4, 5, 3), (5, 8, 4)),
              ("dense_ttdrm", ["dense"], S4, "tt", "tt", (3, 4, 3), (5, 6, 4)),
              ("dense_densedrm", ["dense"], S4, "dense", "dense", (3, 4, 3), (5, 6, 4)),
              ("dense_mixdrm", ["dense"], (6, 5, 4, 7), "dense", "tt", (3, 4, 3), (5, 6, 5)),
              ("sparse_ttdrm", ["sparse"], S4, "tt", "tt", (3, 4, 3), (5, 6, 4)),
              ("sparse_densedrm", ["sparse"], S4, "dense", "dense", (3, 4, 3), (5, 6, 4)),
              ("sparse_hashgauss", ["sparse"], S4, "hashgauss", "hashgauss", (3, 4, 3), (5, 6, 4)),
              ("sparse_hashsign", ["sparse"], S4, "hashsign", "hashsign", (3, 4, 3), (5, 6, 4)),
              ("sparse_tt_hashgauss", ["sparse"], S3, "tt", "hashgauss", (3, 4), (5, 6)),
              ("sparselr_hashgauss", ["sparse_lowrank"], (6, 7, 5), "hashgauss", "hashsign", (4, 3), (6, 5)),
              ("sum_tt3", ["tt", "tt", "tt"], S4, "tt", "tt", (4, 5, 3), (5, 8, 4)),
              ("sum_tt_sparse", ["tt", "sparse"], S3, "tt", "tt", (4, 5), (6, 7)),
              ("sum_tt_cp_dense", ["tt", "cp", "dense"], (4, 5, 6), "tt", "tt", (3, 4), (4, 6)),
              ("tt_d2", ["tt"], (10, 11), "tt", "tt", (3,), (4,)),
              ]
    out = {}
    meta = {}
    for name, kinds, shape, ldk, rdk, lrank, rrank in cases:
        tensors, infos = [], []
        for k in kinds:
            t, info = make_tensor(k, shape, rng)
            tensors.append(t)
            infos.append(info)
        tensor = tensors[0] if len(tensors) == 1 else TensorSum(tensors)
        # Inputs that are not exactly low rank only pin a well-defined answer if every Omega and
        # every Psi unfolding is well conditioned (a 3 x 5 sign matrix is singular one time in
        # five); walk the DRM seeds until the reference's own sketch is.
        generic = any(k in ("sparse", "dense") for k in kinds)
        for bump in range(50):
            ldrm, linfo = make_drm(ldk, lrank, shape, False, rng, seed=11 + bump)
            rdrm, rinfo = make_drm(rdk, rrank, shape, True, rng, seed=23 + bump)
            if not generic:
                break
            sk = general_sketch(tensor, ldrm, rdrm, SketchMethod.streaming)
            mats = list(sk.Omega_mats) + [P.reshape(-1, P.shape[2]) for P in sk.Psi_cores[:-1]]
            sv = [np.linalg.svd(M, compute_uv=False) for M in mats]
            if min(x[-1] / x[0] for x in sv) > 1e-4:
                break
        else:
            raise RuntimeError(f"no well conditioned seed for {name}")
        for i, info in enumerate(infos):
            put(out, f"{name}/tensor{i}", info)
        put(out, f"{name}/left_drm", linfo)
        put(out, f"{name}/right_drm", rinfo)
        methods = ["streaming", "orthogonal", "hmt"]
        if all(a > b for a, b in zip(lrank, rrank)):
            methods = ["streaming"]
        meta[name] = dict(kinds=[k.replace("sparse_lowrank", "sparse") for k in kinds],
                          shape=shape, left_drm=ldk, right_drm=rdk,
                          left_rank=lrank, right_rank=rrank, methods=methods)
        lc = list(get_sketch_method(tensor, ldrm)(tensor))
        rc = list(get_sketch_method(tensor, rdrm)(tensor))
        if len(tensors) == 1:
            put(out, f"{name}/out", {"left_contractions": lc, "right_contractions": rc})
        else:
            for s in range(len(tensors)):
                put(out, f"{name}/out", {f"left_contractions_s{s}": [x[s] for x in lc],
                                         f"right_contractions_s{s}": [x[s] for x in rc]})
        for m in methods:
            left = None if m == "hmt" else ldrm
            sk = general_sketch(tensor, left, rdrm, SketchMethod(m))
            put(out, f"{name}/out/{m}", {"Psi": sk.Psi_cores, "Omega": sk.Omega_mats})
            if m == "streaming":
                for direction in ("left", "right"):
                    cc = assemble_sketched_tt(sk, direction=direction)
                    put(out, f"{name}/out/{m}", {f"C_{direction}": cc})

    # blocked / sliced DRMs: rank_min > 0 (drm_base.py:86-109)
    name = "tt_ttdrm_sliced"
    shape = S4
    t, info = make_tensor("tt", shape, rng)
    put(out, f"{name}/tensor0", info)
    ldrm, linfo = make_drm("tt", (4, 5, 6), shape, False, rng, 11,
                           rank_min=(1, 2, 2), rank_max=(3, 5, 4), true_rank=(4, 5, 6))
    rdrm, rinfo = make_drm("tt", (6, 8, 9), shape, True, rng, 23,
                           rank_min=(2, 0, 3), rank_max=(6, 5, 9), true_rank=(6, 8, 9))
    put(out, f"{name}/left_drm", linfo)
    put(out, f"{name}/right_drm", rinfo)
    sk = general_sketch(t, ldrm, rdrm, SketchMethod.streaming)
    put(out, f"{name}/out/streaming", {"Psi": sk.Psi_cores, "Omega": sk.Omega_mats})
    put(out, f"{name}/out", {"left_contractions": list(ldrm.sketch_tt(t)),
                             "right_contractions": list(rdrm.sketch_tt(t))})
    meta[name] = dict(kinds=["tt"], shape=shape, left_drm="tt", right_drm="tt",
                      left_rank=ldrm.rank, right_rank=tuple(rdrm.rank[::-1]),
                      methods=["streaming"], sliced=True)

    name = "sparse_hash_sliced"
    t, info = make_tensor("sparse", shape, rng)
    put(out, f"{name}/tensor0", info)
    ldrm, linfo = make_drm("hashgauss", (4, 5, 6), shape, False, rng, 11,
                           rank_min=(1, 2, 2), rank_max=(3, 5, 4), true_rank=(4, 5, 6))
    rdrm, rinfo = make_drm("hashsign", (6, 8, 9), shape, True, rng, 23,
                           rank_min=(2, 0, 3), rank_max=(6, 5, 9), true_rank=(6, 8, 9))
    put(out, f"{name}/left_drm", linfo)
    put(out, f"{name}/right_drm", rinfo)
    sk = general_sketch(t, ldrm, rdrm, SketchMethod.streaming)
    put(out, f"{name}/out/streaming", {"Psi": sk.Psi_cores, "Omega": sk.Omega_mats})
    put(out, f"{name}/out", {"left_contractions": list(ldrm.sketch_sparse(t)),
                             "right_contractions": list(rdrm.sketch_sparse(t))})
    meta[name] = dict(kinds=["sparse"], shape=shape, left_drm="hashgauss",
                      right_drm="hashsign", left_rank=ldrm.rank,
                      right_rank=tuple(rdrm.rank[::-1]), methods=["streaming"], sliced=True)

    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "sketch_cases.npz"), **out)
    print("sketch_cases.npz", len(out), "arrays,", len(meta), "cases")


# ------------------------------------------------------------------ tt_gmres
def gmres_fixture():
    """Reference ``tt_sum_gmres`` with the deterministic roundings ("exact", "pairwise") on a small
    preconditioned problem, plus one MPO product and both TTPrecond directions."""
    from tt_sketch.tt_gmres import MPO, TTLinearMapSum, TTPrecond, tt_sum_gmres
    rng = np.random.default_rng(77)
    shape = (6, 5, 4, 5)
    d = len(shape)
    out = {"shape": np.array(shape)}

    def mpo_cores(rank, scale):
        rk = (1,) + (rank,) * (d - 1) + (1,)
        cores = []
        for k, n in enumerate(shape):
            C = rng.standard_normal((rk[k], n, n, rk[k + 1]))
            C = C + C.transpose(0, 2, 1, 3)
            cores.append(C * (scale ** (1 / d)) / np.sqrt(C.size))
        return cores

    maps = [[2.0 ** (k == 0) * np.eye(n).reshape(1, n, n, 1) for k, n in enumerate(shape)],
            mpo_cores(2, 0.4), mpo_cores(3, 0.3)]
    for m, cores in enumerate(maps):
        for k, C in enumerate(cores):
            out[f"map{m}_core{k}"] = C
    P = rng.standard_normal((shape[1], shape[1]))
    P = P @ P.T / shape[1] + np.eye(shape[1])
    out["precond"] = P
    b = TensorTrain.random(shape, 2)
    b.cores = [rng.standard_normal(C.shape) / np.sqrt(C.shape[0] * C.shape[1]) for C in b.cores]
    for k, C in enumerate(b.cores):
        out[f"b_core{k}"] = C

    A = TTLinearMapSum([MPO([C.copy() for C in cores]) for cores in maps])
    pre = TTPrecond(P, shape, mode=1)
    out["mpo_apply"] = MPO(maps[2])(b).to_numpy()
    out["precond_backward"] = pre.backward_call(b).to_numpy()
    out["precond_forward"] = pre.forward_call(b).to_numpy()
    for method in ("exact", "pairwise"):
        for use_pre in (False, True):
            x, hist = tt_sum_gmres(A, b, max_rank=6, precond=pre if use_pre else None, tolerance=1e-8,
                                   maxiter=8, rounding_method=method, save_basis=True)
            key = f"{method}_{int(use_pre)}"
            out[key + "_x"] = x.to_numpy()
            out[key + "_residual_norm"] = np.array(hist["residual_norm"])
            out[key + "_w_norm"] = np.array(hist["w_norm"])
            out[key + "_rank"] = np.array(hist["rank"])
            out[key + "_H"] = hist["H_matrix"]
            out[key + "_y"] = hist["y"]
            print("gmres", key, hist["residual_norm"][-1], hist["rank"][-1])
    # sketched roundings at a rank that represents every iterate exactly: the run is then independent
    # of the random DRMs (to rounding) and comparable across implementations
    for method in ("sketch", "orth_sketch"):
        x, hist = tt_sum_gmres(A, b, max_rank=30, tolerance=1e-9, maxiter=25, rounding_method=method)
        out[method + "_full_x"] = x.to_numpy()
        out[method + "_full_residual_norm"] = np.array(hist["residual_norm"])
        out[method + "_full_rank"] = np.array(hist["rank"])
        print("gmres", method, hist["residual_norm"])
    np.savez_compressed(os.path.join(HERE, "gmres_case.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sampler", "sketch", "gmres"]
    if "sampler" in which:
        sampler_fixture()
    if "sketch" in which:
        sketch_fixture()
    if "gmres" in which:
        gmres_fixture()
