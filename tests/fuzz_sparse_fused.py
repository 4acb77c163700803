"""Randomised sweep of the one-pass-per-mode sparse sketch (csrc/sparse_fused.hip) against the generator path of round 2
(sampler.hip + sparse.hip, itself pinned to the oracle) and, for small cases, the oracle -- a tool, not part of the
collected suite: `python tests/fuzz_sparse_fused.py SEED SECONDS` on a GPU box.  Random orders 2..6, mode sizes 1..400,
nonzero counts 1..3e5 (duplicates included), ranks 1..32 with rank slices, SparseGaussianDRM / SparseSignDRM in every pairing
(sign rows with random non-zero counts), table / in-pass factors in every mix."""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat, sparse_fused
from oracle import ttsk_oracle as orc

nat.call("ttsk_init", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
t0, cases, oracle_cases, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    d = int(rng.integers(2, 7))
    shape = tuple(int(rng.integers(1, 400)) if rng.random() < 0.5 else int(rng.integers(1, 12)) for _ in range(d))
    nnz = int(10 ** rng.uniform(0, 5.5))
    idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
    val = rng.standard_normal(nnz)
    lo_l = tuple(int(rng.integers(0, 3)) for _ in range(d - 1))
    lo_r = tuple(int(rng.integers(0, 3)) for _ in range(d - 1))
    wmax = 33 if rng.random() < 0.5 else 17
    hi_l = tuple(a + int(rng.integers(1, wmax)) for a in lo_l)
    hi_r = tuple(a + int(rng.integers(1, wmax)) for a in lo_r)
    sl, sr = int(rng.integers(0, 2**31)), int(rng.integers(0, 2**31))
    kl, kr = ("s" if rng.random() < 0.4 else "g"), ("s" if rng.random() < 0.4 else "g")
    tr_l = tuple(h + int(rng.integers(0, 3)) for h in hi_l)          # sign rows longer than the slice used
    tr_r = tuple(h + int(rng.integers(0, 3)) for h in hi_r)
    nz_l = tuple(int(rng.integers(0, t + 1)) for t in tr_l)
    nz_r = tuple(int(rng.integers(0, t + 1)) for t in tr_r)

    def one(kind, transpose, seed, lo, hi, tr, nz):
        # (constructor arguments in the tensor's mode order; a transposed DRM stores them reversed, the oracle's classes take
        # them in walking order)
        w = (lambda v: v[::-1]) if transpose else (lambda v: v)
        if kind == "g":
            return (tsa.SparseGaussianDRM(hi, shape, transpose, seed=seed, rank_min=lo, rank_max=hi, true_rank=hi),
                    orc.HashGaussDrm(seed, shape, transpose, w(lo), w(hi)))
        return (tsa.SparseSignDRM(tr, shape, transpose, seed=seed, rank_min=lo, rank_max=hi, true_rank=tr, num_non_zero_per_row=w(nz)),
                orc.HashSignDrm(seed, shape, transpose, w(tr), w(lo), w(hi), w(nz)))
    mk = lambda: (one(kl, False, sl, lo_l, hi_l, tr_l, nz_l)[0], one(kr, True, sr, lo_r, hi_r, tr_r, nz_r)[0])
    os.environ["TTSK_SPARSE_FUSED"] = "1"
    T = tsa.SparseTensor(shape, idx, val)
    ld, rd = mk()
    if sparse_fused.try_sparse_gauss_sketch(T, ld, rd, tsa.SketchMethod.streaming) is None:
        continue
    new = tsa.general_sketch(T, ld, rd, tsa.SketchMethod.streaming)
    os.environ["TTSK_SPARSE_FUSED"] = "0"
    old = tsa.general_sketch(tsa.SparseTensor(shape, idx, val), *mk(), tsa.SketchMethod.streaming)
    cases += 1
    err = max(rel(a, b) for a, b in zip(new.Psi_cores + new.Omega_mats, old.Psi_cores + old.Omega_mats))
    if nnz <= 3000 and int(np.prod(shape, dtype=object)) < 2**31:
        oP, oO = orc.general_sketch("sparse", (shape, idx, val), one(kl, False, sl, lo_l, hi_l, tr_l, nz_l)[1],
                                    one(kr, True, sr, lo_r, hi_r, tr_r, nz_r)[1], "streaming")
        err = max(err, max(rel(a, b) for a, b in zip(new.Psi_cores + new.Omega_mats, oP + oO)))
        oracle_cases += 1
    worst = max(worst, err)
    if not err <= 1e-10:
        print("FAIL", dict(shape=shape, nnz=nnz, kl=kl, kr=kr, lo_l=lo_l, hi_l=hi_l, lo_r=lo_r, hi_r=hi_r, tr_l=tr_l, tr_r=tr_r, nz_l=nz_l, nz_r=nz_r, sl=sl, sr=sr, err=err), flush=True)
print(f"fuzz_sparse_fused: {cases} cases ({oracle_cases} also against the oracle), worst relative difference {worst:.2e}")
