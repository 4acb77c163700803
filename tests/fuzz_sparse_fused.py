"""Randomised sweep of the one-pass-per-mode sparse sketch (csrc/sparse_fused.hip) against the generator path of round 2
(sampler.hip + sparse.hip, itself pinned to the oracle) and, for small cases, the oracle -- a tool, not part of the
collected suite: `python tests/fuzz_sparse_fused.py SEED SECONDS` on a GPU box.  Random orders 2..6, mode sizes 1..400,
nonzero counts 1..3e5 (duplicates included), ranks 1..16 with rank slices, table / in-pass factors in every mix."""
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
    hi_l = tuple(a + int(rng.integers(1, 17)) for a in lo_l)
    hi_r = tuple(a + int(rng.integers(1, 17)) for a in lo_r)
    sl, sr = int(rng.integers(0, 2**31)), int(rng.integers(0, 2**31))
    mk = lambda: (tsa.SparseGaussianDRM(hi_l, shape, False, seed=sl, rank_min=lo_l, rank_max=hi_l, true_rank=hi_l),
                  tsa.SparseGaussianDRM(hi_r, shape, True, seed=sr, rank_min=lo_r, rank_max=hi_r, true_rank=hi_r))
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
        oP, oO = orc.general_sketch("sparse", (shape, idx, val), orc.HashGaussDrm(ld.seed, shape, False, lo_l, hi_l),
                                    orc.HashGaussDrm(rd.seed, shape, True, lo_r[::-1], hi_r[::-1]), "streaming")
        err = max(err, max(rel(a, b) for a, b in zip(new.Psi_cores + new.Omega_mats, oP + oO)))
        oracle_cases += 1
    worst = max(worst, err)
    if not err <= 1e-10:
        print("FAIL", dict(shape=shape, nnz=nnz, lo_l=lo_l, hi_l=hi_l, lo_r=lo_r, hi_r=hi_r, sl=sl, sr=sr, err=err), flush=True)
print(f"fuzz_sparse_fused: {cases} cases ({oracle_cases} also against the oracle), worst relative difference {worst:.2e}")
