"""Randomised sweep of the device-resident TT arithmetic (round_dev vs round: ranks and tensors, orthogonalize_dev,
dot), a tool and not part of the collected suite (`python tests/fuzz_round.py SEED SECONDS` on a GPU box)."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tt_sketch_amd as tsa
from oracle import ttsk_oracle as orc
from tests.golden_io import rel
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); n = bad = rankdiff = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    d = int(rng.integers(2, 6))
    shape = tuple(int(x) for x in rng.integers(2, 12, d))
    ranks = tuple(int(x) for x in rng.integers(1, 25, d - 1))
    tt = tsa.TensorTrain(orc.random_tt(shape, ranks, rng))
    kw = {}
    if rng.random() < 0.6: kw["max_rank"] = int(rng.integers(1, 12))
    if rng.random() < 0.5: kw["eps"] = float(10.0 ** rng.uniform(-8, -1))
    try:
        want = tt.round(**kw); got = tt.round_dev(**kw)
        a, b = got.to_numpy(), want.to_numpy()
        n += 1
        if not np.all(np.isfinite(a)):
            bad += 1; print("NAN", shape, ranks, kw, flush=True)
        elif got.rank != want.rank:
            rankdiff += 1
            # ranks may differ when a singular value sits on the threshold; the tensors must still agree to eps
            tol = max(kw.get("eps", 0) * 10, 1e-9)
            if rel(a, b) > tol * 10 + 1e-9: bad += 1; print("RANK+ERR", shape, ranks, kw, got.rank, want.rank, rel(a, b), flush=True)
        elif rel(a, b) > 1e-8:
            bad += 1; print("ERR", shape, ranks, kw, rel(a, b), flush=True)
        o = tt.orthogonalize_dev()
        if rel(o.to_numpy(), tt.to_numpy()) > 1e-11: bad += 1; print("ORTH", shape, ranks, flush=True)
        x = tt.to_device(); y = tsa.TensorTrain(orc.random_tt(shape, ranks, rng)).to_device()
        if abs(x.dot(y) - float(np.vdot(tt.to_numpy(), y.to_numpy()))) > 1e-11 * x.norm() * y.norm(): bad += 1; print("DOT", shape, ranks, flush=True)
    except Exception as e:
        bad += 1; print("EXC", shape, ranks, kw, repr(e)[:200], flush=True)
print(f"{n} cases, {bad} bad, {rankdiff} rank differences", flush=True)
