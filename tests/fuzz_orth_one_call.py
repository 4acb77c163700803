"""Randomised sweep of the one-call orthogonal / hmt sketch and of the one-call assembly against the oracle
(a tool, not collected by pytest):  python tests/fuzz_orth_one_call.py [seconds] [seed] [big]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ttsk_oracle as orc  # noqa: E402
import tt_sketch_amd as tsa  # noqa: E402
from tt_sketch_amd import _native, tt_fused  # noqa: E402
from tt_sketch_amd.sketch import assemble_sketched_tt  # noqa: E402

_native.call("ttsk_init", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
hits = {"one": 0, "other": 0}
real = tt_fused.try_orth_sketch


def spy(*a, **k):
    out = real(*a, **k)
    hits["one" if out is not None else "other"] += 1
    return out


tt_fused.try_orth_sketch = spy


from test_gpu_c3_solves import tt_rel_diff  # noqa: E402  (|| A - B || / || B || of two TTs by a QR sweep)


def same_tensor(got, want, what, tol=1e-9):
    """tensor-level agreement and orthonormal left unfoldings: what holds whatever the conditioning of the unfoldings"""
    got = [np.asarray(c) for c in got]
    err = tt_rel_diff(got, [np.asarray(c) for c in want])
    assert err <= tol, (what, "tensor", err)
    for k, c in enumerate(got[:-1]):
        q = c.reshape(-1, c.shape[2])
        dev = np.linalg.norm(q.T @ q - np.eye(q.shape[1]))
        assert dev <= 1e-11, (what, "orthonormality", k, dev)


def close(got, want, tol, what):
    assert [np.asarray(c).shape for c in got] == [np.asarray(c).shape for c in want], what
    for k, (g, w) in enumerate(zip(got, want)):
        g, w = np.asarray(g), np.asarray(w)
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-300)
        assert err <= tol, (what, k, err)


t0, cases = time.time(), 0
while time.time() - t0 < budget:
    if BIG:      # few, larger cases: ranks to 200 (two-block factorisations, global-memory sign kernel), modes to 200
        d = int(rng.integers(3, 5))
        shape = tuple(int(x) for x in rng.integers(40, 200, size=d))
        s_in = tuple(int(x) for x in rng.integers(30, 160, size=d - 1))
        hi_l, hi_d = 140, 90
    else:
        d = int(rng.integers(2, 7))
        shape = tuple(int(x) for x in rng.integers(2, 40, size=d))
        s_in = tuple(int(x) for x in rng.integers(1, 25, size=d - 1))
        hi_l, hi_d = 20, 20
    equal = rng.random() < 0.5
    if equal:
        l = (int(rng.integers(1, hi_l)),) * (d - 1)
        r = (l[0] + int(rng.integers(1, hi_d)),) * (d - 1)
    else:
        l = tuple(int(x) for x in rng.integers(1, hi_l, size=d - 1))
        r = tuple(x + int(y) for x, y in zip(l, rng.integers(1, hi_d, size=d - 1)))
    # the reference trims sketch ranks to what the unfoldings allow
    l = tuple(tsa.utils.process_tt_rank(l, shape, trim=True))
    r = tuple(tsa.utils.process_tt_rank(r, shape, trim=True))
    if not all(a < b for a, b in zip(l, r)):
        continue
    # cores are compared entry by entry: the unfoldings have to have full column rank (TT rank >= sketch rank), else the
    # completion of Q is arbitrary
    s_in = tuple(max(a, b) for a, b in zip(s_in, l))
    cores = orc.random_tt(shape, s_in, rng)
    ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
    left = tsa.TensorTrainDRM(l, shape, False, seed=1, cores=[np.array(c) for c in ld.cores])
    right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=[np.array(c) for c in rd.cores])
    X = tsa.TensorTrain(cores)
    what = f"shape {shape} s {s_in} l {l} r {r}"
    try:
        # entry by entry only where every unfolding is comfortably tall: the Q of a (nearly) square unfolding is known to
        # kappa eps only, in numpy as much as here (a 15 x 15 case of this sweep differed by 1.5e-6 with the tensors equal to 1e-14)
        kl = (1,) + l
        tall = all(kl[mu] * shape[mu] >= 2 * l[mu] for mu in range(d - 1))
        want, wom = orc.general_sketch("tt", cores, ld, rd, "orthogonal")
        cond = max(np.linalg.cond(np.asarray(o)) for o in wom) if wom else 1.0      # pinv(Omega) is known to kappa eps only
        tall = tall and cond < 1e5
        got = tsa.orthogonal_sketch(X, l, r, left_drm=left, right_drm=right).cores
        same_tensor(got, want, "orthogonal " + what, max(1e-9, 1e-12 * cond))
        if tall:
            close(got, want, 1e-7, "orthogonal " + what)
        rdh = orc.random_tt_drm(shape, l, True, rng)
        righth = tsa.TensorTrainDRM(l, shape, True, seed=3, cores=[np.array(c) for c in rdh.cores])
        want, _ = orc.general_sketch("tt", cores, None, rdh, "hmt")
        got = tsa.hmt_sketch(X, l, drm=righth).cores
        same_tensor(got, want, "hmt " + what)
        if tall:
            close(got, want, 1e-7, "hmt " + what)
        stt = tsa.stream_sketch(X, l, r, left_drm=left, right_drm=right)
        P, O = orc.general_sketch("tt", cores, ld, rd, "streaming")
        for direction in ("right", "left"):
            one = assemble_sketched_tt(stt.sketch_, direction=direction)
            ref = orc.assemble(P, O, direction=direction) if "direction" in orc.assemble.__code__.co_varnames else None
            os.environ["TTSK_ASSEMBLE_ONE_CALL"] = "0"
            pairs = assemble_sketched_tt(stt.sketch_, direction=direction)
            os.environ["TTSK_ASSEMBLE_ONE_CALL"] = "1"
            close(one, pairs, 1e-8, f"assemble {direction} " + what)
            if ref is not None:
                # since the refinement step of the assembly: lstsq's accuracy as TENSORS whatever kappa(Omega) is (the
                # oracle's own lstsq is good to ~kappa eps of the data); entry by entry only where Omega is well conditioned
                err = tt_rel_diff([np.asarray(c) for c in one], [np.asarray(c) for c in ref])
                assert err <= max(1e-11, 1e-14 * cond), (f"assemble-vs-oracle {direction} " + what, "tensor", err, cond)
                if cond < 1e5:
                    close(one, ref, 1e-6, f"assemble-vs-oracle {direction} " + what)
    except AssertionError:
        print("FAILED", what, flush=True)
        raise
    cases += 1
print(f"{cases} cases green ({hits['one']} sketches through the one-call path, {hits['other']} declined) in {time.time() - t0:.0f} s")
