"""Randomised parity sweep, a tool and not part of the collected suite (`python tests/fuzz_parity.py SEED SECONDS`
on a GPU box): TT / sparse / dense / CP inputs of random shapes and
ranks through the fused and the generator paths against the oracle."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tt_sketch_amd as tsa
from tt_sketch_amd import tt_fused
from tt_sketch_amd.sketch_dispatch import general_sketch_device
from oracle import ttsk_oracle as orc
from tests.gpu_build import make_drm, make_tensor
from tests.golden_io import rel
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
n_cases = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    d = int(rng.integers(3, 6))
    big = rng.random() < 0.4
    shape = tuple(int(x) for x in rng.integers(20, 70 if big else 14, d)) if big else tuple(int(x) for x in rng.integers(4, 14, d))
    hi = 110 if big else 12
    s = tuple(int(x) for x in rng.integers(2, hi, d - 1))
    lr = tuple(int(x) for x in rng.integers(2, hi // 2 + 2, d - 1))
    rr = tuple(int(a + rng.integers(1, hi // 2 + 2)) for a in lr)
    kind = rng.choice(["tt", "tt", "sum", "sparse", "dense", "cp"]) if not big else rng.choice(["tt", "sum"])
    if kind == "tt":
        data = ("tt", orc.random_tt(shape, s, rng))
    elif kind == "sum":
        data = ("sum", [("tt", orc.random_tt(shape, s, rng)) for _ in range(int(rng.integers(2, 5)))])
    elif kind == "sparse":
        nnz = int(rng.integers(50, 3000))
        idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
        data = ("sparse", (shape, idx, rng.standard_normal(nnz)))
    elif kind == "dense":
        shape = shape[:4] if d > 4 else shape
        d = len(shape); s = s[:d - 1]; lr = lr[:d - 1]; rr = rr[:d - 1]
        data = ("dense", rng.standard_normal(shape))
    else:
        R = int(rng.integers(2, 9))
        data = ("cp", [rng.standard_normal((n, R)) for n in shape])
    ld = orc.random_tt_drm(shape, lr, False, rng)
    rd = orc.random_tt_drm(shape, rr, True, rng)
    tol = 1e-11
    if kind == "sparse" and rng.random() < 0.6:      # hash DRMs (Gaussian or sign), with rank slices
        d_ = len(shape)
        lo_l = tuple(int(x) for x in rng.integers(0, 3, d_ - 1))
        lo_r = tuple(int(x) for x in rng.integers(0, 3, d_ - 1))
        hi_l = tuple(a + b for a, b in zip(lo_l, lr))
        hi_r = tuple(a + b for a, b in zip(lo_r, rr))
        if rng.random() < 0.5:
            ld = orc.HashGaussDrm(int(rng.integers(0, 2**31)), shape, False, lo_l, hi_l)
            rd = orc.HashGaussDrm(int(rng.integers(0, 2**31)), shape, True, lo_r, hi_r)
            tol = 1e-10                              # device ndtri within a few ulp of the oracle's
        else:
            ld = orc.HashSignDrm(int(rng.integers(0, 2**31)), shape, False, hi_l, lo_l, hi_l)
            rd = orc.HashSignDrm(int(rng.integers(0, 2**31)), shape, True, hi_r, lo_r, hi_r)
    try:
        T = make_tensor(*data)
        L, Rm = make_drm(ld), make_drm(rd)
        got = general_sketch_device(T, L, Rm, tsa.SketchMethod.streaming)
        oP, oO = orc.general_sketch(data[0], data[1], ld, rd, "streaming")
        errs = [rel(a.get(), c) for a, c in zip(got[0] + got[1], oP + oO)]
        if kind in ("tt", "sum"):
            fused = tt_fused.try_stream_sketch(T, L, Rm, tsa.SketchMethod.streaming)
            if fused is not None:
                errs += [rel(a.get(), c) for a, c in zip(fused[0] + fused[1], oP + oO)]
        n_cases += 1
        if max(errs) > tol:
            bad += 1
            print("MISMATCH", kind, shape, s, lr, rr, max(errs), flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", kind, shape, s, lr, rr, repr(e)[:300], flush=True)
print(f"{n_cases} cases, {bad} bad", flush=True)
