"""orthogonal_sketch / hmt_sketch of one C3 tensor (d = 6, n = 200, TT-rank 100, l = 50, r = 100; hmt: rank 50): ms per call.
Usage: python profiles/scripts/orth_bench.py orth|hmt   (under rocprofv3 --kernel-trace for the per-kernel breakdown)"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd import TensorTrain, orthogonal_sketch, hmt_sketch
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
d, n, s, l, r = 6, 200, 100, 50, 100
ranks = (1,) + (s,) * (d - 1) + (1,)
cores = [rng.standard_normal((ranks[i], n, ranks[i + 1])) / np.sqrt(ranks[i] * n) for i in range(d)]
tt = TensorTrain(cores); tt.prepare_device()
which = sys.argv[1] if len(sys.argv) > 1 else "orth"
f = (lambda: orthogonal_sketch(tt, (l,) * (d - 1), (r,) * (d - 1), seed=1)) if which == "orth" else (lambda: hmt_sketch(tt, l, seed=1))
for _ in range(3): f()
nat.call("ttsk_sync", -1)
t0 = time.perf_counter()
for _ in range(10): f()
nat.call("ttsk_sync", -1)
print(which, (time.perf_counter() - t0) * 100, "ms per call")
