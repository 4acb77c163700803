import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd import TensorTrain, stream_sketch
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
d, n, l, r = 6, 200, 50, 100
for s in (100, 30, 10):
    ranks = (1,) + (s,) * (d - 1) + (1,)
    tt = TensorTrain([rng.standard_normal((ranks[i], n, ranks[i + 1])) / np.sqrt(ranks[i] * n) for i in range(d)])
    tt.prepare_device()
    f = lambda: stream_sketch(tt, (l,) * (d - 1), (r,) * (d - 1), seed=1).to_tt()
    f(); nat.call("ttsk_sync", -1)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o = f(); nat.call("ttsk_sync", -1); ts.append(time.perf_counter() - t0)
    err = o.error(tt.to_device(), relative=True) if s < 50 else float("nan")
    print(f"TT rank {s}: stream_sketch + to_tt {min(ts)*1e3:7.2f} ms   rel. recovery error {err:.2e}", flush=True)
