import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd import TensorTrain, TensorSum, TensorTrainDRM, stream_sketch
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
d, n, s, l, r, nt = 6, 128, 20, 50, 100, 32
shape = (n,) * d
ranks = (1,) + (s,) * (d - 1) + (1,)
terms = []
for t in range(nt):
    cores = [rng.standard_normal((ranks[i], n, ranks[i + 1])) / np.sqrt(ranks[i] * n) for i in range(d)]
    tt = TensorTrain(cores); tt.prepare_device(); terms.append(tt)
X = TensorSum(terms)
left = TensorTrainDRM((l,) * (d - 1), shape, False, seed=1); right = TensorTrainDRM((r,) * (d - 1), shape, True, seed=2)
def T(f, reps=5):
    f(); nat.call("ttsk_sync", -1)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = f(); nat.call("ttsk_sync", -1); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, out
ms, stt = T(lambda: stream_sketch(X, (l,) * (d - 1), (r,) * (d - 1), left_drm=left, right_drm=right))
print(f"C5 stream_sketch(TensorSum of {nt} TT rank {s}, d={d}, n={n}) l={l} r={r}: {ms:.2f} ms  -> {nt * d / ms * 1e3:.0f} TT-cores/s")
os.environ["TTSK_SKINNY"] = "1"
