import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd import TensorTrain, TensorTrainDRM, stream_sketch, orthogonal_sketch, hmt_sketch
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
d, n, s, l, r = 6, 200, 100, 50, 100
shape = (n,) * d
ranks = (1,) + (s,) * (d - 1) + (1,)
cores = [rng.standard_normal((ranks[i], n, ranks[i + 1])) / np.sqrt(ranks[i] * n) for i in range(d)]
tt = TensorTrain(cores)
tt.prepare_device()
def T(f, reps=5):
    f(); nat.call("ttsk_sync", -1)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = f(); nat.call("ttsk_sync", -1); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, out
ms, stt = T(lambda: stream_sketch(tt, (l,) * (d - 1), (r,) * (d - 1), seed=1))
print(f"stream_sketch incl. DRM sampling  {ms:8.2f} ms")
left = TensorTrainDRM((l,) * (d - 1), shape, False, seed=1); right = TensorTrainDRM((r,) * (d - 1), shape, True, seed=2)
ms, stt = T(lambda: stream_sketch(tt, (l,) * (d - 1), (r,) * (d - 1), left_drm=left, right_drm=right))
print(f"stream_sketch, DRMs prebuilt      {ms:8.2f} ms")
ms, rec = T(lambda: stt.to_tt())
print(f"to_tt                             {ms:8.2f} ms   ranks {rec.rank}")
ms, o = T(lambda: orthogonal_sketch(tt, (l,) * (d - 1), (r,) * (d - 1), seed=1), reps=3)
print(f"orthogonal_sketch incl. DRMs      {ms:8.2f} ms")
ms, h = T(lambda: hmt_sketch(tt, l, seed=1), reps=3)
print(f"hmt_sketch incl. DRMs             {ms:8.2f} ms")
