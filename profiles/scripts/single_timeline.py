import sys, os, time, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd import TensorTrain, TensorTrainDRM
from tt_sketch_amd.tt_fused import TTSketchPlan
from tt_sketch_amd.device import sync
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
d, n, s, l, r = 6, 200, 100, 50, 100
shape = (n,) * d
ranks = (1,) + (s,) * (d - 1) + (1,)
cores = [rng.standard_normal((ranks[i], n, ranks[i + 1])) / np.sqrt(ranks[i] * n) for i in range(d)]
tt = TensorTrain(cores); tt.prepare_device()
left = TensorTrainDRM((l,) * (d - 1), shape, False, seed=1); right = TensorTrainDRM((r,) * (d - 1), shape, True, seed=2)
plan = TTSketchPlan(tt.shape, tt.rank, left, right)
ptrs, keep = plan.core_pointers(tt)
out = plan.new_buffer()
for _ in range(20):
    plan.run(ptrs, out); sync()
