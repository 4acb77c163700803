import sys, time
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat
nat.call("ttsk_init", 0)
shape = (200, 150, 100, 120, 300)
rng = np.random.default_rng(4)
nnz = 10_000_000
idx = np.stack([rng.integers(0, n, nnz) for n in shape]).astype(np.int64)
val = rng.standard_normal(nnz)
T = tsa.SparseTensor(shape, idx, val)
T.prepare_device()
ld = tsa.SparseGaussianDRM(10, shape, False, seed=3)
rd = tsa.SparseGaussianDRM(15, shape, True, seed=4)
from tt_sketch_amd.sketch_dispatch import general_sketch_device
for rep in range(3):
    nat.call("ttsk_sync", -1)
    t0 = time.perf_counter()
    P, O = general_sketch_device(T, ld, rd, tsa.SketchMethod.streaming)
    nat.call("ttsk_sync", -1)
    print(f"C4 nnz=1e7 sketch (device resident): {(time.perf_counter() - t0) * 1e3:.1f} ms")
