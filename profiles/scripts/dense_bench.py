import sys, time
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat
from tt_sketch_amd.utils import random_normal_dev
from tt_sketch_amd.sketch_dispatch import general_sketch_device
nat.call("ttsk_init", 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape = (n,) * 5
Xd = random_normal_dev(shape, seed=2)
T = tsa.DenseTensor(np.zeros((1,) * 5))       # placeholder host array; device payload injected
T.shape = shape; T._dev = Xd
for name, cls in (("TensorTrainDRM", tsa.TensorTrainDRM), ("DenseGaussianDRM", tsa.DenseGaussianDRM)):
    t0 = time.perf_counter()
    ld = cls(20, shape, False, seed=3); rd = cls(40, shape, True, seed=4)
    nat.call("ttsk_sync", -1)
    t_drm = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        P, O = general_sketch_device(T, ld, rd, tsa.SketchMethod.streaming)
        nat.call("ttsk_sync", -1)
        dt = time.perf_counter() - t0
        print(f"C2 dense n={n} {name}: DRM sampling {t_drm * 1e3:.0f} ms, sketch {dt * 1e3:.1f} ms "
              f"({9 * Xd.size * 8 / dt * 1e-12:.2f} TB/s over 9 passes of X, {Xd.size * 8 / 1e9:.2f} GB)")
    del ld, rd, P, O
