import sys, time, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat
from tt_sketch_amd.tt_gmres import MPO, TTLinearMapSum, tt_sum_gmres
rng = np.random.default_rng(0)
d, n, mr = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
shape = (n,) * d
def lap(k):
    cores = [np.eye(n).reshape(1, n, n, 1) for _ in range(d)]
    L = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    cores[k] = (L * (n + 1) ** 2 / (n + 1) ** 2).reshape(1, n, n, 1)
    return MPO(cores)
maps = [MPO([np.eye(n).reshape(1, n, n, 1) * (1.0 if k else 4.0) for k in range(d)])] + [lap(k) for k in range(d)]
A = TTLinearMapSum(maps)
b = tsa.TensorTrain([rng.standard_normal((1 if k == 0 else 5, n, 1 if k == d - 1 else 5)) / np.sqrt(5 * n) for k in range(d)])
for method in ("sketch", "orth_sketch", "pairwise"):
    for rep in range(2):
        t = time.perf_counter()
        x, h = tt_sum_gmres(A, b, max_rank=mr, tolerance=1e-10, maxiter=8, rounding_method=method)
        nat.call("ttsk_sync", -1)
        dt = time.perf_counter() - t
    print(method, "total %.1f ms" % (dt * 1e3), "steps(ms):", " ".join("%.1f" % (s * 1e3) for s in h["step_time"]),
          "res %.2e" % h["residual_norm"][-1], "rank", h["rank"][-1], flush=True)
