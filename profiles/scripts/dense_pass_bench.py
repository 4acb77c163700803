"""ttsk_dense_first_pass alone at the C2 shape (64 x 64^3 x 64, ranks 20 / 40): ms per launch pair (kernel + sum of the partial U),
fraction of the fp64 matrix peak, TB/s of its 11.27 GB.  Usage: python profiles/scripts/dense_pass_bench.py [reps] [last mode size]
with TTSK_DP_DBG=1/2/3/8 for the timing experiments of DESIGN.md section 6 (no loads / no stores / neither / loads as a burst)."""
import sys, time, ctypes, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray, sync
from tt_sketch_amd.utils import random_normal_dev
nat.call("ttsk_init", 0)
n0, Q, T, ll, r = 64, 64 ** 3, 64, 20, 40
if len(sys.argv) > 2:
    T = int(sys.argv[2]); Q = 64 ** 4 // T
X = random_normal_dev((n0, Q, T), seed=2)
C = random_normal_dev((n0, ll), seed=3)
P = random_normal_dev((Q, r), seed=4)
Z, U = DevArray.empty((ll, Q, T)), DevArray.empty((n0, r, T))
V = ctypes.c_void_p
def f():
    nat.call("ttsk_dense_first_pass", V(X.ptr), n0, Q, T, V(C.ptr), ll, V(P.ptr), r, V(Z.ptr), V(U.ptr), 0)
f(); sync()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(reps): f()
sync()
dt = (time.perf_counter() - t0) / reps
print("first pass %.3f ms  %.1f TF/s (%.2f of 78.6)  %.2f TB/s" % (dt * 1e3, 129.0e9 / dt * 1e-12, 129.0e9 / dt / 78.6e12, 11.27e9 / dt * 1e-12))
