import sys, time, ctypes
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray, sync
nat.call("ttsk_init", 0)
n = 1 << 30            # 8.6 GB
a = DevArray.empty((n,)); b = DevArray.empty((n,))
P = ctypes.c_void_p
def T(label, f, gb, reps=5):
    f(); sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    sync(); dt = (time.perf_counter() - t0) / reps
    print(f"{label:40s} {dt*1e3:8.3f} ms {gb/dt*1e-3:6.2f} TB/s", flush=True)
T("memset 8.6 GB (write)", lambda: nat.call("ttsk_memset", P(a.ptr), 0, ctypes.c_size_t(8 * n), 0), 8.59)
T("d2d copy 8.6 GB (read + write)", lambda: nat.call("ttsk_d2d", P(b.ptr), P(a.ptr), ctypes.c_size_t(8 * n), 0), 17.18)
T("axpby (2 reads + 1 write)", lambda: nat.call("ttsk_axpby", P(b.ptr), P(a.ptr), 1.0, 1.0, ctypes.c_size_t(n), 0), 25.8)
out = DevArray.empty((n // 64,))
T("sum_slices nb=64 (read 8.6 GB)", lambda: nat.call("ttsk_sum_slices", P(out.ptr), P(a.ptr), 64, ctypes.c_size_t(n // 64), ctypes.c_size_t(n // 64), 0, 0), 8.59 + 0.13)
out = DevArray.empty((n // 1024,))
T("sum_slices nb=1024 (read 8.6 GB)", lambda: nat.call("ttsk_sum_slices", P(out.ptr), P(a.ptr), 1024, ctypes.c_size_t(n // 1024), ctypes.c_size_t(n // 1024), 0, 0), 8.59)
