import sys, os, time, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
for m, n in ((16, 16), (32, 32), (64, 64), (100, 100), (128, 128), (80, 40), (200, 50), (200, 100)):
    A = rng.standard_normal((m, n))
    d = DevArray.from_host(A)
    US, S, Vt = DevArray.empty((m, n)), DevArray.empty((n,)), DevArray.empty((n, n))
    f = lambda: nat.call("ttsk_svd_small", ctypes.c_void_p(d.ptr), m, n, ctypes.c_void_p(US.ptr), ctypes.c_void_p(S.ptr), ctypes.c_void_p(Vt.ptr), 0)
    f(); nat.call("ttsk_sync", -1)
    t = time.perf_counter()
    for _ in range(5): f()
    nat.call("ttsk_sync", -1)
    dt = (time.perf_counter() - t) / 5
    err = np.abs(US.get() @ Vt.get() - A).max()
    rounds = (n + (n & 1) - 1)
    print(f"{m}x{n}: {dt*1e3:7.3f} ms   {dt*1e6/rounds:7.2f} us per (round x sweeps)  err {err:.1e}", flush=True)
