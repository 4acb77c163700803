import sys, ctypes, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import DevArray, contract
nat.call("ttsk_init", 0)
rng = np.random.default_rng(0)
def bench(name, spec, sa, sb, reps=50):
    a = rng.standard_normal(sa); b = rng.standard_normal(sb)
    A = DevArray.from_host(a); B = DevArray.from_host(b)
    out = contract(spec, A, B)
    nat.call("ttsk_sync", -1)
    ref = np.einsum(spec, a, b)
    err = np.abs(out.get() - ref).max() / np.abs(ref).max()
    g = ctypes.c_void_p()
    nat.call("ttsk_graph_begin", 0)
    for _ in range(reps):
        contract(spec, A, B, out=out)
    nat.call("ttsk_graph_end", 0, ctypes.byref(g))
    nat.call("ttsk_graph_launch", g, 0); nat.call("ttsk_sync", -1)
    t0 = time.perf_counter()
    nat.call("ttsk_graph_launch", g, 0); nat.call("ttsk_sync", -1)
    us = (time.perf_counter() - t0) / reps * 1e6
    lhs, o = spec.split("->"); ia, ib = lhs.split(",")
    size = {}
    for idx, shp in ((ia, sa), (ib, sb)):
        for c, n in zip(idx, shp): size[c] = n
    flops = 2.0 * np.prod([size[c] for c in set(ia + ib)])
    print(f"{name:12s} {spec:14s} {us:8.1f} us/launch  {flops / us * 1e-6:7.2f} TF/s  err {err:.1e}", flush=True)
bench("gemm1_R", "pq,np->qn", (100, 100), (20000, 100))
bench("gemm1_L", "pq,pn->qn", (100, 50), (100, 20000))
bench("psi", "mk,kc->mc", (10000, 100), (100, 100))
bench("gemm2_R", "qkp,qkm->pm", (100, 200, 100), (100, 200, 100))
bench("gemm2_L", "kp,kq->pq", (10000, 100), (10000, 50))
bench("g1R_odd", "pq,np->qn", (37, 53), (5001, 37))
bench("g1R_big", "pq,np->qn", (128, 128), (100000, 128))
print("--- again")
bench("gemm1_R", "pq,np->qn", (100, 100), (20000, 100))
bench("gemm1_R", "pq,np->qn", (100, 100), (20000, 100), reps=500)
bench("gemm2_R", "qkp,qkm->pm", (100, 200, 100), (100, 200, 100), reps=500)
