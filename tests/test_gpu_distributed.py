"""GPU tests of the multi-GPU entry points on the one GPU a test box has: a communicator of one
rank exercises every RCCL call of the path (all-reduce, all-gather, max, barrier) through the same
functions the 8-GPU launch uses; two processes on ONE GPU exercise the failure path of
ttsk_comm_init (RCCL refuses the duplicate device) and must still exit cleanly.  Each case runs in a
child process under a time limit: communicator set-up loads RCCL's kernels (~6 s) and must not be able
to stall the suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env=None, timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)


SHARDED = r"""
import numpy as np, sys
sys.path.insert(0, %(root)r)
from oracle import ttsk_oracle as orc
import tt_sketch_amd as tsa
from tt_sketch_amd.distributed import (RcclComm, blocked_stream_sketch_sharded, stream_sketch_sharded)
comm = RcclComm.from_env()
assert (comm.rank, comm.world) == (0, 1)
rng = np.random.default_rng(5)
shape, s, l, r = (12, 9, 11, 10), 4, (5, 6, 5), (8, 9, 8)
terms = [orc.random_tt(shape, s, rng) for _ in range(6)]
ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
left = tsa.TensorTrainDRM(l, shape, False, seed=1, cores=ld.cores)
right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=rd.cores)
whole = tsa.TensorSum([tsa.TensorTrain(c) for c in terms])
stt = stream_sketch_sharded(whole, l, r, comm, left_drm=left, right_drm=right)
oP, oO = orc.general_sketch("sum", [("tt", c) for c in terms], ld, rd, "streaming")
err = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(stt.Psi_cores + stt.Omega_mats, oP + oO))
assert err < 1e-12, err
# seed-built DRMs: the same call twice gives the same sketch (what makes the ranks agree)
idx = np.stack([rng.integers(0, n, 5000) for n in shape]).astype(np.int64)
sp = tsa.SparseTensor(shape, idx, rng.standard_normal(5000))
a = stream_sketch_sharded(sp, l, r, comm, seed=33, left_drm_type=tsa.SparseGaussianDRM)
b = stream_sketch_sharded(sp, l, r, comm, seed=33, left_drm_type=tsa.SparseGaussianDRM)
# (hash-Gaussian DRMs on both sides: the one-pass sparse path sums in a fixed order, no atomics -- bit for bit)
assert all(np.array_equal(x, y) for x, y in zip(a.Psi_cores + a.Omega_mats, b.Psi_cores + b.Omega_mats))
old = orc.HashGaussDrm(a.left_drm.seed, shape, False, (0,) * 3, l)
ord_ = orc.HashGaussDrm(a.right_drm.seed, shape, True, (0,) * 3, r[::-1])
oP, oO = orc.general_sketch("sparse", (shape, idx, np.asarray(sp.entries)), old, ord_, "streaming")
err = max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(a.Psi_cores + a.Omega_mats, oP + oO))
assert err < 1e-11, err
# rank-sharded placement through the all-gather: equals the unblocked sketch (reference tests :137-187)
hl = tsa.SparseGaussianDRM((4, 5, 4), shape, False, seed=21)
hr = tsa.SparseGaussianDRM((6, 7, 6), shape, True, seed=22)
blk = blocked_stream_sketch_sharded(sp, hl, hr, [(0, 0, 0), (2, 2, 1), (4, 5, 4)], [(0, 0, 0), (3, 4, 2), (6, 7, 6)], comm)
ref = tsa.general_sketch(sp, hl, hr, tsa.SketchMethod.streaming)
err = max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(blk.Psi_cores + blk.Omega_mats, ref.Psi_cores + ref.Omega_mats))
assert err < 1e-12, err
assert comm.max_over_ranks(3.5) == 3.5
comm.barrier()
comm.close()
print("sharded ok")
"""


def test_sharded_entry_points_with_one_rank(tmp_path):
    try:
        res = _run(SHARDED % dict(root=ROOT), env=dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TTSK_RDV_DIR=str(tmp_path / "rdv")))
    except subprocess.TimeoutExpired as exc:        # a hang is a failure, not a skip (subprocess.run has killed and reaped the child)
        pytest.fail(f"RCCL communicator set-up did not return within the time limit: stderr {(exc.stderr or b'')[-3000:]!r}")
    assert res.returncode == 0 and "sharded ok" in res.stdout, (res.stdout[-500:], res.stderr[-3000:])


FAIL_RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
from tt_sketch_amd import _native as nat
from tt_sketch_amd.distributed import RcclComm
try:
    RcclComm.from_env(device=0)          # both ranks on GPU 0: ncclCommInitRank refuses the duplicate device
    print("init unexpectedly succeeded")
except nat.TtskError as e:
    print("init failed:", str(e)[:80])
from tt_sketch_amd.device import DevArray
assert float(DevArray.zeros((1000,)).get().sum()) == 0.0      # the device is still usable afterwards
assert nat.lib().ttsk_comm_destroy() == 0                      # nothing half-built is left to destroy
print("rank done")
"""


def test_failed_rccl_init_exits_cleanly(tmp_path):
    """Round 1's `double free or corruption` after a failed ncclCommInitRank: reproduced set-up (two
    ranks, one GPU), minus torch in the process.  Both ranks must report the error and exit 0."""
    env = dict(os.environ, WORLD_SIZE="2", TTSK_RDV_DIR=str(tmp_path / "rdv"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", FAIL_RANK % dict(root=ROOT)], cwd=ROOT, text=True,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    outs, hung = [], None
    try:
        for p in procs:
            try:
                outs.append(p.communicate(timeout=240))
            except subprocess.TimeoutExpired:
                hung = p
                break
    finally:
        for p in procs:                      # nobody keeps the GPU: kill and reap every child, hung or not
            if p.poll() is None:
                p.kill()
        tails = [p.communicate()[1][-2000:] if p is hung or p.returncode is None or len(outs) <= i else "" for i, p in enumerate(procs)]
    if hung is not None:
        pytest.fail(f"a rank did not return from the failed communicator set-up within the time limit; stderr tails: {tails}")
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (p.returncode, so[-500:], se[-2000:])
        assert "rank done" in so and "init failed" in so, (so[-500:], se[-2000:])


def test_pool_does_not_recycle_across_streams():
    """A temporary released while other streams have work in flight is not handed out until they have
    drained (device.py); single-stream use recycles at once."""
    import numpy as np
    from tt_sketch_amd import _native as nat
    from tt_sketch_amd.device import DevArray, contract, sync
    nat.call("ttsk_init", 0)
    sync()
    a = DevArray.zeros((1000,))
    p0 = a.buf.ptr
    del a
    b = DevArray.empty((1000,))
    assert b.buf.ptr == p0                      # stream 0 -> stream 0: immediate reuse
    del b
    rng = np.random.default_rng(0)
    A, B = DevArray.from_host(rng.standard_normal((300, 200))), DevArray.from_host(rng.standard_normal((200, 100)))
    C3 = contract("ij,jk->ik", A, B, stream=3)  # stream 3 now has work in flight
    t = DevArray.empty((1000,))                 # (taken for stream 0 while only streams {3} U {0} are dirty)
    pt = t.buf.ptr
    del t                                       # released with tag {0, 3}
    u = DevArray.empty((1000,))
    assert u.buf.ptr != pt                      # stream 3 has not drained: not recycled
    sync()
    del u
    v = DevArray.empty((1000,))
    w = DevArray.empty((1000,))
    assert pt in (v.buf.ptr, w.buf.ptr)         # after the sync both are back
    assert np.allclose(C3.get(), A.get() @ B.get())


TWO_RANKS = r"""
import numpy as np, os, sys
sys.path.insert(0, %(root)r)
import tt_sketch_amd as tsa
from tt_sketch_amd.distributed import HostComm, blocked_stream_sketch_sharded, stream_sketch_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
comm = HostComm.over_files(rank, world, os.environ["TTSK_RDV_DIR"])
rng = np.random.default_rng(11)                          # same data on every rank
shape, l, r = (30, 24, 28, 26), (6, 7, 6), (9, 10, 9)
terms = [tsa.TensorTrain.random(shape, 5, seed=40 + i) for i in range(7)]
whole = tsa.TensorSum(terms)
left = tsa.TensorTrainDRM(l, shape, False, seed=1)
right = tsa.TensorTrainDRM(r, shape, True, seed=2)
stt = stream_sketch_sharded(whole, l, r, comm, left_drm=left, right_drm=right)      # 4 + 3 terms, HIP path on both
ref = tsa.general_sketch(whole, left, right, tsa.SketchMethod.streaming)
err = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(stt.Psi_cores + stt.Omega_mats, ref.Psi_cores + ref.Omega_mats))
assert err < 1e-12, err
idx = np.stack([rng.integers(0, n, 4000) for n in shape]).astype(np.int64)
sp = tsa.SparseTensor(shape, idx, rng.standard_normal(4000))
a = stream_sketch_sharded(sp, l, r, comm, seed=33, left_drm_type=tsa.SparseGaussianDRM)   # nnz shards
b = tsa.stream_sketch(sp, l, r, seed=33, left_drm_type=tsa.SparseGaussianDRM)
err = max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(a.Psi_cores + a.Omega_mats, b.Psi_cores + b.Omega_mats))
assert err < 1e-12, err
hl = tsa.SparseGaussianDRM((4, 5, 4), shape, False, seed=21)
hr = tsa.SparseGaussianDRM((6, 7, 6), shape, True, seed=22)
blk = blocked_stream_sketch_sharded(sp, hl, hr, [(0, 0, 0), (2, 2, 1), (4, 5, 4)], [(0, 0, 0), (3, 4, 2), (6, 7, 6)], comm)
one = tsa.general_sketch(sp, hl, hr, tsa.SketchMethod.streaming)
err = max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(blk.Psi_cores + blk.Omega_mats, one.Psi_cores + one.Omega_mats))
assert err < 1e-12, err
comm.close()
print("rank", rank, "ok")
"""


def test_two_processes_share_the_work(tmp_path):
    """Two ranks, each its own process with the HIP path on the one GPU of the box, the collectives through files
    (RCCL refuses two ranks on one device): the sharded entry points give the single-process sketch."""
    env = dict(os.environ, WORLD_SIZE="2", TTSK_RDV_DIR=str(tmp_path / "rdv"))
    procs = [subprocess.Popen([sys.executable, "-c", TWO_RANKS % dict(root=ROOT)], cwd=ROOT, text=True,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, RANK=str(r)))
             for r in range(2)]
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish")
        assert p.returncode == 0 and f"rank {r} ok" in so, (so[-300:], se[-2000:])


def _visible_gpus():
    """device count from a child process (hipGetDeviceCount does not initialise a context, but keep pytest's own
    process out of it anyway)"""
    code = ("import ctypes; lib = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0); "
            "rc = lib.hipGetDeviceCount(ctypes.byref(n)); print(n.value if rc == 0 else 0)")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def test_two_gpus_give_the_one_gpu_sketch():
    """RCCL with N > 1: `bench.py --gpus 2 --scaling strong` under the driver's launcher on the first
    box that shows two devices; the sketch of the whole job after the all-reduce must equal the N = 1 run's at 1e-12.
    Skipped on one-GPU boxes (where this suite has run so far).  The launcher starts the ranks before anything touches
    a GPU; nothing re-execs."""
    import json
    if _visible_gpus() < 2:
        pytest.skip("needs two visible GPUs")
    common = ["--scaling", "strong", "--items", "16", "--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu", "--check"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    port = 29500 + os.getpid() % 2000
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, (two.stdout[-1000:], two.stderr[-3000:])
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1
    ca, cb = a["sketch_check"], b["sketch_check"]
    assert abs(ca["norm"] - cb["norm"]) <= 1e-12 * ca["norm"]
    assert abs(ca["probe"] - cb["probe"]) <= 1e-11 * ca["norm"] * (a["config"]["sketch_bytes"] / 8) ** 0.5
    for x, y in zip(ca["head"], cb["head"]):
        assert abs(x - y) <= 1e-12 * max(abs(x), 1e-300) + 1e-12 * ca["norm"] * 1e-3
    # the same for C5 -- the configuration north_star spreads term-per-GPU: 32 terms dealt over the ranks, one reduce
    c5 = ["--config", "c5", "--steps", "2", "--warmup", "1", "--no-cpu", "--check"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + c5, capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port + 1), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + c5,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, (two.stdout[-1000:], two.stderr[-3000:])
    ca = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])["sketch_check"]
    cb = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])["sketch_check"]
    assert abs(ca["norm"] - cb["norm"]) <= 1e-12 * ca["norm"]
    assert abs(ca["probe"] - cb["probe"]) <= 1e-11 * ca["norm"] * ca["size"] ** 0.5


def test_strong_scaling_check_runs_on_one_gpu():
    """the --check leg itself (local sum, no communicator) and, with the collective path forced on one rank, the
    all-reduce: both must give the same numbers"""
    import json
    common = ["--gpus", "1", "--scaling", "strong", "--items", "8", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu", "--check"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = []
    for force, coll in ((False, "reduce"), (True, "reduce"), (True, "allreduce")):
        e = dict(env, TTSK_BENCH_FORCE_COMM="1") if force else env
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--collective", coll], capture_output=True,
                                 text=True, timeout=300, env=e, cwd=ROOT)
        except subprocess.TimeoutExpired as exc:
            pytest.fail(f"bench.py --check (force_comm={force}, {coll}) did not finish: {exc}")
        assert res.returncode == 0, res.stderr[-3000:]
        outs.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])["sketch_check"])
    for o in outs[1:]:
        assert abs(outs[0]["norm"] - o["norm"]) <= 1e-13 * outs[0]["norm"]
        assert outs[0]["head"] == o["head"]
    # C5 with the collective path forced on one rank: the reduce leaves the sum of all 32 terms on rank 0
    c5 = ["--gpus", "1", "--config", "c5", "--steps", "2", "--warmup", "1", "--no-cpu", "--check"]
    c5o = []
    for force in (False, True):
        e = dict(env, TTSK_BENCH_FORCE_COMM="1") if force else env
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + c5, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        c5o.append(json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])["sketch_check"])
    assert abs(c5o[0]["norm"] - c5o[1]["norm"]) <= 1e-13 * c5o[0]["norm"]
