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
# (the sparse Psi flushes run sums with fp64 atomics: equal to rounding, not bit for bit)
assert all(np.linalg.norm(x - y) <= 1e-13 * np.linalg.norm(y) for x, y in zip(a.Psi_cores + a.Omega_mats, b.Psi_cores + b.Omega_mats))
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
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL communicator set-up did not return within the time limit on this box")
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
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            p.kill()
            pytest.skip("RCCL communicator set-up did not return within the time limit on this box")
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
