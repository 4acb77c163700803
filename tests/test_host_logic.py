"""CPU tests: host-side logic of tt_sketch_amd and the C-ABI surface (no compute calls)."""
import ctypes
import os

import numpy as np
import pytest

from tt_sketch_amd import _native as nat
from tt_sketch_amd.device import _view_strides
from tt_sketch_amd.utils import dematricize, matricize, process_tt_rank, trim_ranks


def test_library_exports_every_declared_symbol():
    lib = nat.lib()                      # raises if a symbol of include/ttsk.h is missing
    names = nat.declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n


def test_rank_processing():
    assert process_tt_rank(3, (4, 5, 6), trim=False) == (3, 3)
    assert process_tt_rank((2, 9), (4, 5, 6), trim=False) == (2, 9)
    with pytest.raises(ValueError):
        process_tt_rank((2, 3, 4), (4, 5, 6), trim=False)
    assert trim_ranks((4, 5, 6), (100, 100)) == (4, 6)
    assert trim_ranks((2, 2, 2, 2), (9, 9, 9)) == (2, 4, 2)
    assert trim_ranks((10, 10, 10), (3, 50)) == (3, 10)
    assert process_tt_rank(100, (5, 6, 7, 8), trim=True) == (5, 30, 8)


def test_matricize_roundtrip():
    A = np.arange(2 * 3 * 4 * 5.0).reshape(2, 3, 4, 5)
    for mode in range(4):
        M = matricize(A, mode)
        assert M.shape == (A.shape[mode], A.size // A.shape[mode])
        assert np.array_equal(dematricize(M, mode, A.shape), A)
    assert matricize(A, range(2), mat_shape=True).shape == (6, 20)
    assert matricize(A, range(2)).shape == (2, 3, 20)


def test_view_strides_rule():
    assert _view_strides((3, 4, 5), (20, 5, 1), (12, 5)) == (5, 1)
    assert _view_strides((3, 4, 5), (1, 3, 12), (12, 5)) is None
    assert _view_strides((1, 7, 4), (1, 1, 7), (7, 4)) == (1, 7)
    assert _view_strides((6, 5), (5, 1), (2, 3, 5)) == (15, 5, 1)


def test_drm_bookkeeping_matches_reference_conventions():
    from tt_sketch_amd.drm_base import DRM
    d = DRM((3, 4, 5), (6, 7, 8, 9), transpose=True, seed=12,
            rank_min=(1, 0, 2), rank_max=(3, 4, 5), true_rank=(3, 4, 5))
    assert d.rank_min == (2, 0, 1) and d.rank_max == (5, 4, 3) and d.rank == (3, 4, 2)
    assert d.true_rank == (5, 4, 3)
    t = d.T
    assert t.transpose is False and t.rank == (2, 4, 3) and t.rank_min == (1, 0, 2)
    assert DRM(3, (4, 5), False, seed=2**32 + 5).seed == (2**32 + 5) % (2**32 - 1)


def test_container_arithmetic_and_packing():
    from tt_sketch_amd import SketchContainer
    rng = np.random.default_rng(0)
    a = SketchContainer.zero((4, 5, 6), (2, 3), (3, 4))
    b = SketchContainer([rng.standard_normal(p.shape) for p in a.Psi_cores],
                        [rng.standard_normal(o.shape) for o in a.Omega_mats])
    c = (b + b) * 0.5 - b / 1.0
    assert all(np.allclose(x, 0) for x in c.Psi_cores + c.Omega_mats)
    assert b.left_rank == (2, 3) and b.right_rank == (3, 4) and b.shape == (4, 5, 6)
    t = b.T
    assert t.left_rank == (4, 3) and t.right_rank == (3, 2)
    buf = b.pack()
    assert buf.size == sum(x.size for x in b.Psi_cores + b.Omega_mats)
    r = a.unpack(buf)
    assert all(np.array_equal(x, y) for x, y in zip(r.Psi_cores + r.Omega_mats,
                                                     b.Psi_cores + b.Omega_mats))


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU every compute entry point must raise, never fall back."""
    import tt_sketch_amd as tsa
    lib = nat.lib()
    if lib.ttsk_init(0) == 0:
        pytest.skip("a GPU is present")
    X = tsa.TensorTrain.random((4, 5, 6), 2, seed=1)
    with pytest.raises(nat.TtskError):
        tsa.stream_sketch(X, 2, 3, seed=1)


def test_read_tns(tmp_path):
    """FROSTT text format (reference scripts/frostt.py:51-65): 1-based indices, value last."""
    import gzip
    from tt_sketch_amd.io import read_tns
    text = "# comment\n1 2 3 0.5\n4 1 2 -1.25\n\n2 2 2 3e-2\n"
    p = tmp_path / "t.tns"
    p.write_text(text)
    T = read_tns(str(p))
    assert T.shape == (4, 2, 3) and T.indices.shape == (3, 3)
    assert np.array_equal(T.indices, np.array([[0, 3, 1], [1, 0, 1], [2, 1, 1]]))
    assert np.allclose(T.entries, [0.5, -1.25, 0.03])
    g = tmp_path / "t.tns.gz"
    with gzip.open(g, "wb") as f:
        f.write(text.encode())
    T2 = read_tns(str(g), shape=(5, 5, 5))
    assert T2.shape == (5, 5, 5) and np.array_equal(T2.indices, T.indices)
    with pytest.raises(ValueError):
        read_tns(str(p), shape=(2, 2, 2))
    bad = tmp_path / "bad.tns"
    bad.write_text("1 2 3 0.5\n1 2 0.5\n")
    with pytest.raises(ValueError):
        read_tns(str(bad))
    zero = tmp_path / "zero.tns"
    zero.write_text("0 1 1 1.0\n")
    with pytest.raises(ValueError):
        read_tns(str(zero))


def test_pool_reuse_is_stream_ordered():
    """device.py hands a released buffer out again only to the same stream, or after every stream that
    may still be touching it has been drained (cross-stream reuse)."""
    from tt_sketch_amd import device
    saved = (set(nat._dirty), list(nat._stream_gen))
    saved_join = dict(nat._joined_into)
    try:
        nat._dirty.clear()
        nat._joined_into.clear()
        nat._mark("ttsk_axpby", (None, None, 1.0, 1.0, 0, 0))        # work on stream 0 only
        tag0 = nat.dirty_snapshot()
        assert set(tag0) == {0}
        assert device._reusable(tag0, 0) and not device._reusable(tag0, 3)
        nat._mark("ttsk_gemm", (None, None, None, None, None, 3))    # a multi-stream region opens
        nat._mark("ttsk_tt_sketch", (None,) * 14 + (4,))             # forks a helper on stream 5, joined back into 4
        assert nat._dirty >= {0, 3, 4, 5}
        tag = nat.dirty_snapshot()
        assert set(tag) == {0, 3, 4}                                 # the joined helper counts as its caller's stream
        assert not any(device._reusable(tag, s) for s in range(nat.NUM_STREAMS))
        nat._stream_gen[3] += 1                                      # as ttsk_sync(3) does
        nat._dirty.discard(3)
        assert not device._reusable(tag, 0)
        for s in range(nat.NUM_STREAMS):                             # as ttsk_sync(-1) does
            nat._stream_gen[s] += 1
        nat._dirty.clear()
        assert all(device._reusable(tag, s) for s in range(nat.NUM_STREAMS))
        assert device._size_class(100) == 256 and device._size_class(9 << 20) == 10 << 20
    finally:
        nat._dirty.clear()
        nat._dirty.update(saved[0])
        nat._stream_gen[:] = saved[1]
        nat._joined_into.clear()
        nat._joined_into.update(saved_join)


def test_helper_stream_of_a_one_call_sketch_is_drained_with_its_caller():
    """ttsk_tt_sketch* fork stream + 1 and join it back; a blocking read / sync of the CALLER's stream must
    release buffers tagged with the helper too (they used to wait for a device-wide sync), but not if something else
    was queued on the helper directly."""
    from tt_sketch_amd import _native as nat
    saved = (set(nat._dirty), list(nat._stream_gen), dict(nat._joined_into))
    try:
        nat._dirty.clear(); nat._joined_into.clear()
        nat._mark("ttsk_tt_sketch_batch", (1, 2, 3, 0))          # last argument: stream 0 -> helper 1
        assert nat._dirty == {0, 1}
        tag = nat.dirty_snapshot()
        nat._drained(0)                                          # what ttsk_d2h(..., 0) / ttsk_sync(0) do
        assert nat._dirty == set() and all(nat.drained_since(s, g) for s, g in tag.items())
        nat._mark("ttsk_tt_sketch", (0,))
        nat._mark("ttsk_gemm", (1,))                             # independent work on the helper stream
        tag = nat.dirty_snapshot()
        nat._drained(0)
        assert nat._dirty == {1} and not nat.drained_since(1, tag[1])
    finally:
        nat._dirty.clear(); nat._dirty.update(saved[0])
        nat._stream_gen[:] = saved[1]
        nat._joined_into.clear(); nat._joined_into.update(saved[2])


def _rdv_worker(rank, world, directory, out):
    from tt_sketch_amd.rendezvous import FileRendezvous
    r = FileRendezvous(rank, world, directory=directory, timeout=30)
    got = r.broadcast(b"\x01" * 128 if rank == 0 else None)
    parts = r.allgather(bytes([rank]) * 3)
    r.barrier()
    r.close()
    out.put((rank, got == b"\x01" * 128, parts == [bytes([k]) * 3 for k in range(world)]))


def test_file_rendezvous_carries_the_id_between_processes(tmp_path):
    """The torch-free exchange of the 128-byte RCCL id (bench.py, RcclComm.from_env)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdv_worker, args=(r, 3, str(tmp_path / "rdv"), q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True), (2, True, True)]


def test_file_rendezvous_ignores_the_leftovers_of_a_crashed_launch(tmp_path):
    """a directory that still holds an earlier launch's files (crashed before close(), or a fixed
    TTSK_RDV_DIR): the new launch must neither read the stale NCCL id nor a stale gather piece, and the directory is
    gone after a clean close."""
    import multiprocessing as mp
    import os
    d = tmp_path / "rdv"
    d.mkdir()
    # what a launch of the old naming scheme and a crashed launch of the new one leave behind
    (d / "000001_bcast.0").write_bytes(b"\x07" * 128)
    (d / "000002_gather.1").write_bytes(b"stale")
    (d / "hello.0").write_bytes(b"deadbeefdeadbeef")
    (d / "hello.1").write_bytes(b"0123456789abcdef")
    (d / "hello.2").write_bytes(b"fedcba9876543210")
    (d / "session").write_bytes(b"aaaaaaaaaaaaaaaa deadbeefdeadbeef 0123456789abcdef fedcba9876543210")
    (d / "ack.1").write_bytes(b"aaaaaaaaaaaaaaaa deadbeefdeadbeef 0123456789abcdef fedcba9876543210")
    (d / "aaaaaaaaaaaaaaaa_000001_bcast.0").write_bytes(b"\x09" * 128)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdv_worker, args=(r, 3, str(d), q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True), (2, True, True)]
    assert not os.path.exists(d)              # the last rank out removed it, leftovers included


def test_failed_comm_init_leaves_the_process_exit_clean():
    """ttsk_comm_init with a bad rank / without a device returns an error code, leaves no
    communicator behind (destroy is a no-op, init can be called again) and the interpreter exits 0
    (round 1: the only N = 2 run died in `double free or corruption` after a failed init)."""
    import subprocess
    import sys
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
from tt_sketch_amd import _native as nat
lib = nat.lib()
uid = (ctypes.c_char * 128)()
rcs = [lib.ttsk_comm_init(uid, 5, 2), lib.ttsk_comm_init(uid, -1, 2), lib.ttsk_comm_init(None, 0, 1)]
assert all(rc != 0 for rc in rcs), rcs
assert lib.ttsk_comm_destroy() == 0
assert lib.ttsk_comm_allreduce_sum(None, 4, 0) != 0          # no communicator: an error, not a crash
assert lib.ttsk_comm_init(uid, 7, 3) != 0 and lib.ttsk_comm_destroy() == 0
print("clean")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
    assert "clean" in out.stdout


def test_test_tensor_generators():
    """hilbert_tensor / sqrt_tensor (reference utils.py:20-39) from their definitions, incl. the swapped first two
    modes of sqrt_tensor (np.meshgrid's 'xy' default in the reference)."""
    from tt_sketch_amd import utils
    H = utils.hilbert_tensor(3, 4)
    i, j, k = np.ogrid[:4, :4, :4]
    assert H.shape == (4, 4, 4) and np.array_equal(H, 1.0 / (i + j + k + 1.0))
    S = utils.sqrt_tensor((3, 4, 5), a=0.5, b=3)
    assert S.shape == (4, 3, 5) and abs(np.linalg.norm(S) - 1) < 1e-14
    t = [np.linspace(0.5, 3, n) for n in (4, 3, 5)]
    want = np.sqrt(np.abs(t[0][:, None, None] + t[1][None, :, None] + t[2][None, None, :]))
    assert np.allclose(S, want / np.linalg.norm(want), rtol=1e-15, atol=0)
    assert utils.sqrt_tensor((7,)).shape == (7,)


def test_public_names_of_the_reference_are_present():
    """The reference's public names on and around the path (modules sketch, sketch_dispatch, drm, drm_base, tensor,
    utils, sketch_container, tt_svd, tt_gmres; listed from its sources) exist under the same module paths."""
    import importlib
    expected = {
        "sketch": ["stream_sketch", "orthogonal_sketch", "hmt_sketch", "blocked_stream_sketch", "assemble_sketched_tt",
                   "SketchedTensorTrain"],
        "sketch_dispatch": ["general_sketch", "SketchMethod", "get_sketch_method", "orth_step", "OrthogTTDRM", "sum_sketch",
                            "sketch_omega_sum", "sketch_psi_sum", "OMEGA_METHODS", "PSI_METHODS",
                            "DRM_SKETCH_METHOD_DISPATCH"],
        "sketch_container": ["SketchContainer"],
        "drm": ["ALL_DRM", "DenseGaussianDRM", "SparseGaussianDRM", "SparseSignDRM", "TensorTrainDRM"],
        "drm.fast_lazy_gaussian": ["hash_int_c", "_inds_to_rand_double", "inds_to_normal", "inds_to_sparse_sign"],
        "drm_base": ["DRM", "CanSlice", "CanIncreaseRank", "handle_transpose"],
        "tensor": ["Tensor", "DenseTensor", "SparseTensor", "TensorTrain", "CPTensor", "TuckerTensor", "TensorSum"],
        "utils": ["hilbert_tensor", "sqrt_tensor", "power_decay_tensor", "matricize", "dematricize", "right_mul_pinv",
                  "left_mul_pinv", "projector", "trim_ranks", "process_tt_rank", "random_normal"],
        "tt_svd": ["tt_svd"],
        "tt_gmres": ["TTLinearMap", "MPO", "TTPrecond", "TTLinearMapSum", "round_tt_sum", "tt_sum_gmres"],
        "sketching_methods.abstract_methods": ["CansketchSparse", "CansketchTT", "CansketchCP", "CansketchDense",
                                               "CanSketchTucker"],
    }
    for mod, names in expected.items():
        m = importlib.import_module("tt_sketch_amd." + mod)
        missing = [n for n in names if not hasattr(m, n)]
        assert not missing, (mod, missing)
    from tt_sketch_amd.sketch import SketchedTensorTrain
    for attr in ("left_rank", "right_rank", "Psi_cores", "Omega_mats", "C_cores", "T", "to_tt", "to_numpy", "increase_rank",
                 "error", "dense"):
        assert hasattr(SketchedTensorTrain, attr), attr


def test_shard_bounds_cover_everything():
    from tt_sketch_amd.distributed import shard_bounds
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sparse_one_pass_path_declines_a_mode_beyond_its_sort_key():
    """The mode-order sort key of the one-pass sparse path holds the mode index in 24 bits (ttsk_sparse_mode_order): a
    longer mode is declined BEFORE anything touches the device, so that the panel path takes over."""
    from tt_sketch_amd import SparseGaussianDRM, SparseTensor, sparse_fused
    from tt_sketch_amd.sketch_dispatch import SketchMethod
    shape = (5, (1 << 24) + 1, 4)
    idx = np.array([[0, 1], [3, (1 << 24)], [2, 1]])
    X = SparseTensor(shape, idx, np.array([1.0, -2.0]))
    left = SparseGaussianDRM((2, 2), shape, transpose=False, seed=1)
    right = SparseGaussianDRM((3, 3), shape, transpose=True, seed=2)
    assert sparse_fused.try_sparse_gauss_sketch(X, left, right, SketchMethod.streaming) is None
