"""World-size-2 `gloo` test of the multi-GPU path's host logic (no GPU): every rank sketches its
shard with the CPU oracle, packs, one all_reduce, and the result equals the sketch of the whole
input (linearity, reference tests/test_sketching_matrix.py:410-419,602-631)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_comm(dist, rank, world):
    """HostComm over a gloo group: the host all-reduce / all-gather the package is handed in CPU tests."""
    import torch
    from tt_sketch_amd.distributed import HostComm

    def allreduce(buf):
        t = torch.from_numpy(np.array(buf, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def allgather(buf):
        t = torch.from_numpy(np.array(buf, dtype=np.float64))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    def reduce(buf, root):
        t = torch.from_numpy(np.array(buf, dtype=np.float64))
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        return t.numpy() if rank == root else np.array(buf, dtype=np.float64)     # (gloo leaves partial sums on the others)
    return HostComm(rank, world, allreduce, allgather, reduce)


def _oracle_drm(d):
    from oracle import ttsk_oracle as orc
    from tt_sketch_amd import SparseGaussianDRM, TensorTrainDRM
    if isinstance(d, SparseGaussianDRM):
        return orc.HashGaussDrm(d.seed, d.shape, d.transpose, tuple(d.rank_min), tuple(d.rank_max))
    if isinstance(d, TensorTrainDRM):
        return orc.TTDrm([np.asarray(c) for c in d.cores], d.shape, d.transpose, tuple(d.rank_min), tuple(d.rank_max))
    raise TypeError(type(d))


def _oracle_data(t):
    from tt_sketch_amd import SparseTensor, TensorSum, TensorTrain
    if isinstance(t, TensorTrain):
        return "tt", [np.asarray(c) for c in t.cores]
    if isinstance(t, SparseTensor):
        return "sparse", (t.shape, np.asarray(t.indices), np.asarray(t.entries))
    if isinstance(t, TensorSum):
        return "sum", [_oracle_data(x) for x in t.tensors]
    raise TypeError(type(t))


def _oracle_sketch(tensor, left_drm, right_drm):
    """sketch_fn of the sharded entry points: the CPU oracle instead of the HIP path."""
    from oracle import ttsk_oracle as orc
    from tt_sketch_amd import SketchContainer
    kind, data = _oracle_data(tensor)
    P, O = orc.general_sketch(kind, data, _oracle_drm(left_drm), _oracle_drm(right_drm), "streaming")
    return SketchContainer(P, O)


def _maxrel(a, b):
    return max(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300)
               for x, y in zip(a.Psi_cores + a.Omega_mats, b.Psi_cores + b.Omega_mats))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import ttsk_oracle as orc
    from tt_sketch_amd import (SketchContainer, SparseGaussianDRM, SparseTensor, TensorSum, TensorTrain,
                               TensorTrainDRM)
    from tt_sketch_amd.distributed import (blocked_stream_sketch_sharded, shard_bounds, shard_tensor,
                                           stream_sketch_sharded)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = _gloo_comm(dist, rank, world)
    rng = np.random.default_rng(7)                      # identical on all ranks
    shape, s, l, r = (6, 7, 5, 8), 3, (3, 4, 3), (5, 6, 5)
    terms = [orc.random_tt(shape, s, rng) for _ in range(5)]
    ld = orc.random_tt_drm(shape, l, False, rng)
    rd = orc.random_tt_drm(shape, r, True, rng)
    idx = np.stack([rng.integers(0, n, 101) for n in shape])
    val = rng.standard_normal(101)
    left = TensorTrainDRM(l, shape, False, seed=1, cores=ld.cores)
    right = TensorTrainDRM(r, shape, True, seed=2, cores=rd.cores)

    results = {}
    # (1) TensorSum of TTs: summands dealt to ranks, ONE all-reduce of the packed sketch, through the
    # same entry point the GPU ranks call (the oracle stands in for the HIP sketch)
    whole = TensorSum([TensorTrain(c) for c in terms])
    lo, hi = shard_bounds(len(terms), rank, world)
    assert shard_tensor(whole, rank, world).num_summands == hi - lo
    stt = stream_sketch_sharded(whole, l, r, comm, left_drm=left, right_drm=right, sketch_fn=_oracle_sketch)
    ref = _oracle_sketch(whole, left, right)
    results["tt_sum"] = _maxrel(stt.sketch_, ref)
    # (2) nnz shards of a sparse tensor, hash DRMs built from the seed on every rank
    sp = SparseTensor(shape, idx, val)
    stt = stream_sketch_sharded(sp, l, r, comm, seed=11, left_drm_type=SparseGaussianDRM, sketch_fn=_oracle_sketch)
    ref = _oracle_sketch(sp, stt.left_drm, stt.right_drm)
    results["sparse"] = _maxrel(stt.sketch_, ref)
    # (3) more ranks than summands: the empty share contributes zeros
    one = TensorSum([TensorTrain(terms[0])])
    stt = stream_sketch_sharded(one, l, r, comm, left_drm=left, right_drm=right, sketch_fn=_oracle_sketch)
    results["one_term"] = _maxrel(stt.sketch_, _oracle_sketch(one, left, right))
    # (4) rank-sharded placement: blocks of DRM rank slices dealt to the ranks, ONE all-gather, no sum
    hl = SparseGaussianDRM((4, 5, 4), shape, False, seed=21)
    hr = SparseGaussianDRM((6, 7, 6), shape, True, seed=22)
    lsl = [(0, 0, 0), (2, 2, 1), (4, 5, 4)]
    rsl = [(0, 0, 0), (3, 4, 2), (5, 5, 5), (6, 7, 6)]
    blk = blocked_stream_sketch_sharded(sp, hl, hr, lsl, rsl, comm, sketch_fn=_oracle_sketch)
    from tt_sketch_amd.distributed import HostComm
    alone = HostComm(0, 1, lambda b: b, lambda b: [b])          # the same blocks computed by one process
    ref = blocked_stream_sketch_sharded(sp, hl, hr, lsl, rsl, alone, sketch_fn=_oracle_sketch)
    results["blocked"] = max(float(np.max(np.abs(a - b))) for a, b in
                             zip(blk.Psi_cores + blk.Omega_mats, ref.Psi_cores + ref.Omega_mats))   # placement: exact
    results["blocked_vs_whole"] = _maxrel(blk, _oracle_sketch(sp, hl, hr))     # blocked == unblocked (reference tests :137-187)
    # (5) the single REDUCE to the rank that assembles (north_star: "a single RCCL reduce"): rank 1 holds the sketch of the
    # whole sum, rank 0 keeps its own partial sketch
    stt = stream_sketch_sharded(whole, l, r, comm, left_drm=left, right_drm=right, sketch_fn=_oracle_sketch, root=1)
    mine = shard_tensor(whole, rank, world)
    want = _oracle_sketch(whole, left, right) if rank == 1 else _oracle_sketch(mine, left, right)
    results["reduce_root"] = _maxrel(stt.sketch_, want)
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.array([results[k] for k in ("tt_sum", "sparse", "one_term", "blocked_vs_whole", "reduce_root", "blocked")]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partial_sketch_sum(tmp_path):
    """Two gloo ranks, launched from a CHILD interpreter: torch brings its own bundled ROCm runtime, and a process
    that holds both that and libttsk.so (the rest of this suite loads it) aborts at interpreter exit
    -- the pytest process itself never imports torch."""
    import subprocess
    import sys
    import __graft_entry__ as ge
    ge.build_oracle()
    port = _free_port()
    res = subprocess.run([sys.executable, os.path.abspath(__file__), str(port), str(tmp_path)], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    for rank in range(2):
        errs = np.load(tmp_path / f"r{rank}.npy")
        assert np.all(errs[:5] < 1e-12), errs
        assert errs[5] == 0.0, errs          # placement moves blocks, it adds nothing


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, int(sys.argv[1]), sys.argv[2]), nprocs=2, join=True)
