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


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import ttsk_oracle as orc
    from tt_sketch_amd import SketchContainer, SparseTensor, TensorSum, TensorTrain
    from tt_sketch_amd.distributed import allreduce_container, shard_bounds, shard_tensor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)                      # identical on all ranks
    shape, s, l, r = (6, 7, 5, 8), 3, (3, 4, 3), (5, 6, 5)
    terms = [orc.random_tt(shape, s, rng) for _ in range(5)]
    ld = orc.random_tt_drm(shape, l, False, rng)
    rd = orc.random_tt_drm(shape, r, True, rng)
    idx = np.stack([rng.integers(0, n, 101) for n in shape])
    val = rng.standard_normal(101)

    def sketch(kind, data):
        P, O = orc.general_sketch(kind, data, ld, rd, "streaming")
        return SketchContainer(P, O)

    results = {}
    # (1) TensorSum of TTs: summands dealt to ranks
    whole = TensorSum([TensorTrain(c) for c in terms])
    mine = shard_tensor(whole, rank, world)
    lo, hi = shard_bounds(len(terms), rank, world)
    assert mine.num_summands == hi - lo
    local = sketch("sum", [("tt", t.cores) for t in mine.tensors]) if mine.num_summands else \
        SketchContainer.zero(shape, l, r)
    total = allreduce_container(local)
    ref = sketch("sum", [("tt", c) for c in terms])
    results["tt_sum"] = max(np.max(np.abs(a - b)) / np.max(np.abs(b))
                            for a, b in zip(total.Psi_cores + total.Omega_mats, ref.Psi_cores + ref.Omega_mats))
    # (2) nnz shards of a sparse tensor
    sp = SparseTensor(shape, idx, val)
    part = shard_tensor(sp, rank, world).tensors[0]
    local = sketch("sparse", (shape, np.asarray(part.indices), np.asarray(part.entries)))
    total = allreduce_container(local)
    ref = sketch("sparse", (shape, idx, val))
    results["sparse"] = max(np.max(np.abs(a - b)) / np.max(np.abs(b))
                            for a, b in zip(total.Psi_cores + total.Omega_mats, ref.Psi_cores + ref.Omega_mats))
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.array([results["tt_sum"], results["sparse"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partial_sketch_sum(tmp_path):
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    ge.build_oracle()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        errs = np.load(tmp_path / f"r{rank}.npy")
        assert np.all(errs < 1e-12), errs


def test_shard_bounds_cover_everything():
    from tt_sketch_amd.distributed import shard_bounds
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
