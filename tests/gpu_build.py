"""Build tt_sketch_amd tensors / DRMs from the golden fixtures or oracle descriptors."""
import numpy as np

import tt_sketch_amd as tsa
from oracle import ttsk_oracle as orc


def make_tensor(kind, data):
    if kind == "tt":
        return tsa.TensorTrain([np.array(c) for c in data])
    if kind == "cp":
        return tsa.CPTensor([np.array(c) for c in data])
    if kind == "tucker":
        return tsa.TuckerTensor([np.array(u) for u in data[0]], np.array(data[1]))
    if kind == "dense":
        return tsa.DenseTensor(np.array(data))
    if kind == "sparse":
        return tsa.SparseTensor(data[0], np.array(data[1]), np.array(data[2]))
    if kind == "sum":
        return tsa.TensorSum([make_tensor(k, t) for k, t in data])
    raise ValueError(kind)


def _user(t, transpose):
    return tuple(t[::-1]) if transpose else tuple(t)


def make_drm(o):
    """oracle DRM descriptor -> tt_sketch_amd DRM with the same injected data."""
    tr = o.transpose
    if isinstance(o, orc.TTDrm):
        true = tuple(c.shape[2] for c in o.cores)
        return tsa.TensorTrainDRM(_user(true, tr), o.shape, tr, seed=1, cores=[np.array(c) for c in o.cores],
                                  rank_min=_user(o.rank_min, tr), rank_max=_user(o.rank_max, tr),
                                  true_rank=_user(true, tr))
    if isinstance(o, orc.DenseDrm):
        d = tsa.DenseGaussianDRM(_user(o.rank, tr), o.shape, tr, seed=1)
        d.sketching_mats = [np.array(m) for m in o.mats]
        return d
    if isinstance(o, orc.HashGaussDrm):
        return tsa.SparseGaussianDRM(_user(o.rank_max, tr), o.shape, tr, seed=o.seed,
                                     rank_min=_user(o.rank_min, tr), rank_max=_user(o.rank_max, tr),
                                     true_rank=_user(o.rank_max, tr))
    if isinstance(o, orc.HashSignDrm):
        return tsa.SparseSignDRM(_user(o.true_rank, tr), o.shape, tr, seed=o.seed,
                                 num_non_zero_per_row=o.nnz, rank_min=_user(o.rank_min, tr),
                                 rank_max=_user(o.rank_max, tr), true_rank=_user(o.true_rank, tr))
    raise ValueError(type(o))
