"""Classical TT-SVD of a dense array (reference ``tt_sketch/tt_svd.py:10-49``).

Kept for API completeness: it is a LAPACK SVD sweep over unfoldings, not a sketch, and is not
on the accelerated path; host NumPy.
"""
from typing import Optional

import numpy as np

from .tensor import Tensor, TensorTrain
from .utils import TTRank, process_tt_rank


def tt_svd(tensor: Tensor, rank: Optional[TTRank] = None) -> TensorTrain:
    """Left-to-right sweep: SVD of the (r_{k-1} n_k) x rest unfolding, keep <= rank[k] columns."""
    X = np.asarray(tensor.to_numpy(), dtype=np.float64)
    shape = tuple(tensor.shape)
    d = len(shape)
    if rank is None:
        rank = (int(np.prod(shape, dtype=np.int64)),) * (d - 1)
    cap = process_tt_rank(rank, shape, trim=True)
    cores = []
    rest = X.reshape(1, -1)
    r_prev = 1
    for k in range(d - 1):
        M = rest.reshape(r_prev * shape[k], -1)
        U, S, Vt = np.linalg.svd(M, full_matrices=False)
        r = max(min(U.shape[1], cap[k]), 1)
        cores.append(U[:, :r].reshape(r_prev, shape[k], r))
        rest = S[:r, None] * Vt[:r]
        r_prev = r
    cores.append(rest.reshape(r_prev, shape[-1], 1))
    return TensorTrain(cores)
