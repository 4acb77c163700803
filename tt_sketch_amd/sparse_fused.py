"""Fast path: SparseTensor input with hashed sparse DRMs (SparseGaussianDRM / SparseSignDRM, any pair) on both sides,
streaming method.

One pass per mode over a resident, mode-ordered stream of the nonzeros (``csrc/sparse_fused.hip``): the DRM rows are
sampled (or gathered from a small per-prefix table) where they are consumed, Psi_mu and one Omega come out of the
same pass, nothing of size nnz x rank is ever written.  Numerically the same sums as
``SparseGaussianDRM.sketch_sparse`` / ``SparseSignDRM.sketch_sparse`` + ``sketch_omega_sparse`` / ``sketch_psi_sparse``
(reference sparse_gaussian_drm.py:29-44, sparse_sign_drm.py:34-51, sparse_sketch.py:8-69) with bit-identical samples; the summation order is fixed
(no atomics), so two runs agree bit for bit.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _native as nat
from .device import DevArray, axpby
from .drm.sparse_gaussian_drm import SparseGaussianDRM
from .drm.sparse_sign_drm import SparseSignDRM
from .tensor import SparseTensor, TensorSum

last_plan: dict = {}             # what the last sketch did (bench.py reads it): sampled columns per nonzero, table rows
MAX_WIDTH = 32                   # columns per DRM factor the pass kernel takes (one 16-column matrix tile, or two)
MAX_MODE = 1 << 24               # the mode-order sort key holds the mode index in 24 bits (ttsk_sparse_mode_order)
TABLE_BYTES = 32 << 20           # a per-prefix table larger than this is sampled per nonzero instead (it would leave the L2 / MALL)


class _Factor(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("w", ctypes.c_int), ("rank_min", ctypes.c_int), ("src", ctypes.c_int),
                ("mul", ctypes.c_uint64), ("seed", ctypes.c_uint64), ("table", ctypes.c_void_p),
                ("full", ctypes.c_int), ("nnz", ctypes.c_int)]


def _u64(vals):
    return (ctypes.c_uint64 * len(vals))(*[int(v) for v in vals])


def _flat_mult(shape) -> List[int]:
    out = (ctypes.c_uint64 * len(shape))()
    nat.call("ttsk_sparse_flat_mult", _u64(shape), len(shape), out)
    return [int(v) for v in out]


class _Side:
    """One DRM as the passes see it: factor k = the DRM's k-th sketching matrix (k + 1 index rows)."""

    def __init__(self, drm, shape: Tuple[int, ...], nnz: int):
        self.drm = drm
        self.sign = type(drm) is SparseSignDRM
        self.shape = tuple(int(n) for n in shape)          # in the order the DRM walks the tensor
        self.nnz = nnz
        self.cache = drm.__dict__.setdefault("_sg_tables", {})

    def width(self, k: int) -> int:
        return int(self.drm.rank_max[k] - self.drm.rank_min[k])

    def staged(self, k: int) -> int:
        """columns of the staged tile a factor sampled in the pass needs: a sign row is made whole (its swaps reach
        every position), a Gaussian row only where it is used"""
        return int(self.drm.true_rank[k]) if self.sign else self.width(k)

    def seed(self, k: int) -> int:
        return (k + int(self.drm.seed)) % 2**63            # sparse_gaussian_drm.py:34-36, sparse_sign_drm.py:39-41

    def prefixes(self, k: int) -> int:
        return int(np.prod([int(n) for n in self.shape[:k + 1]], dtype=object))

    def use_table(self, k: int) -> bool:
        P = self.prefixes(k)
        return P < 2**31 and 2 * P <= self.nnz and P * self.width(k) * 8 <= TABLE_BYTES

    def table(self, k: int) -> DevArray:
        drm = self.drm
        key = (k, self.shape[:k + 1], int(drm.rank_min[k]), int(drm.rank_max[k]), int(drm.seed))
        if self.sign:
            key += (int(drm.true_rank[k]), int(drm.nnz[k]))
        if key not in self.cache:
            # (one spare row: the pass kernel fetches rows in 16-byte units, the last unit of an odd row reaches 8 bytes on)
            out = DevArray.empty((self.prefixes(k) + 1, self.width(k)))
            if self.sign:
                nat.call("ttsk_sparse_sign_table", _u64(self.shape[:k + 1]), k + 1, int(drm.true_rank[k]), int(drm.rank_min[k]),
                         int(drm.rank_max[k]), int(drm.nnz[k]), ctypes.c_uint64(self.seed(k)), ctypes.c_void_p(out.ptr), 0)
            else:
                nat.call("ttsk_sparse_normal_table", _u64(self.shape[:k + 1]), k + 1, int(drm.rank_min[k]),
                         int(drm.rank_max[k]), ctypes.c_uint64(self.seed(k)), ctypes.c_void_p(out.ptr), 0)
            self.cache[key] = out
        return self.cache[key]

    def factor(self, k: int, src: int, mul: int = 0) -> Tuple[_Factor, Optional[DevArray]]:
        tab = self.table(k) if self.use_table(k) else None
        kind = 1 if tab is not None else (3 if self.sign else 2)
        f = _Factor(kind, self.width(k), int(self.drm.rank_min[k]), src, mul, self.seed(k),
                    tab.ptr if tab is not None else None,
                    int(self.drm.true_rank[k]) if self.sign else 0, int(self.drm.nnz[k]) if self.sign else 0)
        return f, tab

    def cost(self, k: int) -> int:
        return 0 if self.use_table(k) else self.width(k)

    def covered(self, k: int) -> bool:
        if not 1 <= self.width(k) <= MAX_WIDTH:
            return False
        if self.sign and not self.use_table(k):
            return self.staged(k) <= MAX_WIDTH and 0 <= int(self.drm.nnz[k]) <= int(self.drm.true_rank[k])
        return True


def _mode_stream(tensor: SparseTensor, mu: int):
    """(fl, fr, j, val) of mode ``mu`` in mode order; built once per tensor and mode (shared with views)."""
    d, N = len(tensor.shape), tensor.nnz
    order = tensor.dev_row_order
    cache = tensor._upload()[2]
    key = ("stream", tuple(order), mu)
    if key not in cache:
        idx, val = tensor.dev_indices(), tensor.dev_entries()
        l_rows, l_shape = list(order[:mu]), list(tensor.shape[:mu])
        r_rows = [order[d - 1 - i] for i in range(d - 1 - mu)]
        r_shape = [tensor.shape[d - 1 - i] for i in range(d - 1 - mu)]
        ints = lambda v: (ctypes.c_int * max(len(v), 1))(*v)
        # mode order with the suffix as the secondary key (not the plain mode sort of dev_mode_perm): inside a slice the
        # rows of the right-hand DRM tables are then visited in ascending order
        perm = DevArray.empty((N,), dtype=np.int64)
        nat.call("ttsk_sparse_mode_order", ctypes.c_void_p(idx.ptr), N, ctypes.c_size_t(N), ints(r_rows), _u64(r_shape or [1]),
                 len(r_rows), int(order[mu]), int(tensor.shape[mu]), ctypes.c_void_p(perm.ptr), 0)
        # 32-bit flat indices where every prefix / suffix extent stays below 2^31 (20 instead of 28 bytes per record and pass)
        small = (int(np.prod(l_shape or [1], dtype=object)) < 2**31 and int(np.prod(r_shape or [1], dtype=object)) < 2**31
                 and os.environ.get("TTSK_SPARSE_U32", "1") != "0")
        words = (N + 1) // 2 if small else N
        fl, fr = DevArray.empty((words,), dtype=np.int64), DevArray.empty((words,), dtype=np.int64)
        jj = DevArray.empty(((N + 1) // 2,), dtype=np.int64)          # int32 records
        vv = DevArray.empty((N,))
        nat.call("ttsk_sparse_mode_stream_u32" if small else "ttsk_sparse_mode_stream", ctypes.c_void_p(idx.ptr), N,
                 ctypes.c_void_p(perm.ptr), ctypes.c_size_t(N),
                 ints(l_rows), _u64(l_shape or [1]), len(l_rows), ints(r_rows), _u64(r_shape or [1]), len(r_rows), int(order[mu]),
                 ctypes.c_void_p(val.ptr), ctypes.c_void_p(fl.ptr), ctypes.c_void_p(fr.ptr), ctypes.c_void_p(jj.ptr),
                 ctypes.c_void_p(vv.ptr), 0)
        cache[key] = (fl, fr, jj, vv, small)
    return cache[key]


def try_sparse_gauss_sketch(tensor, left_drm, right_drm, method):
    """(Psi, Omega) device arrays through the one-pass-per-mode path, or None if it does not apply."""
    from .sketch_dispatch import SketchMethod
    if method != SketchMethod.streaming or os.environ.get("TTSK_SPARSE_FUSED", "1") == "0":
        return None
    if type(left_drm) not in (SparseGaussianDRM, SparseSignDRM) or type(right_drm) not in (SparseGaussianDRM, SparseSignDRM):
        return None
    if type(tensor) is TensorSum and tensor.tensors and all(type(t) is SparseTensor for t in tensor.tensors):
        # a sum of sparse tensors (the nnz shards of distributed.shard_tensor, reference tensor.py:215-234): every summand
        # through the passes, summed in the order of the summands -- as the reference's += (sketch_dispatch.py:85-139)
        # and, like the passes themselves, the same bits in every run
        parts = [try_sparse_gauss_sketch(t, left_drm, right_drm, method) for t in tensor.tensors if t.nnz]
        if not parts or any(p is None for p in parts):
            return None
        Psi, Omega = parts[0]
        for P2, O2 in parts[1:]:
            for y, x in zip(Psi + Omega, P2 + O2):
                axpby(y, x)
        return Psi, Omega
    if type(tensor) is not SparseTensor:
        return None
    if left_drm.transpose or not right_drm.transpose:
        return None
    shape = tuple(int(n) for n in tensor.shape)
    d, N = len(shape), tensor.nnz
    if d < 2 or N == 0 or N >= 2**31:
        return None
    if max(shape) > MAX_MODE:
        return None
    if tuple(left_drm.shape) != shape or tuple(right_drm.shape) != shape:
        raise ValueError(f"Shape {left_drm.shape} of DRM doesn't match tensor's shape {tensor.shape}")
    L = _Side(left_drm, shape, N)
    R = _Side(right_drm, shape[::-1], N)          # factor nu of the right DRM = suffix of d - 1 - nu.. = R_mu with mu = d - 2 - nu
    if any(not L.covered(k) or not R.covered(k) for k in range(d - 1)):
        return None
    tensor.prepare_device()
    multL, multR = _flat_mult(shape), _flat_mult(shape[::-1])
    # which pass carries Omega_mu: pass mu (with L_mu as the extra factor, sharing R_mu) or pass mu + 1 (with R_mu as
    # the extra factor, sharing L_mu) -- whichever extra factor is cheaper to make; a pass carries one Omega
    rider = {}
    for mu in range(d - 1):
        nu = d - 2 - mu
        here = (mu not in rider, L.cost(mu))
        there = (R.cost(nu),)
        if here[0] and here[1] <= there[0]:
            rider[mu] = ("left", mu)
        else:
            rider[mu + 1] = ("right", mu)
    lw = lambda mu: L.width(mu)
    rw = lambda mu: R.width(d - 2 - mu)
    Psi, Omega, keep = [], [None] * (d - 1), []
    sampled, table_rows = 0, 0
    for mu in range(d):
        fl, fr, jj, vv, small = _mode_stream(tensor, mu)
        A = B = C = None
        if mu > 0:
            A, t = L.factor(mu - 1, 0)
            keep.append(t)
        if mu < d - 1:
            B, t = R.factor(d - 2 - mu, 1)
            keep.append(t)
        psi = DevArray.zeros((lw(mu - 1) if mu > 0 else 1, shape[mu], rw(mu) if mu < d - 1 else 1))
        om, c_left = None, 0
        if mu in rider:
            side, k = rider[mu]
            om = DevArray.zeros((lw(k), rw(k)))
            Omega[k] = om
            if side == "left":          # Omega_mu = L_mu (x) R_mu: L_mu's flat index = prefix + j * multiplier of mode mu
                C, t = L.factor(k, 2, multL[k])
                c_left = 1
            else:                       # Omega_{mu-1} = L_{mu-1} (x) R_{mu-1}: R_{mu-1}'s flat index = suffix + j * multiplier
                C, t = R.factor(d - 2 - k, 3, multR[d - 2 - k])
            keep.append(t)
        for f in (A, B, C):
            if f is not None and f.kind >= 2:
                sampled += f.w
        P = ctypes.c_void_p
        ref = lambda f: None if f is None else ctypes.byref(f)
        nat.call("ttsk_sparse_gauss_pass_u32" if small else "ttsk_sparse_gauss_pass", P(fl.ptr), P(fr.ptr), P(jj.ptr), P(vv.ptr),
                 ctypes.c_size_t(N), int(shape[mu]),
                 ref(A), ref(B), ref(C), c_left, P(psi.ptr), None if om is None else P(om.ptr), 0)
        Psi.append(psi)
    last_plan.clear()
    last_plan.update(sampled_columns_per_nonzero=sampled, passes=d, stream_bytes_per_nonzero_and_pass=20 if small else 28,
                     riders={int(k): v for k, v in rider.items()})
    return Psi, Omega
