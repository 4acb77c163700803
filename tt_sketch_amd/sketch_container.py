"""``SketchContainer``: the Psi cores and Omega matrices of one sketch.

Same role and arithmetic as the reference's ``tt_sketch/sketch_container.py:11-89`` (whose
``__mul__`` is broken at :78; fixed here).  Arrays are host NumPy (picklable state, as in the
reference); ``pack`` / ``unpack`` give the contiguous ``[Psi_0..Psi_{d-1}, Omega_0..]`` buffer that
the multi-GPU partial-sketch sum reduces with a single collective.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .device import DevArray, to_host


class SketchContainer:
    def __init__(self, Psi_cores, Omega_mats, shape: Optional[Tuple[int, ...]] = None,
                 left_rank: Optional[Tuple[int, ...]] = None,
                 right_rank: Optional[Tuple[int, ...]] = None) -> None:
        self.Psi_cores = [to_host(P) for P in Psi_cores]
        self.Omega_mats = [to_host(O) for O in Omega_mats]
        P = self.Psi_cores
        self.shape = tuple(int(p.shape[1]) for p in P) if shape is None else tuple(shape)
        self.left_rank = (tuple(int(p.shape[0]) for p in P[1:]) if left_rank is None
                          else tuple(left_rank))
        self.right_rank = (tuple(int(p.shape[2]) for p in P[:-1]) if right_rank is None
                           else tuple(right_rank))

    @classmethod
    def zero(cls, shape, left_rank, right_rank) -> "SketchContainer":
        lr, rr = (1,) + tuple(left_rank), tuple(right_rank) + (1,)
        Psi = [np.zeros((a, n, b)) for a, n, b in zip(lr, shape, rr)]
        Om = [np.zeros((a, b)) for a, b in zip(left_rank, right_rank)]
        return cls(Psi, Om, shape, left_rank, right_rank)

    def __add__(self, other: "SketchContainer") -> "SketchContainer":
        return SketchContainer([a + b for a, b in zip(self.Psi_cores, other.Psi_cores)],
                               [a + b for a, b in zip(self.Omega_mats, other.Omega_mats)])

    @property
    def T(self) -> "SketchContainer":
        return SketchContainer([P.transpose(2, 1, 0) for P in self.Psi_cores[::-1]],
                               [O.T for O in self.Omega_mats[::-1]])

    def __mul__(self, other: float) -> "SketchContainer":
        return SketchContainer([P * other for P in self.Psi_cores],
                               [O * other for O in self.Omega_mats])

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1

    def __sub__(self, other):
        return self + (-other)

    def __truediv__(self, other: float):
        return self * (1 / other)

    # ---- packed form for the RCCL partial-sketch sum
    def pack(self) -> np.ndarray:
        parts = [np.ascontiguousarray(a).ravel() for a in self.Psi_cores + self.Omega_mats]
        return np.concatenate(parts) if parts else np.zeros(0)

    def unpack(self, buf: np.ndarray) -> "SketchContainer":
        out, off = [], 0
        for a in self.Psi_cores + self.Omega_mats:
            out.append(np.array(buf[off:off + a.size]).reshape(a.shape))
            off += a.size
        d = len(self.Psi_cores)
        return SketchContainer(out[:d], out[d:])
