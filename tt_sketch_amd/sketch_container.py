"""``SketchContainer``: the Psi cores and Omega matrices of one sketch.

Same role and arithmetic as the reference's ``tt_sketch/sketch_container.py:11-89`` (whose
``__mul__`` is broken at :78; fixed here).  ``Psi_cores`` / ``Omega_mats`` are host NumPy arrays as in
the reference (picklable, mutable in place), but a container built from device arrays keeps them on
the device until one of those lists is first read: ``stream_sketch(...).to_tt()`` never moves the
32 MB sketch over PCIe.  ``pack`` / ``unpack`` give the contiguous ``[Psi_0..Psi_{d-1}, Omega_0..]``
buffer that the multi-GPU partial-sketch sum reduces with a single collective.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .device import DevArray, to_host


class SketchContainer:
    def __init__(self, Psi_cores, Omega_mats, shape: Optional[Tuple[int, ...]] = None,
                 left_rank: Optional[Tuple[int, ...]] = None,
                 right_rank: Optional[Tuple[int, ...]] = None) -> None:
        self._psi = list(Psi_cores)       # DevArray until the host lists are first read
        self._omega = list(Omega_mats)
        if not any(isinstance(a, DevArray) for a in self._psi + self._omega):
            self._to_host()
        P = self._psi
        self.shape = tuple(int(p.shape[1]) for p in P) if shape is None else tuple(shape)
        self.left_rank = (tuple(int(p.shape[0]) for p in P[1:]) if left_rank is None
                          else tuple(left_rank))
        self.right_rank = (tuple(int(p.shape[2]) for p in P[:-1]) if right_rank is None
                           else tuple(right_rank))

    def _to_host(self) -> None:
        """From here on the host lists are the state (callers may write into them)."""
        self._psi = [to_host(P) for P in self._psi]
        self._omega = [to_host(O) for O in self._omega]

    @property
    def Psi_cores(self) -> List[np.ndarray]:
        self._to_host()
        return self._psi

    @Psi_cores.setter
    def Psi_cores(self, value) -> None:
        self._psi = [to_host(P) for P in value]

    @property
    def Omega_mats(self) -> List[np.ndarray]:
        self._to_host()
        return self._omega

    @Omega_mats.setter
    def Omega_mats(self, value) -> None:
        self._omega = [to_host(O) for O in value]

    def device_arrays(self, stream: int = 0):
        """(Psi, Omega) as device arrays, uploading whatever lives on the host; state unchanged."""
        from .device import as_dev
        return [as_dev(P, stream) for P in self._psi], [as_dev(O, stream) for O in self._omega]

    def __getstate__(self):
        self._to_host()
        return self.__dict__

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
