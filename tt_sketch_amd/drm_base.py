"""Dimension reduction matrix (DRM) base classes -- the plug-in surface.

Same protocol as the reference's ``tt_sketch/drm_base.py``: a DRM is constructed as
``(rank, shape, transpose, seed=None, rank_min=, rank_max=, true_rank=, **kwargs)``,
exposes ``rank / rank_min / rank_max / true_rank / shape / transpose / seed / T`` and
implements ``sketch_<kind>(tensor)`` generators (see ``sketching_methods.abstract_methods``)
that yield d-1 partial contractions, left to right.  Rank bookkeeping of a right
(``transpose=True``) DRM is stored reversed, i.e. in the order of the transposed
tensor it actually walks (reference drm_base.py:52-58).

Native DRMs yield ``DevArray`` views that stay in HBM; a user-defined DRM may yield
NumPy arrays, they are uploaded where they are consumed.
"""
from __future__ import annotations

import copy
from typing import Callable, Optional, Tuple

import numpy as np

from .utils import TTRank, process_tt_rank


class DRM:
    rank: Tuple[int, ...]
    rank_min: Tuple[int, ...]
    rank_max: Tuple[int, ...]
    true_rank: Tuple[int, ...]
    shape: Tuple[int, ...]
    transpose: bool
    seed: int

    def __init__(self, rank: TTRank, shape: Tuple[int, ...], transpose: bool,
                 seed: Optional[int] = None, rank_min: Optional[Tuple[int, ...]] = None,
                 rank_max: Optional[Tuple[int, ...]] = None,
                 true_rank: Optional[Tuple[int, ...]] = None, **kwargs) -> None:
        self.transpose = bool(transpose)
        self.shape = tuple(shape)
        full = process_tt_rank(rank, self.shape, trim=False)
        lo = tuple(rank_min) if rank_min is not None else (0,) * (len(self.shape) - 1)
        hi = tuple(rank_max) if rank_max is not None else full
        tr = tuple(true_rank) if true_rank is not None else full
        if self.transpose:  # keep everything in the order of the transposed tensor
            lo, hi, tr = lo[::-1], hi[::-1], tr[::-1]
        self.rank_min, self.rank_max, self.true_rank = lo, hi, tr
        self.rank = tuple(b - a for a, b in zip(lo, hi))
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        self.seed = int(seed) % (2**32 - 1)

    @property
    def T(self):
        """The same DRM regarded from the other side (reference drm_base.py:65-73)."""
        other = copy.copy(self)
        other.transpose = not self.transpose
        for name in ("true_rank", "rank_min", "rank_max", "rank"):
            setattr(other, name, getattr(self, name)[::-1])
        return other

    def __repr__(self) -> str:
        side = "Right" if self.transpose else "Left"
        return (f"<{side} {type(self).__name__} of rank {self.rank} and shape {self.shape}"
                f" at {hex(id(self))}>")


class CanSlice(DRM):
    """DRM whose rank range can be restricted to ``[start_rank, end_rank)`` -- the blocked
    sketch (reference drm_base.py:86-109).  Generic implementation: rebuild with the same
    seed and the slice bounds; the sampled data must be a function of the seed only."""

    def slice(self, start_rank, end_rank) -> DRM:
        full = self.true_rank[::-1] if self.transpose else self.true_rank
        return type(self)(rank=self.rank, shape=self.shape, transpose=self.transpose, seed=self.seed,
                          rank_min=start_rank, rank_max=end_rank, true_rank=full)


class CanIncreaseRank(CanSlice):
    """DRM whose leading block is unchanged when the rank grows (reference drm_base.py:112-119)."""

    def increase_rank(self, new_rank) -> DRM:
        return type(self)(new_rank, self.shape, self.transpose, self.seed)


def handle_transpose(sketch: Callable) -> Callable:
    """Right sketches reuse the left-to-right code on the transposed tensor and hand their
    list back reversed (reference drm_base.py:122-145).  The tensor is made resident first
    so that ``tensor.T`` is a set of strided views on the same HBM buffers."""

    def wrapper(self, tensor):
        if tuple(self.shape) != tuple(tensor.shape):
            raise ValueError(
                f"Shape {self.shape} of DRM doesn't match tensor's shape {tensor.shape}")
        if not self.transpose:
            yield from sketch(self, tensor)
            return
        tensor.prepare_device()
        for mat in list(sketch(self, tensor.T))[::-1]:
            yield mat

    wrapper.__name__ = getattr(sketch, "__name__", "sketch")
    wrapper.__doc__ = sketch.__doc__
    return wrapper
