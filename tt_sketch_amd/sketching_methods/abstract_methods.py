"""Capability ABCs of the DRM plug-in surface (reference
``tt_sketch/sketching_methods/abstract_methods.py:15-63``): which tensor kinds a DRM can
form partial contractions with.  Each method is a generator of d-1 arrays in left-to-right
order; shapes as documented per method."""
from abc import ABC, abstractmethod

from ..drm_base import DRM


class CansketchTT(DRM, ABC):
    @abstractmethod
    def sketch_tt(self, tensor):
        """yields (tensor.rank[mu], drm.rank[mu]): DRM_mu^T contracted with cores 0..mu."""


class CansketchSparse(DRM, ABC):
    @abstractmethod
    def sketch_sparse(self, tensor):
        """yields (drm.rank[mu], tensor.nnz): DRM rows sampled at the nonzero positions."""


class CansketchDense(DRM, ABC):
    @abstractmethod
    def sketch_dense(self, tensor):
        """yields (drm.rank[mu], prod(shape[:mu+1])): the DRM as a dense matrix."""


class CansketchCP(DRM, ABC):
    @abstractmethod
    def sketch_cp(self, tensor):
        """yields (tensor.rank, drm.rank[mu]): one contracted row per CP term."""


class CanSketchTucker(DRM, ABC):
    @abstractmethod
    def sketch_tucker(self, tensor):
        """yields (prod(tensor.rank[:mu+1]), drm.rank[mu]): DRM against the factor matrices."""
