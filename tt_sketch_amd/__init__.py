"""tt_sketch_amd: the streaming TT-sketch hot path of RikVoorhaar/tt-sketch on AMD MI355X.

Python host code with the reference's module layout and API (``sketch``, ``sketch_dispatch``,
``drm``, ``drm_base``, ``sketching_methods``, ``tensor``, ``utils``, ``sketch_container``,
``tt_svd``) over a C-ABI HIP library (``include/ttsk.h`` -> ``tt_sketch_amd/libttsk.so``).
There is no CPU fallback: compute entry points raise if the library or a GPU is missing.
"""
from .drm import ALL_DRM, DenseGaussianDRM, SparseGaussianDRM, SparseSignDRM, TensorTrainDRM
from .sketch import (SketchedTensorTrain, assemble_sketched_tt, blocked_stream_sketch, hmt_sketch,
                     orthogonal_sketch, orthogonal_sketch_batch, hmt_sketch_batch, stream_sketch, stream_sketch_batch)
from .sketch_container import SketchContainer
from .sketch_dispatch import SketchMethod, general_sketch
from .tensor import (CPTensor, DenseTensor, SparseTensor, Tensor, TensorSum, TensorTrain,
                     TuckerTensor)
from .tt_svd import tt_svd

__all__ = [
    "ALL_DRM", "DenseGaussianDRM", "SparseGaussianDRM", "SparseSignDRM", "TensorTrainDRM",
    "SketchedTensorTrain", "assemble_sketched_tt", "blocked_stream_sketch", "hmt_sketch",
    "orthogonal_sketch", "orthogonal_sketch_batch", "hmt_sketch_batch", "stream_sketch", "stream_sketch_batch", "SketchContainer", "SketchMethod", "general_sketch",
    "CPTensor", "DenseTensor", "SparseTensor", "Tensor", "TensorSum", "TensorTrain", "TuckerTensor",
    "tt_svd",
]
