from .dense_gaussian_drm import DenseGaussianDRM
from .sparse_gaussian_drm import SparseGaussianDRM
from .sparse_sign_drm import SparseSignDRM
from .tensor_train_drm import TensorTrainDRM

ALL_DRM = (DenseGaussianDRM, SparseGaussianDRM, TensorTrainDRM, SparseSignDRM)
