"""MI355X-native ResNet forward path behind the reference's ops/nn/tensor API.

The arithmetic lives in ``csrc/`` (hand-written HIP for gfx950) behind the
C-ABI declared in ``include/rn_hip.h``; this package is the Python host-side
mirror of the reference's operator interface (olehskip/resnet.c: cuda/ops.cuh,
cuda/nn.cuh, cuda/tensor.cuh, cuda/inference/main.cu).  Importing the package is
cheap; the shared library is loaded on first use and its absence is a hard
error -- there is no CPU fallback.
"""
from . import weights, preprocess  # noqa: F401  (pure-numpy helpers)
from . import _lib, tensor, nn, ops, model  # noqa: F401
from ._lib import RnError  # noqa: F401
from .tensor import Device, Shape, Tensor, FloatTensor, Context, get_ctx, set_device  # noqa: F401
from .nn import (Conv2d, BatchNorm2d, Pool2d, Linear, reluForward, addForward,  # noqa: F401
                 convOutputSize)
from .model import (ResnetModel, createResnet, createResnet152, layerForward,  # noqa: F401
                    resnetForward, NativeModel, Pipeline, Graph, ShardedModel)

__version__ = "0.1.0"
