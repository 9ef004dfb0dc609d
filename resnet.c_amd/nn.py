"""Host-side mirror of the reference's nn.cuh / nn.cu layer wrappers.

Same classes, member names and argument order as cuda/nn.cuh:8-136:
``Conv2d``, ``BatchNorm2d``, ``Pool2d``, ``Linear``, ``reluForward``,
``addForward``.  ``forward(x, out)`` never allocates: the caller owns the
activations (nn.cu throughout).  Each forward is one call into the C-ABI, which
launches hand-written gfx950 kernels; nothing here computes on the CPU.

Differences from the reference, all at the edges:
  * precondition failures raise AssertionError / RnError instead of abort();
  * no device synchronisation after each op unless
    ``get_ctx().set_sync_each_op(True)`` asks for the reference's behaviour;
  * tensors carry a layout tag; an op runs in the layout of its input.
"""
from __future__ import annotations

from . import _lib as L
from .tensor import Device, FloatTensor, Shape, Tensor, get_ctx

WEIGHTS_DIR = "weights_bin/"  # nn.cuh:21,58-61,113,117


def convOutputSize(x: int, kernel_size: int, stride: int, padding: int) -> int:
    """cuda/ops.cuh:9-13."""
    return int(L.lib().rn_conv_output_size(x, kernel_size, stride, padding))


def _call(fn_name: str, layout: int, *args) -> None:
    ctx = get_ctx()
    ctx.set_layout(layout)
    L.check(getattr(L.lib(), fn_name)(ctx.handle, *args), fn_name, ctx.handle)


def _gpu(*tensors: Tensor) -> None:
    for t in tensors:
        assert t.device == Device.GPU, "forward expects GPU tensors"
        assert bool(t), "forward on an empty tensor"


class Conv2d:
    def __init__(self, weight: FloatTensor, in_channels: int, out_channels: int, kernel_size: int,
                 stride: int = 1, padding: int = 0):
        self.weight = weight
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    @staticmethod
    def loadWeightToCuda(name: str, in_channels: int, out_channels: int, kernel_size: int,
                         stride: int = 1, padding: int = 0) -> "Conv2d":
        weight = FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".weight").view(
            Shape((out_channels, in_channels, kernel_size, kernel_size)))
        return Conv2d(weight, in_channels, out_channels, kernel_size, stride, padding)

    def getOutShape(self, x_shape: Shape) -> Shape:
        assert len(x_shape) == 4
        assert x_shape[1] == self.in_channels
        return Shape((x_shape[0], self.out_channels,
                      convOutputSize(x_shape[2], self.kernel_size, self.stride, self.padding),
                      convOutputSize(x_shape[3], self.kernel_size, self.stride, self.padding)))

    def forward(self, x: FloatTensor, out: FloatTensor) -> None:
        _gpu(x, out, self.weight)
        B, _c, H, W = x.shape().as_tuple(4)
        _b, _oc, h_out, w_out = out.shape().as_tuple(4)
        out.layout = x.layout
        _call("rn_conv2d_forward", x.layout, x.data(), out.data(), self.weight.data(),
              self.kernel_size, self.stride, self.padding, h_out, w_out, B, self.in_channels,
              self.out_channels, H, W)


class BatchNorm2d:
    def __init__(self, weight: FloatTensor, bias: FloatTensor, mean: FloatTensor,
                 var: FloatTensor, channels_num: int):
        self.weight, self.bias, self.mean, self.var = weight, bias, mean, var
        self.channels_num = channels_num
        for t in (weight, bias, mean, var):  # nn.cuh:50-53
            assert t.shape() == Shape((channels_num,))

    @staticmethod
    def loadWeightToCuda(name: str, channels_num: int) -> "BatchNorm2d":
        return BatchNorm2d(FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".weight"),
                           FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".bias"),
                           FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".running_mean"),
                           FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".running_var"),
                           channels_num)

    def forward(self, x: FloatTensor, out: FloatTensor) -> None:
        _gpu(x, out)
        B, _C, h, w = x.shape().as_tuple(4)
        out.layout = x.layout
        _call("rn_batchnorm2d_forward", x.layout, x.data(), out.data(), self.weight.data(),
              self.bias.data(), self.mean.data(), self.var.data(), B, self.channels_num, h * w)


class Pool2d:
    def __init__(self, channels: int, kernel_size: int, stride: int = 1, padding: int = 0):
        self.channels, self.kernel_size = channels, kernel_size
        self.stride, self.padding = stride, padding

    def outSideSize(self, side_size: int) -> int:
        return convOutputSize(side_size, self.kernel_size, self.stride, self.padding)

    def getOutShape(self, x_shape: Shape) -> Shape:
        assert len(x_shape) == 4
        assert x_shape[1] == self.channels
        return Shape((x_shape[0], self.channels, self.outSideSize(x_shape[2]),
                      self.outSideSize(x_shape[3])))

    def _run(self, fn: str, x: FloatTensor, out: FloatTensor) -> None:
        _gpu(x, out)
        B, C, H, W = x.shape().as_tuple(4)
        out_h, out_w = out.shape()[2], out.shape()[3]
        out.layout = x.layout
        _call(fn, x.layout, x.data(), out.data(), self.kernel_size, self.stride, self.padding,
              out_h, out_w, B, C, H, W)

    def maxforward(self, x: FloatTensor, out: FloatTensor) -> None:
        self._run("rn_maxpool2d_forward", x, out)

    def avgforward(self, x: FloatTensor, out: FloatTensor) -> None:
        self._run("rn_avgpool2d_forward", x, out)


class Linear:
    def __init__(self, weight: FloatTensor, bias: FloatTensor, in_features: int,
                 out_features: int):
        self.weight, self.bias = weight, bias
        self.in_features, self.out_features = in_features, out_features
        assert weight.shape() == Shape((out_features, in_features))  # nn.cuh:107-108
        assert bias.shape() == Shape((out_features,))

    @staticmethod
    def loadWeightToCuda(name: str, in_features: int, out_features: int) -> "Linear":
        weight = FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".weight").view(
            Shape((out_features, in_features)))
        bias = FloatTensor.loadToCuda(WEIGHTS_DIR + name + ".bias").view(Shape((out_features,)))
        return Linear(weight, bias, in_features, out_features)

    def getOutShape(self, x_shape: Shape) -> Shape:
        assert len(x_shape) == 2
        assert x_shape[1] == self.in_features
        return Shape((x_shape[0], self.out_features))

    def forward(self, x: FloatTensor, out: FloatTensor) -> None:
        _gpu(x, out)
        B = x.shape()[0]
        _call("rn_linear_forward", x.layout, x.data(), out.data(), self.weight.data(),
              self.bias.data(), B, self.in_features, self.out_features)


def reluForward(x: FloatTensor, out: FloatTensor) -> None:
    assert x.shape() == out.shape()  # nn.cu:68
    _gpu(x, out)
    out.layout = x.layout
    _call("rn_relu_forward", x.layout, x.data(), out.data(), x.numel())


def addForward(a: FloatTensor, b: FloatTensor, out: FloatTensor) -> None:
    assert a.shape() == b.shape()  # nn.cu:79-80
    assert a.shape() == out.shape()
    _gpu(a, b, out)
    assert a.layout == b.layout, "addForward on tensors of different layouts"
    out.layout = a.layout
    _call("rn_add_forward", a.layout, a.data(), b.data(), out.data(), a.numel())
