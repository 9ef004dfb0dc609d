"""Functional forms of the seven reference ops on host arrays.

Each function uploads its NCHW numpy operands, makes ONE call into the C-ABI
entry point that replaces the reference kernel (include/rn_hip.h), and
downloads the result.  They exist for tests, notebooks and small tools; the
throughput path is ``model.NativeModel``.  ``layout`` selects how the device
side holds activations: "nchw" (what the reference kernels see) or "nhwc" (the
engine layout; arrays are transposed on the host around the call).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib as L
from .tensor import Device, FloatTensor, get_ctx

_LAYOUT = {"nchw": L.RN_LAYOUT_NCHW, "nhwc": L.RN_LAYOUT_NHWC}


def _up(a: np.ndarray, layout: str) -> FloatTensor:
    a = np.asarray(a, dtype=np.float32)
    if layout == "nhwc" and a.ndim == 4:
        a = a.transpose(0, 2, 3, 1)
    return FloatTensor.from_numpy(a, Device.GPU)


def _down(t: FloatTensor, shape_nchw, layout: str) -> np.ndarray:
    flat = t.cpu()._storage
    if layout == "nhwc" and len(shape_nchw) == 4:
        B, C, H, W = shape_nchw
        return flat.reshape(B, H, W, C).transpose(0, 3, 1, 2).copy()
    return flat.reshape(shape_nchw).copy()


def _run(name: str, layout: str, *args) -> None:
    ctx = get_ctx()
    ctx.set_layout(_LAYOUT[layout])
    L.check(getattr(L.lib(), name)(ctx.handle, *args), name, ctx.handle)
    ctx.sync()


def conv_output_size(x: int, k: int, stride: int, pad: int) -> int:
    return int(L.lib().rn_conv_output_size(x, k, stride, pad))


def conv2d(x, w, stride: int = 1, pad: int = 0, layout: str = "nchw") -> np.ndarray:
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    dx, dw = _up(x, layout), _up(w, "nchw")
    out = FloatTensor((B, Cout, ho, wo), Device.GPU)
    _run("rn_conv2d_forward", layout, dx.data(), out.data(), dw.data(), k, stride, pad, ho, wo, B,
         Cin, Cout, H, W)
    return _down(out, (B, Cout, ho, wo), layout)


def _pool(name, x, k, stride, pad, layout):
    B, C, H, W = x.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    dx = _up(x, layout)
    out = FloatTensor((B, C, ho, wo), Device.GPU)
    _run(name, layout, dx.data(), out.data(), k, stride, pad, ho, wo, B, C, H, W)
    return _down(out, (B, C, ho, wo), layout)


def maxpool2d(x, k, stride=1, pad=0, layout="nchw"):
    return _pool("rn_maxpool2d_forward", x, k, stride, pad, layout)


def avgpool2d(x, k, stride=1, pad=0, layout="nchw"):
    return _pool("rn_avgpool2d_forward", x, k, stride, pad, layout)


def linear(x, w, b: Optional[np.ndarray]) -> np.ndarray:
    B, fin = x.shape
    fout = w.shape[0]
    dx, dw = _up(x, "nchw"), _up(w, "nchw")
    db = None if b is None else _up(b, "nchw")
    out = FloatTensor((B, fout), Device.GPU)
    _run("rn_linear_forward", "nchw", dx.data(), out.data(), dw.data(),
         None if db is None else db.data(), B, fin, fout)
    return _down(out, (B, fout), "nchw")


def relu(x, inplace: bool = True) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    dx = _up(x.reshape(-1), "nchw")
    out = dx if inplace else FloatTensor((x.size,), Device.GPU)
    _run("rn_relu_forward", "nchw", dx.data(), out.data(), x.size)
    return _down(out, x.shape, "nchw")


def add(a, b, inplace: bool = True) -> np.ndarray:
    a = np.asarray(a, dtype=np.float32)
    da, db = _up(a.reshape(-1), "nchw"), _up(np.asarray(b).reshape(-1), "nchw")
    out = da if inplace else FloatTensor((a.size,), Device.GPU)
    _run("rn_add_forward", "nchw", da.data(), db.data(), out.data(), a.size)
    return _down(out, a.shape, "nchw")


def batchnorm2d(x, w, b, mean, var, layout: str = "nchw", inplace: bool = True) -> np.ndarray:
    B, C = x.shape[0], x.shape[1]
    N = int(np.prod(x.shape[2:]))
    dx = _up(x, layout)
    out = dx if inplace else FloatTensor(x.shape, Device.GPU)
    p = [_up(v, "nchw") for v in (w, b, mean, var)]
    _run("rn_batchnorm2d_forward", layout, dx.data(), out.data(), *(t.data() for t in p), B, C, N)
    return _down(out, x.shape, layout)


def argmax(logits) -> np.ndarray:
    logits = np.asarray(logits, dtype=np.float32)
    B, C = logits.shape
    dl = _up(logits, "nchw")
    idx = FloatTensor((2 * B,), Device.GPU)  # B uint64 slots
    _run("rn_argmax_forward", "nchw", dl.data(), idx.data(), B, C)
    raw = idx.cpu()._storage
    return raw.view(np.uint64).astype(np.int64)


def conv2d_nhwc_fused(x, w, stride=1, pad=0, scale=None, shift=None, residual=None,
                      relu_: bool = False) -> np.ndarray:
    """rn_conv2d_pack_weight + rn_conv2d_nhwc_forward with an epilogue (NCHW host arrays)."""
    ctx = get_ctx()
    lib = L.lib()
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    cs = int(lib.rn_conv2d_input_channels(Cin))
    xp = np.zeros((B, H, W, cs), dtype=np.float32)
    xp[..., :Cin] = np.asarray(x, dtype=np.float32).transpose(0, 2, 3, 1)
    dx = FloatTensor.from_numpy(xp, Device.GPU)
    dw = _up(w, "nchw")
    packed = FloatTensor((int(lib.rn_conv2d_packed_weight_numel(Cin, Cout, k)),), Device.GPU)
    L.check(lib.rn_conv2d_pack_weight(ctx.handle, dw.data(), packed.data(), Cin, Cout, k),
            "rn_conv2d_pack_weight", ctx.handle)
    keep = [_up(v, "nchw") if v is not None else None for v in (scale, shift)]
    dres = _up(residual, "nhwc") if residual is not None else None
    ep = L.Epilogue(keep[0].data() if keep[0] else None, keep[1].data() if keep[1] else None,
                    dres.data() if dres else None, int(relu_))
    out = FloatTensor((B, Cout, ho, wo), Device.GPU)
    L.check(lib.rn_conv2d_nhwc_forward(ctx.handle, dx.data(), out.data(), packed.data(), k, stride,
                                       pad, ho, wo, B, Cin, Cout, H, W, ctypes.byref(ep)),
            "rn_conv2d_nhwc_forward", ctx.handle)
    ctx.sync()
    return _down(out, (B, Cout, ho, wo), "nhwc")
