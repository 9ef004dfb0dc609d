"""CPU checker: ctypes binding of liboracle.so plus the model graph.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

The per-op functions are thin wrappers over oracle/rn_oracle.c (which cites
the reference kernel each one follows).  ``resnet_forward`` restates the
sequencing of the reference driver, cuda/inference/main.cu:127-226:

    conv1 -> bn1 -> relu -> maxpool(3,2,1)                       main.cu:179-192
    per bottleneck block                                          main.cu:127-166
        [downsample conv -> bn]           (block 0 of a stage)
        conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3
        add(shortcut) -> relu
    avgpool(7) -> flatten -> fc                                   main.cu:213-224
    argmax with strict '<' (first maximum wins)                   main.cu:243-251

BN, ReLU and add run in place there (main.cu:138,145-146,162-163); in numpy the
same buffers are reused, which exercises the same aliasing contract.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib: Optional[ctypes.CDLL] = None

_u64 = ctypes.c_uint64
_fp = ctypes.POINTER(ctypes.c_float)


def build() -> str:
    """Compile liboracle.so in place (gcc only; no GPU, no reference needed)."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.rn_oracle_conv_output_size.restype = _u64
        L.rn_oracle_conv_output_size.argtypes = [_u64] * 4
        L.rn_oracle_conv2d.argtypes = [_fp, _fp, _fp] + [_u64] * 10
        L.rn_oracle_maxpool2d.argtypes = [_fp, _fp] + [_u64] * 9
        L.rn_oracle_avgpool2d.argtypes = [_fp, _fp] + [_u64] * 9
        L.rn_oracle_linear.argtypes = [_fp, _fp, _fp, _fp] + [_u64] * 3
        L.rn_oracle_relu.argtypes = [_fp, _fp, _u64]
        L.rn_oracle_batchnorm2d.argtypes = [_fp] * 6 + [_u64] * 3
        L.rn_oracle_add.argtypes = [_fp, _fp, _fp, _u64]
        L.rn_oracle_argmax.argtypes = [_fp, _u64, _u64, ctypes.POINTER(_u64)]
        for f in ("conv2d", "maxpool2d", "avgpool2d", "linear", "relu", "batchnorm2d", "add",
                  "argmax"):
            getattr(L, "rn_oracle_" + f).restype = None
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def conv_output_size(x: int, k: int, stride: int, pad: int) -> int:
    return int(lib().rn_oracle_conv_output_size(x, k, stride, pad))


def conv2d(x: np.ndarray, w: np.ndarray, stride: int = 1, pad: int = 0) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    B, Cin, H, W = x.shape
    Cout, Cin2, k, k2 = w.shape
    assert Cin == Cin2 and k == k2
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    out = np.empty((B, Cout, ho, wo), dtype=np.float32)
    lib().rn_oracle_conv2d(_p(x), _p(out), _p(w), k, stride, pad, ho, wo, B, Cin, Cout, H, W)
    return out


def _pool(fn, x: np.ndarray, k: int, stride: int, pad: int) -> np.ndarray:
    x = _f32(x)
    B, C, H, W = x.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    out = np.empty((B, C, ho, wo), dtype=np.float32)
    fn(_p(x), _p(out), k, stride, pad, ho, wo, B, C, H, W)
    return out


def maxpool2d(x, k: int, stride: int = 1, pad: int = 0) -> np.ndarray:
    return _pool(lib().rn_oracle_maxpool2d, x, k, stride, pad)


def avgpool2d(x, k: int, stride: int = 1, pad: int = 0) -> np.ndarray:
    return _pool(lib().rn_oracle_avgpool2d, x, k, stride, pad)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    B, fin = x.shape
    fout = w.shape[0]
    out = np.empty((B, fout), dtype=np.float32)
    lib().rn_oracle_linear(_p(x), _p(out), _p(w), _p(b), B, fin, fout)
    return out


def relu_(x: np.ndarray) -> np.ndarray:
    lib().rn_oracle_relu(_p(x), _p(x), x.size)
    return x


def relu(x) -> np.ndarray:
    return relu_(_f32(x).copy())


def batchnorm2d_(x: np.ndarray, w, b, mean, var) -> np.ndarray:
    B, C = x.shape[0], x.shape[1]
    N = x.size // (B * C)
    lib().rn_oracle_batchnorm2d(_p(x), _p(x), _p(_f32(w)), _p(_f32(b)), _p(_f32(mean)),
                                _p(_f32(var)), B, C, N)
    return x


def batchnorm2d(x, w, b, mean, var) -> np.ndarray:
    return batchnorm2d_(_f32(x).copy(), w, b, mean, var)


def add_(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    assert a.shape == b.shape
    lib().rn_oracle_add(_p(a), _p(_f32(b)), _p(a), a.size)
    return a


def add(a, b) -> np.ndarray:
    return add_(_f32(a).copy(), b)


def argmax(logits: np.ndarray) -> np.ndarray:
    logits = _f32(logits)
    B, C = logits.shape
    out = np.zeros(B, dtype=np.uint64)
    lib().rn_oracle_argmax(_p(logits), B, C, out.ctypes.data_as(ctypes.POINTER(_u64)))
    return out.astype(np.int64)


# --------------------------------------------------------------------------
# graph
# --------------------------------------------------------------------------
def _bn(state: Dict[str, np.ndarray], name: str, x: np.ndarray) -> np.ndarray:
    return batchnorm2d_(x, state[name + ".weight"], state[name + ".bias"],
                        state[name + ".running_mean"], state[name + ".running_var"])


def bottleneck(state, pre: str, x: np.ndarray, stride: int, has_ds: bool,
               taps: Optional[dict] = None) -> np.ndarray:
    """One block of layerForward (main.cu:127-166)."""
    shortcut = x
    if has_ds:
        shortcut = conv2d(x, state[pre + ".downsample.0.weight"], stride, 0)
        _bn(state, pre + ".downsample.1", shortcut)
    y = conv2d(x, state[pre + ".conv1.weight"], 1, 0)
    relu_(_bn(state, pre + ".bn1", y))
    y = conv2d(y, state[pre + ".conv2.weight"], stride, 1)
    relu_(_bn(state, pre + ".bn2", y))
    y = conv2d(y, state[pre + ".conv3.weight"], 1, 0)
    _bn(state, pre + ".bn3", y)
    add_(y, shortcut)
    relu_(y)
    if taps is not None:
        taps[pre] = y
    return y


_DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def blocks_of(arch: str):
    """(prefix, stride, has_downsample) per block: createLayer, main.cu:53-89,116-119.
    Every stage changes the channel count or the stride, so block 0 always
    carries the projection shortcut (main.cu:71)."""
    for li, (n, stride) in enumerate(zip(_DEPTHS[arch], (1, 2, 2, 2)), start=1):
        for bi in range(n):
            yield f"layer{li}.{bi}", (stride if bi == 0 else 1), bi == 0


def resnet_forward(state: Dict[str, np.ndarray], x: np.ndarray, arch: str = "resnet50",
                   taps: Optional[dict] = None) -> np.ndarray:
    """Logits [B,1000] for NCHW input x, sequenced like resnet152Forward."""
    y = conv2d(x, state["conv1.weight"], 2, 3)
    relu_(_bn(state, "bn1", y))
    if taps is not None:
        taps["stem"] = y
    y = maxpool2d(y, 3, 2, 1)
    if taps is not None:
        taps["maxpool"] = y
    for pre, stride, has_ds in blocks_of(arch):
        y = bottleneck(state, pre, y, stride, has_ds, taps)
    y = avgpool2d(y, 7, 1, 0)
    y = y.reshape(y.shape[0], -1)
    if taps is not None:
        taps["avgpool"] = y
    return linear(y, state["fc.weight"], state["fc.bias"])
