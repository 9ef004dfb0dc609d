"""Host-side mirror of the reference's tensor.cuh: Device, Shape, Tensor.

Same names and meaning as cuda/tensor.cuh:15-247; storage is a numpy array on
the CPU and an ``rn_malloc`` buffer on the GPU, shared between views like the
reference's ``shared_ptr`` (tensor.cuh:165-170).  Where the reference asserts or
aborts, this raises.

One addition: a GPU tensor carries a ``layout`` tag (NCHW like the reference, or
the engine's NHWC) so the nn layer wrappers can pick the matching kernel.
"""
from __future__ import annotations

import ctypes
import enum
import threading
from functools import reduce
from typing import Optional, Sequence

import numpy as np

from . import _lib as L


class Device(enum.Enum):  # tensor.cuh:15-19
    CPU = 0
    GPU = 1


class Shape(tuple):
    """tensor.cuh:21-57.  ``numel`` of the reference accumulates in int
    (tensor.cuh:28); Python integers do not overflow, every tensor of the
    configs is far below 2**31 anyway."""

    def __new__(cls, dims: Sequence[int] = ()):
        return super().__new__(cls, (int(d) for d in dims))

    def numel(self) -> int:
        assert len(self) != 0
        return reduce(lambda a, b: a * b, self, 1)

    def as_tuple(self, n: int) -> tuple:
        if len(self) != n:  # tensor.cuh:36-38 aborts
            raise ValueError(f"shape {tuple(self)} does not have {n} dimensions")
        return tuple(self)

    def __repr__(self) -> str:
        return "(" + ", ".join(str(d) for d in self) + ")"


# ---------------------------------------------------------------------------
# context: one per host thread (threading)
# ---------------------------------------------------------------------------
class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        lib = L.lib()
        h = ctypes.c_void_p()
        L.check(lib.rn_ctx_create(ctypes.byref(h), device, stream), "rn_ctx_create")
        self.handle = h
        self.device = device

    def sync(self) -> None:
        L.check(L.lib().rn_sync(self.handle), "rn_sync", self.handle)

    def set_layout(self, layout: int) -> None:
        L.check(L.lib().rn_ctx_set_layout(self.handle, layout), "rn_ctx_set_layout", self.handle)

    def set_sync_each_op(self, on: bool) -> None:
        L.check(L.lib().rn_ctx_set_sync_each_op(self.handle, int(on)), "rn_ctx_set_sync_each_op")

    def set_deferred(self, on: bool) -> None:
        """rn_ctx_set_deferred: the seven reference ops record their calls; the list runs when something is
        observed, with conv + in-place batch-norm / add / ReLU as one fused NHWC launch (rn_hip.h)."""
        L.check(L.lib().rn_ctx_set_deferred(self.handle, int(on)), "rn_ctx_set_deferred", self.handle)

    def flush(self) -> None:
        L.check(L.lib().rn_flush(self.handle), "rn_flush", self.handle)

    def observe(self, dev_ptr) -> None:
        L.check(L.lib().rn_observe(self.handle, dev_ptr), "rn_observe", self.handle)

    def deferred_stats(self) -> dict:
        import ctypes
        v = [ctypes.c_uint64() for _ in range(5)]
        L.check(L.lib().rn_ctx_deferred_stats(self.handle, *(ctypes.byref(x) for x in v)), "rn_ctx_deferred_stats")
        return dict(zip(("pending_ops", "nhwc_buffers", "fused_launches", "literal_launches", "transposes"),
                        (int(x.value) for x in v)))

    def set_xcd_groups(self, groups: int) -> None:
        """Tile order of the contraction over the 8 XCDs: 0 = chosen per launch, 1 = M panel major, 2 / 4 / 8 groups
        of N tiles (rn_ctx_set_xcd_groups).  Changes which block computes a tile, never a bit."""
        L.check(L.lib().rn_ctx_set_xcd_groups(self.handle, int(groups)), "rn_ctx_set_xcd_groups", self.handle)

    def set_nchw_taps(self, mode: int) -> None:
        """k x k convolutions of rn_conv2d_forward on NCHW tensors: 0 = transpose the input and contract in NHWC,
        2 = gather the taps from the channel planes, 1 = gather on large planes only (rn_ctx_set_nchw_taps).  Changes
        the route, never a bit."""
        L.check(L.lib().rn_ctx_set_nchw_taps(self.handle, int(mode)), "rn_ctx_set_nchw_taps", self.handle)

    def set_weight_cache(self, on: bool) -> None:
        """rn_conv2d_forward (OIHW weights) packs each weight buffer once instead of per call."""
        L.check(L.lib().rn_ctx_set_weight_cache(self.handle, int(on)), "rn_ctx_set_weight_cache")

    def set_split_k(self, max_splits: int) -> None:
        """Latency mode for small batches: split under-filled contractions along K."""
        L.check(L.lib().rn_ctx_set_split_k(self.handle, int(max_splits)), "rn_ctx_set_split_k",
                self.handle)

    def close(self) -> None:
        if self.handle:
            L.lib().rn_ctx_destroy(self.handle)
            self.handle = None


_tls = threading.local()


def get_ctx() -> Context:
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = Context(getattr(_tls, "device", 0))
    return ctx


def set_device(device: int) -> None:
    """Device of this thread's default context (before first use)."""
    old = getattr(_tls, "ctx", None)
    if old is not None and old.device != device:
        old.close()
        _tls.ctx = None
    _tls.device = device


class _DeviceBuffer:
    """rn_malloc'd block freed when the last view goes away (tensor.cuh:81-86)."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        p = ctypes.c_void_p()
        L.check(L.lib().rn_malloc(ctx.handle, ctypes.byref(p), nbytes), "rn_malloc", ctx.handle)
        self.ptr = p.value or 0
        self.nbytes = nbytes

    @classmethod
    def adopt(cls, ctx: Context, ptr: int, nbytes: int) -> "_DeviceBuffer":
        self = cls.__new__(cls)
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes
        return self

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                L.lib().rn_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
        self.ptr = 0


class Tensor:
    """Tensor<float> (FloatTensor, tensor.cuh:247).  ``Tensor(Device.GPU)`` is
    the empty tensor: shape (0,), no data, falsy (tensor.cuh:62-65,222-225)."""

    dtype = np.float32

    def __init__(self, shape=None, device: Device = Device.CPU, *, layout: int = L.RN_LAYOUT_NCHW,
                 _storage=None):
        if isinstance(shape, Device):  # Tensor(Device.GPU)
            device, shape = shape, None
        self.device = device
        self.layout = layout
        if shape is None:
            self._shape = Shape((0,))
            self._storage = None
            return
        self._shape = Shape(shape)
        assert len(self._shape) != 0
        if _storage is not None:
            self._storage = _storage
        elif self.numel() == 0:
            self._storage = None
        elif device == Device.CPU:
            self._storage = np.empty(self.numel(), dtype=np.float32)
        else:
            self._storage = _DeviceBuffer(get_ctx(), self.size())

    # -- factories (tensor.cuh:126-152) ---------------------------------
    @staticmethod
    def loadToCpu(file_name: str) -> "Tensor":
        try:
            arr = np.fromfile(file_name, dtype=np.float32)
        except OSError as e:  # reference: "Can't open" + abort (tensor.cuh:129-132)
            raise FileNotFoundError(f"Can't open {file_name}") from e
        assert arr.size > 0
        return Tensor((arr.size,), Device.CPU, _storage=arr)

    @staticmethod
    def loadToCuda(file_name: str) -> "Tensor":
        ctx = get_ctx()
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        L.check(L.lib().rn_load_f32_file(ctx.handle, file_name.encode(), ctypes.byref(p),
                                         ctypes.byref(n)), "rn_load_f32_file", ctx.handle)
        return Tensor((n.value,), Device.GPU,
                      _storage=_DeviceBuffer.adopt(ctx, p.value, n.value * 4))

    @staticmethod
    def from_numpy(arr: np.ndarray, device: Device = Device.CPU, layout: int = L.RN_LAYOUT_NCHW):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        t = Tensor(a.shape if a.ndim else (1,), Device.CPU, _storage=a.reshape(-1).copy())
        return t.cuda(layout) if device == Device.GPU else t

    # -- reference methods ------------------------------------------------
    def save(self, file_name: str) -> None:  # tensor.cuh:154-163
        assert self.device == Device.CPU
        self._storage.tofile(file_name)

    def view(self, new_shape) -> "Tensor":  # tensor.cuh:165-170: shares storage
        new_shape = Shape(new_shape)
        assert len(new_shape) != 0
        assert self._shape.numel() == new_shape.numel()
        return Tensor(new_shape, self.device, layout=self.layout, _storage=self._storage)

    def numel(self) -> int:
        return self._shape.numel()

    def size(self) -> int:
        return self.numel() * 4

    def toDevice(self, new_device: Device, layout: Optional[int] = None) -> "Tensor":
        # tensor.cuh:184-199: CPU<->GPU only, synchronous
        ctx = get_ctx()
        ctx.sync()
        ret = Tensor(self._shape, new_device, layout=self.layout if layout is None else layout)
        if self.device == Device.CPU and new_device == Device.GPU:
            if self.numel():
                L.check(L.lib().rn_memcpy_h2d(ctx.handle, ret.data(), self._storage.ctypes.data,
                                              self.size()), "rn_memcpy_h2d", ctx.handle)
        elif self.device == Device.GPU and new_device == Device.CPU:
            if self.numel():
                L.check(L.lib().rn_memcpy_d2h(ctx.handle, ret._storage.ctypes.data, self.data(),
                                              self.size()), "rn_memcpy_d2h", ctx.handle)
        else:
            raise RuntimeError("Unsupported device transfer combination")
        return ret

    def cuda(self, layout: Optional[int] = None) -> "Tensor":
        return self.toDevice(Device.GPU, layout)

    def cpu(self) -> "Tensor":
        return self.toDevice(Device.CPU)

    def __bool__(self) -> bool:
        return self._storage is not None

    def shape(self) -> Shape:
        return self._shape

    def data(self):
        """Raw address: device pointer (GPU) or host pointer (CPU); None when empty."""
        if self._storage is None:
            return None
        if self.device == Device.GPU:
            return self._storage.ptr
        return self._storage.ctypes.data

    def numpy(self) -> np.ndarray:
        """Host copy with this tensor's shape (NCHW tensors only)."""
        t = self if self.device == Device.CPU else self.cpu()
        return t._storage.reshape(tuple(self._shape)).copy()

    def move_from(self, other: "Tensor") -> None:
        """operator=(Tensor&&) of the reference (tensor.cuh:212-220)."""
        assert self.device == other.device
        self._storage, self._shape, self.layout = other._storage, other._shape, other.layout
        assert len(self._shape) != 0
        other._storage, other._shape = None, Shape((0,))


FloatTensor = Tensor
