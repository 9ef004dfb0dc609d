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


def conv2d_nhwc_exact(x, w, stride=1, pad=0, scale=None, shift=None, residual=None,
                      relu_: bool = False) -> np.ndarray:
    """Small-Cin convolution in the exact-K form: rn_nchw_to_nhwc_pad_dt (physical border) +
    rn_conv2d_pack_weight_exact + rn_conv2d_nhwc_exact_forward.  NCHW fp32 host arrays."""
    ctx, lib = get_ctx(), L.lib()
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Hp, Wp = H + 2 * pad, W + 2 * pad
    ho, wo = conv_output_size(Hp, k, stride, 0), conv_output_size(Wp, k, stride, 0)
    dx32 = _up(np.asarray(x, dtype=np.float32), "nchw")
    dx = FloatTensor((B, Hp, Wp, Cin), Device.GPU)
    L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, L.RN_DTYPE_F32, dx32.data(), dx.data(), B, Cin,
                                       H, W, Cin, pad), "rn_nchw_to_nhwc_pad_dt", ctx.handle)
    dw = _up(w, "nchw")
    packed = FloatTensor((int(lib.rn_conv2d_packed_weight_numel_exact(Cin, Cout, k)),), Device.GPU)
    L.check(lib.rn_conv2d_pack_weight_exact(ctx.handle, dw.data(), packed.data(), Cin, Cout, k),
            "rn_conv2d_pack_weight_exact", ctx.handle)
    keep = [_up(v, "nchw") if v is not None else None for v in (scale, shift)]
    dres = _up(residual, "nhwc") if residual is not None else None
    ep = L.Epilogue(keep[0].data() if keep[0] else None, keep[1].data() if keep[1] else None,
                    dres.data() if dres else None, int(relu_))
    out = FloatTensor((B, Cout, ho, wo), Device.GPU)
    L.check(lib.rn_conv2d_nhwc_exact_forward(ctx.handle, dx.data(), out.data(), packed.data(), k,
                                             stride, ho, wo, B, Cin, Cout, Hp, Wp, ctypes.byref(ep)),
            "rn_conv2d_nhwc_exact_forward", ctx.handle)
    ctx.sync()
    return _down(out, (B, Cout, ho, wo), "nhwc")


# ---------------------------------------------------------------------------
# bf16 storage (element-type tagged entry points)
# ---------------------------------------------------------------------------
def to_bf16_bits(a) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def from_bf16_bits(b) -> np.ndarray:
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def bf16_round(a) -> np.ndarray:
    """fp32 values rounded to the nearest bf16, still stored as fp32."""
    return from_bf16_bits(to_bf16_bits(a)).reshape(np.shape(a))


def _up_raw(arr: np.ndarray):
    from .tensor import _DeviceBuffer
    ctx = get_ctx()
    arr = np.ascontiguousarray(arr)
    buf = _DeviceBuffer(ctx, max(arr.nbytes, 16))
    L.check(L.lib().rn_memcpy_h2d(ctx.handle, buf.ptr, arr.ctypes.data, arr.nbytes), "h2d", ctx.handle)
    return buf


def _down_raw(buf, dtype, count: int) -> np.ndarray:
    ctx = get_ctx()
    out = np.empty(count, dtype=dtype)
    L.check(L.lib().rn_memcpy_d2h(ctx.handle, out.ctypes.data, buf.ptr, out.nbytes), "d2h", ctx.handle)
    return out


def conv2d_nhwc_bf16(x, w, stride=1, pad=0, scale=None, shift=None, residual=None,
                     relu_: bool = False, out_f32: bool = False) -> np.ndarray:
    """bf16 activations/weights through rn_conv2d_pack_weight_dt + rn_conv2d_nhwc_forward_dt.
    NCHW fp32 host arrays in (rounded to bf16 on upload), NCHW fp32 host array out.  The
    small-Cin stem form gets a physically zero-padded image, as the model driver builds it."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    c4 = Cin <= 4 and k <= 8
    dx32 = _up(np.asarray(x, dtype=np.float32), "nchw")
    if c4:
        Hp, Wp = H + 2 * pad, W + 2 * pad
        dx = _DeviceBuffer(ctx, B * Hp * Wp * 4 * 2)
        L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, L.RN_DTYPE_BF16, dx32.data(), dx.ptr, B, Cin,
                                           H, W, 4, pad), "pad_dt", ctx.handle)
        Hin, Win, pad_arg = Hp, Wp, 0
    else:
        dx = _up_raw(to_bf16_bits(np.asarray(x, dtype=np.float32).transpose(0, 2, 3, 1)))
        Hin, Win, pad_arg = H, W, pad
    dw = _up(w, "nchw")
    pn = int(lib.rn_conv2d_packed_weight_numel_dt(L.RN_DTYPE_BF16, Cin, Cout, k))
    packed = _DeviceBuffer(ctx, pn * 2)
    L.check(lib.rn_conv2d_pack_weight_dt(ctx.handle, L.RN_DTYPE_BF16, dw.data(), packed.ptr, Cin,
                                         Cout, k), "pack_dt", ctx.handle)
    keep = [_up(v, "nchw") if v is not None else None for v in (scale, shift)]
    dres = None
    if residual is not None:
        r = np.asarray(residual, dtype=np.float32).transpose(0, 2, 3, 1)
        dres = _up_raw(r if out_f32 else to_bf16_bits(r))
    ep = L.Epilogue(keep[0].data() if keep[0] else None, keep[1].data() if keep[1] else None,
                    dres.ptr if dres else None, int(relu_))
    n_out = B * Cout * ho * wo
    out = _DeviceBuffer(ctx, n_out * (4 if out_f32 else 2))
    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, L.RN_DTYPE_BF16,
                                          L.RN_DTYPE_F32 if out_f32 else L.RN_DTYPE_BF16, dx.ptr,
                                          out.ptr, packed.ptr, k, stride, pad_arg, ho, wo, B, Cin,
                                          Cout, Hin, Win, ctypes.byref(ep)),
            "rn_conv2d_nhwc_forward_dt", ctx.handle)
    ctx.sync()
    if out_f32:
        y = _down_raw(out, np.float32, n_out)
    else:
        y = from_bf16_bits(_down_raw(out, np.uint16, n_out))
    return y.reshape(B, ho, wo, Cout).transpose(0, 3, 1, 2).copy()


def conv_chain_bf16(t2, x, w3, scale3, shift3, w1, scale1, shift1):
    """rn_conv_chain_forward_dt: relu(bn3(conv1x1(t2, w3)) + x) -> y and relu(bn1(conv1x1(y, w1)))
    -> t1 as one launch (bf16 storage).  NCHW fp32 host arrays in (rounded to bf16 on upload);
    returns (y, t1) as NCHW fp32 host arrays."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cm, H, W = t2.shape
    C, N1 = w3.shape[0], w1.shape[0]
    rows = B * H * W

    def up_act(a):
        return _up_raw(to_bf16_bits(np.asarray(a, dtype=np.float32).transpose(0, 2, 3, 1)))

    def pack(w, cin, cout):
        dw = _up(w, "nchw")
        pk = _DeviceBuffer(ctx, int(lib.rn_conv2d_packed_weight_numel_dt(L.RN_DTYPE_BF16, cin, cout, 1)) * 2)
        L.check(lib.rn_conv2d_pack_weight_dt(ctx.handle, L.RN_DTYPE_BF16, dw.data(), pk.ptr, cin, cout, 1),
                "pack_dt", ctx.handle)
        return pk

    dt2, dx = up_act(t2), up_act(x)
    p3, p1 = pack(w3, Cm, C), pack(w1, C, N1)
    keep = [_up(np.asarray(v, dtype=np.float32), "nchw") if v is not None else None
            for v in (scale3, shift3, scale1, shift1)]
    ptr = [k.data() if k else None for k in keep]
    y, t1 = _DeviceBuffer(ctx, rows * C * 2), _DeviceBuffer(ctx, rows * N1 * 2)
    L.check(lib.rn_conv_chain_forward_dt(ctx.handle, L.RN_DTYPE_BF16, dt2.ptr, dx.ptr, y.ptr, p3.ptr, ptr[0],
                                         ptr[1], t1.ptr, p1.ptr, ptr[2], ptr[3], rows, Cm, C, N1),
            "rn_conv_chain_forward_dt", ctx.handle)
    ctx.sync()
    yh = from_bf16_bits(_down_raw(y, np.uint16, rows * C)).reshape(B, H, W, C).transpose(0, 3, 1, 2).copy()
    th = from_bf16_bits(_down_raw(t1, np.uint16, rows * N1)).reshape(B, H, W, N1).transpose(0, 3, 1, 2).copy()
    return yh, th


def conv_chain_f32(t2, x, w3, scale3, shift3, w1, scale1, shift1):
    """rn_conv_chain_forward_dt with fp32 storage (64 -> 256 -> 64 | 128 channels): (y, t1) as NCHW
    fp32 host arrays."""
    ctx, lib = get_ctx(), L.lib()
    B, Cm, H, W = t2.shape
    C, N1 = w3.shape[0], w1.shape[0]
    rows = B * H * W

    def pack(w, cin, cout):
        dw = _up(w, "nchw")
        pk = FloatTensor((int(lib.rn_conv2d_packed_weight_numel(cin, cout, 1)),), Device.GPU)
        L.check(lib.rn_conv2d_pack_weight(ctx.handle, dw.data(), pk.data(), cin, cout, 1), "pack", ctx.handle)
        return pk

    dt2, dx = _up(np.asarray(t2, dtype=np.float32), "nhwc"), _up(np.asarray(x, dtype=np.float32), "nhwc")
    p3, p1 = pack(w3, Cm, C), pack(w1, C, N1)
    keep = [_up(np.asarray(v, dtype=np.float32), "nchw") if v is not None else None
            for v in (scale3, shift3, scale1, shift1)]
    ptr = [k.data() if k else None for k in keep]
    y, t1 = FloatTensor((B, C, H, W), Device.GPU), FloatTensor((B, N1, H, W), Device.GPU)
    L.check(lib.rn_conv_chain_forward_dt(ctx.handle, L.RN_DTYPE_F32, dt2.data(), dx.data(), y.data(), p3.data(),
                                         ptr[0], ptr[1], t1.data(), p1.data(), ptr[2], ptr[3], rows, Cm, C, N1),
            "rn_conv_chain_forward_dt", ctx.handle)
    ctx.sync()
    return _down(y, (B, C, H, W), "nhwc"), _down(t1, (B, N1, H, W), "nhwc")


def conv_chain_pair(t2, x2, w3, scale3, wd, scaled, shift, w1, scale1, shift1, bf16: bool = True):
    """rn_conv_chain_pair_forward_dt: relu(conv1x1(t2, w3*scale3) + conv1x1(x2, wd*scaled) + shift) -> y
    and relu(bn1(conv1x1(y, w1))) -> t1 as one launch.  NCHW fp32 host arrays; returns (y, t1)."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cm, H, W = t2.shape
    C, C2, N1 = w3.shape[0], x2.shape[1], w1.shape[0]
    rows = B * H * W
    dt, es = (L.RN_DTYPE_BF16, 2) if bf16 else (L.RN_DTYPE_F32, 4)

    def up_act(a):
        a = np.asarray(a, dtype=np.float32).transpose(0, 2, 3, 1)
        return _up_raw(to_bf16_bits(a) if bf16 else a)

    dt2, dx2 = up_act(t2), up_act(x2)
    dw3, dwd, dw1 = _up(w3, "nchw"), _up(wd, "nchw"), _up(w1, "nchw")
    keep = [_up(np.asarray(v, dtype=np.float32), "nchw") if v is not None else None
            for v in (scale3, scaled, shift, scale1, shift1)]
    ptr = [k.data() if k else None for k in keep]
    pp = _DeviceBuffer(ctx, int(lib.rn_conv2d_packed_pair_weight_numel(Cm, C, 1, C2)) * es)
    L.check(lib.rn_conv2d_pack_weight_pair_dt(ctx.handle, dt, dw3.data(), ptr[0], dwd.data(), ptr[1],
                                              pp.ptr, Cm, C, 1, C2), "pack_pair", ctx.handle)
    p1 = _DeviceBuffer(ctx, int(lib.rn_conv2d_packed_weight_numel_dt(dt, C, N1, 1)) * es)
    L.check(lib.rn_conv2d_pack_weight_dt(ctx.handle, dt, dw1.data(), p1.ptr, C, N1, 1), "pack_dt", ctx.handle)
    y, t1 = _DeviceBuffer(ctx, rows * C * es), _DeviceBuffer(ctx, rows * N1 * es)
    L.check(lib.rn_conv_chain_pair_forward_dt(ctx.handle, dt, dt2.ptr, dx2.ptr, y.ptr, pp.ptr, ptr[2],
                                              t1.ptr, p1.ptr, ptr[3], ptr[4], rows, Cm, C2, C, N1),
            "rn_conv_chain_pair_forward_dt", ctx.handle)
    ctx.sync()

    def down(b, n, c):
        h = from_bf16_bits(_down_raw(b, np.uint16, n)) if bf16 else _down_raw(b, np.float32, n)
        return h.reshape(B, H, W, c).transpose(0, 3, 1, 2).copy()

    return down(y, rows * C, C), down(t1, rows * N1, N1)


def conv_chain_pair_bf16(*a):
    return conv_chain_pair(*a, bf16=True)


def conv2d_nhwc_pair(x, w, x2, w2, stride=1, pad=0, stride2=1, scale=None, scale2=None, shift=None,
                     residual=None, relu_: bool = False, bf16: bool = False) -> np.ndarray:
    """epilogue(conv(x, w * scale) + conv1x1(x2, w2 * scale2)) as ONE contraction:
    rn_conv2d_pack_weight_pair_dt + rn_conv2d_nhwc_pair_forward_dt.  NCHW fp32 host arrays."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    _, Cin2, H2, W2 = x2.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    dt, es = (L.RN_DTYPE_BF16, 2) if bf16 else (L.RN_DTYPE_F32, 4)

    def up_act(a):
        a = np.asarray(a, dtype=np.float32).transpose(0, 2, 3, 1)
        return _up_raw(to_bf16_bits(a) if bf16 else a)

    dx, dx2 = up_act(x), up_act(x2)
    dw, dw2 = _up(w, "nchw"), _up(w2, "nchw")
    keep = [_up(v, "nchw") if v is not None else None for v in (scale, scale2, shift)]
    pn = int(lib.rn_conv2d_packed_pair_weight_numel(Cin, Cout, k, Cin2))
    packed = _DeviceBuffer(ctx, pn * es)
    L.check(lib.rn_conv2d_pack_weight_pair_dt(ctx.handle, dt, dw.data(),
                                              keep[0].data() if keep[0] else None, dw2.data(),
                                              keep[1].data() if keep[1] else None, packed.ptr, Cin,
                                              Cout, k, Cin2), "pack_pair", ctx.handle)
    dres = up_act(residual) if residual is not None else None
    ep = L.Epilogue(None, keep[2].data() if keep[2] else None, dres.ptr if dres else None,
                    int(relu_))
    second = L.ConvSecond(dx2.ptr, Cin2, H2, W2, stride2)
    n_out = B * Cout * ho * wo
    out = _DeviceBuffer(ctx, n_out * es)
    L.check(lib.rn_conv2d_nhwc_pair_forward_dt(ctx.handle, dt, dt, dx.ptr, out.ptr, packed.ptr, k,
                                               stride, pad, ho, wo, B, Cin, Cout, H, W,
                                               ctypes.byref(second), ctypes.byref(ep)),
            "rn_conv2d_nhwc_pair_forward_dt", ctx.handle)
    ctx.sync()
    y = from_bf16_bits(_down_raw(out, np.uint16, n_out)) if bf16 else _down_raw(out, np.float32, n_out)
    return y.reshape(B, ho, wo, Cout).transpose(0, 3, 1, 2).copy()


def pool_nhwc_bf16(x, k, stride=1, pad=0, is_max=True) -> np.ndarray:
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, C, H, W = x.shape
    ho, wo = conv_output_size(H, k, stride, pad), conv_output_size(W, k, stride, pad)
    dx = _up_raw(to_bf16_bits(np.asarray(x, dtype=np.float32).transpose(0, 2, 3, 1)))
    out = _DeviceBuffer(ctx, B * C * ho * wo * 2)
    fn = lib.rn_maxpool2d_nhwc_forward_dt if is_max else lib.rn_avgpool2d_nhwc_forward_dt
    L.check(fn(ctx.handle, L.RN_DTYPE_BF16, dx.ptr, out.ptr, k, stride, pad, ho, wo, B, C, H, W),
            "pool_dt", ctx.handle)
    ctx.sync()
    y = from_bf16_bits(_down_raw(out, np.uint16, B * C * ho * wo))
    return y.reshape(B, ho, wo, C).transpose(0, 3, 1, 2).copy()


def stem_pool(x, w, scale=None, shift=None, relu_: bool = True, bf16: bool = False,
              from_nchw: bool = False) -> np.ndarray:
    """conv 7x7/2 pad 3 + per-channel affine + ReLU + max-pool 3x3/2/1 through the fused launch
    (rn_stem_pool_pack_weight_dt + rn_nchw_to_nhwc_pad_dt + rn_stem_pool_forward_dt, or with
    ``from_nchw`` rn_stem_pool_nchw_forward_dt on the NCHW image itself).  NCHW fp32 host arrays
    in, NCHW fp32 host array [B,64,PH,PW] out."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cin, H, W = x.shape
    assert w.shape == (64, Cin, 7, 7)
    dt, es, cpad = (L.RN_DTYPE_BF16, 2, 4) if bf16 else (L.RN_DTYPE_F32, 4, 3)
    Hp, Wp = H + 6, W + 6
    ho, wo = conv_output_size(Hp, 7, 2, 0), conv_output_size(Wp, 7, 2, 0)
    ph, pw = conv_output_size(ho, 3, 2, 1), conv_output_size(wo, 3, 2, 1)
    xin = FloatTensor.from_numpy(np.ascontiguousarray(x, dtype=np.float32), Device.GPU)
    if not from_nchw:
        xp = _DeviceBuffer(ctx, B * Hp * Wp * cpad * es)
        L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, dt, xin.data(), xp.ptr, B, Cin, H, W, cpad, 3), "pad", ctx.handle)
    wd = FloatTensor.from_numpy(np.ascontiguousarray(w, dtype=np.float32), Device.GPU)
    wp = _DeviceBuffer(ctx, int(lib.rn_stem_pool_packed_weight_numel(dt)) * es)
    L.check(lib.rn_stem_pool_pack_weight_dt(ctx.handle, dt, wd.data(), wp.ptr, Cin), "pack", ctx.handle)
    sc = FloatTensor.from_numpy(np.asarray(scale, dtype=np.float32), Device.GPU) if scale is not None else None
    sh = FloatTensor.from_numpy(np.asarray(shift, dtype=np.float32), Device.GPU) if shift is not None else None
    out = _DeviceBuffer(ctx, B * ph * pw * 64 * es)
    if from_nchw:
        L.check(lib.rn_stem_pool_nchw_forward_dt(ctx.handle, dt, xin.data(), out.ptr, wp.ptr,
                                                 sc.data() if sc else None, sh.data() if sh else None,
                                                 int(relu_), B, Cin, H, W), "stem_pool_nchw", ctx.handle)
    else:
        L.check(lib.rn_stem_pool_forward_dt(ctx.handle, dt, xp.ptr, out.ptr, wp.ptr, sc.data() if sc else None,
                                            sh.data() if sh else None, int(relu_), B, Hp, Wp), "stem_pool", ctx.handle)
    host = np.empty(B * ph * pw * 64, dtype=np.uint16 if bf16 else np.float32)
    L.check(lib.rn_memcpy_d2h(ctx.handle, host.ctypes.data, out.ptr, host.nbytes), "d2h", ctx.handle)
    if bf16:
        host = (host.astype(np.uint32) << 16).view(np.float32)
    return np.ascontiguousarray(host.reshape(B, ph, pw, 64).transpose(0, 3, 1, 2))


def stem_conv_pool(x, w, scale=None, shift=None):
    """rn_stem_conv_pool_nchw_forward: the fused stem launch that also writes the stem tensor.  NCHW fp32 host
    arrays in; (stem [B,64,Ho,Wo], pooled [B,64,PH,PW]) NCHW fp32 host arrays out."""
    from .tensor import _DeviceBuffer
    ctx, lib = get_ctx(), L.lib()
    B, Cin, H, W = x.shape
    assert w.shape == (64, Cin, 7, 7)
    ho, wo = conv_output_size(H + 6, 7, 2, 0), conv_output_size(W + 6, 7, 2, 0)
    ph, pw = conv_output_size(ho, 3, 2, 1), conv_output_size(wo, 3, 2, 1)
    xin = FloatTensor.from_numpy(np.ascontiguousarray(x, dtype=np.float32), Device.GPU)
    wd = FloatTensor.from_numpy(np.ascontiguousarray(w, dtype=np.float32), Device.GPU)
    wp = _DeviceBuffer(ctx, int(lib.rn_stem_pool_packed_weight_numel(L.RN_DTYPE_F32)) * 4)
    L.check(lib.rn_stem_pool_pack_weight_dt(ctx.handle, L.RN_DTYPE_F32, wd.data(), wp.ptr, Cin), "pack", ctx.handle)
    sc = FloatTensor.from_numpy(np.asarray(scale, dtype=np.float32), Device.GPU) if scale is not None else None
    sh = FloatTensor.from_numpy(np.asarray(shift, dtype=np.float32), Device.GPU) if shift is not None else None
    y, out = _DeviceBuffer(ctx, B * ho * wo * 64 * 4), _DeviceBuffer(ctx, B * ph * pw * 64 * 4)
    L.check(lib.rn_memset(ctx.handle, y.ptr, 0xFF, B * ho * wo * 64 * 4), "memset", ctx.handle)   # NaNs: every element must be written
    L.check(lib.rn_stem_conv_pool_nchw_forward(ctx.handle, xin.data(), y.ptr, out.ptr, wp.ptr, sc.data() if sc else None,
                                               sh.data() if sh else None, B, Cin, H, W), "stem_conv_pool", ctx.handle)
    hy, hp = np.empty(B * ho * wo * 64, dtype=np.float32), np.empty(B * ph * pw * 64, dtype=np.float32)
    L.check(lib.rn_memcpy_d2h(ctx.handle, hy.ctypes.data, y.ptr, hy.nbytes), "d2h", ctx.handle)
    L.check(lib.rn_memcpy_d2h(ctx.handle, hp.ctypes.data, out.ptr, hp.nbytes), "d2h", ctx.handle)
    return (np.ascontiguousarray(hy.reshape(B, ho, wo, 64).transpose(0, 3, 1, 2)),
            np.ascontiguousarray(hp.reshape(B, ph, pw, 64).transpose(0, 3, 1, 2)))
