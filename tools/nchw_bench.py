#!/usr/bin/env python3
"""The convolutions of ResNet-50 on the literal drop-in route (rn_conv2d_forward on NCHW tensors,
NCHW-native kernel; RN_NCHW_TAPS=0 in the environment puts the 3x3 layers back on the route that
transposes their input through scratch) next to the engine's NHWC contraction of the same shape,
and the NCHW batch-norm of each activation shape.

    python tools/nchw_bench.py [--batch 256] [--reps 10]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer

# (H, Cin, Cout, kernel, stride, count in the network)
SHAPES = [
    (56, 64, 64, 1, 1, 1), (56, 64, 256, 1, 1, 4), (56, 256, 64, 1, 1, 2), (56, 256, 128, 1, 1, 1), (56, 256, 512, 1, 2, 1),
    (28, 128, 512, 1, 1, 4), (28, 512, 128, 1, 1, 3), (28, 512, 256, 1, 1, 1), (28, 512, 1024, 1, 2, 1),
    (14, 256, 1024, 1, 1, 6), (14, 1024, 256, 1, 1, 5), (14, 1024, 512, 1, 1, 1), (14, 1024, 2048, 1, 2, 1),
    (7, 512, 2048, 1, 1, 3), (7, 2048, 512, 1, 1, 2),
    (56, 64, 64, 3, 1, 3), (56, 128, 128, 3, 2, 1), (28, 128, 128, 3, 1, 3), (28, 256, 256, 3, 2, 1),
    (14, 256, 256, 3, 1, 5), (14, 512, 512, 3, 2, 1), (7, 512, 512, 3, 1, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    B = a.batch
    lib, ctx = L.lib(), R.get_ctx()
    rng = np.random.default_rng(0)

    def buf(n):
        b = _DeviceBuffer(ctx, n * 4)
        h = rng.standard_normal(n, dtype=np.float32) * 0.5
        L.check(lib.rn_memcpy_h2d(ctx.handle, b.ptr, h.ctypes.data, h.nbytes), "h2d", ctx.handle)
        return b

    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.rn_event_create(ctx.handle, ctypes.byref(e0))
    lib.rn_event_create(ctx.handle, ctypes.byref(e1))

    def timed(run):
        for _ in range(3):
            run()
        lib.rn_event_record(ctx.handle, e0)
        for _ in range(a.reps):
            run()
        lib.rn_event_record(ctx.handle, e1)
        ms = ctypes.c_float()
        lib.rn_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        return ms.value / a.reps

    ctx.set_deferred(False)  # every call is a launch of its own here
    ctx.set_weight_cache(True)  # the 3x3 panels packed once, as the veneer runs
    tot = {1: [0.0, 0.0], 3: [0.0, 0.0]}
    print(f"RN_NCHW_TAPS={os.environ.get('RN_NCHW_TAPS', '1')}")
    print(f"{'shape':34s} {'x':>2s} {'NCHW us':>9s} {'TF/s':>7s} {'GB/s':>6s} {'NHWC us':>9s} {'TF/s':>7s}")
    for H, Cin, Cout, k, s, count in SHAPES:
        pad = k // 2
        Ho = (H + 2 * pad - k) // s + 1
        x = buf(B * Cin * H * H)
        w = buf(Cout * Cin * k * k)
        out = _DeviceBuffer(ctx, B * Cout * Ho * Ho * 4)
        wp = _DeviceBuffer(ctx, int(lib.rn_conv2d_packed_weight_numel_dt(L.RN_DTYPE_F32, Cin, Cout, k)) * 4)
        L.check(lib.rn_conv2d_pack_weight_dt(ctx.handle, L.RN_DTYPE_F32, w.ptr, wp.ptr, Cin, Cout, k), "pack", ctx.handle)
        flops = 2.0 * B * Ho * Ho * Cout * Cin * k * k
        bytes_ = 4.0 * (B * Cin * H * H / (s * s if k == 1 else 1) + Cout * Cin * k * k + B * Cout * Ho * Ho)

        def nchw():
            ctx.set_layout(L.RN_LAYOUT_NCHW)
            L.check(lib.rn_conv2d_forward(ctx.handle, x.ptr, out.ptr, w.ptr, k, s, pad, Ho, Ho, B, Cin, Cout, H, H),
                    "conv", ctx.handle)

        def nhwc():
            L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, L.RN_DTYPE_F32, L.RN_DTYPE_F32, x.ptr, out.ptr, wp.ptr,
                                                  k, s, pad, Ho, Ho, B, Cin, Cout, H, H, None), "conv", ctx.handle)

        tn, th = timed(nchw), timed(nhwc)
        tot[k][0] += tn * count
        tot[k][1] += th * count
        print(f"{H:3d}x{H:<3d} {Cin:4d} -> {Cout:4d} {k}x{k} stride {s}  {count:2d} {tn*1e3:9.1f} {flops/tn/1e9:7.1f} "
              f"{bytes_/tn/1e6:6.0f} {th*1e3:9.1f} {flops/th/1e9:7.1f}")
    for k in (1, 3):
        print(f"network total, {k}x{k} layers: NCHW {tot[k][0]:.2f} ms, NHWC {tot[k][1]:.2f} ms")

    print("batch-norm on NCHW planes (in place):")
    for H, C, count in ((112, 64, 1), (56, 64, 6), (56, 256, 4), (28, 128, 8), (28, 512, 5), (14, 256, 12),
                        (14, 1024, 7), (7, 512, 6), (7, 2048, 4)):
        x = buf(B * C * H * H)
        prm = [R.FloatTensor.from_numpy((rng.random(C, dtype=np.float32) + 0.5), R.Device.GPU) for _ in range(4)]

        def bn():
            ctx.set_layout(L.RN_LAYOUT_NCHW)
            L.check(lib.rn_batchnorm2d_forward(ctx.handle, x.ptr, x.ptr, *(t.data() for t in prm), B, C, H * H),
                    "bn", ctx.handle)

        t = timed(bn)
        print(f"{H:3d}x{H:<3d} {C:4d} ch  {count:2d}x  {t*1e3:8.1f} us  {8.0*B*C*H*H/t/1e6:6.0f} GB/s")
    ctx.set_layout(L.RN_LAYOUT_NCHW)


if __name__ == "__main__":
    main()
