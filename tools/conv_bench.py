#!/usr/bin/env python3
"""Time one convolution shape through rn_conv2d_nhwc_forward_dt for every tile candidate.

    python tools/conv_bench.py B H W Cin Cout k stride pad [--dtype bf16] [--residual] [--relu]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dims", type=int, nargs=8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--relu", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cand", type=int, default=-1, help="time this candidate only")
    ap.add_argument("--exact", action="store_true", help="small-Cin exact-K form (fp32)")
    a = ap.parse_args()
    B, H, W, Cin, Cout, k, s, p = a.dims
    lib, ctx = L.lib(), R.get_ctx()
    dt = L.RN_DTYPE_BF16 if a.dtype == "bf16" else L.RN_DTYPE_F32
    es = 2 if a.dtype == "bf16" else 4
    ho, wo = int(lib.rn_conv_output_size(H, k, s, p)), int(lib.rn_conv_output_size(W, k, s, p))
    rng = np.random.default_rng(0)
    def buf(n):
        b = _DeviceBuffer(ctx, n * es)
        h = (rng.standard_normal(n, dtype=np.float32) * 0.5)
        if es == 2:
            h = R.ops.to_bf16_bits(h)
        L.check(lib.rn_memcpy_h2d(ctx.handle, b.ptr, h.ctypes.data, h.nbytes), "h2d", ctx.handle)
        return b
    Hp, Wp = H + 2 * p, W + 2 * p
    x = buf(B * Hp * Wp * Cin) if a.exact else buf(B * H * W * max(Cin, 4))
    wn = int(lib.rn_conv2d_packed_weight_numel_exact(Cin, Cout, k)) if a.exact else \
        int(lib.rn_conv2d_packed_weight_numel_dt(dt, Cin, Cout, k))
    w = buf(wn)
    out = _DeviceBuffer(ctx, B * ho * wo * Cout * es)
    res = buf(B * ho * wo * Cout) if a.residual else None
    sc = R.FloatTensor.from_numpy(np.ones(Cout, np.float32), R.Device.GPU)
    sh = R.FloatTensor.from_numpy(np.zeros(Cout, np.float32), R.Device.GPU)
    ep = L.Epilogue(sc.data(), sh.data(), res.ptr if res else None, int(a.relu))
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.rn_event_create(ctx.handle, ctypes.byref(e0)); lib.rn_event_create(ctx.handle, ctypes.byref(e1))
    flops = 2.0 * B * ho * wo * Cout * Cin * k * k
    bytes_ = es * (B * H * W * Cin + wn + B * ho * wo * Cout * (2 if a.residual else 1))
    # the chip takes a few milliseconds of load to reach its clock: without this the first
    # candidate timed reads 15-20 % slow
    lib.rn_ctx_set_conv_tile(ctx.handle, 0)
    for _ in range(40):
        if a.exact:
            L.check(lib.rn_conv2d_nhwc_exact_forward(ctx.handle, x.ptr, out.ptr, w.ptr, k, s, ho, wo, B, Cin, Cout,
                                                     Hp, Wp, ctypes.byref(ep)), "conv", ctx.handle)
        else:
            L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, dt, dt, x.ptr, out.ptr, w.ptr, k, s, p, ho, wo, B, Cin,
                                                  Cout, H, W, ctypes.byref(ep)), "conv", ctx.handle)
    ctx.sync()
    names = ["auto", "128x128", "128x64", "64x128", "64x64", "P128x128", "P128x64", "P64x128", "P64x64",
             "W256x256", "W256x128", "W128x256", "W256x64", "W224x256", "W128x128", "strip"]
    for cand in (range(0, lib.rn_conv_tile_candidates() + 1) if a.cand < 0 else [a.cand]):
        lib.rn_ctx_set_conv_tile(ctx.handle, cand)
        def run():
            if a.exact:
                L.check(lib.rn_conv2d_nhwc_exact_forward(ctx.handle, x.ptr, out.ptr, w.ptr, k, s, ho, wo,
                                                         B, Cin, Cout, Hp, Wp, ctypes.byref(ep)), "conv", ctx.handle)
                return
            L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, dt, dt, x.ptr, out.ptr, w.ptr, k, s, p, ho, wo,
                                                  B, Cin, Cout, H, W, ctypes.byref(ep)), "conv", ctx.handle)
        run(); run()
        lib.rn_event_record(ctx.handle, e0)
        for _ in range(a.reps):
            run()
        lib.rn_event_record(ctx.handle, e1)
        ms = ctypes.c_float()
        lib.rn_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        t = ms.value / a.reps
        print(f"{names[cand]:9s} {t*1e3:8.1f} us  {flops/t/1e9:7.1f} TF/s  {bytes_/t/1e6:7.0f} GB/s")
    lib.rn_ctx_set_conv_tile(ctx.handle, 0)


if __name__ == "__main__":
    main()
