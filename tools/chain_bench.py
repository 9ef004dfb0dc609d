#!/usr/bin/env python3
"""conv3 -> conv1 chain (rn_conv_chain_forward_dt) against the two launches it replaces, bf16, at B
images: the stage-1 shape (mid 64, 56x56), stage 2 (mid 128, 28x28) or stage 3 (mid 256, 14x14).

    python tools/chain_bench.py [--batch 256] [--mid 64|128|256] [--next-mid 64|128] [--reps 30]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--mid", type=int, default=64)
ap.add_argument("--next-mid", type=int, default=0, help="default: mid")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
MID = a.mid
C = 4 * MID
B, H, W, N1 = a.batch, {64: 56, 128: 28, 256: 14}[MID], {64: 56, 128: 28, 256: 14}[MID], a.next_mid or MID
rows = B * H * W
lib, ctx = L.lib(), R.get_ctx()
rng = np.random.default_rng(0)
def buf(n, scale=0.5):
    b = _DeviceBuffer(ctx, n * 2)
    h = R.ops.to_bf16_bits(rng.standard_normal(n, dtype=np.float32) * scale)
    L.check(lib.rn_memcpy_h2d(ctx.handle, b.ptr, h.ctypes.data, h.nbytes), "h2d", ctx.handle)
    return b
t2, x = buf(rows * MID), buf(rows * C)
w3, w1 = buf(C * MID, 0.1), buf(N1 * C, 0.06)
y, t1 = _DeviceBuffer(ctx, rows * C * 2), _DeviceBuffer(ctx, rows * N1 * 2)
one = R.FloatTensor.from_numpy(np.ones(C, np.float32), R.Device.GPU)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
lib.rn_event_create(ctx.handle, ctypes.byref(e0)); lib.rn_event_create(ctx.handle, ctypes.byref(e1))
BF = L.RN_DTYPE_BF16
ep3 = L.Epilogue(one.data(), one.data(), x.ptr, 1)
ep1 = L.Epilogue(one.data(), one.data(), None, 1)

def separate():
    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, BF, BF, t2.ptr, y.ptr, w3.ptr, 1, 1, 0, H, W, B, MID, C, H, W,
                                          ctypes.byref(ep3)), "conv3", ctx.handle)
    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, BF, BF, y.ptr, t1.ptr, w1.ptr, 1, 1, 0, H, W, B, C, N1, H, W,
                                          ctypes.byref(ep1)), "conv1", ctx.handle)

def chained():
    L.check(lib.rn_conv_chain_forward_dt(ctx.handle, BF, t2.ptr, x.ptr, y.ptr, w3.ptr, one.data(), one.data(),
                                         t1.ptr, w1.ptr, one.data(), one.data(), rows, MID, C, N1), "chain", ctx.handle)

def timed(f):
    for _ in range(5): f()
    lib.rn_event_record(ctx.handle, e0)
    for _ in range(a.reps): f()
    lib.rn_event_record(ctx.handle, e1)
    ms = ctypes.c_float(); lib.rn_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / a.reps

for _ in range(40): separate()
best = None
for c3 in (0, 1, 2, 5, 6, 9, 10, 13, 14):
    lib.rn_ctx_set_conv_tile(ctx.handle, c3)
    ms = timed(separate)
    best = ms if best is None or ms < best else best
    print(f"separate, tile candidate {c3:2d} for both: {ms*1e3:8.1f} us")
lib.rn_ctx_set_conv_tile(ctx.handle, 0)
ms = timed(chained)
gb = (rows * (MID + C + C + N1) * 2) / 1e9
print(f"chained: {ms*1e3:8.1f} us   {gb/ms:6.2f} TB/s algorithmic   (best separate {best*1e3:.1f} us)")
