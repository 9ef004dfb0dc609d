#!/usr/bin/env python3
"""The fused stem + max-pool launch alone (rn_stem_pool_forward_dt) at B images of 224x224:
microseconds per launch by HIP events, matrix rate and algorithmic GB/s.

    python tools/stem_bench.py [--batch 256] [--dtype f32|bf16|both] [--reps 30]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="both")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
B = a.batch
lib, ctx = L.lib(), R.get_ctx()
x = R.weights.generate_input(min(B, 32), seed=5)
x = np.concatenate([x] * ((B + 31) // 32))[:B]
w = R.weights.generate_tensor("conv1.weight", (64, 3, 7, 7), 0)
xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
wd = R.FloatTensor.from_numpy(w, R.Device.GPU)
sc = R.FloatTensor.from_numpy(np.full(64, 1.1, np.float32), R.Device.GPU)
sh = R.FloatTensor.from_numpy(np.full(64, 0.05, np.float32), R.Device.GPU)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
lib.rn_event_create(ctx.handle, ctypes.byref(e0)); lib.rn_event_create(ctx.handle, ctypes.byref(e1))
for name in (["f32", "bf16"] if a.dtype == "both" else [a.dtype]):
    dt, es, cpad = (L.RN_DTYPE_BF16, 2, 4) if name == "bf16" else (L.RN_DTYPE_F32, 4, 3)
    xp = _DeviceBuffer(ctx, B * 230 * 230 * cpad * es)
    L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, dt, xin.data(), xp.ptr, B, 3, 224, 224, cpad, 3), "pad", ctx.handle)
    wp = _DeviceBuffer(ctx, int(lib.rn_stem_pool_packed_weight_numel(dt)) * es)
    L.check(lib.rn_stem_pool_pack_weight_dt(ctx.handle, dt, wd.data(), wp.ptr, 3), "pack", ctx.handle)
    out = _DeviceBuffer(ctx, B * 56 * 56 * 64 * es)

    def run():
        L.check(lib.rn_stem_pool_forward_dt(ctx.handle, dt, xp.ptr, out.ptr, wp.ptr, sc.data(), sh.data(), 1, B, 230, 230),
                "stem", ctx.handle)

    for _ in range(10):
        run()
    lib.rn_event_record(ctx.handle, e0)
    for _ in range(a.reps):
        run()
    lib.rn_event_record(ctx.handle, e1)
    ms = ctypes.c_float()
    lib.rn_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    us = ms.value / a.reps * 1e3
    flops = 2.0 * B * 112 * 112 * 64 * 147
    gb = (B * 230 * 230 * cpad * es + B * 56 * 56 * 64 * es) / 1e9
    print(f"stem+pool {name} B={B}: {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s   {gb / us * 1e6:7.1f} GB/s algorithmic", flush=True)
