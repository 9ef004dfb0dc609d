#!/usr/bin/env python3
"""A/B of the fused stem's two inputs on the whole forward (ResNet-50, B = 256): padded NHWC image
written by the layout launch (1) against patches fetched from the NCHW input (2).

    python tools/stem_ab.py [f32] [bf16]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import resnet_c_amd as R
from resnet_c_amd.tensor import get_ctx

B, STEPS = 256, 30
for dtype in (sys.argv[1:] or ["f32", "bf16"]):
    m = R.NativeModel("resnet50", state=R.weights.generate_state("resnet50", 0), dtype=dtype)
    m.set_streams(1)
    x = R.FloatTensor.from_numpy(R.weights.generate_input(B, 0), R.Device.GPU)
    out = R.FloatTensor((B, 1000), R.Device.GPU)
    ref = None
    for mode in (1, 2, 1, 2):
        m.set_stem_pool_fusion(mode)
        m.tune(x.data(), B, out.data(), True)
        for _ in range(5):
            m.forward_ptr(x.data(), B, out.data(), True)
        got = out.numpy()
        ref = got if ref is None else ref
        assert np.array_equal(got, ref)
        get_ctx().sync()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            m.forward_ptr(x.data(), B, out.data(), True)
        get_ctx().sync()
        ms = (time.perf_counter() - t0) / STEPS * 1e3
        m.set_profiling(True)
        m.forward_ptr(x.data(), B, out.data(), True)
        front = [(r["layer"], round(r["ms"], 3)) for r in m.profile()[:2]]
        m.set_profiling(False)
        print(f"{dtype} stem mode {mode}: {ms:.3f} ms/forward  {B / ms * 1e3:.0f} img/s  first ops {front}", flush=True)
    m.close()
