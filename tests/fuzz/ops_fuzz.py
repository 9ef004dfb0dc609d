#!/usr/bin/env python3
"""Randomised parity run of the element-wise, pooling, batch-norm and linear entry points against the
CPU oracle, both layouts (test infrastructure under tests/).

    python tests/fuzz/ops_fuzz.py [--seconds 60] [--seed 0]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import resnet_c_amd as R
from oracle import oracle as O
from resnet_c_amd import _lib as L
from resnet_c_amd import ops
from resnet_c_amd.tensor import _DeviceBuffer


def layout_case(g):
    """rn_nchw_to_nhwc / rn_nhwc_to_nchw / rn_nchw_to_nhwc_pad_dt against numpy: pure data movement."""
    lib, ctx = L.lib(), R.get_ctx()
    B, C = int(g.integers(1, 6)), int(g.choice([1, 2, 3, 4, 5, 8, 31, 32, 64, 100]))
    H, W = int(g.integers(1, 30)), int(g.integers(1, 30))
    x = g.standard_normal((B, C, H, W), dtype=np.float32)
    src = R.FloatTensor.from_numpy(x, R.Device.GPU)
    which = str(g.choice(["plain", "back", "pad"]))
    if which == "plain":
        dst = R.FloatTensor((B, H, W, C), R.Device.GPU)
        L.check(lib.rn_nchw_to_nhwc(ctx.handle, src.data(), dst.data(), B, C, H, W), "nchw_to_nhwc", ctx.handle)
        ctx.sync()
        assert np.array_equal(dst.numpy().reshape(B, H, W, C), x.transpose(0, 2, 3, 1)), f"nchw_to_nhwc {x.shape}"
    elif which == "back":
        nhwc = R.FloatTensor.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(B, H, W, C), R.Device.GPU)
        dst = R.FloatTensor((B, C, H, W), R.Device.GPU)
        L.check(lib.rn_nhwc_to_nchw(ctx.handle, nhwc.data(), dst.data(), B, C, H, W), "nhwc_to_nchw", ctx.handle)
        ctx.sync()
        assert np.array_equal(dst.numpy().reshape(B, C, H, W), x), f"nhwc_to_nchw {x.shape}"
    else:
        bf16 = bool(g.integers(0, 2))
        Cpad = int(g.choice([C, C + 1, max(C, 4), ((C + 3) // 4) * 4]))
        border = int(g.integers(0, 4))
        Hp, Wp = H + 2 * border, W + 2 * border
        want = np.zeros((B, Hp, Wp, Cpad), dtype=np.float32)
        want[:, border:border + H, border:border + W, :C] = x.transpose(0, 2, 3, 1)
        n = B * Hp * Wp * Cpad
        dst = _DeviceBuffer(ctx, n * (2 if bf16 else 4))
        L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, L.RN_DTYPE_BF16 if bf16 else L.RN_DTYPE_F32, src.data(), dst.ptr,
                                           B, C, H, W, Cpad, border), "pad_dt", ctx.handle)
        ctx.sync()
        if bf16:
            h = np.empty(n, dtype=np.uint16)
            L.check(lib.rn_memcpy_d2h(ctx.handle, h.ctypes.data, dst.ptr, h.nbytes), "d2h", ctx.handle)
            assert np.array_equal(h, ops.to_bf16_bits(want.reshape(-1))), f"pad_dt bf16 {x.shape} Cpad={Cpad} border={border}"
        else:
            h = np.empty(n, dtype=np.float32)
            L.check(lib.rn_memcpy_d2h(ctx.handle, h.ctypes.data, dst.ptr, h.nbytes), "d2h", ctx.handle)
            assert np.array_equal(h, want.reshape(-1)), f"pad_dt f32 {x.shape} Cpad={Cpad} border={border}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    t0, n = time.time(), {"maxpool": 0, "avgpool": 0, "batchnorm": 0, "linear": 0, "relu/add": 0, "layout": 0}
    while time.time() - t0 < a.seconds:
        what = str(g.choice(list(n)))
        layout = str(g.choice(["nchw", "nhwc"]))
        if what in ("maxpool", "avgpool"):
            k = int(g.choice([1, 2, 3, 3, 7]))
            s = int(g.choice([1, 2, 3, 7]))
            p = int(g.integers(0, k // 2 + 1))
            B, C = int(g.integers(1, 9)), int(g.choice([1, 3, 4, 8, 12, 64, 100, 256, 2048]))
            H, W = int(g.integers(max(1, k - 2 * p), 20)), int(g.integers(max(1, k - 2 * p), 20))
            if C >= 256:
                H, W = min(H, 8), min(W, 8)
            elif what == "maxpool" and g.random() < 0.3:  # the network's pool on taller images: the column walk
                k, s, p, H = 3, 2, 1, int(g.integers(15, 60))
            if what == "avgpool" and g.random() < 0.4:  # the global pool of the network's tail
                H = W = k = 7
                s, p = int(g.choice([1, 7])), 0
            x = g.standard_normal((B, C, H, W), dtype=np.float32)
            got = (ops.maxpool2d if what == "maxpool" else ops.avgpool2d)(x, k, s, p, layout)
            want = (O.maxpool2d if what == "maxpool" else O.avgpool2d)(x, k, s, p)
            assert np.array_equal(got, want), f"{what} {layout} {x.shape} k={k} s={s} p={p}"
        elif what == "batchnorm":
            B, C = int(g.integers(1, 40)), int(g.choice([1, 2, 3, 4, 5, 8, 33, 64, 70, 256]))
            H, W = int(g.integers(1, 20)), int(g.integers(1, 20))
            x = g.standard_normal((B, C, H, W), dtype=np.float32) * 3
            w, b = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
            m, v = g.standard_normal(C, dtype=np.float32), g.random(C, dtype=np.float32) + 0.5
            want = O.batchnorm2d(x, w, b, m, v)
            got = ops.batchnorm2d(x, w, b, m, v, layout, inplace=bool(g.integers(0, 2)))
            # ops.cu:150 type by type in the oracle and in the kernels (fp32 subtraction, then double, one fma, one
            # rounding): the same bits
            assert np.array_equal(got, want), f"batchnorm {layout} {x.shape}"
        elif what == "linear":
            B, I, Oo = int(g.integers(1, 70)), int(g.choice([1, 7, 32, 64, 96, 300, 2048])), int(g.integers(1, 130))
            x, w = g.standard_normal((B, I), dtype=np.float32), g.standard_normal((Oo, I), dtype=np.float32) / np.sqrt(I)
            b = g.standard_normal(Oo, dtype=np.float32) if g.random() < 0.7 else None
            want = O.linear(x, w, b)
            got = ops.linear(x, w, b)
            tol = 3e-7 * np.sqrt(I) * (float(np.abs(want).max()) + 1e-6) + 1e-6
            assert got.shape == want.shape and float(np.abs(got - want).max()) <= tol, f"linear {B}x{I}->{Oo}"
        elif what == "layout":
            layout_case(g)
        else:
            nel = int(g.choice([1, 3, 4, 5, 63, 64, 1000, 4097, 70001]))
            x, y = g.standard_normal(nel, dtype=np.float32), g.standard_normal(nel, dtype=np.float32)
            inplace = bool(g.integers(0, 2))
            assert np.array_equal(ops.relu(x, inplace), O.relu(x)) and np.array_equal(ops.add(x, y, inplace), O.add(x, y))
        n[what] += 1
    print(f"ops_fuzz: {n}, seed {a.seed}: pools / relu / add / batch-norm / layout changes bit-exact, linear within tolerance")


if __name__ == "__main__":
    main()
