#!/usr/bin/env python3
"""Randomised parity run of the convolution entry points against the CPU oracle (test
infrastructure under tests/: nothing outside tests/, smoke() and bench.py's CPU baseline touches oracle/).  Shapes, strides, paddings, layouts,
storage types, epilogues and tile candidates are drawn at random; every tile candidate of a case
must give the bits of the first one.

    python tests/fuzz/conv_fuzz.py [--seconds 60] [--seed 0]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    lib, ctx = L.lib(), R.get_ctx()
    ncand = int(lib.rn_conv_tile_candidates())
    t0, cases, launches = time.time(), 0, 0
    kinds = {"nchw": 0, "nhwc": 0, "fused": 0, "bf16": 0, "pair": 0, "exact": 0}
    while time.time() - t0 < a.seconds:
        k = int(g.choice([1, 1, 3, 3, 5, 7]))
        stride = int(g.choice([1, 1, 2, 3]))
        pad = int(g.integers(0, k // 2 + 2))
        B = int(g.integers(1, 6))
        H, W = int(g.integers(max(1, k - 2 * pad), 24)), int(g.integers(max(1, k - 2 * pad), 24))
        Cin = int(g.choice([1, 3, 4, 8, 16, 24, 32, 64, 96, 128, 160]))
        Cout = int(g.choice([1, 4, 8, 20, 32, 64, 72, 128, 256]))
        if H + 2 * pad < k or W + 2 * pad < k:
            continue
        x = g.standard_normal((B, Cin, H, W), dtype=np.float32)
        w = g.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)
        K = Cin * k * k
        want = O.conv2d(x, w, stride, pad)
        scale = float(np.abs(want).max()) + 1e-6
        kind = str(g.choice(["nchw", "nhwc", "fused", "bf16", "pair", "exact"]))
        if kind == "bf16" and (Cin % 64 or Cout % 8):  # the bf16 contraction: whole 128-byte channel segments
            kind = "fused"
        if kind == "exact" and not (Cin <= 4 and k <= 8 and Cout % 4 == 0):  # the small-Cin exact-K form
            kind = "nhwc"
        if kind == "pair" and not (Cin % 32 == 0 and Cout % 4 == 0 and stride == 1):
            kind = "fused"
        kinds[kind] += 1
        first = None
        cands = [0] + [int(c) for c in g.choice(np.arange(1, ncand + 1), size=3, replace=False)]
        try:
            for c in cands:
                lib.rn_ctx_set_conv_tile(ctx.handle, c)
                if kind in ("nchw", "nhwc"):
                    # (NCHW, k x k: the gathering kernel, the transposing route or the default choice between them)
                    ctx.set_nchw_taps(int(g.integers(0, 3)))
                    got = ops.conv2d(x, w, stride, pad, kind)
                    ctx.set_nchw_taps(1)
                    ref, tol = want, 3e-7 * np.sqrt(K) * scale + 1e-6
                elif kind == "fused":
                    sc = g.random(Cout, dtype=np.float32) + 0.5
                    sh = g.standard_normal(Cout, dtype=np.float32)
                    res = g.standard_normal(want.shape, dtype=np.float32)
                    # (the same epilogue arrays for every candidate of the case)
                    if first is None:
                        keep = (sc, sh, res)
                    sc, sh, res = keep
                    got = ops.conv2d_nhwc_fused(x, w, stride, pad, sc, sh, res, True)
                    ref = np.maximum(want * sc[None, :, None, None] + sh[None, :, None, None] + res, 0)
                    tol = 3e-7 * np.sqrt(K) * (float(np.abs(ref).max()) + scale) + 2e-6
                elif kind == "exact":
                    got = ops.conv2d_nhwc_exact(x, w, stride, pad)
                    ref, tol = want, 3e-7 * np.sqrt(K) * scale + 1e-6
                elif kind == "pair":
                    # + a 1x1 convolution (stride s2) of a second tensor in the same K loop, scales folded
                    if first is None:
                        s2, Cin2 = int(g.choice([1, 2])), int(g.choice([32, 64, 96]))
                        ho, wo = want.shape[2], want.shape[3]
                        x2 = g.standard_normal((B, Cin2, (ho - 1) * s2 + 1, (wo - 1) * s2 + 1), dtype=np.float32)
                        w2 = g.standard_normal((Cout, Cin2, 1, 1), dtype=np.float32) / np.sqrt(Cin2)
                        sc1, sc2 = g.random(Cout, dtype=np.float32) + 0.5, g.random(Cout, dtype=np.float32) + 0.5
                        shf = g.standard_normal(Cout, dtype=np.float32)
                        ref = np.maximum(O.conv2d(x, w * sc1[:, None, None, None], stride, pad) +
                                         O.conv2d(x2, w2 * sc2[:, None, None, None], s2, 0) + shf[None, :, None, None], 0)
                        keep = (x2, w2, s2, sc1, sc2, shf, ref)
                    x2, w2, s2, sc1, sc2, shf, ref = keep
                    got = ops.conv2d_nhwc_pair(x, w, x2, w2, stride, pad, s2, sc1, sc2, shf, None, True)
                    tol = 3e-7 * np.sqrt(K + x2.shape[1] + 4) * (float(np.abs(ref).max()) + scale) * 2 + 2e-6
                else:
                    got = ops.conv2d_nhwc_bf16(x, w, stride, pad, None, None, None, False, out_f32=True)
                    ref = O.conv2d(ops.bf16_round(x), ops.bf16_round(w), stride, pad)
                    tol = 3e-7 * np.sqrt(K) * scale + 1e-6
                launches += 1
                err = float(np.abs(got - ref).max())
                assert got.shape == ref.shape and err <= tol, (
                    f"{kind} B={B} {Cin}->{Cout} {H}x{W} k={k} s={stride} p={pad} cand={c}: err {err:.3e} > {tol:.3e}")
                if first is None:
                    first = got
                else:
                    assert np.array_equal(got, first), (
                        f"{kind} B={B} {Cin}->{Cout} {H}x{W} k={k} s={stride} p={pad}: candidate {c} != candidate {cands[0]}")
        finally:
            lib.rn_ctx_set_conv_tile(ctx.handle, 0)
        cases += 1
    print(f"conv_fuzz: {cases} cases, {launches} launches, {kinds}, seed {a.seed}: all within tolerance, "
          f"every candidate bit-identical per case")


if __name__ == "__main__":
    main()
