#!/usr/bin/env python3
"""Per-op table of one instrumented forward: ms, TFLOP/s, GB/s (HIP events per op).

    python tools/layer_report.py [--arch resnet50] [--batch 256] [--mode fused|ops] [--reps 3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import resnet_c_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--mode", default="fused")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    state = R.weights.generate_state(a.arch, 0)
    m = R.NativeModel(a.arch, state=state, dtype=a.dtype)
    x = R.FloatTensor.from_numpy(R.weights.generate_input(a.batch, 0), R.Device.GPU)
    out = R.FloatTensor((a.batch, 1000), R.Device.GPU)
    fused = a.mode == "fused"
    if a.tune:
        m.tune(x.data(), a.batch, out.data(), fused)
    for _ in range(2):
        m.forward_ptr(x.data(), a.batch, out.data(), fused)
    m.set_profiling(True)
    acc = None
    for _ in range(a.reps):
        m.forward_ptr(x.data(), a.batch, out.data(), fused)
        recs = m.profile()
        if acc is None:
            acc = recs
        else:
            for r, n in zip(acc, recs):
                r["ms"] = min(r["ms"], n["ms"])
    tot = sum(r["ms"] for r in acc)
    print(f"{'op':18s} {'layer':24s} {'ms':>8s} {'TF/s':>7s} {'GB/s':>8s} {'ideal_ms':>8s}")
    ideal_tot = 0
    mfma_peak = 2500e9 if a.dtype == "bf16" else 157.3e9  # flop per ms (MI355X_MICROARCH.md)
    for r in acc:
        tf = r["flops"] / r["ms"] / 1e9 if r["ms"] > 0 else 0
        gb = r["bytes"] / r["ms"] / 1e6 if r["ms"] > 0 else 0
        # ms at the dense MFMA peak of the element type / at the achievable HBM rate
        ideal = max(r["flops"] / mfma_peak, r["bytes"] / 6.3e9)
        ideal_tot += ideal
        print(f"{r['op']:18s} {r['layer']:24s} {r['ms']:8.3f} {tf:7.1f} {gb:8.0f} {ideal:8.3f}")
    print(f"total {tot:.3f} ms  ({a.batch / tot * 1e3:.0f} img/s by events)  roofline-ideal {ideal_tot:.3f} ms")


if __name__ == "__main__":
    main()
