#!/usr/bin/env python3
"""Experiment: the batch as S independent sub-batches on S streams (one model instance and one
context each) against one stream, same total images.  Tails of one stream's kernels can be
filled by the other stream's blocks.

    python tools/two_stream.py [--batch 256] [--streams 2] [--steps 30] [--dtype f32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
state = R.weights.generate_state("resnet50", 0)
x_host = R.weights.generate_input(a.batch, 0)
for S in a.streams:
    b = a.batch // S
    ctxs = [R.Context(0) for _ in range(S)]
    models = [R.NativeModel("resnet50", state=state, ctx=c, dtype=a.dtype) for c in ctxs]
    xs = [R.FloatTensor.from_numpy(x_host[i * b:(i + 1) * b], R.Device.GPU) for i in range(S)]
    outs = [R.FloatTensor((b, 1000), R.Device.GPU) for _ in range(S)]
    for m, x, o in zip(models, xs, outs):
        m.tune(x.data(), b, o.data(), True)
    def step():
        for m, x, o in zip(models, xs, outs):
            m.forward_ptr(x.data(), b, o.data(), True)
    for _ in range(5): step()
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps): step()
    for c in ctxs: c.sync()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"streams {S}  sub-batch {b:4d}  {dt*1e3:8.3f} ms/step  {a.batch/dt:9.1f} img/s", flush=True)
    for m in models: m.close()
    for c in ctxs: c.close()
