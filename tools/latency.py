#!/usr/bin/env python3
"""Small-batch latency of one forward: eager launches vs the captured graph (rn_model_capture).

    python tools/latency.py [--arch resnet50] [--dtype f32] [--batches 1 2 4 8 16 32]
Host-clock time of N back-to-back forwards + one sync, divided by N (so queueing hides the
launch cost when the GPU is the bound), and of single forwards each followed by a sync
(the latency a caller sees)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="resnet50")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32])
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--split-k", type=int, default=16)
a = ap.parse_args()
m = R.NativeModel(a.arch, state=R.weights.generate_state(a.arch, 0), dtype=a.dtype)
ctx = m.ctx
ap2 = None
print(f"{'B':>4s} {'eager us':>10s} {'graph us':>10s} {'eager sync us':>14s} {'graph sync us':>14s}  nodes"
      f" {'split-K eager sync us':>22s} {'split-K graph sync us':>22s}  nodes")
for B in a.batches:
    x = R.FloatTensor.from_numpy(R.weights.generate_input(B, 0), R.Device.GPU)
    out = R.FloatTensor((B, 1000), R.Device.GPU)
    m.tune(x.data(), B, out.data(), True)
    g = R.Graph(m, x.data(), B, out.data(), True)
    def timed(fn, each_sync):
        for _ in range(10): fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
            if each_sync: ctx.sync()
        ctx.sync()
        return (time.perf_counter() - t0) / a.reps * 1e6
    eager = lambda: m.forward_ptr(x.data(), B, out.data(), True)
    line = (f"{B:4d} {timed(eager, False):10.1f} {timed(g.launch, False):10.1f} "
            f"{timed(eager, True):14.1f} {timed(g.launch, True):14.1f}  {g.node_count()}")
    g.close()
    # latency mode: K loops of under-filled layers split over more blocks (rn_ctx_set_split_k)
    ctx.set_split_k(a.split_k)
    m.tune(x.data(), B, out.data(), True)
    g = R.Graph(m, x.data(), B, out.data(), True)
    line += f" {timed(eager, True):22.1f} {timed(g.launch, True):22.1f}  {g.node_count()}"
    g.close()
    ctx.set_split_k(0)
    print(line, flush=True)
