#!/usr/bin/env python3
"""Throughput of the literal drop-in route: the reference-shaped graph (createResnet /
resnetForward), one C-ABI call per reference op on NCHW tensors, against the model driver.

    python tools/veneer_rate.py [--batch 64] [--steps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--route-only", action="store_true", help="only the NCHW op-by-op route, asynchronous (for a kernel trace)")
ap.add_argument("--deferred-only", action="store_true", help="only the deferred route (for a kernel trace)")
a = ap.parse_args()
state = R.weights.generate_state("resnet50", 0)
x_host = R.weights.generate_input(a.batch, 0)
ctx = R.get_ctx()
x = R.FloatTensor.from_numpy(x_host, R.Device.GPU)
m = R.createResnet("resnet50", state)
for sync, cache in (() if a.deferred_only else ((False, False),) if a.route_only else ((True, False), (False, False), (False, True))):
    ctx.set_sync_each_op(sync)
    ctx.set_weight_cache(cache)
    for _ in range(2):
        out = R.resnetForward(m, x)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = R.resnetForward(m, x)
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"NCHW op-by-op graph, sync after each op = {sync}, packed-weight cache = {cache}: "
          f"{dt*1e3:8.2f} ms/forward  {a.batch/dt:8.1f} img/s")
if a.route_only:
    sys.exit(0)
# the same unchanged caller on a deferred context (rn_ctx_set_deferred: what the C++ veneer runs by default):
# every call is recorded; the list runs when the logits are observed, conv + in-place bn / add / relu as one
# fused NHWC launch in the caller's own buffers
ctx.set_sync_each_op(False)
ctx.set_weight_cache(True)
ctx.set_deferred(True)
for _ in range(3):
    out = R.resnetForward(m, x)
    ctx.flush()
ctx.sync()
s0 = ctx.deferred_stats()
t0 = time.perf_counter()
for _ in range(a.steps):
    out = R.resnetForward(m, x)
    ctx.flush()
ctx.sync()
dt = (time.perf_counter() - t0) / a.steps
s1 = ctx.deferred_stats()
per = {k: (s1[k] - s0[k]) // a.steps for k in ("fused_launches", "literal_launches", "transposes")}
print(f"NCHW op-by-op graph, DEFERRED (conv + in-place bn / add / relu folded, NHWC kept in the caller's buffers): "
      f"{dt*1e3:8.2f} ms/forward  {a.batch/dt:8.1f} img/s   per forward: {per}")
if a.deferred_only:
    sys.exit(0)
logits_deferred = out.numpy()
ctx.set_deferred(False)
logits_literal = R.resnetForward(m, x).numpy()
print(f"   max |deferred - literal| over the logits: {np.abs(logits_deferred - logits_literal).max():.2e}, "
      f"same top-1 on {int((logits_deferred.argmax(1) == logits_literal.argmax(1)).sum())} of {a.batch}")
ctx.set_sync_each_op(False)
ctx.set_weight_cache(False)
nm = R.NativeModel("resnet50", state=state)
lg = R.FloatTensor((a.batch, 1000), R.Device.GPU)
for fused in (False, True):
    nm.tune(x.data(), a.batch, lg.data(), fused)
    for _ in range(5):
        nm.forward_ptr(x.data(), a.batch, lg.data(), fused)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        nm.forward_ptr(x.data(), a.batch, lg.data(), fused)
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"model driver (NHWC arenas), fused = {fused}: {dt*1e3:8.2f} ms/forward  {a.batch/dt:8.1f} img/s")
