#!/usr/bin/env python3
"""What each fusion of the model driver is worth at ResNet-50 fp32 B = 256, added one at a time (one stream):
fused epilogues only (what the deferred route of the unchanged op-by-op caller can reach: every tensor the
caller named stays materialised), + tuned tiles, + stem / max-pool, + conv3 / downsample pair, + block-boundary
chains, + two streams.  `--layers` prints the per-layer table of every rung.

    python tools/fusion_ladder.py [--layers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import resnet_c_amd as R

state = R.weights.generate_state("resnet50", 0)
B = 256
x = R.FloatTensor.from_numpy(R.weights.generate_input(B, 0), R.Device.GPU)
out = R.FloatTensor((B, 1000), R.Device.GPU)
ctx = R.get_ctx()
layers = "--layers" in sys.argv


def rate(m, label):
    for _ in range(3):
        m.forward_ptr(x.data(), B, out.data(), True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        m.forward_ptr(x.data(), B, out.data(), True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    print(f"{label:78s} {dt*1e3:7.2f} ms  {B/dt:8.0f} img/s", flush=True)
    if layers:
        m.set_profiling(True)
        m.forward_ptr(x.data(), B, out.data(), True)
        for r in m.profile():
            print(f"      {r['layer']:44s} {r['ms']:7.3f}")
        m.set_profiling(False)


m = R.NativeModel("resnet50", state=state)
m.set_streams(1)
m.set_chain(False); m.set_pair_fusion(False); m.set_stem_pool_fusion(False)
rate(m, "fused epilogues only (no chain / pair / stem+pool), one stream, untuned tiles")
m.tune(x.data(), B, out.data(), True)
rate(m, "  ... tuned tiles")
m.set_stem_pool_fusion(True); m.tune(x.data(), B, out.data(), True)
rate(m, "  + stem + bn + relu + max-pool as one launch")
m.set_chain(True); m.tune(x.data(), B, out.data(), True)
rate(m, "  + conv3 of a block chained with conv1 of the next (stage 1)")
m.set_pair_fusion(True); m.tune(x.data(), B, out.data(), True)
rate(m, "  + conv3 + projection shortcut as one contraction (the default set)")
m.set_streams(2); m.tune(x.data(), B, out.data(), True)
rate(m, "  + the batch as two parts on two streams (the bench configuration)")
m.set_streams(1); m.set_chain(False); m.tune(x.data(), B, out.data(), True)
rate(m, "  (pairs without chains, one stream)")
