#!/usr/bin/env python3
"""Images/s of the multi-device library path (rn_shard_*) WITH the input upload and the logits
download, on the devices listed (default: device 0 once): every device owns an rn_pipeline
(pinned staging, copy stream, two slots).

    python tools/shard_rate.py [--devices 0[,0,...]] [--batch 256] [--steps 10]

  one-shot      rn_shard_forward per batch: pageable host array -> pinned staging -> upload ->
                forward -> download, nothing overlapped across batches (main.cu:236-240 per device)
  stream/copy   rn_shard_submit(batch) / rn_shard_collect, two batches in flight: the copy into
                pinned staging (by each device's host thread) and the upload of batch i+1 run
                beside the forward of batch i
  stream/inplace  the same with the producer writing into the pinned staging buffers
                (rn_shard_stream_buffer) -- here they are filled once, outside the timed loop
  resident      rn_model_forward on images already in HBM (what bench.py's `value` measures)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--devices", default="0")
ap.add_argument("--batch", type=int, default=256, help="images per device")
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
devices = [int(d) for d in a.devices.split(",")]
B = a.batch * len(devices)
state = R.weights.generate_state("resnet50", 0)
x = np.concatenate([R.weights.generate_input(32, seed=9)] * ((B + 31) // 32))[:B]
for dtype in ("f32", "bf16"):
    g = R.ShardedModel(devices, "resnet50", state=state, dtype=dtype)
    g.tune(x, fused=True)
    for _ in range(2):
        g.forward(x, fused=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.forward(x, fused=True)
    one_shot = B * a.steps / (time.perf_counter() - t0)
    g.stream_open(B, fused=True)
    g.tune(x, fused=True)   # with a stream open: the tiles of a whole shard per launch (not of the one-shot form's chunks)
    g.submit(x); g.submit(x); g.collect(); g.collect()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if g.in_flight() == 2:
            g.collect()
        g.submit(x)
    while g.in_flight():
        g.collect()
    stream_copy = B * a.steps / (time.perf_counter() - t0)
    for _ in range(2):  # both slots' staging buffers hold the batch now; fill them explicitly all the same
        for r in range(len(devices)):
            buf, lo, hi = g.stream_buffer(r)
            if buf is not None:
                buf[...] = x[lo:hi]
        g.submit(None)
    g.collect(); g.collect()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if g.in_flight() == 2:
            g.collect()
        g.submit(None)
    while g.in_flight():
        g.collect()
    stream_inplace = B * a.steps / (time.perf_counter() - t0)
    g.stream_close()
    g_place = [g.placement(r) for r in range(len(devices))]
    g.close()
    m = R.NativeModel("resnet50", state=state, dtype=dtype)
    xd = R.FloatTensor.from_numpy(x[:a.batch], R.Device.GPU)
    out = R.FloatTensor((a.batch, 1000), R.Device.GPU)
    m.tune(xd.data(), a.batch, out.data(), True)
    for _ in range(3):
        m.forward_ptr(xd.data(), a.batch, out.data(), True)
    m.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.forward_ptr(xd.data(), a.batch, out.data(), True)
    m.ctx.sync()
    resident = a.batch * a.steps / (time.perf_counter() - t0)
    m.close()
    print(f"{dtype} placement: " + "; ".join("shard %d on device %d, NUMA node %d, cpus [%s]" % ((r,) + g_place[r])
                                              for r in range(len(devices))))
    print(f"{dtype} devices {devices} B={B}: one-shot {one_shot:9.0f}  stream/copy {stream_copy:9.0f}  "
          f"stream/inplace {stream_inplace:9.0f}  img/s with upload + download;  resident (one device) {resident:9.0f} img/s",
          flush=True)
