#!/usr/bin/env python3
"""bf16 forward with EVERY contraction forced onto one tile candidate (through an edited tuning table,
rn_model_import_tuning; a layer the candidate is not eligible for falls back to its per-launch choice),
against the tuner's per-layer picks, for 1 / 2 / 4 batch parts on streams: does a tile that shares a CU
(the 128x128 tile on 80 KB of LDS, two blocks per CU) pay off once the parts of a batch run side by side,
even where it loses layer by layer?

    python tools/force_tile_ab.py [--dtype bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import resnet_c_amd as R

dtype = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "bf16"
B = 256
x = R.FloatTensor.from_numpy(R.weights.generate_input(B, 0), R.Device.GPU)
out = R.FloatTensor((B, 1000), R.Device.GPU)
ctx = R.get_ctx()
m = R.NativeModel("resnet50", state=R.weights.generate_state("resnet50", 0), dtype=dtype)


def rate(label):
    for _ in range(5):
        m.forward_ptr(x.data(), B, out.data(), True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        m.forward_ptr(x.data(), B, out.data(), True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    print(f"{label:64s} {dt*1e3:7.3f} ms  {B/dt:8.0f} img/s", flush=True)


names = {9: "W256x256", 10: "W256x128", 13: "W224x256", 14: "W128x128 (2 blocks/CU)", 5: "P128x128 (4-wave)", 8: "P64x64 (4-wave)"}
for streams in (1, 2, 4):
    m.set_streams(streams)
    m.tune(x.data(), B, out.data(), True)
    rate(f"streams {streams}, tuned per layer")
    tuned = m.export_tuning()
    n = (len(tuned) - 10) // 4
    for cand, nm in names.items():
        w = tuned.copy()
        for i in range(n):
            w[10 + 4 * i] = cand
            w[10 + 4 * i + 2] = cand
        m.import_tuning(w)
        rate(f"streams {streams}, every layer forced to {nm}")
    m.import_tuning(tuned)
