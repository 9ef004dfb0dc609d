#!/usr/bin/env python3
"""Per layer: the per-launch (untuned) tile choice against the tuner's pick -- which layers the heuristic of
launch_gemm gets wrong and what that costs (one stream, per-op events, min of 5 forwards).

    python tools/tune_report.py [--dtype f32|bf16] [--batch 256]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import resnet_c_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
B = a.batch
names = ["auto", "128x128", "128x64", "64x128", "64x64", "P128x128", "P128x64", "P64x128", "P64x64",
         "W256x256", "W256x128", "W128x256", "W256x64", "W224x256", "W128x128", "strip"]
m = R.NativeModel("resnet50", state=R.weights.generate_state("resnet50", 0), dtype=a.dtype)
m.set_streams(1)
x = R.FloatTensor.from_numpy(R.weights.generate_input(B, 0), R.Device.GPU)
out = R.FloatTensor((B, 1000), R.Device.GPU)


def table():
    for _ in range(2):
        m.forward_ptr(x.data(), B, out.data(), True)
    m.set_profiling(True)
    best = {}
    order = []
    for _ in range(5):
        m.forward_ptr(x.data(), B, out.data(), True)
        for r in m.profile():
            if r["layer"] not in best:
                order.append(r["layer"])
            best[r["layer"]] = min(best.get(r["layer"], 1e9), r["ms"])
    m.set_profiling(False)
    return order, best


order, untuned = table()
m.tune(x.data(), B, out.data(), True)
_, tuned = table()
words = m.export_tuning()
# header 10 words, then (tile, B) x 2 per convolution in the model's convolution order, then per block pair
print(f"{'layer':44s} {'untuned ms':>10s} {'tuned ms':>9s} {'gain us':>8s}")
tot_u = tot_t = 0.0
for k in order:
    u, t = untuned[k], tuned.get(k, float('nan'))
    tot_u += u
    tot_t += t
    mark = "  <--" if u - t > 0.004 else ""
    print(f"{k:44s} {u:10.3f} {t:9.3f} {(u - t) * 1e3:8.1f}{mark}")
print(f"{'total':44s} {tot_u:10.3f} {tot_t:9.3f} {(tot_u - tot_t) * 1e3:8.1f}")
n_convs = int(words[6])
picks = [int(words[10 + 4 * i]) for i in range(n_convs)]
print("tuned candidates in the model's convolution order:", " ".join(names[p] if p < len(names) else str(p) for p in picks))
