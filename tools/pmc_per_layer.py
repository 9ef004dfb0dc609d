#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of one forward launch by launch, next to each layer's algorithmic bytes.

    python tools/pmc_per_layer.py <FETCH dir> <WRITE dir> <layers.txt> <launches per forward> > out.txt

<layers.txt> is tools/layer_report.py's table of the same configuration (its GB/s x ms are the
algorithmic bytes of a layer: activations in and out once, residual, weights once).  The launches of
the last forward of each --pmc pass are walked in order; a splitk_finish_kernel belongs to the
launch in front of it.  FETCH_SIZE counts L2 misses, Infinity-Cache hits included
(MI355X_MICROARCH.md, HBM): a stage whose weights and activations fit the 256 MB Infinity Cache can
show several times its algorithmic bytes here without reading HBM more than once."""
import csv
import glob
import os
import re
import sys

FAMILY = ("conv_gemm_kernel", "conv_wide_kernel", "conv_strip_kernel", "conv_strip128_kernel", "chain_kernel",
          "chain32_", "splitk_finish_kernel", "stem_pool_kernel")


def last_forward(d, counter, n):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    return [r for r in rows if any(k in r["Kernel_Name"] for k in FAMILY)][-n:]


def short(name):
    if name.startswith("_Z"):  # a name the profiler could not demangle (__bf16 arguments): family<integers>
        fam = next(k for k in FAMILY if k in name)
        ints = re.findall(r"L[ib](\d+)E", name.split(fam, 1)[1])
        return (fam.replace("conv_gemm_kernel", "gemm") + "<bf16, " + ", ".join(ints) + ">")[:44]
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rn_gemm::", "")
    return n.split("(")[0].replace("conv_gemm_kernel", "gemm").replace("float", "f32")[:44]


def main():
    n = int(sys.argv[4])
    fe, wr = last_forward(sys.argv[1], "FETCH_SIZE", n), last_forward(sys.argv[2], "WRITE_SIZE", n)
    layers = []
    for line in open(sys.argv[3]):
        p = line.split()
        if len(p) >= 6 and (p[0].startswith("conv2d") or p[0] == "linear"):
            layers.append((p[1], float(p[2]) * float(p[4]) * 1e6))  # ms x GB/s -> bytes
    groups = []  # (kernel, fetch, write) per contraction launch, finish launches folded in
    for f, w in zip(fe, wr):
        fb, wb = float(f["Counter_Value"]) * 2048, float(w["Counter_Value"]) * 1024
        if "splitk_finish" in f["Kernel_Name"] and groups:
            groups[-1][1] += fb
            groups[-1][2] += wb
            groups[-1][0] += " +finish"
        else:
            groups.append([short(f["Kernel_Name"]), fb, wb])
    if len(groups) != len(layers):
        print(f"# {len(groups)} contraction launches against {len(layers)} table rows: names not attached")
        layers = [("?", 0.0)] * len(groups)
    print(f"{'layer':44s} {'fetch MB':>9s} {'write MB':>9s} {'algorithmic MB':>14s} {'ratio':>6s}  kernel (FETCH pass)")
    stage = {}
    for (name, alg), (k, fb, wb) in zip(layers, groups):
        print(f"{name:44s} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {alg / 1e6:14.1f} {(fb + wb) / alg if alg else 0:6.2f}  {k}")
        s = name.split(".")[0] if name.startswith("layer") else ("stem" if name.startswith("conv1") else name)
        a = stage.setdefault(s, [0.0, 0.0, 0.0])
        a[0] += fb
        a[1] += wb
        a[2] += alg
    print()
    tot = [0.0, 0.0, 0.0]
    for s, (fb, wb, alg) in stage.items():
        print(f"{s:10s} fetch {fb / 1e9:7.2f} GB  write {wb / 1e9:6.2f} GB  algorithmic {alg / 1e9:6.2f} GB  "
              f"ratio {(fb + wb) / alg if alg else 0:5.2f}")
        tot = [tot[0] + fb, tot[1] + wb, tot[2] + alg]
    print(f"{'total':10s} fetch {tot[0] / 1e9:7.2f} GB  write {tot[1] / 1e9:6.2f} GB  algorithmic {tot[2] / 1e9:6.2f} GB  "
          f"ratio {(tot[0] + tot[1]) / tot[2] if tot[2] else 0:5.2f}")


if __name__ == "__main__":
    main()
