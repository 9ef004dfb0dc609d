#!/usr/bin/env python3
"""HBM traffic of the contraction kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950: TCC slot limit).  Uses the contraction dispatches
(conv_gemm_kernel, conv_wide_kernel, conv_strip_kernel, chain_kernel, splitk_finish_kernel, stem_pool_kernel) of the LAST forward of each run;
FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide coalesced read stream,
MI355X_MICROARCH.md section HBM); both counters are in KiB.

    python tools/pmc_traffic.py <FETCH dir> <WRITE dir> out.json [launches per forward] [algorithmic bytes] [note]

The JSON is stamped with the digest of the kernel sources (resnet_c_amd._lib.source_digest) and
the launch count: bench.py reports the figure only for the build and launch count it was
measured on (otherwise `traffic` is null and `traffic_stale` true).  A per-kernel table (bytes
fetched and written per forward by kernel instantiation) goes next to it as <out>.per_kernel.txt.
"""
import csv
import glob
import json
import os
import sys


FAMILY = ("conv_gemm_kernel", "conv_wide_kernel", "conv_strip_kernel", "conv_strip128_kernel", "chain_kernel", "chain32_", "splitk_finish_kernel",
          "stem_pool_kernel")


def last_forward(d, counter, n, names=None):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in FAMILY)][-n:]
    if names is not None:
        names.extend(r["Kernel_Name"] for r in rows)
    return [float(r["Counter_Value"]) for r in rows]


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rn_gemm::", "")
    return n.split("(")[0][:110]


def main():
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 72
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from resnet_c_amd._lib import source_digest

    names_f, names_w = [], []
    fe = last_forward(sys.argv[1], "FETCH_SIZE", n, names_f)
    wr = last_forward(sys.argv[2], "WRITE_SIZE", n, names_w)
    fetch = sum(fe) * 1024 * 2
    write = sum(wr) * 1024
    per = {}
    for nm, v in zip(names_f, fe):
        per.setdefault(short(nm), [0, 0.0, 0.0])
        per[short(nm)][0] += 1
        per[short(nm)][1] += v * 2048
    for nm, v in zip(names_w, wr):
        per.setdefault(short(nm), [0, 0.0, 0.0])[2] += v * 1024
    with open(os.path.splitext(sys.argv[3])[0] + ".per_kernel.txt", "w") as fh:
        fh.write("launches  fetch_MB(x2)  write_MB  kernel instantiation (one forward)\n")
        for k, (c, f_, w_) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            fh.write(f"{c:8d} {f_ / 1e6:13.1f} {w_ / 1e6:9.1f}  {k}\n")
    out = {"kernel": " + ".join(FAMILY), "launches": len(fe), "source_digest": source_digest(),
           "fetch_bytes_per_forward_x2_corrected": fetch, "write_bytes_per_forward": write,
           "traffic_bytes_per_launch": (fetch + write) / len(fe),
           "algorithmic_bytes_per_forward": float(sys.argv[5]) if len(sys.argv) > 5 else 24.97e9,
           "note": sys.argv[6] if len(sys.argv) > 6 else
           "ResNet-50 fp32 B=256 fused mode; separate --pmc passes; FETCH_SIZE x2"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
