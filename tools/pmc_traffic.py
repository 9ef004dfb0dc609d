#!/usr/bin/env python3
"""HBM traffic of the contraction kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950: TCC slot limit).  Uses the contraction dispatches
(conv_gemm_kernel, conv_wide_kernel, conv_strip_kernel, chain_kernel, splitk_finish_kernel, stem_pool_kernel) of the LAST forward of each run;
FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide coalesced read stream,
MI355X_MICROARCH.md section HBM); both counters are in KiB.

    python tools/pmc_traffic.py <FETCH dir> <WRITE dir> out.json [launches per forward] [algorithmic bytes] [note]
"""
import csv
import glob
import json
import os
import sys


FAMILY = ("conv_gemm_kernel", "conv_wide_kernel", "conv_strip_kernel", "conv_strip128_kernel", "chain_kernel", "chain32_", "splitk_finish_kernel",
          "stem_pool_kernel")


def last_forward(d, counter, n):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    vals = [float(r["Counter_Value"]) for r in rows if any(k in r["Kernel_Name"] for k in FAMILY)]
    return vals[-n:]


def main():
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 72
    fe = last_forward(sys.argv[1], "FETCH_SIZE", n)
    wr = last_forward(sys.argv[2], "WRITE_SIZE", n)
    fetch = sum(fe) * 1024 * 2
    write = sum(wr) * 1024
    out = {"kernel": " + ".join(FAMILY), "launches": len(fe),
           "fetch_bytes_per_forward_x2_corrected": fetch, "write_bytes_per_forward": write,
           "traffic_bytes_per_launch": (fetch + write) / len(fe),
           "algorithmic_bytes_per_forward": float(sys.argv[5]) if len(sys.argv) > 5 else 24.97e9,
           "note": sys.argv[6] if len(sys.argv) > 6 else
           "ResNet-50 fp32 B=256 fused mode; separate --pmc passes; FETCH_SIZE x2"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
