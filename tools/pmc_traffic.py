#!/usr/bin/env python3
"""HBM traffic of the contraction kernel from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950: TCC slot limit).  Uses the 50 conv_gemm dispatches
of the LAST forward of each run; FETCH_SIZE is doubled (gfx950 reports half the bytes of a
wide coalesced read stream, MI355X_MICROARCH.md section HBM); both counters are in KiB.

    python tools/pmc_traffic.py gpurun_out/pmc_r1_FETCH_SIZE gpurun_out/pmc_r1_WRITE_SIZE out.json
"""
import csv
import glob
import json
import os
import sys


def last_forward(d, counter, n=50):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "conv_gemm_kernel" in r["Kernel_Name"]]
    return vals[-n:]


def main():
    fe = last_forward(sys.argv[1], "FETCH_SIZE")
    wr = last_forward(sys.argv[2], "WRITE_SIZE")
    fetch = sum(fe) * 1024 * 2
    write = sum(wr) * 1024
    out = {"kernel": "conv_gemm_kernel", "launches": len(fe),
           "fetch_bytes_per_forward_x2_corrected": fetch, "write_bytes_per_forward": write,
           "traffic_bytes_per_launch": (fetch + write) / len(fe),
           "algorithmic_bytes_per_forward": 24.97e9,
           "note": "ResNet-50 fp32 B=256 fused mode; separate --pmc passes; FETCH_SIZE x2"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
