#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name.

    python tools/pmc_summary.py gpurun_out/pmc1 [--skip-first N]
Prints, per kernel, the sum of each counter over its dispatches and a few ratios.
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.split("(")[0][-60:] if len(name) > 60 else name
            for fam in ("conv_gemm_kernel", "conv_wide_kernel", "stem_pool_kernel"):
                if fam in name:
                    short = fam + name.split(fam)[1].split(">")[0] + ">"
            if "conv_strip_kernel" in name:
                short = "conv_strip_kernel"
            if "conv_strip128_kernel" in name:
                short = "conv_strip128_kernel"
            for fam in ("chain_kernel", "chain32_kernel", "chain32_pair_kernel"):
                if fam in name:
                    short = fam + name.split(fam)[1].split("(")[0]
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[short] += 1
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"== {k}  dispatches={calls[k]}")
        for n, v in sorted(c.items()):
            print(f"   {n:32s} {v:18.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
            print(f"   mfma_busy/busy_cu_cycles = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CU_CYCLES']:.3f}")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    print(f"   {n}/WAVE_CYCLES = {c[n] / c['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   lds_conflict/idx_active = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")


if __name__ == "__main__":
    main()
