#!/usr/bin/env python3
"""Mean of a rocprofv3 --pmc counter per (kernel instantiation, grid size), in dispatch order of
first appearance -- for A/B runs of one shape over tile candidates (tools/conv_bench.py).

    python tools/pmc_per_dispatch.py <dir> [COUNTER] [scale]     (FETCH_SIZE: KiB, x2 on gfx950 -> scale 2048)"""
import collections, csv, glob, os, sys

d = sys.argv[1]
counter = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else (2048.0 if counter == "FETCH_SIZE" else 1024.0)
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("rn_gemm::", "").split("(")[0]
    k = (n[:90], r.get("Grid_Size", "?"))
    agg.setdefault(k, []).append(float(r["Counter_Value"]) * scale)
for (n, g), v in agg.items():
    print(f"{len(v):4d} x  {sum(v) / len(v) / 1e6:10.1f} MB  grid {g:>9s}  {n}")
