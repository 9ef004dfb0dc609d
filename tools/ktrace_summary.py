#!/usr/bin/env python3
"""Per-kernel durations of the TIMED forwards in a rocprofv3 --kernel-trace of bench.py.

bench.py launches every tile candidate while it tunes, so the --stats table of the whole
process mixes those launches in.  This takes the last F forwards of the trace (F = --steps +
--profile-forwards; a forward = the launches between two input-layout kernels) and reports,
per kernel family, launches per forward and the average duration: the number bench.py's
`roofline` is computed from (its HIP events) must agree with it.

    python tools/ktrace_summary.py <dir with *_kernel_trace.csv> F [out.json]"""
import csv, glob, json, os, sys


def family(name):
    if "conv_gemm_kernel" in name:
        return "conv_gemm_kernel"
    if "conv_wide_kernel" in name:
        return "conv_wide_kernel"
    if "stem_pool_kernel" in name:
        return "stem_pool_kernel"
    if "conv_strip_kernel" in name or "conv_strip128_kernel" in name:
        return "conv_strip_kernel"
    if "chain_kernel" in name or "chain32_" in name:
        return "chain_kernel"
    if "maxpool3_nhwc_kernel" in name:
        return "maxpool3_nhwc_kernel"
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0].split("<")[0]


def main():
    d, F = sys.argv[1], int(sys.argv[2])
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
                   for r in csv.DictReader(open(f))), key=lambda r: r[0])
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[2]]
    first = starts[-F]
    sel = rows[first:]
    fam = {}
    for s, e, n in sel:
        a = fam.setdefault(family(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    span_ns = sel[-1][1] - sel[0][0]
    out = {"forwards": F, "span_ms_per_forward": span_ns / F / 1e6,
           "kernels": {k: {"launches_per_forward": v[0] / F, "avg_us": v[1] / v[0] / 1e3,
                           "ms_per_forward": v[1] / F / 1e6} for k, v in sorted(fam.items())}}
    # the contraction family as bench.py's `roofline` counts it: the implicit-GEMM launches plus
    # the launches that add the K-chunk pieces of cut tail tiles
    con = [v for k, v in fam.items()
           if k in ("conv_gemm_kernel", "conv_wide_kernel", "conv_strip_kernel", "chain_kernel", "splitk_finish_kernel", "stem_pool_kernel")]
    if con:
        n, ns = sum(v[0] for v in con), sum(v[1] for v in con)
        out["contraction_family"] = {"launches_per_forward": n / F, "avg_us": ns / n / 1e3,
                                     "ms_per_forward": ns / F / 1e6}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
