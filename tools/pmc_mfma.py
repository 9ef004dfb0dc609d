#!/usr/bin/env python3
"""Matrix-pipe utilisation of the contraction kernels from ONE rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE: 8 SQ slots; GRBM_GUI_ACTIVE rides in the GRBM block).

    python tools/pmc_mfma.py <pmc dir> out.json <launches per forward> [note] [layers.txt]

(layers.txt: tools/layer_report.py's table of the same configuration; with it the launches of the forward are also
written one by one, in order, under their layer names, as <out minus .json>.per_layer.txt)

Uses the contraction dispatches of the LAST forward of the run (as tools/pmc_traffic.py does).  Per kernel
instantiation and for the family as a whole:

  mfma_busy            SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles =
                       GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs): the share of ALL the
                       chip's SIMD-cycles during the launches in which a matrix pipe was busy -- times the
                       clock the chip held, this is what the achieved TFLOP/s is made of;
  mfma_busy_in_busy_cu SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the same inside the CUs that had
                       a wave (tails and ramps taken out);
  wait_inst / wait_any SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES (issue stalls; parked on s_waitcnt or
                       a barrier), lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.

The JSON carries the digest of the kernel sources (resnet_c_amd._lib.source_digest) and the launch count;
bench.py quotes roofline.mfma_busy from it only for the build it was measured on.  The per-kernel table goes
next to it as <out minus .json>.per_kernel.txt.
"""
import collections
import csv
import glob
import json
import os
import sys

FAMILY = ("conv_gemm_kernel", "conv_wide_kernel", "conv_strip_kernel", "conv_strip128_kernel", "chain_kernel",
          "chain32_", "splitk_finish_kernel", "stem_pool_kernel", "conv_nchw_kernel", "conv_pair_kernel",
          "conv_fused23_kernel")
CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    """rocprofv3 leaves names with a bf16 template argument mangled (_ZN12_GLOBAL__N_116conv_wide_kernelIDF16bLi224E...)."""
    import re
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(.*?)EEvN", name)
    if m:
        args = []
        for a in re.findall(r"DF16b|Li\d+E|Lb[01]E|f", m.group(2)):
            args.append("bf16" if a == "DF16b" else "float" if a == "f" else a[2:-1] if a[1] == "i" else
                        ("true" if a[2] == "1" else "false"))
        return f"{m.group(1)}<{', '.join(args)}>"
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rn_gemm::", "")
    return n.split("(")[0][:110]


def dispatches(d):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = by.setdefault(k, {"name": r["Kernel_Name"], "c": collections.defaultdict(float)})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def ratios(c):
    out = {}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    if cyc > 0:
        out["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * CUS * cyc)
        out["cu_busy"] = c.get("SQ_BUSY_CU_CYCLES", 0.0) / (CUS * cyc)
    if c.get("SQ_BUSY_CU_CYCLES"):
        out["mfma_busy_in_busy_cu"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * c["SQ_BUSY_CU_CYCLES"])
    if c.get("SQ_WAVE_CYCLES"):
        out["wait_inst"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        out["wait_any"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    return out


def main():
    d, out_path, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from resnet_c_amd._lib import source_digest

    rows = [e for e in dispatches(d) if any(k in e["name"] for k in FAMILY)][-n:]
    total = collections.defaultdict(float)
    per = collections.OrderedDict()
    for e in rows:
        p = per.setdefault(short(e["name"]), {"launches": 0, "c": collections.defaultdict(float)})
        p["launches"] += 1
        for k, v in e["c"].items():
            p["c"][k] += v
            total[k] += v
    table = os.path.splitext(out_path)[0] + ".per_kernel.txt"
    with open(table, "w") as fh:
        fh.write("# one forward; mfma_busy = MFMA busy cycles / (4 SIMDs x 256 CUs x kernel cycles); in_busy_cu = / (4 x CU-busy cycles)\n")
        fh.write("launches  kernel_us mfma_busy in_busy_cu cu_busy wait_inst wait_any lds_confl  MFMA insts   kernel instantiation\n")
        for k, p in sorted(per.items(), key=lambda kv: -kv[1]["c"].get("GRBM_GUI_ACTIVE", 0.0)):
            r, c = ratios(p["c"]), p["c"]
            fh.write(f"{p['launches']:8d} {c.get('GRBM_GUI_ACTIVE', 0.0) / XCDS / 2.4e3:10.1f} {r.get('mfma_busy', 0):9.3f} "
                     f"{r.get('mfma_busy_in_busy_cu', 0):10.3f} {r.get('cu_busy', 0):7.3f} {r.get('wait_inst', 0):9.3f} "
                     f"{r.get('wait_any', 0):8.3f} {r.get('lds_conflict', 0):9.3f} {c.get('SQ_INSTS_MFMA', 0):12.0f}   {k}\n")
        fh.write("# kernel_us: GRBM_GUI_ACTIVE / 8 at a nominal 2.4 GHz (the chip holds less under load: durations come from the traces)\n")
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        layers = []
        for line in open(sys.argv[5]):
            q = line.split()
            if len(q) >= 6 and (q[0].startswith("conv2d") or q[0] == "linear"):
                layers.append((q[1], float(q[2]), float(q[3])))  # name, ms by events, TFLOP/s
        groups = []  # a splitk_finish_kernel belongs to the launch in front of it
        for e in rows:
            if "splitk_finish" in e["name"] and groups:
                for k, v in e["c"].items():
                    groups[-1]["c"][k] += v
                groups[-1]["name"] += " +finish"
            else:
                groups.append({"name": short(e["name"]), "c": collections.defaultdict(float, e["c"])})
        with open(os.path.splitext(out_path)[0] + ".per_layer.txt", "w") as fh:
            if len(groups) != len(layers):
                fh.write(f"# {len(groups)} contraction launches against {len(layers)} table rows: names not attached\n")
                layers = [("?", 0.0, 0.0)] * len(groups)
            fh.write("# one forward, launch by launch; ms / TF/s: HIP events of an un-profiled run (tools/layer_report.py)\n")
            fh.write(f"{'layer':42s} {'ms':>6s} {'TF/s':>6s} {'mfma_busy':>9s} {'in_busy_cu':>10s} {'cu_busy':>7s} {'wait_inst':>9s} "
                     f"{'wait_any':>8s} {'lds_confl':>9s}  kernel\n")
            for (name, ms, tf), g in zip(layers, groups):
                r = ratios(g["c"])
                fh.write(f"{name:42s} {ms:6.3f} {tf:6.1f} {r.get('mfma_busy', 0):9.3f} {r.get('mfma_busy_in_busy_cu', 0):10.3f} "
                         f"{r.get('cu_busy', 0):7.3f} {r.get('wait_inst', 0):9.3f} {r.get('wait_any', 0):8.3f} "
                         f"{r.get('lds_conflict', 0):9.3f}  {g['name']}\n")
    r = ratios(total)
    out = {"kernel": " + ".join(FAMILY), "launches": len(rows), "source_digest": source_digest(),
           "mfma_busy": round(r.get("mfma_busy", 0.0), 4),
           "mfma_busy_in_busy_cu": round(r.get("mfma_busy_in_busy_cu", 0.0), 4),
           "cu_busy": round(r.get("cu_busy", 0.0), 4),
           "wait_inst_over_wave_cycles": round(r.get("wait_inst", 0.0), 4),
           "wait_any_over_wave_cycles": round(r.get("wait_any", 0.0), 4),
           "lds_bank_conflict_over_idx_active": round(r.get("lds_conflict", 0.0), 4),
           "counters": {k: v for k, v in sorted(total.items())},
           "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8), contraction "
                         "launches of the last forward of one rocprofv3 --pmc pass",
           "note": note}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("launches", "mfma_busy", "mfma_busy_in_busy_cu", "cu_busy")}))


if __name__ == "__main__":
    main()
