#!/usr/bin/env python3
"""Register / LDS budget of every kernel in a hipcc -S dump (amdhsa metadata).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only -o /tmp/x.s file.hip
    python tools/isa_regs.py /tmp/x.s"""
import re, sys
cur = {}
for line in open(sys.argv[1]):
    m = re.match(r"\s+[-.]?\s*\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
    if not m:
        m2 = re.match(r"\s+- \.(agpr_count):\s+(\S+)", line)
        if m2: m = m2
        else: continue
    k, v = m.group(1), m.group(2)
    if k == "agpr_count" and cur.get("agpr_count") is not None and "name" in cur and "vgpr_count" in cur:
        pass
    cur[k] = v
    if k == "vgpr_spill_count":
        pass
    if all(x in cur for x in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "group_segment_fixed_size")):
        n = cur["name"]
        n = re.sub(r"_ZN\d+_GLOBAL__N_\d+", "", n)
        print(f"{n[:70]:70s} vgpr {cur['vgpr_count']:>4s} agpr {cur.get('agpr_count','-'):>3s} sgpr {cur['sgpr_count']:>4s} spill {cur['vgpr_spill_count']:>3s} lds {cur['group_segment_fixed_size']}")
        cur = {}
