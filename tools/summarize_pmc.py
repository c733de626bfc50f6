#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs (kernel stats + per-dispatch PMC) of a tools/profile_gpu.sh run.
usage: tools/summarize_pmc.py gpurun_out/prof_<tag> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "render_kernel<false, false>"
for f in glob.glob(os.path.join(root, "trace", "*", "*_kernel_stats.csv")):
    print("== kernel stats:", f)
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 6:
            print(f"  {r['Name'][:90]:90s} calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} pct={r['Percentage']}")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    if agg:
        print("== pmc:", os.path.basename(os.path.dirname(os.path.dirname(f))), "grid,wg,vgpr,agpr,sgpr,lds,scratch =", meta)
        for k, v in agg.items():
            print(f"  {k:28s} dispatches={len(v):3d} mean={sum(v) / len(v):.6g}")
