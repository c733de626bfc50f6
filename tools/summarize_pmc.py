#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs (kernel stats + per-dispatch PMC) of a tools/profile_gpu.sh run.
usage: tools/summarize_pmc.py gpurun_out/prof_<tag> [--traffic profiles/traffic.json <workload>] [kernel-substring]

--traffic records the render kernel's HBM bytes per launch for bench.py's roofline.traffic:
2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; the factor 2 is MI355X_MICROARCH.md's gfx950
correction for FETCH_SIZE, calibrated there on wide streaming reads)."""
import collections
import csv
import glob
import json
import os
import sys

args = sys.argv[1:]
root = args.pop(0)
traffic_out = workload = None
if args and args[0] == "--traffic":
    traffic_out, workload = args[1], args[2]
    args = args[3:]
needle = args[0] if args else "render_kernel<false, false"
def newest(pattern):
    """one file per directory: the most recent (gpurun merges new results into old local directories)"""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest(os.path.join(root, "trace", "*", "*_kernel_stats.csv")):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print(f"  {r['Name'][:90]:90s} calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} pct={r['Percentage']}")
vals = {}
for f in newest(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    if agg:
        print("== pmc:", os.path.basename(os.path.dirname(os.path.dirname(f))), "kernel", needle, "grid,wg,vgpr,agpr,sgpr,lds,scratch =", meta)
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
            print(f"  {k:28s} dispatches={len(v):3d} mean={sum(v) / len(v):.6g}")
if traffic_out and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    try:
        t = json.load(open(traffic_out))
    except (OSError, ValueError):
        t = {}
    t[workload] = {
        "kernel": needle, "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
        "hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
        # executed view: wave64 VALU instructions per launch and the share of SIMD issue time they take
        # (SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs)
        "valu_insts_per_launch": vals.get("SQ_INSTS_VALU"),
        "valu_busy_frac": (vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
                          if "SQ_ACTIVE_INST_VALU" in vals and vals.get("GRBM_GUI_ACTIVE") else None,
        "source": os.path.basename(os.path.normpath(root)),
    }
    json.dump(t, open(traffic_out, "w"), indent=1, sort_keys=True)
    print("== traffic ->", traffic_out, t[workload])
