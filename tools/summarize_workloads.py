#!/usr/bin/env python3
"""Summarise tools/profile_workloads.sh output: per workload and render-kernel instantiation, the traced
duration of the timed launches and the PMC counters per launch, written to profiles/<tag>/ (text summary,
kernel stats CSVs) and profiles/traffic.json (keyed by workload, then kernel variant; bench.py reads it).

usage: tools/summarize_workloads.py gpurun_out/prof_<tag> profiles/<tag>

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KB; the factor 2 is MI355X_MICROARCH.md's gfx950
correction for FETCH_SIZE).  Executed lane-operations per launch = SQ_THREAD_CYCLES_VALU (one per active lane
and VALU instruction); VALU busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
PEAK_LANE_OPS = 39.3e12
traffic_path = os.path.join(os.path.dirname(os.path.normpath(out)), "traffic.json")
try:
    traffic = json.load(open(traffic_path))
except (OSError, ValueError):
    traffic = {}
lines = []


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None


def canon(name):
    """'void rtx950::(anonymous namespace)::render_kernel<false, false, true, false>(rtx950::Params)' -> 'render_kernel<false,false,true,false>'"""
    i = name.find("render_kernel<")
    return name[i: name.index(">", i) + 1].replace(" ", "") if i >= 0 else None


def timed_region(trace_csv, bench):
    """Average duration (us) of the plain render kernel over the timed launches of a --lean run, per kernel."""
    rows = [r for r in csv.DictReader(open(trace_csv)) if canon(r["Kernel_Name"]) == bench["roofline"]["kernel"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    w, k = bench["warmup"] + bench["config"].get("settle_frames", 0), bench["steps"]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[w:w + k]]
    period = (int(rows[w + k - 1]["End_Timestamp"]) - int(rows[w]["Start_Timestamp"])) / k / 1e3 if len(rows) >= w + k else None
    return (sum(d) / len(d) if d else None), len(d), len(rows), period


for wdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(wdir):
        continue
    w = os.path.basename(wdir)
    try:
        bench = json.loads(open(os.path.join(wdir, "bench_trace.json")).read().strip().splitlines()[-1])
        bench_one = json.loads(open(os.path.join(wdir, "bench_trace_one.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError) as e:
        lines.append(f"== {w}: no bench line ({e})")
        continue
    kernel = bench["roofline"]["kernel"]
    lines.append(f"== {w}: kernel {kernel}; bench.py --lean: {bench['ms_per_step']:.4f} ms/frame with {bench['config']['frames_in_flight']} in flight, "
                 f"{bench_one['ms_per_step']:.4f} ms/frame one at a time; HIP-event kernel time {bench['roofline']['kernel_avg_ms'] * 1e3:.1f} / {bench_one['roofline']['kernel_avg_ms'] * 1e3:.1f} us")
    entry = {"source": os.path.basename(os.path.normpath(out))}
    for tag, b in (("trace", bench), ("trace_one", bench_one)):
        st = newest(os.path.join(wdir, tag, "*", "*_kernel_stats.csv"))
        tr = newest(os.path.join(wdir, tag, "*", "*_kernel_trace.csv"))
        if st:
            shutil.copy(st, os.path.join(out, f"{w}_{tag}_kernel_stats.csv"))
            for i, r in enumerate(csv.DictReader(open(st))):
                if i < 4:
                    lines.append(f"   [{tag}] {r['Name'][:100]:100s} calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} pct={r['Percentage']}")
        if tr:
            avg, n, total, period = timed_region(tr, b)
            if avg:
                lines.append(f"   [{tag}] timed region: {n} of {total} launches, avg {avg:.1f} us, frame period {period:.1f} us | bench.py HIP events {b['roofline']['kernel_avg_ms'] * 1e3:.1f} us")
                entry["kernel_us_traced" if tag == "trace" else "kernel_us_traced_one_in_flight"] = avg
    vals = {}
    for f in sorted(glob.glob(os.path.join(wdir, "pmc_*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if canon(r["Kernel_Name"]) == kernel:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = v[len(v) // 4:]  # the run-in launches' tile order is still settling
            vals[k] = sum(v) / len(v)
            lines.append(f"   pmc {k:26s} launches={len(v):3d} mean={vals[k]:.6g}")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        entry.update(fetch_size_kb=vals["FETCH_SIZE"], write_size_kb=vals["WRITE_SIZE"],
                     hbm_bytes_per_launch=(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
    if "SQ_INSTS_VALU" in vals:
        entry["valu_insts_per_launch"] = vals["SQ_INSTS_VALU"]
        if "SQ_THREAD_CYCLES_VALU" in vals:
            entry["lane_ops_per_launch"] = vals["SQ_THREAD_CYCLES_VALU"]
            entry["active_lanes_per_inst"] = vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_INSTS_VALU"]
            dur = entry.get("kernel_us_traced_one_in_flight")
            if dur:
                entry["lane_ops_per_s"] = vals["SQ_THREAD_CYCLES_VALU"] / (dur * 1e-6)
                entry["frac_of_lane_op_ceiling"] = entry["lane_ops_per_s"] / PEAK_LANE_OPS
                entry["valu_issue_frac"] = vals["SQ_INSTS_VALU"] * 4.0 / (dur * 1e-6 * 2.4e9 * 1024.0)  # wave64 VALU = 4 cycles of a SIMD, 1024 SIMDs at 2.4 GHz
    if "SQ_ACTIVE_INST_VALU" in vals and vals.get("GRBM_GUI_ACTIVE"):
        entry["valu_busy_frac"] = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    ex = bench_one["roofline"].get("executed_flops_per_launch")
    if ex and entry.get("lane_ops_per_launch"):
        entry["modelled_fp32_flops_per_launch"] = ex
        entry["modelled_flops_per_executed_lane_op"] = ex / entry["lane_ops_per_launch"]
    traffic.setdefault(w, {}).setdefault("kernels", {})[kernel] = entry
    lines.append("   -> " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in entry.items()}))

open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)
print("\n".join(lines))
