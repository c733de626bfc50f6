"""Average duration of the plain render kernel per region of one traced bench.py run (rocprofv3
--kernel-trace): warm-up, the timed region (what roofline.kernel_avg_ms measures with HIP events), and the
one-frame-in-flight side measurement.  Usage: trace_regions.py <kernel_trace.csv> <bench json> """
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "render_kernel<false, false" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bench = json.load(open(sys.argv[2]))
w, k = bench["warmup"] + bench["config"].get("settle_frames", 0), bench["steps"]  # run-in + warm-up, timed
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
start = [int(r["Start_Timestamp"]) for r in rows]
end = [int(r["End_Timestamp"]) for r in rows]


def region(name, a, b, events_ms):
    if b > len(dur) or b <= a:
        return
    d = dur[a:b]
    line = f"  {name:<38} launches {a + 1}..{b}: avg {sum(d) / len(d):7.1f} us  min {min(d):7.1f}  max {max(d):7.1f}  frame period {(end[b - 1] - start[a]) / (b - a) / 1e3:7.1f} us"
    if events_ms:
        line += f"   | bench.py HIP events: {events_ms * 1e3:.1f} us"
    print(line)


print(f"== render kernel by region of the traced run ({len(rows)} launches; run-in + warmup {w}, steps {k}, {bench['config'].get('frames_in_flight')} frame(s) in flight)")
region("run-in + warm-up", 0, w, None)
region("timed region", w, w + k, bench["roofline"]["kernel_avg_ms"])
side = bench["config"].get("one_frame_in_flight")
if side:
    n_side = len(rows) - (w + k)
    skip = max(0, n_side - k)  # the side measurement's own warm-up and the frame check come first / last
    region("one frame in flight (side measurement)", w + k + skip - 1 if skip else w + k, w + 2 * k + skip - 1 if skip else w + 2 * k, side["kernel_avg_ms"])
