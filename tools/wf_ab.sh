#!/bin/bash
# A/B of the wavefront path against the tile walk: ms per frame per workload, one and two frames in flight
mkdir -p gpurun_out
for wl in minecraft_1080p minecraft_4k minecraft_water_4k synthetic_4k minecraft_8k; do
  for inf in 1 2; do
    for wf in 0 1; do
      timeout -k 10 120 python bench.py --workload $wl --inflight $inf --wavefront $wf --lean --steps 200 --cpu-seconds 0 > gpurun_out/wf_ab_tmp.json 2>gpurun_out/wf_ab_err.txt || { echo "FAILED $wl $inf $wf"; tail -3 gpurun_out/wf_ab_err.txt; exit 1; }
      python - "$wl" "$inf" "$wf" <<'PY'
import json,sys
r=json.load(open("gpurun_out/wf_ab_tmp.json"))
print(f"{sys.argv[1]:22s} inflight {sys.argv[2]} wavefront {sys.argv[3]}: {r['ms_per_step']:.4f} ms/frame  kernel {r['roofline']['kernel_avg_ms']:.4f} ms", flush=True)
PY
    done
  done
done
