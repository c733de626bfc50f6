#!/bin/bash
# Executed-work evidence per workload and kernel variant: for every bench workload a
# rocprofv3 kernel trace of `bench.py --lean` (the timed launches and nothing else of the render kernel) and
# four PMC passes of the same command with fewer steps (counter passes serialise the launches), each in a
# run of its own as the harness asks.  Usage (on the GPU box): bash tools/profile_workloads.sh <tag> [workloads...]
#   -> gpurun_out/prof_<tag>/<workload>/{trace,pmc_*}; summarise with tools/summarize_workloads.py
set -o pipefail
# (a gpurun merge adds files to what an earlier run of the same tag left in gpurun_out/: delete the local directory first)
TAG=${1:-r2}; shift || true
WORKLOADS=${*:-"minecraft_4k minecraft_1080p minecraft_water_4k synthetic_4k minecraft_8k_orbit"}
export TMPDIR=/tmp
for w in $WORKLOADS; do
  OUT=$PWD/gpurun_out/prof_$TAG/$w
  rm -rf "$OUT"; mkdir -p "$OUT"
  steps=400; psteps=60
  case $w in synthetic_4k|minecraft_8k_orbit) steps=200; psteps=30;; esac
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --workload $w --lean --steps $steps > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { echo "trace failed: $w"; exit 1; }
  # the same command one frame at a time: the kernel's duration with the GPU to itself
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_one" -- python3 bench.py --workload $w --lean --inflight 1 --steps $steps > "$OUT/bench_trace_one.json" 2> "$OUT/trace_one.err" || { echo "trace (one in flight) failed: $w"; exit 1; }
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
              "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --workload $w --lean --inflight 1 --steps $psteps --warmup 5 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || { echo "pmc pass failed: $w $pass"; exit 1; }
  done
  echo "profiled $w"
done
