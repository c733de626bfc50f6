#!/bin/bash
# Everything profiles/<tag>/ holds, in one gpurun call: the traced + PMC'd default command (profile_gpu.sh),
# traces of the one-frame-in-flight and synthetic runs, and every workload at 1 and 2 frames in flight.
# Usage (on the GPU box): bash tools/profile_round.sh <tag>
set -e -o pipefail
TAG=${1:-r1}
export TMPDIR=/tmp
bash tools/profile_gpu.sh $TAG
for v in "one --inflight 1" "syn --workload synthetic_4k"; do
  set -- $v; name=$1; shift
  mkdir -p gpurun_out/prof_${TAG}_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$name/trace -- python3 bench.py --cpu-seconds 0 "$@" > gpurun_out/prof_${TAG}_$name/bench.json 2> gpurun_out/prof_${TAG}_$name/err.txt
done
for k in 1 2; do
  : > gpurun_out/prof_$TAG/bench_workloads_inflight$k.jsonl
  for w in minecraft_4k minecraft_1080p minecraft_water_4k minecraft_8k minecraft_8k_orbit synthetic_4k; do
    timeout -k 10 300 python3 bench.py --workload $w --inflight $k --cpu-seconds 0 2>/dev/null >> gpurun_out/prof_$TAG/bench_workloads_inflight$k.jsonl
  done
done
echo "profile_round done"
