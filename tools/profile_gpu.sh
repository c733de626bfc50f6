#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -e -o pipefail
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS="--cpu-seconds 0 $*"  # bench.py's default steps and warmup: the traced command is the measured one
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $BENCH_ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace done"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pass failed: $pass"
  echo "pmc pass done: $pass"
done
find "$OUT" -name "*.csv" | head -50
