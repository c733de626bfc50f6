#!/bin/bash
# Executed VALU instructions per launch of the render kernel (rocprofv3 SQ_INSTS_VALU: deterministic, unlike timings)
# for each library given, on one workload.  Usage (on the GPU box): WL=minecraft_4k bash tools/ab_valu.sh abtest/a.so abtest/b.so
set -o pipefail
export TMPDIR=/tmp
WL=${WL:-minecraft_4k}
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  d=$PWD/gpurun_out/valu_$tag
  rm -rf "$d"
  RTX950_LIB=$PWD/$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$d" -- python3 bench.py --workload $WL --lean --inflight 1 --steps 20 --warmup 5 --cpu-seconds 0 > "$d.log" 2>&1 || { echo "pmc pass failed: $lib"; exit 1; }
  echo "$lib $WL: $(python3 tools/pmc_kernels.py "$d" 2 | grep 'render_kernel<')"
done
