#!/bin/bash
# A/B of library builds x kernel variants (RT_TUNE_CULL_REFINE): tools/ab_refine.sh lib.so ... ; WL, REF (list of refine values), KS (frames in flight)
for lib in "$@"; do
  for w in ${WL:-minecraft_4k minecraft_1080p minecraft_water_4k}; do
    for r in ${REF:-1 2}; do
      for k in ${KS:-2 1}; do
        RTX950_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $w --lean --inflight $k --steps 300 --cpu-seconds 0 --refine $r 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w refine $r inflight $k: %.4f ms/frame  kernel %.1f us' % (r['ms_per_step'], r['roofline']['kernel_avg_ms']*1e3))"
      done
    done
  done
done
