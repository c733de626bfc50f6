#!/bin/bash
# A/B: bench --lean of workloads with alternative builds of the library (RTX950_LIB)
for lib in "$@"; do
  for w in ${WL:-minecraft_4k minecraft_1080p}; do
    for k in 2 1; do
      RTX950_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $w --lean --inflight $k --steps 300 ${EXTRA} 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w inflight $k: %.4f ms/frame  kernel %.1f us' % (r['ms_per_step'], r['roofline']['kernel_avg_ms']*1e3))"
    done
  done
done
