#!/bin/bash
# ms per frame of every benchmark workload with the in-tree library, two and one frame in flight: tools/ab_all.sh [bench args]
for w in ${WL:-minecraft_4k minecraft_1080p minecraft_water_4k synthetic_4k minecraft_8k minecraft_8k_orbit}; do
  for k in 2 1; do
    timeout -k 10 120 python3 bench.py --workload $w --lean --inflight $k --steps 300 --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w inflight $k: %.4f ms/frame  kernel %.1f us  %s' % (r['ms_per_step'], r['roofline']['kernel_avg_ms']*1e3, r['roofline']['kernel']))"
  done
done
