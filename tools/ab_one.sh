#!/bin/bash
# One-frame-at-a-time frame time of library builds on a few workloads: tools/ab_one.sh lib.so ... (WL = workloads)
for lib in "$@"; do
  line="$lib:"
  for w in ${WL:-minecraft_4k minecraft_1080p synthetic_4k}; do
    v=$(RTX950_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload $w --lean --inflight 1 --steps 200 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % r['ms_per_step'])")
    line="$line $w $v"
  done
  echo "$line"
done
