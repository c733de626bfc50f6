#!/bin/bash
# A/B several builds of librtx950 in one gpurun call: tools/ab_bench.sh lib1.so lib2.so ... [-- bench args]
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for round in 1 2; do
for l in "${libs[@]}"; do
  RTX950_LIB=$PWD/$l timeout -k 10 300 python -u bench.py --steps 400 --warmup 40 --cpu-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', 'ms=%.4f'%d['ms_per_step'], 'kern_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'Mrays/s=%.0f'%d['value'])"
done; done
