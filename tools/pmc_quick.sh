#!/bin/bash
# tools/pmc_quick.sh <tag> <lib.so>: one SQ counter pass of bench.py for a given library build
TAG=$1; LIB=$2
OUT=$PWD/gpurun_out/pmcq_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
export RTX950_LIB=$PWD/$LIB
timeout -s KILL 240 rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU} --output-format csv -d "$OUT" -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 > "$OUT/bench.json" 2> "$OUT/err.txt" || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'render_kernel<false, fal' in k: print(k[-30:], {c: '%.4g'%(sum(x)/len(x)) for c,x in v.items()}, 'n=%d'%len(next(iter(v.values()))))
PY
