#!/bin/bash
# tools/pmc_script.sh <tag> <script.py>: SQ instruction counters per dispatch for a python script
TAG=$1; SCRIPT=$2
OUT=$PWD/gpurun_out/pmcs_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT" -- python3 $SCRIPT > "$OUT/out.txt" 2> "$OUT/err.txt" || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    per=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'render_kernel<false, false' not in r['Kernel_Name']: continue
        per.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
    last=None
    for d,v in per.items():
        key=(round(v.get('SQ_INSTS_VALU',0)/1e5),)
        if key!=last:
            print('dispatch',d,{k:'%.4g'%x for k,x in v.items()})
            last=key
PY
