for sc in 0.5 1 2 4 8 16 1e9; do
  echo "scale $sc"; RTX950_COHERENT_SCALE=$sc bash tools/ab_bench.sh raytracing_amd/librtx950.so | head -1
  RTX950_COHERENT_SCALE=$sc timeout -k 10 200 python bench.py --workload synthetic_4k --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   synthetic ms=%.3f'%d['ms_per_step'])"
done
