for g in 24576 49152 98304; do for w in minecraft_8k minecraft_water_4k minecraft_1080p synthetic_4k; do
 echo -n "grid $g $w: "; RTX950_GRID=$g timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms=%.4f'%d['ms_per_step'])"
done; done
