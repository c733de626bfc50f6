#!/bin/bash
# tools/sweep_grid.sh: render_kernel grid size sweep (RTX950_GRID, read by rt_ctx_create)
for g in 4096 8192 12288 16384 24576 32768 65536 200000; do
  echo -n "grid $g: "; RTX950_GRID=$g bash tools/ab_bench.sh raytracing_amd/librtx950.so | head -1
done
