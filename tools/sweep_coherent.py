#!/usr/bin/env python3
"""Frame time of the 10k-cube world at 4K (two frames in flight, grid limit as bench.py) against the factor on the
long-list traversal threshold (RT_TUNE_COHERENT_SCALE: below it a wave takes the chunk traversal, above it every lane
walks the BVH): python tools/sweep_coherent.py [scale ...]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
import raytracing_amd as rt  # noqa: E402

scales = [float(a) for a in sys.argv[1:]] or [0.25, 0.5, 1.0, 2.0, 4.0]
kind, kwargs, W, H = bench.WORKLOADS["synthetic_4k"]
scene = bench.make_scene(rt, kind, kwargs)
frame = scene.frame(W, H)
for sc in scales:
    ctxs = [rt.Context(0) for _ in range(2)]
    outs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in ctxs]
    for c in ctxs:
        c.set_tuning(rt.RT_TUNE_COHERENT_SCALE, sc)
        c.upload(scene)
        c.set_grid_limit(12288)

    def run(n):
        for k in range(n):
            ctxs[k % 2].render_device(frame, outs[k % 2].data_ptr())
        for c in ctxs:
            c.sync()

    run(30)
    t0 = time.perf_counter()
    run(120)
    dt = time.perf_counter() - t0
    print(f"scale {sc:g}: {dt / 120 * 1e3:.4f} ms per frame", flush=True)
    for c in ctxs:
        c.close()
