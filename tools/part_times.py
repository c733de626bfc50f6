#!/usr/bin/env python3
"""tools/part_times.py: what one GPU of an N-GPU job has to do, measured on one GPU: for N = 1, 2, 4, 8 render
part p of N of the workload's frame (rt_render_device_part) and report the frame time (3 launches,
host-timed over 20 frames) and the render kernel's HIP-event time, for every part."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import raytracing_amd as rt  # noqa: E402
from raytracing_amd import multigpu  # noqa: E402
from bench import WORKLOADS, make_scene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "minecraft_4k"
kind, kwargs, W, H = WORKLOADS[name]
scene = make_scene(rt, kind, kwargs)
ctx = rt.Context(0)
ctx.upload(scene)
f = scene.frame(W, H)
out = {}
for n in (1, 2, 4, 8):
    buf = torch.zeros((multigpu.max_part_rows(H, n), W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rows = []
    for p in range(n):
        for _ in range(3):
            ctx.render_device(f, buf.data_ptr(), n_parts=n, part=p)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.kernel_time_ms()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.render_device(f, buf.data_ptr(), n_parts=n, part=p)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        k, _ = ctx.kernel_time_ms()
        ctx.enable_timing(False)
        rows.append((round(dt, 4), round(k, 4)))
    out[n] = rows
    print(f"{name} N={n}: frame ms max {max(r[0] for r in rows):.4f}  kernel ms max {max(r[1] for r in rows):.4f}  per part {rows}", flush=True)
print(json.dumps({name: out}))
