#!/usr/bin/env python3
"""Per-lane events the counting variant of the kernel executes for one frame of a bench workload
(rt_count_executed): python tools/executed_events.py synthetic_4k"""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
import raytracing_amd as rt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_4k"
kind, kwargs, W, H = bench.WORKLOADS[name]
scene = bench.make_scene(rt, kind, kwargs)
with rt.Context(0) as ctx:
    ctx.upload(scene)
    _, ex = ctx.count_executed(scene.frame(W, H))
    print(json.dumps({name: {k: getattr(ex, k) for k, _ in ex._fields_}}))
