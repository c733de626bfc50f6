"""What one rank of an N-GPU job can do by itself (no collective): its part of the 4K frame rendered
back-to-back with 1, 2 or 3 frames in flight.  Prints ms per frame, slowest part of each N.
Usage: python tools/sweep_part_depth.py [workload] [grid limit with frames in flight, 0 = library default] [comm|-] [RT_TUNE_HEAVY_CLASS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracing_amd as rt  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "minecraft_4k"
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind, kwargs, W, H = bench.WORKLOADS[workload]
scene = bench.make_scene(rt, kind, kwargs)
frame = scene.frame(W, H)
dev = torch.device("cuda", 0)
ctxs = [rt.Context(0) for _ in range(4)]
heavy = int(sys.argv[4]) if len(sys.argv) > 4 else -1
for c in ctxs:
    c.upload(scene)
    c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, heavy)
bufs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in ctxs]
torch.cuda.synchronize()
frames = 200
# "comm" as a third argument: a stand-in for RCCL's stream - after every frame a 2 MB device copy on a
# stream of its own, fenced against the frame's render stream both ways (as ProcessGroupNCCL orders a
# collective), so that the hardware queues are shared as they would be in an N-rank run
comm = torch.cuda.Stream(device=dev) if (len(sys.argv) > 3 and sys.argv[3] == "comm") else None
streams = [torch.cuda.ExternalStream(c.stream(), device=dev) for c in ctxs]
msg = [torch.empty(512 * 1024, dtype=torch.int32, device=dev) for _ in range(2)]
for n in (1, 2, 4, 8):
    for depth in ((2, 3, 4) if comm is not None else (1, 2, 3)):
        worst = 0.0
        for c in ctxs:
            c.set_grid_limit(limit if depth > 1 else 0)
        for part in range(n):
            for rep in range(2):  # the first repetition settles the tile costs
                for c in ctxs:
                    c.sync()
                t0 = time.perf_counter()
                for f in range(frames):
                    s = f % depth
                    ctxs[s].render_device(frame, bufs[s].data_ptr(), n_parts=n, part=part)
                    if comm is not None:
                        e1 = torch.cuda.Event(); e1.record(streams[s]); comm.wait_event(e1)
                        with torch.cuda.stream(comm):
                            msg[1].copy_(msg[0])
                        e2 = torch.cuda.Event(); e2.record(comm); streams[s].wait_event(e2)
                for c in ctxs:
                    c.sync()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / frames * 1e3
            worst = max(worst, ms)
        print(f"{workload} grid_limit={limit if depth > 1 else 0} heavy_class={heavy} N={n} frames_in_flight={depth}: slowest part {worst:.4f} ms/frame", flush=True)
