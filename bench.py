#!/usr/bin/env python3
"""bench.py — Mrays/s of the MI355X render path (BASELINE.json metric) on N GPUs of one node.

A step = one frame of the workload rendered through the C-ABI (rt_render_device_part).  Two frames are
in flight on one GPU, three per rank of an N-GPU job (one library context = one stream each, frames
alternate; --inflight), every frame complete before the timed region ends.  At N>1 the
frame's 16-row blocks are dealt round-robin to the ranks (strong scaling, total work fixed) and each
rank's compact band (the rows that can hold geometry) is gathered to GPU0 over RCCL/xGMI inside the
timed region, overlapped with the rendering of the next frame.

    python bench.py --gpus 1 --steps 400 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = unique rays of the frame (nearest-hit + shadow list traversals,
SURVEY.md section 8(d)) x K / max-over-ranks wall time.  roofline: algorithmic FP32 ops
(26/cube test + 25/sphere test + 60 x lights/shaded hit) of rank 0's launch / its HIP-event time,
against the 157.3 TFLOP/s FP32 vector peak.  cpu_baseline: the oracle (reference algorithm,
per-light re-casting, all host threads) on a bounded row sample of the same frame, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene factory kwargs, width, height)
    "minecraft_4k": ("default", {}, 3840, 2160),
    "minecraft_water_4k": ("default", {"with_water": True}, 3840, 2160),
    "minecraft_1080p": ("default", {}, 1920, 1080),
    "minecraft_720p": ("default", {}, 1080, 720),
    "synthetic_4k": ("synthetic", {}, 3840, 2160),
    "minecraft_8k": ("default", {}, 7680, 4320),
    # BASELINE.json config 5: camera orbit of the Minecraft scene at 8K, one degree of yaw per step
    # (Camera::rotate_cam, reference src/camera.rs:61-86); rays/frame are counted on the first frame
    "minecraft_8k_orbit": ("default", {"orbit": True}, 7680, 4320),
}

PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (vector), packed FMA
PEAK_LANE_OPS_T = 39.3       # 256 CU x 64 lanes x 2.4 GHz: one non-packed, non-FMA op per lane-clock


def algorithmic_flops(c, n_lights):
    """26 FP32 ops per ray-cube test, 25 per ray-sphere test, ~60 per shaded hit per light."""
    return 26.0 * c.cube_tests + 25.0 * c.sphere_tests + 60.0 * n_lights * c.shaded_hits


def profile_field(workload, field):
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get(field)
    except (OSError, ValueError):
        return None


def traffic_from_profile(workload):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/traffic.json, written by tools/summarize_pmc.py: 2 x FETCH_SIZE KB + WRITE_SIZE KB
    as MI355X_MICROARCH.md prescribes for gfx950); None when no pass is on file for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        return t.get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def make_scene(rt, kind, kwargs):
    kwargs = {k: v for k, v in kwargs.items() if k != "orbit"}
    return rt.default_scene(**kwargs) if kind == "default" else rt.synthetic_scene(**kwargs)


def cpu_baseline(scene, width, height, target_s, max_threads):
    """Reference algorithm on the host (oracle, literal mode = per-light re-casting like the reference):
    whole frames of the same workload, repeated until about target_s seconds of CPU work are done.  A
    frame that alone would take longer than target_s is sampled by rows 0, step, 2*step, ... instead."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    O = orc.OracleScene(scene)
    frame = scene.frame(width, height)
    threads = max(1, min(max_threads, len(os.sched_getaffinity(0))))
    probe_step = max(1, height // 24)
    t0 = time.perf_counter()
    _, _, cnt, used = O.render(frame, reuse_secondary=False, row_step=probe_step, threads=threads)
    t_probe = time.perf_counter() - t0
    est_frame = t_probe * probe_step  # seconds for a whole frame
    step = max(1, int(round(est_frame / target_s))) if est_frame > target_s else 1
    rays, secs, frames = 0, 0.0, 0
    while True:
        t0 = time.perf_counter()
        _, _, cnt, used = O.render(frame, reuse_secondary=False, row_step=step, threads=threads)
        secs += time.perf_counter() - t0
        rays += cnt.nearest_rays + cnt.shadow_rays  # literal mode: what the reference casts
        frames += 1
        if secs >= 0.85 * target_s or frames >= 64:
            break
    n_rows = (height + step - 1) // step
    what = f"{frames} x the whole {width}x{height} frame" if step == 1 else \
           f"{frames} x rows 0::{step} ({n_rows} of {height} rows) of the {width}x{height} frame"
    return {
        "value": rays / secs / 1e6,
        "unit": "Mrays/s",
        "cores": int(used),
        "kind": "port",
        "sample": f"{what}: {rays} reference rays in {secs:.2f} s, oracle literal mode (per-light re-casting), pthreads over rows",
        "seconds": secs,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400,
                    help="timed frames; the default is long enough for the GPU's clocks to settle (20 frames are over in 4 ms and "
                         "measure 8 %% slower)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="minecraft_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target size of the CPU baseline sample; 0 disables it")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per GPU (one library context and stream each: the next frame starts under the current "
                         "frame's slowest tiles); 0 = the default: 2 at N = 1, 3 per rank at N > 1; 1 = strictly one frame at a time")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket every k-th launch of the render kernel with HIP events (roofline.kernel_avg_ms); the two event "
                         "records cost the stream about 6 us per launch, so timing every launch slows the frame it measures by 4 %%")
    ap.add_argument("--grid-limit", type=int, default=-1,
                    help="rt_set_grid_limit of every context; -1 = 12288 with frames in flight (two render kernels share the GPU), "
                         "the library's default (32768) with one")
    ap.add_argument("--extra", action="store_true", help="also time the other Minecraft workloads (N=1 only, untimed region)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracing_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU rendering path)", file=sys.stderr)
        sys.exit(2)
    # Rehearsal of the N-GPU path on a box with one GPU: RTX950_BENCH_BACKEND=gloo RTX950_BENCH_DEVICE=0 puts
    # every rank on the same card and moves the messages through host memory (RCCL refuses two ranks on
    # one device).  Not a measurement.
    backend = os.environ.get("RTX950_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("RTX950_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the small bookkeeping collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    kind, kwargs, W, H = WORKLOADS[args.workload]
    scene = make_scene(rt, kind, kwargs)
    ctx = rt.Context(device_index)
    ctx.upload(scene)
    frame = scene.frame(W, H)
    n_lights = len(scene.lights)

    from raytracing_amd.multigpu import FrameGather, FramePipeline
    # the library's own stream as a torch stream: the gather of a frame is ordered after its render by events
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    with torch.cuda.stream(ext):
        # default: 2 on one GPU (the frame is then bound by instruction issue); 3 per rank of an N-GPU job, whose
        # 1/N of the frame is bound by its slowest tile (tools/sweep_part_depth.py with a stand-in for RCCL's
        # stream: N = 4 and 8 gain 9-30 % over 2, N = 2 0-9 %, depending on how streams fall on hardware queues)
        inflight = args.inflight if args.inflight > 0 else (2 if world == 1 else 3)
        piped = world > 1  # N = 1 renders straight into its frames: no message, no scatter
        fg = FrameGather(H, W, world, rank, dev, depth=inflight)
        outs = fg.locals  # N = 1: one whole frame per frame in flight ([max_part_rows, W] = [H, W] there)
    out = fg.local  # [max_part_rows, W] int32, padded so every rank sends the same size
    # N > 1: frame f+1 renders while frame f is gathered to GPU0 and while f's slowest tiles finish (a rank
    # holds 1/N of the work but the whole critical path); every frame is whole on GPU0 before the timed
    # region ends (sync() below waits for all streams).  One library context (own stream, tile queue, tile
    # costs) per message buffer.  (GPU_MAX_HW_QUEUES = 4: with RCCL's stream and the null stream, 3 render
    # streams mean that two streams share a hardware queue; measured with a stand-in for RCCL's stream, the
    # third frame in flight still gains more than the sharing costs - see the default above.)
    ctxs, streams = [ctx], [ext]
    if inflight > 1:
        for _ in range(len(fg.locals) - 1):
            c2 = rt.Context(device_index)
            c2.upload(scene)
            ctxs.append(c2)
            streams.append(torch.cuda.ExternalStream(c2.stream(), device=dev))
    grid_limit = args.grid_limit if args.grid_limit >= 0 else (12288 if inflight > 1 else 0)
    for c in ctxs:
        c.set_grid_limit(grid_limit)
    pipe = FramePipeline(fg, lambda buf, slot: ctxs[slot].render_device(frame, buf.data_ptr(), n_parts=world, part=rank), streams) if piped else None

    orbit = bool(kwargs.get("orbit"))
    state = {"cam": scene.camera, "n": 0}

    def step():
        if orbit:  # every rank advances the same camera
            state["cam"] = rt.rotate_cam(state["cam"], 3.14159265358979 / 180.0, 0.0)
            frame.camera = state["cam"]
        if pipe is None:
            slot = state["n"] % inflight  # N = 1 with --inflight k: frames alternate over k contexts (streams)
            state["n"] += 1
            ctxs[slot].render_device(frame, outs[slot].data_ptr())
        else:
            # render this rank's blocks, then (gather stream) RCCL gather of the rows that can hold geometry
            # to GPU0 + scatter into frame order; GPU0 paints the sky rows itself (rt_frame_row_extent)
            pipe.submit(ctx.row_extent(frame), frame.sky_0rgb)

    def sync():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)  # torch's buffer fills are not ordered against the library's streams
    # The GPU's clocks take 50-100 frames of this load (about 15 ms) to settle, and frames run 8 % slower
    # until then: a fixed run-in ahead of the W warm-up steps, so that a short --warmup measures the same
    # steady state as a long one.  Untimed, reported as config.settle_frames; the orbit restarts after it.
    settle = max(0, 100 - args.warmup)
    for _ in range(settle):
        step()
    if orbit:
        state["cam"] = scene.camera
        frame.camera = scene.camera
    for _ in range(args.warmup):
        step()
    sync()
    for c in ctxs:
        c.enable_timing(max(1, args.time_every))
        c.kernel_time_ms()  # drop anything recorded so far
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kt = [c.kernel_time_ms() for c in ctxs]
    kern_n = sum(n for _, n in kt)
    kern_ms = sum(ms * n for ms, n in kt) / kern_n if kern_n else 0.0  # with frames in flight on two streams, launches overlap and each takes longer
    for c in ctxs:
        c.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([kern_ms], dtype=torch.float64, device=cdev)
        km_list = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(km_list, km)
        kern_all = [float(x.item()) for x in km_list]
    else:
        kern_all = [kern_ms]

    # ---- accounting (outside the timed region)
    result = None
    if rank == 0:
        if orbit:
            frame.camera = scene.camera
        if pipe is not None or inflight > 1:
            # the gathered (or last in-flight) frame against one rendered whole on this GPU (same camera: the last step's)
            last = outs[(state["n"] - 1) % inflight]
            if pipe is not None:
                pipe.drain()
                last = fg.frame
            whole = torch.empty((H, W), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)  # nothing of torch's may still be writing when the library's stream renders into it
            ctx.render_device(frame if not orbit else scene.frame(W, H, camera=state["cam"]), whole.data_ptr())
            ctx.sync()
            torch.cuda.synchronize(dev)
            if not torch.equal(whole, last):
                bad = int((whole != last).sum().item())
                print(f"bench.py: the frame gathered from {world} rank(s) differs from the single-GPU frame in {bad} pixels", file=sys.stderr)
                sys.exit(3)
        total = ctx.count_rays(frame)
        mine = ctx.count_rays(frame, n_parts=world, part=0) if world > 1 else total
        rays_uniq, rays_ref = total.rays, total.ref_rays
        ms_per_step = elapsed / args.steps * 1e3
        flops_launch = algorithmic_flops(mine, n_lights)
        achieved_tf = flops_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        result = {
            "metric": "Mrays/s (primary+shadow+secondary, unique rays)",
            "value": rays_uniq * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference default diorama (platform materials pinned by PCG32 0xC0FFEE, reference textures baked to RGB8)"
                    if kind == "default" else "synthetic voxel world, PCG32 0x5EED5EED",
            "config": {
                "workload": args.workload, **({"rehearsal": f"backend {backend}, every rank on cuda:{device_index}: not a measurement"} if backend != "nccl" else {}), "width": W, "height": H, "n_cubes": scene.n_cubes, "n_spheres": scene.n_spheres,
                "n_lights": n_lights, "max_depth": 3,
                "partition": f"16-row blocks round-robin over {world} GPU(s); rows that can hold geometry gathered to GPU0 (RCCL), {inflight} frames in flight per GPU" if world > 1
                             else ("whole frame on one GPU" + (f", {inflight} frames in flight" if inflight > 1 else "")),
                "rays_per_frame_unique": rays_uniq, "rays_per_frame_reference": rays_ref,
                "mrays_per_s_reference_accounting": rays_ref * args.steps / elapsed / 1e6,
                "cube_tests_per_frame": int(total.cube_tests), "sphere_tests_per_frame": int(total.sphere_tests),
                "kernel_ms_per_rank": kern_all, "frames_in_flight": inflight,
            },
            "roofline": {
                "bound": "valu_fp32", "achieved": achieved_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP32_TFLOPS, "traffic": traffic_from_profile(args.workload),
                "kernel": "render_kernel<false,false>", "kernel_avg_ms": kern_ms, "launches_timed": kern_n, "timed_every": max(1, args.time_every), "launches_overlap": inflight > 1,
                "algorithmic_flops_per_launch": flops_launch,
                "frac_of_nonfma_lane_op_ceiling": achieved_tf / PEAK_LANE_OPS_T,
                # executed view from the committed rocprofv3 PMC pass of this command (profiles/traffic.json)
                "valu_busy_frac_pmc": profile_field(args.workload, "valu_busy_frac"),
                "valu_insts_per_launch_pmc": profile_field(args.workload, "valu_insts_per_launch"),
                "note": "algorithmic FP32 ops of the reference's brute-force walk: 26/ray-cube test + 25/ray-sphere test + 60*lights/shaded hit (SURVEY 8d), rank 0's launch; the kernel proves most tests fail without executing them, so frac can exceed 1 (executed view: valu_busy_frac_pmc)",
            },
        }
        result["config"]["settle_frames"] = settle
        if pipe is not None or inflight > 1:
            result["config"]["grid_limit"] = grid_limit if grid_limit > 0 else 32768
            result["config"]["frame_check"] = "gathered frame identical to the frame rendered whole on GPU0 (checked after the timed region)"
        if world == 1 and args.cpu_seconds > 0:
            cb = cpu_baseline(scene, W, H, args.cpu_seconds, args.cpu_threads)
            result["cpu_baseline"] = cb
            result["config"]["gpu_over_cpu_reference_accounting"] = (rays_ref * args.steps / elapsed / 1e6) / cb["value"]
        if world == 1 and inflight > 1:
            # The same workload strictly one frame at a time (one context, one stream), measured apart: how long a
            # frame takes from launch to completion, and the render kernel's duration alone on the GPU.
            if orbit:
                frame.camera = scene.camera
            ctx.set_grid_limit(0)  # the library's default, which is the best for one frame at a time
            for _ in range(3):
                ctx.render_device(frame, out.data_ptr())
            ctx.sync()
            ctx.enable_timing(max(1, args.time_every))
            ctx.kernel_time_ms()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ctx.render_device(frame, out.data_ptr())
            ctx.sync()
            dt = (time.perf_counter() - t1) / args.steps
            solo_ms, _ = ctx.kernel_time_ms()
            ctx.enable_timing(False)
            ctx.set_grid_limit(grid_limit)
            result["config"]["one_frame_in_flight"] = {"ms_per_step": dt * 1e3, "value": rays_uniq / dt / 1e6, "unit": "Mrays/s",
                                                       "kernel_avg_ms": solo_ms,
                                                       "roofline_frac": (flops_launch / (solo_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if solo_ms > 0 else None,
                                                       "note": "--inflight 1 makes this the timed mode"}
        if world == 1 and args.extra:
            extra = {}
            for name in ("minecraft_1080p", "minecraft_water_4k"):
                k2, kw2, w2, h2 = WORKLOADS[name]
                sc2 = make_scene(rt, k2, kw2)
                for c in ctxs:
                    c.upload(sc2)
                f2 = sc2.frame(w2, h2)
                o2 = [torch.zeros((h2, w2), dtype=torch.int32, device=dev) for _ in ctxs]
                torch.cuda.synchronize(dev)  # torch's fills are not ordered against the library's streams
                for i in range(2 * len(ctxs)):
                    ctxs[i % len(ctxs)].render_device(f2, o2[i % len(ctxs)].data_ptr())
                for c in ctxs:
                    c.sync()
                t1 = time.perf_counter()
                for i in range(10 * len(ctxs)):
                    ctxs[i % len(ctxs)].render_device(f2, o2[i % len(ctxs)].data_ptr())
                for c in ctxs:
                    c.sync()
                dt = (time.perf_counter() - t1) / (10 * len(ctxs))
                c2 = ctx.count_rays(f2)
                extra[name] = {"ms_per_frame": dt * 1e3, "mrays_per_s": c2.rays / dt / 1e6, "rays_per_frame_unique": c2.rays, "frames_in_flight": len(ctxs)}
            result["config"]["extra"] = extra
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
