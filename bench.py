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
SURVEY.md section 8(d)) x K / max-over-ranks wall time.  roofline: the FP32 operations rank 0's launch
EXECUTES (events counted by the kernel's counting variant, rt_count_executed, priced per event: EVENT_FLOPS
below and DESIGN.md) / the launch's HIP-event time, against the 157.3 TFLOP/s FP32 vector peak; the
reference's brute-force operation count over the same time is reported beside it as algorithmic_speedup
(the cull proves most of those tests away, so that ratio is not an efficiency).  cpu_baseline: the oracle
(reference algorithm, per-light re-casting, the host cores this job may use) on a bounded row sample of the
same frame, N=1 only.
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


# FP32 operations per executed event (rt_exec_counters), counted from the source of
# raytracing_amd/csrc/rtx950_kernels.hip with an FMA as 2, a correctly rounded division or square root as 1
# (they take ~10 instructions each) and compares/min/max/abs as 1; integer, control and data movement are
# not counted.  Derivation per event in DESIGN.md, "Roofline".
EVENT_FLOPS = {
    "cull_tests": 33,        # cull_test: 3 axes x (2 FMA + add + 2 FMA + |g| > l) + 3 abs
    "chunk_tests": 39,       # chunk_test: 3 axes x (2 FMA + add + 3 FMA + compare) + 3 abs
    "candidate_evals": 31,   # cube_slab_approx (9 FMA, max3, min3) + gap, 2 decisions, distance select, fold
    "exact_evals": 26,       # cube_exact = Cube::ray_intersect's slab test
    "bvh_steps": 52,         # two box_slab (9 FMA + max3 + min3 + gap + compare each) + ordering compares
    "sphere_evals": 25,      # sphere_exact
    "queries": 37,           # ray_pre (3 rcp, 3 mul, error bound) + cull_pre (moments, radii)
    "light_iterations": 60,  # Phong fold per light
    "resolved_hits": 64,     # cube_distance (3 refined reciprocals, 3 shared divisions), hit point, face tests, uv
    "blockers": 65,          # the blocker's cube_distance + distance2 + shadow term
    "child_setups": 60,      # reflect + normalize + refract + two origins
    "primary_rays": 40,      # ray generation: two normalisations and the change of basis
}


def executed_flops(ex):
    d = ex.as_dict()
    return float(sum(EVENT_FLOPS[k] * d[k] for k in EVENT_FLOPS))


def profile_entry(workload, kernel):
    """The committed rocprofv3 PMC passes of this command (profiles/traffic.json, written by
    tools/summarize_pmc.py), keyed by workload and kernel instantiation; {} when none is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}
    e = t.get(workload, {})
    if "kernels" in e:
        return e["kernels"].get(kernel, {})
    return {}


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
    allowed = len(os.sched_getaffinity(0))
    # 0 = every core this job may use: the affinity mask, capped by the box's per-job share when it states
    # one (the GPU box exports OMP_NUM_THREADS = its CPU share per GPU while nproc shows the whole machine)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max_threads if max_threads > 0 else (min(allowed, share) if share > 0 else allowed)
    threads = max(1, min(threads, allowed))
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
        "host_logical_cores": os.cpu_count(),
        "cores_allowed": allowed,
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
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline; 0 = all cores this job may use (affinity mask, capped by OMP_NUM_THREADS "
                         "when the box sets it as the job's CPU share)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per GPU (one library context and stream each: the next frame starts under the current "
                         "frame's slowest tiles); 0 = the default: 2 at N = 1, 3 per rank at N > 1; 1 = strictly one frame at a time")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket every k-th launch of the render kernel with HIP events (roofline.kernel_avg_ms); the two event "
                         "records cost the stream about 6 us per launch, so timing every launch slows the frame it measures by 4 %%")
    ap.add_argument("--grid-limit", type=int, default=-1,
                    help="rt_set_grid_limit of every context; -1 = 12288 with frames in flight (two render kernels share the GPU), "
                         "the library's default (32768) with one")
    ap.add_argument("--heavy-class", type=int, default=-1,
                    help="rt_set_tuning(RT_TUNE_HEAVY_CLASS) of every context (tiles of this cost class and above go to the side kernel: "
                         "three-wave workgroups on a second stream; 8 = none); -1 = the library's default (automatic: off for these "
                         "asynchronous launches)")
    ap.add_argument("--wavefront", type=int, default=-1,
                    help="rt_set_tuning(RT_TUNE_WAVEFRONT) of every context: 1 = one launch per ray-tree level over bundles of coherent "
                         "rays instead of the tile-resident walk, 0 = never, -1 = the library's default")
    ap.add_argument("--refine", type=int, default=-1, help="rt_set_tuning(RT_TUNE_CULL_REFINE) of every context (-1: the library's choice per launch)")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs (tools/profile_workloads.sh): only run-in, warm-up, the timed region and the counting launch; "
                         "no CPU baseline and none of the side measurements, so that a rocprofv3 trace of the command holds the timed "
                         "launches and nothing else of the render kernel")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip config.extra: the metric's other sizes and scenes (1080p, 4K with water, the 10k-cube world, the 8K orbit), "
                         "measured after the timed region at N = 1 (about 20 s)")
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
        c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, args.heavy_class)
        c.set_tuning(rt.RT_TUNE_WAVEFRONT, args.wavefront)
        c.set_tuning(rt.RT_TUNE_CULL_REFINE, args.refine)
    pipe = FramePipeline(fg, lambda buf, slot: ctxs[slot].render_device(frame, buf.data_ptr(), n_parts=world, part=rank), streams,
                         time_every=max(1, args.time_every)) if piped else None

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
    if pipe is not None:
        pipe.gather_stats()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issued = time.perf_counter()  # everything is enqueued; the host has not waited yet (unless a queue filled up)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kt = [c.kernel_time_ms() for c in ctxs]
    timed_kernel = ctx.kernel_name()  # of the timed launches (the frame check below renders a whole frame, which can take another variant)
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
    gather_ms, gather_n, sent_bytes = (pipe.gather_stats() if pipe is not None else (0.0, 0, 0))
    if world > 1:
        g = torch.tensor([gather_ms, float(sent_bytes)], dtype=torch.float64, device=cdev)
        g_list = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(g_list, g)
        gather_all = [float(x[0].item()) for x in g_list]
        sent_all = [int(x[1].item()) for x in g_list]
    else:
        gather_all, sent_all = [gather_ms], [sent_bytes]

    ok = True
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
                ok = False
    if world > 1:
        # every rank learns the verdict before any of them leaves (a lone exit would strand the others in the barrier)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if not ok:
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)

    if rank == 0:
        total, total_ex = ctx.count_executed(frame)
        mine, mine_ex = ctx.count_executed(frame, n_parts=world, part=0) if world > 1 else (total, total_ex)
        rays_uniq, rays_ref = total.rays, total.ref_rays
        ms_per_step = elapsed / args.steps * 1e3
        kernel = timed_kernel
        flops_ref = algorithmic_flops(mine, n_lights)     # what the reference's brute force would do for this launch
        flops_exec = executed_flops(mine_ex)              # what the kernel does
        achieved_tf = flops_exec / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        prof = profile_entry(args.workload, kernel)
        result = {
            "metric": "Mrays/s (primary+shadow+secondary; unique rays ANSWERED per second: a ray = one nearest-hit or shadow query of the "
                      "reference's ray tree, SURVEY 8d, whether the kernel traverses the object list for it or proves the answer; "
                      "config.mrays_per_s_traced counts only the queries the kernel executes)",
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
                # queries the kernel EXECUTES per frame, per lane (nearest hit, exact re-run, shadow): the sphere pass over an empty
                # list and the tiles classified as sky are answered without one
                "traced_rays_per_frame": int(total_ex.queries), "mrays_per_s_traced": total_ex.queries * args.steps / elapsed / 1e6,
                "cube_tests_per_frame": int(total.cube_tests), "sphere_tests_per_frame": int(total.sphere_tests),
                "kernel_ms_per_rank": kern_all, "frames_in_flight": inflight,
                # host time this rank spent issuing a frame (Python + ctypes + the library's three launches; the library's own
                # share is 7.7 us per frame, tools/host_issue.cpp); an upper bound when the host ran ahead into a full queue
                "host_issue_us_per_frame": (t_issued - t0) / args.steps * 1e6,
            },
            "roofline": {
                "bound": "valu_fp32", "achieved": achieved_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP32_TFLOPS,
                "traffic": prof.get("hbm_bytes_per_launch"),
                "kernel": kernel, "kernel_avg_ms": kern_ms, "launches_timed": kern_n, "timed_every": max(1, args.time_every), "launches_overlap": inflight > 1,
                "executed_flops_per_launch": flops_exec,
                "executed_events_per_launch": mine_ex.as_dict(),
                # the same executed operations against the ceiling of one plain (unpacked, non-FMA) op per lane and clock
                "frac_of_nonfma_lane_op_ceiling": achieved_tf / PEAK_LANE_OPS_T,
                # the reference's brute-force walk of the same launch over the same time: not an efficiency (the cull,
                # the slab filter and the tile classification prove most of those tests away), a speed-up of the method
                "algorithmic_flops_per_launch": flops_ref,
                "algorithmic_tflops_equivalent": flops_ref / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                "algorithmic_speedup": flops_ref / flops_exec if flops_exec > 0 else None,
                # executed view from the committed rocprofv3 PMC passes of this command and kernel (profiles/traffic.json)
                "pmc": {k: prof.get(k) for k in ("valu_insts_per_launch", "valu_busy_frac", "active_lanes_per_inst", "lane_ops_per_s", "frac_of_lane_op_ceiling", "source")} if prof else None,
                "note": "achieved = FP32 operations the launch executes (rt_count_executed events x EVENT_FLOPS, an FMA as 2, a division as 1) / its HIP-event duration; "
                        "algorithmic_* = the reference's brute force (26/ray-cube test + 25/ray-sphere test + 60*lights/shaded hit, SURVEY 8d) for comparison",
            },
        }
        result["config"]["settle_frames"] = settle
        if world > 1:
            result["config"]["exchange"] = {
                "gather_ms_per_rank": gather_all, "gathers_timed": gather_n, "bytes_sent_per_frame_per_rank": sent_all,
                "bytes_into_gpu0_per_frame": int(sum(sent_all)),
                "note": "gather_ms: events on the frame's stream around dist.gather (+ rank 0's sky fill and row scatter), i.e. from the end of the rank's render to the end of its exchange",
            }
        if pipe is not None or inflight > 1:
            result["config"]["grid_limit"] = grid_limit if grid_limit > 0 else 32768
            result["config"]["frame_check"] = "gathered frame identical to the frame rendered whole on GPU0 (checked after the timed region)"
        if world == 1 and args.cpu_seconds > 0 and not args.lean:
            cb = cpu_baseline(scene, W, H, args.cpu_seconds, args.cpu_threads)
            result["cpu_baseline"] = cb
            result["config"]["gpu_over_cpu_reference_accounting"] = (rays_ref * args.steps / elapsed / 1e6) / cb["value"]
        if world == 1 and not args.lean:
            # ---- latency side of the same workload, measured apart (one context, one stream)
            if orbit:
                frame.camera = scene.camera
            ctx.set_grid_limit(0)  # the library's default, which is the best for one frame at a time
            # ... with and without the side kernel for the heaviest tiles (RT_TUNE_HEAVY_CLASS: three-wave workgroups on
            # a second stream; automatic only in the synchronous rt_render, so both settings are measured here)
            def one_at_a_time(heavy_class):
                ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, heavy_class)
                for _ in range(8):
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
                # a frame from launch to completion with the host waiting for it (what an interactive host sees)
                lat = []
                for _ in range(50):
                    t1 = time.perf_counter()
                    ctx.render_device(frame, out.data_ptr())
                    ctx.sync()
                    lat.append(time.perf_counter() - t1)
                lat.sort()
                return dt, solo_ms, lat[len(lat) // 2]
            dt, solo_ms, lat_off = one_at_a_time(8)
            dt_on, solo_on_ms, lat_on = one_at_a_time(5)
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, args.heavy_class)
            if ctx.debug_status() != 0:
                print("bench.py: a side-kernel wait gave up (rt_debug_status)", file=sys.stderr)
                sys.exit(3)
            result["config"]["one_frame_in_flight"] = {"ms_per_step": dt * 1e3, "value": rays_uniq / dt / 1e6, "unit": "Mrays/s",
                                                       "side_kernel": "off (the library's default behind the asynchronous entry points)",
                                                       "ms_per_step_side_kernel_off": dt * 1e3, "ms_per_step_side_kernel_on": dt_on * 1e3,
                                                       "kernel_avg_ms": solo_ms, "kernel_avg_ms_side_kernel_on": solo_on_ms,
                                                       "roofline_frac": (flops_exec / (solo_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if solo_ms > 0 else None,
                                                       "note": "--inflight 1 makes this the timed mode; side kernel = rt_set_tuning(RT_TUNE_HEAVY_CLASS, 5): the tiles that were "
                                                               "slowest in the previous launch on three-wave workgroups on a second stream (kernel_avg_ms is render_kernel's alone)"}
            result["single_frame_latency_ms"] = lat_off * 1e3  # the library's default for rt_render_device
            result["config"]["single_frame_latency"] = {"side_kernel_off_ms": lat_off * 1e3, "side_kernel_on_ms": lat_on * 1e3,
                                                        "note": "median of 50 frames, launch to completion with the host waiting"}
            # the first frame after a camera jump: the queue order comes from the previous launch's tile costs, so
            # a new view starts with a stale order (and a new set of traced tiles)
            jumps, steady = [], []
            cam_a = scene.camera
            for k in range(12):
                cam_b = rt.rotate_cam(cam_a, 0.9 + 0.37 * k, 0.11 * ((k % 3) - 1))
                f2 = scene.frame(W, H, camera=cam_b)
                t1 = time.perf_counter()
                ctx.render_device(f2, out.data_ptr())
                ctx.sync()
                jumps.append(time.perf_counter() - t1)
                for _ in range(3):
                    ctx.render_device(f2, out.data_ptr())
                ctx.sync()
                t1 = time.perf_counter()
                ctx.render_device(f2, out.data_ptr())
                ctx.sync()
                steady.append(time.perf_counter() - t1)
                cam_a = cam_b
            result["config"]["camera_jump"] = {"first_frame_ms": sum(jumps) / len(jumps) * 1e3, "same_view_steady_ms": sum(steady) / len(steady) * 1e3,
                                               "note": "12 jumps to unrelated views, one frame at a time, host waits per frame"}
            # rt_render, the reference's synchronous render(): kernels + the single read-back over PCIe
            def timed_rt_render(target):
                ctx.render(frame, out=target)
                t1 = time.perf_counter()
                for _ in range(10):
                    ctx.render(frame, out=target)
                return (time.perf_counter() - t1) / 10 * 1e3
            import numpy as np
            pageable = np.empty((H, W), dtype=np.uint32)
            pinned = rt.PinnedFrame(H, W)
            result["config"]["rt_render_ms"] = {"pageable_host_memory": timed_rt_render(pageable), "pinned_host_memory (rt_host_alloc)": timed_rt_render(pinned.array),
                                                "bytes_read_back": H * W * 4, "note": "synchronous like the reference's render(): launches + D2H copy + wait; never `value`"}
            pinned.close()
            ctx.set_grid_limit(grid_limit)
        if world == 1 and not args.no_extra and not args.lean:
            # BASELINE.json's metric is quoted "at 1080p/4K" and its configs name four more frames: the same measurement
            # (frames in flight as above, HIP-event kernel time, executed-work fraction) on each, outside the timed region
            extra = {}
            for name in ("minecraft_1080p", "minecraft_water_4k", "synthetic_4k", "minecraft_8k_orbit"):
                k2, kw2, w2, h2 = WORKLOADS[name]
                sc2 = make_scene(rt, k2, kw2)
                for c in ctxs:
                    c.upload(sc2)
                    c.set_grid_limit(grid_limit)
                f2 = sc2.frame(w2, h2)
                o2 = [torch.zeros((h2, w2), dtype=torch.int32, device=dev) for _ in ctxs]
                torch.cuda.synchronize(dev)  # torch's fills are not ordered against the library's streams
                orbit2 = bool(kw2.get("orbit"))
                cam2 = [sc2.camera]

                def frames2(n):
                    for i in range(n):
                        if orbit2:
                            cam2[0] = rt.rotate_cam(cam2[0], 3.14159265358979 / 180.0, 0.0)
                            f2.camera = cam2[0]
                        ctxs[i % len(ctxs)].render_device(f2, o2[i % len(ctxs)].data_ptr())
                    for c in ctxs:
                        c.sync()
                n_warm, n_timed = (20, 100) if name == "synthetic_4k" else (30, 120)  # about 0.1 s each
                frames2(n_warm)
                for c in ctxs:
                    c.enable_timing(4)
                    c.kernel_time_ms()
                cam2[0] = sc2.camera
                t1 = time.perf_counter()
                frames2(n_timed)
                dt = (time.perf_counter() - t1) / n_timed
                kt2 = [c.kernel_time_ms() for c in ctxs]
                for c in ctxs:
                    c.enable_timing(False)
                kn2 = sum(n for _, n in kt2)
                kms2 = sum(ms * n for ms, n in kt2) / kn2 if kn2 else 0.0
                f2.camera = sc2.camera
                c2, e2 = ctx.count_executed(f2)
                fl2 = executed_flops(e2)
                extra[name] = {"width": w2, "height": h2, "n_cubes": sc2.n_cubes, "n_spheres": sc2.n_spheres,
                               "ms_per_frame": dt * 1e3, "mrays_per_s": c2.rays / dt / 1e6, "mrays_per_s_traced": e2.queries / dt / 1e6,
                               "rays_per_frame_unique": c2.rays, "frames_in_flight": len(ctxs), "frames_timed": n_timed,
                               "kernel": ctx.kernel_name(), "kernel_avg_ms": kms2,
                               "executed_flops_per_launch": fl2,
                               "roofline_frac": (fl2 / (kms2 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if kms2 > 0 else None}
                if orbit2:
                    extra[name]["note"] = "one degree of yaw per frame (Camera::rotate_cam); rays and executed work counted on the first view"
            result["config"]["extra"] = extra
            result["config"]["extra_note"] = ("the metric's other frames, measured like the headline (frames in flight, HIP-event kernel time every 4th launch) after "
                                              "the timed region; executed work is counted by the kernel's counting variant, which has no cull refinement")
            for c in ctxs:
                c.upload(scene)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
