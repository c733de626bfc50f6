"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): hit indices bit-exact; 8-bit colour within 1 LSB per channel on
>= 99.9% of pixels.  The kernels claim more - colours identical to the oracle's in every bit - and
assert_parity holds them to that (exact=True is its default).  Every comparison renders the frame twice:
with the debug variant of the kernel (hit planes, ray-tree hashes) and with the PLAIN variant that
rt_render / bench.py ship, and requires the two frames to be equal.
The oracle is "parity unpinned" except for Color*f32 (see oracle/rt_oracle.c).
"""
import os

import numpy as np
import pytest

import raytracing_amd as rt
from parity import assert_parity, colour_report, random_scene

pytestmark = pytest.mark.gpu


def _both(ctx, orc, scene, w, h, **kw):
    ctx.upload(scene)
    f = scene.frame(w, h, **kw)
    gfb, gpl = ctx.render(f, debug=True)
    plain = ctx.render(f)  # the kernel variant that ships (no debug planes): must paint the same frame
    assert np.array_equal(plain, gfb), f"plain and debug kernels differ on {int((plain != gfb).sum())} pixels ({scene.name} {w}x{h})"
    O = orc.OracleScene(scene)
    ofb, opl, ocnt, _ = O.render(f, debug=True, reuse_secondary=True)
    return f, gfb, gpl, ofb, opl, ocnt


@pytest.fixture
def tuned_ctx(gpu_ctx):
    """The session's context with rt_set_tuning restored to the defaults afterwards."""
    yield gpu_ctx
    gpu_ctx.reset_tuning()


@pytest.mark.parametrize("w,h", [(270, 180), (1080, 720)])
def test_default_scene_matches_oracle(gpu_ctx, oracle_mod, default_scene, w, h):
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, default_scene, w, h)
    rep = assert_parity(gfb, gpl, ofb, opl, f"default {w}x{h}")
    print("colour parity", rep)
    # counters of the unique-ray accounting
    c = gpu_ctx.count_rays(f)
    assert c.uniq_dict() == ocnt.uniq_dict()


def test_water_variant_matches_oracle(gpu_ctx, oracle_mod, water_scene):
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, water_scene, 480, 270)
    rep = assert_parity(gfb, gpl, ofb, opl, "minecraft+water 480x270")
    print("colour parity", rep)
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_debug_and_plain_kernels_agree(gpu_ctx, default_scene):
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(320, 200)
    a = gpu_ctx.render(f)
    b, _ = gpu_ctx.render(f, debug=True)
    assert np.array_equal(a, b)


def test_unpainted_pixel_quirk(gpu_ctx, default_scene):
    # reference src/render.rs:316-322: pixel i == width (x=0,y=1) keeps the clear colour 0x000000
    gpu_ctx.upload(default_scene)
    fb = gpu_ctx.render(default_scene.frame(64, 48))
    assert fb[1, 0] == 0
    assert fb[0, 0] == default_scene.sky_0rgb and fb[1, 1] == default_scene.sky_0rgb


def test_bands_equal_full_frame(gpu_ctx, default_scene):
    gpu_ctx.upload(default_scene)
    w, h = 200, 150
    full = gpu_ctx.render(default_scene.frame(w, h))
    parts = [gpu_ctx.render(default_scene.frame(w, h, y0=y0, y1=y1)) for y0, y1 in [(0, 1), (1, 37), (37, 38), (38, 150)]]
    assert np.array_equal(np.concatenate(parts, axis=0), full)


def test_synthetic_small_matches_oracle(gpu_ctx, oracle_mod):
    scene = rt.synthetic_scene(n_cubes=600, n_spheres=24)
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, scene, 320, 180)
    rep = assert_parity(gfb, gpl, ofb, opl, "synthetic 600c/24s 320x180")
    print("colour parity", rep)
    assert (gpl["hit_sphere"] != rt.RT_NO_HIT).any(), "test scene should show spheres"
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_night_and_orbit_match_oracle(gpu_ctx, oracle_mod, default_scene):
    import ctypes as C
    # TimeToNight (reference src/main.rs:299-314) + a camera orbit step (src/camera.rs:61-86)
    s = default_scene
    lights = (rt.Light * 2)(s.lights[0], rt.Light(rt.Vec3(0.0, 20.0, 0.0), 0x000000, 0.5))
    cam = rt.rotate_cam(s.camera, 0.7, -0.3)
    night = rt.Scene(s.cubes, s.spheres, s.materials, lights, cam, s.ambient_light, rt.NIGHT_COLOR, s.atlas, s.tex_w,
                     s.tex_h, s.sprite, "night")
    f, gfb, gpl, ofb, opl, _ = _both(gpu_ctx, oracle_mod, night, 300, 200)
    assert_parity(gfb, gpl, ofb, opl, "night orbit")


def test_golden_fixtures(gpu_ctx):
    """GPU vs the committed oracle outputs (tests/golden/*.npz, made by tests/make_golden.py)."""
    import os
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    cases = [("minecraft_270x180.npz", rt.default_scene(), 270, 180),
             ("minecraft_water_240x135.npz", rt.default_scene(with_water=True), 240, 135),
             ("synthetic_600c_24s_160x90.npz", rt.synthetic_scene(n_cubes=600, n_spheres=24), 160, 90)]
    for name, scene, w, h in cases:
        g = np.load(os.path.join(gdir, name))
        gpu_ctx.upload(scene)
        f = scene.frame(w, h)
        fb, planes = gpu_ctx.render(f, debug=True)
        assert_parity(fb, planes, g["fb"], {k: g[k] for k in ("hit_cube", "hit_sphere", "tree_hash")}, name)
        assert np.array_equal(gpu_ctx.render(f), g["fb"]), name  # plain kernel against the committed frame
        c = gpu_ctx.count_rays(f)
        assert list(c.uniq_dict().values()) == g["uniq"].tolist()
        assert list(c.ref_dict().values()) == g["ref"].tolist()  # n_lights^depth weighting == the reference's re-casting


def test_interleaved_parts_reassemble(gpu_ctx, default_scene):
    """rt_render_device_part for every part of an N-way split == the whole frame (N-GPU path on one GPU)."""
    import torch
    from raytracing_amd import multigpu
    gpu_ctx.upload(default_scene)
    w, h = 200, 150
    f = default_scene.frame(w, h)
    full = gpu_ctx.render(f)
    for n in (2, 3, 8):
        frame = np.zeros((h, w), dtype=np.uint32)
        total = rt.Counters()
        for p in range(n):
            rows = multigpu.part_rows(h, n, p)
            buf = torch.zeros((multigpu.max_part_rows(h, n), w), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # torch's fill is not ordered against the library's stream
            gpu_ctx.render_device(f, buf.data_ptr(), n_parts=n, part=p)
            gpu_ctx.sync()
            frame[multigpu.part_row_indices(h, n, p)] = buf[:rows].cpu().numpy().view(np.uint32)
        assert np.array_equal(frame, full)
        parts = [gpu_ctx.count_rays(f, n_parts=n, part=p) for p in range(n)]
        assert sum(c.rays for c in parts) == gpu_ctx.count_rays(f).rays


_EXTENT_CAMERAS = [
    ("default", None),
    ("looking away", ((0.0, 0.0, 10.0), (0.0, 0.0, 20.0))),
    ("eye inside the diorama", ((0.2, 1.0, 0.3), (3.0, 0.0, -3.0))),
    ("far away", ((0.0, 5.0, 200.0), (0.0, 0.0, 0.0))),
    ("scene below the frame", ((0.0, 0.0, 40.0), (0.0, 120.0, 0.0))),
    ("from above", ((0.0, 30.0, 0.01), (0.0, 0.0, 0.0))),
    ("NaN basis", ((0.0, 0.0, 10.0), (0.0, 0.0, 10.0))),
]


@pytest.mark.parametrize("name,cam", _EXTENT_CAMERAS)
def test_row_extent_bounds_the_geometry(gpu_ctx, default_scene, name, cam):
    """rt_frame_row_extent: rows outside the extent are sky in the rendered frame (the unpainted pixel
    aside), for whole frames and bands; the bound is tight enough to be useful for the default view."""
    gpu_ctx.upload(default_scene)
    w, h = 320, 200
    for y0, y1 in ((0, h), (37, 150), (120, 121)):
        f = default_scene.frame(w, h, y0=y0, y1=y1, camera=_cam(*cam) if cam else None)
        a, b = gpu_ctx.row_extent(f)
        assert y0 <= a <= b <= y1
        fb = gpu_ctx.render(f)
        outside = np.concatenate([fb[: a - y0], fb[b - y0:]]).copy().reshape(-1, w)
        expect = np.full_like(outside, f.sky_0rgb)
        rows_out = list(range(y0, a)) + list(range(b, y1))
        if 1 in rows_out:
            expect[rows_out.index(1), 0] = 0
        assert np.array_equal(outside, expect), (name, y0, y1, a, b)
        if name == "default" and (y0, y1) == (0, h):
            hit_rows = np.where((fb != f.sky_0rgb).any(axis=1))[0]
            assert a >= hit_rows.min() - 8 and b <= hit_rows.max() + 8 and (b - a) < 0.6 * h
        if name in ("looking away", "eye inside the diorama", "NaN basis"):
            assert (a, b) == (y0, y1)  # corners behind the eye / no basis: no conclusion
        if name == "scene below the frame":
            assert a == b


def test_row_extent_with_spheres_is_the_whole_band(gpu_ctx):
    s = rt.synthetic_scene(n_cubes=50, n_spheres=3)
    gpu_ctx.upload(s)
    assert gpu_ctx.row_extent(s.frame(64, 48, y0=5, y1=40)) == (5, 40)


def test_frame_pipeline_on_one_gpu(gpu_ctx, default_scene):
    """FramePipeline (double-buffered messages, gather stream, extent-only scatter) over an orbiting camera
    == direct renders; world 1, so the collective is absent but buffers, events and fills are the real ones."""
    import torch
    from raytracing_amd import multigpu
    gpu_ctx.upload(default_scene)
    w, h = 328, 200
    dev = torch.device("cuda", 0)
    ext = torch.cuda.ExternalStream(gpu_ctx.stream(), device=dev)
    with torch.cuda.stream(ext):
        fg = multigpu.FrameGather(h, w, 1, 0, dev, depth=2)
    f = default_scene.frame(w, h)
    # one context (and stream) per message buffer, as bench.py does for N > 1
    ctx2 = rt.Context(0)
    ctx2.upload(default_scene)
    ctxs = [gpu_ctx, ctx2]
    streams = [ext, torch.cuda.ExternalStream(ctx2.stream(), device=dev)]
    torch.cuda.synchronize()  # the buffers' zero fills ran on the first context's stream only
    pipe = multigpu.FramePipeline(fg, lambda buf, slot: ctxs[slot].render_device(f, buf.data_ptr()), streams)
    cams, got = [], []
    cam = default_scene.camera
    for i in range(7):
        cam = rt.rotate_cam(cam, 0.3, 0.17 if i < 4 else -0.4)
        f.camera = cam
        cams.append(cam)
        pipe.submit(gpu_ctx.row_extent(f), f.sky_0rgb)
        if i in (2, 6):  # reading the frame needs a drain; in between, frames overlap
            pipe.drain()
            got.append((i, fg.frame.cpu().numpy().view(np.uint32).copy()))
    for i, img in got:
        f.camera = cams[i]
        assert np.array_equal(img, gpu_ctx.render(f)), i
    ctx2.close()


def test_host_mirror_render_path(default_scene, gpu_ctx):
    """raytracer::render(framebuffer, model) of the C++ host mirror == direct C-ABI calls."""
    import ctypes as C
    h = rt._ffi.host()
    w, hh = 160, 100
    out = np.zeros((hh, w), dtype=np.uint32)
    assert h.rth_render(default_scene._model, w, hh, out.ctypes.data_as(C.c_void_p), None, None, None) == 0, h.rth_last_error()
    gpu_ctx.upload(default_scene)
    assert np.array_equal(out, gpu_ctx.render(default_scene.frame(w, hh)))


def test_full_size_properties(gpu_ctx, default_scene):
    """Size-independent checks at BASELINE sizes (no oracle run): bands of the 4K frame equal the full frame,
    the sky rows are sky, the unpainted pixel is black, counters are consistent."""
    gpu_ctx.upload(default_scene)
    W, H = 3840, 2160
    full = gpu_ctx.render(default_scene.frame(W, H))
    assert full[1, 0] == 0 and (full[:400] == default_scene.sky_0rgb).sum() == 400 * W - 1
    band = gpu_ctx.render(default_scene.frame(W, H, y0=1000, y1=1100))
    assert np.array_equal(band, full[1000:1100])
    c = gpu_ctx.count_rays(default_scene.frame(W, H))
    assert c.nearest_rays >= 2 * W * H and c.cube_tests >= W * H * default_scene.n_cubes
    assert c.ref_rays >= c.rays and c.shadow_rays == 2 * c.shaded_hits


def test_config2_1080p_against_the_oracle(gpu_ctx, oracle_mod, default_scene):
    """BASELINE.json config 2 at its exact size: the diorama at 1920x1080, depth 3 - debug and plain kernels (the
    plain one is the variant with cull refinement and bundles at this size) against the oracle run here on every pixel: hit planes,
    ray-tree hashes and colours bit-exact, counters equal (about half a second of host threads)."""
    import torch
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, default_scene, 1920, 1080)
    rep = assert_parity(gfb, gpl, ofb, opl, "config 2: 1920x1080")
    assert rep["exact_fraction"] == 1.0, rep
    # _both's plain frame came from the synchronous rt_render (side kernel for the slowest tiles, hence no cull
    # refinement); the asynchronous entry point takes the variant with refinement and bundles: the same frame, twice in a row
    assert gpu_ctx.kernel_name() == "render_kernel<false,false,0,false>"
    out = torch.zeros((1080, 1920), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        gpu_ctx.render_device(f, out.data_ptr())
        gpu_ctx.sync()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), gfb)
    assert gpu_ctx.kernel_name() == "render_kernel<false,false,3,false>"  # at most 65536 tiles: refinement and bundles
    c = gpu_ctx.count_rays(f)
    assert c.uniq_dict() == ocnt.uniq_dict()
    assert gfb[1, 0] == 0


@pytest.mark.parametrize("water", [False, True])
def test_config3_4k_against_the_oracle(gpu_ctx, oracle_mod, default_scene, water_scene, water):
    """BASELINE.json configs 2/3 at 3840x2160, with and without the water platform, against the oracle run here on all
    8.3 M pixels (a few seconds of host threads in once-per-hit mode): hit planes, ray-tree hashes and colours
    bit-exact, counters equal; the plain frame (the headline's kernel, render_kernel<..,2,..>, through the
    asynchronous entry point) equals the debug frame."""
    import torch
    sc = water_scene if water else default_scene
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, 3840, 2160)
    rep = assert_parity(gfb, gpl, ofb, opl, f"config 3: 3840x2160 water={water}")
    assert rep["exact_fraction"] == 1.0, rep
    out = torch.zeros((2160, 3840), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_device(f, out.data_ptr())
    gpu_ctx.sync()
    assert gpu_ctx.kernel_name() == f"render_kernel<false,false,2,{'true' if water else 'false'}>"
    assert np.array_equal(out.cpu().numpy().view(np.uint32), gfb)
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_config5_8k_against_the_oracle(gpu_ctx, oracle_mod, default_scene):
    """BASELINE.json config 5's frame size, 7680x4320 (the orbit's first view), against the oracle on all 33 M pixels
    (several seconds of host threads): hit planes, ray-tree hashes and colours bit-exact, counters equal."""
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, default_scene, 7680, 4320)
    rep = assert_parity(gfb, gpl, ofb, opl, "config 5: 7680x4320")
    assert rep["exact_fraction"] == 1.0, rep
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_config3_water_4k_properties(gpu_ctx, water_scene):
    """BASELINE.json config 3 at its exact size (3840x2160, the water platform: reflection + refraction to depth 3):
    plain == debug on every pixel, bands == the full frame, the parts of a 3-way split reassemble, the unpainted
    pixel, counters consistent with the tree shape (every shaded hit casts one shadow ray per light; water pixels
    have two children)."""
    import torch
    gpu_ctx.upload(water_scene)
    W, H = 3840, 2160
    f = water_scene.frame(W, H)
    bufs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(2)]
    hsh = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_device(f, bufs[0].data_ptr())
    gpu_ctx.render_device(f, bufs[1].data_ptr(), d_tree_hash=hsh.data_ptr())
    gpu_ctx.sync()
    assert torch.equal(bufs[0], bufs[1])
    assert gpu_ctx.kernel_name() == "render_kernel<false,false,2,true>"  # no side kernel behind rt_render_device: cull refinement
    full = bufs[0].cpu().numpy().view(np.uint32)
    assert full[1, 0] == 0 and full[0, 0] == water_scene.sky_0rgb
    for y0, y1 in ((0, 97), (1000, 1100), (2100, 2160)):
        assert np.array_equal(gpu_ctx.render(water_scene.frame(W, H, y0=y0, y1=y1)), full[y0:y1]), (y0, y1)
    part = torch.zeros((rt.part_rows(H, 3, 0), W), dtype=torch.int32, device="cuda")
    for p in range(3):
        torch.cuda.synchronize()
        gpu_ctx.render_device(f, part.data_ptr(), n_parts=3, part=p)
        gpu_ctx.sync()
        rows = rt.part_rows(H, 3, p)
        assert np.array_equal(part[:rows].cpu().numpy().view(np.uint32), full[rt.part_row_indices(H, 3, p)]), p
    c = gpu_ctx.count_rays(f)
    d = gpu_ctx.count_rays(gpu_ctx.scene.frame(W, H, max_depth=0))
    assert c.shadow_rays == 2 * c.shaded_hits and c.ref_rays > c.rays > d.rays
    assert c.nearest_rays > d.nearest_rays == 2 * W * H  # depth 0: one primary ray per pass and pixel
    # the wavefront launches paint the same frame at this size (deep trees: every level has items)
    gpu_ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
    try:
        torch.cuda.synchronize()
        gpu_ctx.render_device(f, bufs[1].data_ptr())
        gpu_ctx.sync()
        assert torch.equal(bufs[0], bufs[1])
        assert all(len(gpu_ctx.wavefront_items(level)) > 0 for level in (1, 2, 3))
    finally:
        gpu_ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, -1)


def test_error_paths(gpu_ctx, default_scene):
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(64, 48)
    f.max_depth = 4
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.render(f)
    assert e.value.code == rt._ffi.RT_ERR_UNSUPPORTED
    f = default_scene.frame(64, 48, y0=30, y1=30)
    with pytest.raises(rt.RtError) as e:
        gpu_ctx.render(f)
    assert e.value.code == rt._ffi.RT_ERR_INVALID_ARG
    with rt.Context(0) as fresh:
        with pytest.raises(rt.RtError) as e:
            fresh.render(default_scene.frame(8, 8))
        assert e.value.code == rt._ffi.RT_ERR_NO_SCENE
        # more cubes than the BVH walk's 32-bit pair offsets reach are refused before any array is read
        import ctypes as C
        s = default_scene
        rc = fresh._lib.rt_scene_upload(fresh._ctx, C.cast(s.cubes, C.POINTER(rt.Cube)), (1 << 26) + 1, None, 0,
                                        C.cast(s.materials, C.POINTER(rt.Material)), len(s.materials), None, 0, 0, 0, 16)
        assert rc == rt._ffi.RT_ERR_UNSUPPORTED and b"2^26" in fresh._lib.rt_last_error(fresh._ctx)


def test_empty_scene_and_ragged_sizes(gpu_ctx, oracle_mod, default_scene):
    """Edge cases: no objects at all, a frame that is not a multiple of the 16x16 tile, a 1-row frame."""
    s = default_scene
    empty = rt.make_scene([], [], [s.materials[0]], list(s.lights), s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    gpu_ctx.upload(empty)
    fb = gpu_ctx.render(empty.frame(37, 19))
    assert fb[1, 0] == 0 and (fb == s.sky_0rgb).sum() == 37 * 19 - 1
    for w, h in [(37, 19), (129, 1), (1, 5), (17, 33)]:
        f, gfb, gpl, ofb, opl, _ = _both(gpu_ctx, oracle_mod, s, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"ragged {w}x{h}")


def _cam(eye, center, up=(0.0, 1.0, 0.0)):
    return rt.Camera(rt.Vec3(*eye), rt.Vec3(*center), rt.Vec3(*up))


@pytest.mark.parametrize("name,eye,center", [
    ("looking away (cubes behind the eye are hit at negative t)", (0.0, 0.0, 10.0), (0.0, 0.0, 20.0)),
    ("eye inside the diorama", (0.2, 1.0, 0.3), (3.0, 0.0, -3.0)),
    ("far away", (0.0, 5.0, 200.0), (0.0, 0.0, 0.0)),
    ("from above, nearly straight down", (0.0, 30.0, 0.01), (0.0, 0.0, 0.0)),
    ("axis-aligned view, eye on a cube face plane", (0.75, 3.0, 12.0), (0.75, 3.0, 0.0)),
    ("grazing the platform top", (-9.0, -0.75, 0.0), (9.0, -0.75, 0.0)),
])
def test_cameras_around_the_scene(gpu_ctx, oracle_mod, default_scene, name, eye, center):
    """Camera placements that exercise the tile classification's give-up paths (corners beside/behind the
    eye), negative-distance hits and rays with exactly-zero direction components (even frame width)."""
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, default_scene, 256, 160, camera=_cam(eye, center))
    assert_parity(gfb, gpl, ofb, opl, name)
    c = gpu_ctx.count_rays(f)
    assert c.uniq_dict() == ocnt.uniq_dict()


def test_degenerate_cubes(gpu_ctx, oracle_mod, default_scene):
    """Negative-size, zero-size, duplicated and eye-containing cubes, plus axis-parallel rays."""
    s = default_scene
    m = list(s.materials)
    V = rt.Vec3
    cubes = [rt.Cube(0, V(0.0, 0.0, 0.0), 2.0, 0),
             rt.Cube(1, V(0.0, 0.0, 0.0), 2.0, 1 % len(m)),            # exact duplicate: lowest index must win
             rt.Cube(2, V(3.0, 0.0, 0.0), -2.0, 2 % len(m)),           # min > max
             rt.Cube(3, V(-3.0, 0.0, 0.0), 0.0, 3 % len(m)),           # a point
             rt.Cube(4, V(0.0, 3.0, 1.0), 1.0, 4 % len(m)),
             rt.Cube(5, V(0.0, 0.0, 9.0), 4.0, 5 % len(m)),            # contains the eye
             rt.Cube(6, V(1.0, 0.0, 0.0), 2.0, 6 % len(m))]            # shares the face x = 1 ... 2 with cube 0/2
    sc = rt.make_scene(cubes, [], m, list(s.lights), _cam((0.0, 0.0, 10.0), (0.0, 0.0, 0.0)), s.ambient_light, s.sky_0rgb, s.atlas)
    for w, h in [(128, 96), (97, 64)]:
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"degenerate cubes {w}x{h}")
        assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_shared_reciprocal_division_is_ieee(gpu_ctx):
    """The kernels divide several numerators by one denominator with a shared refined reciprocal; on the
    device that must equal the compiler's IEEE division in every bit (2^32 pseudo-random cases)."""
    for seed in (1, 2):
        bad, n = gpu_ctx.selftest_division(1 << 31, seed=seed)
        assert n > (1 << 30) and bad == 0, (bad, n)


@pytest.mark.parametrize("dark", [0, 1])
@pytest.mark.parametrize("scale,shadow_list", [(0.0, 1), (1e9, 1), (1e9, 0)])
def test_long_list_paths_forced(tuned_ctx, oracle_mod, scale, shadow_list, dark):
    """Long cube lists have two traversals (wave-cooperative over Morton-ordered chunks with chunk boxes,
    per-lane BVH walk for incoherent bundles); the choice per wave is a threshold.  Force each one for
    every wave (RT_TUNE_COHERENT_SCALE, applied by rt_scene_upload) and require the same parity.  Shadow
    queries of the cooperative traversal walk a list-ordered or the Morton-ordered stream, a choice per
    scene (RT_TUNE_SHADOW_LIST forces it); the plain kernel has a variant per RT_TUNE_DARK_KERNEL.  _both
    compares the plain kernel's frame too."""
    tuned_ctx.set_tuning(rt.RT_TUNE_COHERENT_SCALE, scale)
    tuned_ctx.set_tuning(rt.RT_TUNE_SHADOW_LIST, shadow_list)
    tuned_ctx.set_tuning(rt.RT_TUNE_DARK_KERNEL, dark)
    for n_cubes, n_spheres, w, h in ((900, 8, 200, 120), (600, 24, 160, 90)):
        scene = rt.synthetic_scene(n_cubes=n_cubes, n_spheres=n_spheres)
        f, gfb, gpl, ofb, opl, ocnt = _both(tuned_ctx, oracle_mod, scene, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"synthetic {n_cubes}c, coherent scale {scale}, shadow list {shadow_list}, dark {dark}")
        assert tuned_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_shuffled_long_list(tuned_ctx, oracle_mod):
    """A long list in random order (consecutive entries far apart): rt_scene_upload must then keep the
    Morton-ordered stream for shadows; results equal the oracle's either way, also when the list-ordered
    stream is forced on such a list."""
    base = rt.synthetic_scene(n_cubes=700, n_spheres=0)
    rng = np.random.default_rng(5)
    order = rng.permutation(len(base.cubes))
    cubes = [base.cubes[int(i)] for i in order]
    sc = rt.make_scene(cubes, [], list(base.materials), list(base.lights), base.camera, base.ambient_light, base.sky_0rgb, base.atlas)
    for forced in (-1, 1):
        tuned_ctx.set_tuning(rt.RT_TUNE_SHADOW_LIST, forced)
        f, gfb, gpl, ofb, opl, ocnt = _both(tuned_ctx, oracle_mod, sc, 160, 96)
        assert_parity(gfb, gpl, ofb, opl, f"shuffled long list, shadow list {forced}")
        assert tuned_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


@pytest.mark.parametrize("scale", [0.0, 1.0, 1e9])
def test_long_list_with_duplicate_cubes(tuned_ctx, oracle_mod, scale):
    """Exact duplicates in a long list: in the Morton order two equal cubes are neighbours, so they end up as the two
    LEAVES of one pair record, and a ray that meets one meets both - the walk's second leaf evaluation of a step, the
    contested nearest hit (equal distances: the lowest list index wins, src/render.rs:94-101) and the shadow rule
    (lowest index with a different id, src/render.rs:44-70, :113) all in one scene.  A third of the copies keep the
    original's id (excluded together with it), the others get a new one; some copies are a little larger (overlap
    without equality).  Each traversal forced for every wave, and the library's own choice."""
    from raytracing_amd._ffi import Camera, Cube, Vec3
    base = rt.synthetic_scene(n_cubes=420, n_spheres=6)
    rng = np.random.default_rng(11)
    cubes = [base.cubes[i] for i in range(len(base.cubes))]
    copies = []
    for i, c in enumerate(cubes):
        k = Cube()
        k.id = c.id if i % 3 == 0 else 100000 + i
        k.center = c.center
        k.size = c.size * (1.25 if i % 7 == 0 else 1.0)
        k.material = base.cubes[int(rng.integers(len(cubes)))].material
        copies.append(k)
    order = rng.permutation(len(copies))
    cubes = cubes + [copies[int(j)] for j in order]
    spheres = [base.spheres[i] for i in range(len(base.spheres))]
    cam = Camera(Vec3(3.0, 5.0, 12.0), Vec3(0.0, -2.0, 0.0), Vec3(0.0, 1.0, 0.0))  # close to the small terrain, looking down on it
    sc = rt.make_scene(cubes, spheres, list(base.materials), list(base.lights), cam, base.ambient_light, base.sky_0rgb, base.atlas)
    tuned_ctx.set_tuning(rt.RT_TUNE_COHERENT_SCALE, scale)
    f, gfb, gpl, ofb, opl, ocnt = _both(tuned_ctx, oracle_mod, sc, 192, 108)
    assert_parity(gfb, gpl, ofb, opl, f"long list with duplicates, coherent scale {scale}")
    assert tuned_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()
    hc = gpl["hit_cube"]
    assert (hc != rt.RT_NO_HIT).mean() > 0.15 and (hc[hc != rt.RT_NO_HIT] >= 420).any()  # copies win where they are larger


@pytest.mark.parametrize("dark,scale", [(-1, 1.0), (0, 1.0), (1, 1.0), (-1, 0.0), (-1, 1e9)])
def test_config4_world_at_size(tuned_ctx, oracle_mod, dark, scale):
    """BASELINE.json config 4 at its size - 10 000 cubes (157 chunks: three 64-chunk groups of the chunk
    traversal, a 14-level BVH) + 256 spheres - on a frame the oracle finishes in seconds, against the oracle
    run here AND the committed golden frame (tests/golden/synthetic_10000c_256s_320x180.npz), debug and
    plain kernels, each plain variant (RT_TUNE_DARK_KERNEL) and each traversal forced for every wave.
    Replaces the brute-force walks of reference src/render.rs:87-101 and :44-70."""
    tuned_ctx.set_tuning(rt.RT_TUNE_DARK_KERNEL, dark)
    tuned_ctx.set_tuning(rt.RT_TUNE_COHERENT_SCALE, scale)
    scene = rt.synthetic_scene()
    assert scene.n_cubes == 10000 and scene.n_spheres == 256
    f, gfb, gpl, ofb, opl, ocnt = _both(tuned_ctx, oracle_mod, scene, 320, 180)
    assert_parity(gfb, gpl, ofb, opl, f"config 4, dark {dark}, coherent scale {scale}")
    assert tuned_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "synthetic_10000c_256s_320x180.npz"))
    assert np.array_equal(gfb, g["fb"]) and np.array_equal(tuned_ctx.render(f), g["fb"])
    for k in ("hit_cube", "hit_sphere", "tree_hash"):
        assert np.array_equal(gpl[k], g[k]), k
    assert (gpl["hit_sphere"] != rt.RT_NO_HIT).any() and (gpl["hit_cube"] != rt.RT_NO_HIT).mean() > 0.05


def test_bvh_depth_limit_falls_back_to_the_chunk_traversal(tuned_ctx, oracle_mod):
    """rt_scene_upload sends every wave through the chunk traversal when the BVH is deeper than the walk's
    per-lane stack (kBvhStackMax = 28 levels: unreachable with real lists, so the limit is a tuning key)."""
    tuned_ctx.set_tuning(rt.RT_TUNE_BVH_DEPTH_LIMIT, 5)
    scene = rt.synthetic_scene(n_cubes=4200, n_spheres=3)  # 66 chunks: a second 64-chunk group
    f, gfb, gpl, ofb, opl, ocnt = _both(tuned_ctx, oracle_mod, scene, 200, 112)
    assert_parity(gfb, gpl, ofb, opl, "bvh depth limit")
    _, ex = tuned_ctx.count_executed(f)
    assert ex.bvh_steps == 0 and ex.chunk_tests > 0
    tuned_ctx.reset_tuning()
    tuned_ctx.upload(scene)
    _, ex = tuned_ctx.count_executed(f)
    assert ex.bvh_steps > 0


def test_config4_bands_of_the_4k_frame_against_the_oracle(gpu_ctx, oracle_mod):
    """BASELINE.json config 4 at its exact size, 3840x2160: the whole frame is minutes of brute force for the oracle,
    so two 32-row bands of it through the terrain (the same rays as in the
    whole frame) are compared on every pixel: hit planes, ray-tree hashes, colours, counters."""
    scene = rt.synthetic_scene()
    for y0 in (1000, 1232):
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, scene, 3840, 2160, y0=y0, y1=y0 + 32)
        rep = assert_parity(gfb, gpl, ofb, opl, f"config 4: rows {y0}..{y0 + 32} of 3840x2160")
        assert rep["exact_fraction"] == 1.0, rep
        assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()
        assert (gpl["hit_cube"] != rt.RT_NO_HIT).mean() > 0.1, y0


def test_config4_plain_equals_debug_at_4k(gpu_ctx):
    """The plain long-list kernel at BASELINE's size (3840x2160, 10k cubes + 256 spheres; what
    `bench.py --workload synthetic_4k` times) paints the frame the debug variant paints; size-independent
    properties beside it: bands equal the full frame, the unpainted pixel."""
    import torch
    scene = rt.synthetic_scene()
    gpu_ctx.upload(scene)
    W, H = 3840, 2160
    f = scene.frame(W, H)
    bufs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(2)]
    planes = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    gpu_ctx.render_device(f, bufs[0].data_ptr())
    gpu_ctx.render_device(f, bufs[1].data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr())
    gpu_ctx.sync()
    assert torch.equal(bufs[0], bufs[1])
    full = bufs[0].cpu().numpy().view(np.uint32)
    assert full[1, 0] == 0 and full[0, 0] == scene.sky_0rgb
    assert (planes[0].cpu().numpy().view(np.uint32) != rt.RT_NO_HIT).mean() > 0.05
    band = gpu_ctx.render(scene.frame(W, H, y0=1203, y1=1260))
    assert np.array_equal(band, full[1203:1260])


def test_config5_frame_properties_at_8k(gpu_ctx, default_scene):
    """BASELINE.json config 5's frame (7680x4320) after an orbit step: bands == the full frame, the parts of
    an 8-way split reassemble to it, sky rows are sky, the unpainted pixel is black, plain == debug."""
    import torch
    from raytracing_amd import multigpu
    gpu_ctx.upload(default_scene)
    W, H = 7680, 4320
    cam = rt.rotate_cam(default_scene.camera, 3.14159265358979 / 180.0 * 37, 0.0)
    f = default_scene.frame(W, H, camera=cam)
    full_t = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    dbg_t = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    hash_t = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_device(f, full_t.data_ptr())
    gpu_ctx.render_device(f, dbg_t.data_ptr(), d_tree_hash=hash_t.data_ptr())
    gpu_ctx.sync()
    assert torch.equal(full_t, dbg_t)
    full = full_t.cpu().numpy().view(np.uint32)
    a, b = gpu_ctx.row_extent(f)
    assert 0 < a < b <= H  # from this angle the platform's near corner leaves the frame at the bottom
    sky = np.full((W,), default_scene.sky_0rgb, dtype=np.uint32)
    assert full[1, 0] == 0 and (full[0] == sky).all() and (full[2:a] == sky).all() and (full[b:] == sky).all()
    assert (full[a:b] != default_scene.sky_0rgb).any()
    for y0, y1 in ((0, 540), (2153, 2171), (3780, 4320)):
        assert np.array_equal(gpu_ctx.render(default_scene.frame(W, H, y0=y0, y1=y1, camera=cam)), full[y0:y1]), (y0, y1)
    part = torch.zeros((multigpu.max_part_rows(H, 8), W), dtype=torch.int32, device="cuda")
    for p in (0, 5):
        torch.cuda.synchronize()
        gpu_ctx.render_device(f, part.data_ptr(), n_parts=8, part=p)
        gpu_ctx.sync()
        rows = multigpu.part_rows(H, 8, p)
        assert np.array_equal(part[:rows].cpu().numpy().view(np.uint32), full[multigpu.part_row_indices(H, 8, p)]), p


def test_golden_bmp_of_the_default_frame(gpu_ctx, default_scene):
    """BASELINE.json config 1: the default diorama at main.rs's framebuffer size (1080x720, reference
    src/main.rs:25-29) -> BMP.  GPU frame -> bmp::encode of the host mirror (write_bmp_file, reference
    src/bmp.rs:19-113 via Framebuffer::save, src/framebuffer.rs:212-221) -> sha256 equals the committed
    oracle frame's BMP (tests/golden/minecraft_1080x720.npz, made by tests/make_golden.py)."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "minecraft_1080x720.npz"))
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(1080, 720)
    fb, planes = gpu_ctx.render(f, debug=True)
    plain = gpu_ctx.render(f)
    assert np.array_equal(plain, fb)
    assert np.array_equal(planes["hit_cube"], g["hit_cube"])
    assert int(np.bitwise_xor.reduce(planes["tree_hash"].ravel())) == int(g["tree_hash_xor"])
    bmp = rt.bmp_encode(plain, 1080, 720)
    assert len(bmp) == 54 + 1080 * 720 * 3
    assert hashlib.sha256(bmp).hexdigest() == str(g["bmp_sha256"])
    assert hashlib.sha256(plain.tobytes()).hexdigest() == str(g["fb_sha256"])


def test_render_into_pinned_memory(gpu_ctx, default_scene):
    """rt_render into page-locked memory from rt_host_alloc (the fast read-back) == into pageable memory."""
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(328, 200)
    want = gpu_ctx.render(f)
    pin = rt.PinnedFrame(200, 328)
    try:
        pin.array[:] = 0x55AA55
        got = gpu_ctx.render(f, out=pin.array)
        assert got is pin.array and np.array_equal(got, want)
    finally:
        pin.close()


@pytest.mark.parametrize("heavy", [0, 2, 8, -1])
def test_heavy_tiles_on_three_waves(tuned_ctx, oracle_mod, default_scene, water_scene, heavy):
    """RT_TUNE_HEAVY_CLASS: tiles of cost class >= the key (costs of the context's previous launch) are traced
    by render_heavy_kernel on the second stream - a workgroup of three waves per tile, a tracer that posts each
    node's light fold to two shader waves, the general cull, bundles walked one by one - and the others by
    render_kernel.  As many tiles as fit (0), the expensive ones (2), none (8), automatic (-1: on for the
    synchronous rt_render of a small frame): the frame is the debug variant's (one wave per tile, inline fold)
    and the oracle's, for the diorama, the water variant (two children per node), a long list with spheres,
    1 and 3 and 16 lights, and repeated launches (the second and later ones use the list the one before wrote)."""
    ctx = tuned_ctx
    ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, heavy)
    s = default_scene
    lights16 = [rt.Light(rt.Vec3(-8.0 + k, 6.0 + 0.5 * k, 9.0 - 1.3 * k), 0x102030 * (k % 7 + 1) & 0xFFFFFF, 0.05 + 0.07 * k) for k in range(16)]
    many = rt.make_scene(list(s.cubes), [], list(s.materials), lights16, s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    one = rt.make_scene(list(s.cubes), [], list(s.materials), lights16[:1], s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    three = rt.make_scene(list(s.cubes), [], list(s.materials), lights16[:3], s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    syn = rt.synthetic_scene(n_cubes=1500, n_spheres=12)
    for sc, w, h in ((s, 480, 270), (water_scene, 480, 270), (syn, 320, 180), (one, 160, 96), (three, 160, 96), (many, 160, 96)):
        f, gfb, gpl, ofb, opl, ocnt = _both(ctx, oracle_mod, sc, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"heavy class {heavy}: {sc.name} {w}x{h}")
        for _ in range(3):  # the plain launches' own tile costs now select the heavy tiles
            assert np.array_equal(ctx.render(f), gfb)
        assert ctx.debug_status() == 0
    if heavy == 0:
        costs = ctx.tile_costs()
        assert (costs > 0).any()


@pytest.mark.parametrize("sync", [1, 0])
def test_heavy_tiles_asynchronous_entry_points(tuned_ctx, default_scene, water_scene, sync):
    """The side kernel behind rt_render_device / rt_render_device_part: the heavy list a launch uses was written
    by the launch before it, for that launch's camera - a moving camera, a changed frame size (the list is then
    dropped), the parts of a split, frames queued back to back into one buffer without a host wait (the side
    stream is fenced against the context's stream both ways), a debug launch in between.  Every frame equals
    the one the same context renders with the side kernel off.  Both ways of ordering the two streams
    (RT_TUNE_HEAVY_SYNC: 1 device-side gate and join kernels, 0 events - what the library falls back to under
    a profiler that serialises kernels)."""
    import torch
    ctx = tuned_ctx
    ctx.set_tuning(rt.RT_TUNE_HEAVY_SYNC, sync)
    for scene in (default_scene, water_scene):
        ctx.upload(scene)
        cams = []
        for k in range(6):
            c = rt.Camera(); C0 = scene.camera
            c.eye = rt.Vec3(C0.eye.x + 0.7 * k, C0.eye.y + 0.3 * k, C0.eye.z - 0.5 * k); c.center = C0.center; c.up = C0.up
            cams.append(c)
        sizes = [(480, 272), (480, 272), (480, 272), (320, 200), (320, 200), (480, 272)]
        frames = [scene.frame(w, h, camera=c) for (w, h), c in zip(sizes, cams)]
        ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 8)
        want = [ctx.render(f) for f in frames]
        for heavy in (0, 3):
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, heavy)
            # 1. one buffer per frame, everything queued, one wait at the end
            outs = [torch.zeros((f.height, f.width), dtype=torch.int32, device="cuda") for f in frames]
            torch.cuda.synchronize()
            for rep in range(2):
                for f, o in zip(frames, outs):
                    ctx.render_device(f, o.data_ptr())
            ctx.sync()
            for f, o, wnt in zip(frames, outs, want):
                assert np.array_equal(o.cpu().numpy().view(np.uint32), wnt), f"heavy {heavy}: {scene.name} {f.width}x{f.height}"
            # 2. the same camera again and again into ONE buffer, a debug launch in between, no host wait
            f = frames[0]
            o = outs[0]
            hc = torch.zeros_like(o)
            for rep in range(5):
                ctx.render_device(f, o.data_ptr())
                if rep == 2:
                    ctx.render_device(f, o.data_ptr(), d_hit_cube=hc.data_ptr())
            ctx.sync()
            assert np.array_equal(o.cpu().numpy().view(np.uint32), want[0])
            # 3. the parts of a 3-way split, three frames each
            f = frames[1]
            for part in range(3):
                rows = rt.part_rows(f.y1 - f.y0, 3, part)
                po = torch.zeros((rows, f.width), dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                for rep in range(3):
                    ctx.render_device(f, po.data_ptr(), n_parts=3, part=part)
                ctx.sync()
                idx = rt.part_row_indices(f.y1 - f.y0, 3, part)
                assert np.array_equal(po.cpu().numpy().view(np.uint32), want[1][idx]), f"heavy {heavy}: part {part}"
            assert ctx.debug_status() == 0


def test_heavy_tiles_three_contexts_in_flight(default_scene):
    """Three contexts on one GPU, each with the side kernel on, frames interleaved without host waits: six
    streams on four hardware queues.  The waits between a context's two streams are device-side polls; with the
    side stream at a raised priority the polling gate starved the kernel it waited for (every frame then took
    seconds, until the poll gave up).  Frames equal the single-context frame and no wait gives up (rt_debug_status)."""
    import time
    import torch
    W, H = 960, 544
    f = default_scene.frame(W, H)
    ctxs = [rt.Context(0) for _ in range(3)]
    try:
        outs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in ctxs]
        torch.cuda.synchronize()
        for c in ctxs:
            c.upload(default_scene)
            c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 8)
        want = ctxs[0].render(f)
        for c in ctxs:
            c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 2)
        t0 = time.perf_counter()
        for k in range(60):
            ctxs[k % 3].render_device(f, outs[k % 3].data_ptr())
        for c in ctxs:
            c.sync()
        dt = time.perf_counter() - t0
        for c, o in zip(ctxs, outs):
            assert c.debug_status() == 0
            assert np.array_equal(o.cpu().numpy().view(np.uint32), want)
        print(f"60 frames over three contexts: {dt * 1e3:.1f} ms")  # a gave-up poll (seconds) shows in debug_status above
    finally:
        for c in ctxs:
            c.close()


def test_side_kernel_give_up_is_never_a_silent_wrong_frame(default_scene):
    """A device-side wait of the side kernel that gives up (RT_TUNE_HEAVY_SPIN_LIMIT = 1: the join kernel reports
    every launch with a side kernel as given up, without looking) must not reach the caller as a frame: the
    synchronous rt_render renders the frame again without the side kernel, takes it out of service for the
    context and says so in rt_last_error; behind the asynchronous entry points rt_sync returns RT_ERR_DEVICE_WAIT.
    Setting the key back to 0 restores the side kernel."""
    import torch
    W, H = 320, 200
    f = default_scene.frame(W, H)
    with rt.Context(0) as ctx:
        ctx.upload(default_scene)
        ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 8)
        want = ctx.render(f)
        ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 0)
        ctx.set_tuning(rt.RT_TUNE_HEAVY_SYNC, 1)
        ctx.set_tuning(rt.RT_TUNE_HEAVY_SPIN_LIMIT, 1)
        assert np.array_equal(ctx.render(f), want)      # writes the first heavy list; no side kernel yet
        assert "gave up" not in ctx.last_error()
        assert np.array_equal(ctx.render(f), want)      # side kernel + impatient join -> rendered again without it
        assert "gave up" in ctx.last_error() and "disabled" in ctx.last_error()
        assert ctx.debug_status() == 0                    # the word was dealt with
        for _ in range(3):                                # out of service: plain launches
            assert np.array_equal(ctx.render(f), want)
        # asynchronous entry points: rt_sync reports, the caller renders again
        ctx.set_tuning(rt.RT_TUNE_HEAVY_SPIN_LIMIT, 0)    # back in service ...
        ctx.set_tuning(rt.RT_TUNE_HEAVY_SPIN_LIMIT, 1)    # ... with an impatient join
        out = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.render_device(f, out.data_ptr())
        ctx.sync()                                        # the list's first launch: nothing to wait for
        ctx.render_device(f, out.data_ptr())
        with pytest.raises(rt.RtError) as e:
            ctx.sync()
        assert e.value.code == rt._ffi.RT_ERR_DEVICE_WAIT
        ctx.render_device(f, out.data_ptr())
        ctx.sync()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
        ctx.set_tuning(rt.RT_TUNE_HEAVY_SPIN_LIMIT, 0)    # patient again: the side kernel works
        for _ in range(3):
            assert np.array_equal(ctx.render(f), want)
        assert "gave up" in ctx.last_error()              # the old warning stays until the next failing call
        assert ctx.debug_status() == 0


def test_side_kernel_moving_camera_empty_and_full_lists():
    """Several contexts with the side kernel on, a camera that moves every frame, and views without geometry in
    between: a launch whose heavy list is empty is followed by one whose list is not.  The join must outlast
    every workgroup of the side kernel, or a workgroup that starts late reads the count and list of the launch
    after next from the reused header bank and paints its tiles with the wrong camera.  Every frame lands in a
    buffer of its own and must equal the frame rendered without the side kernel."""
    import torch
    s = rt.default_scene()
    W, H = 640, 360
    cams = []
    cam = s.camera
    for k in range(12):
        if k % 4 == 2:   # looking at the sky: nothing to trace, an empty heavy list for the next launch
            cams.append(_cam((0.0, 0.0, 40.0), (0.0, 120.0, 0.0)))
        else:
            cam = rt.rotate_cam(cam, 0.21, 0.05 if k % 2 else -0.04)
            cams.append(cam)
    frames = [s.frame(W, H, camera=c) for c in cams]
    ctxs = [rt.Context(0) for _ in range(3)]
    try:
        for c in ctxs:
            c.upload(s)
            c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 8)
        want = [ctxs[0].render(f) for f in frames]
        assert (want[2] == s.sky_0rgb).sum() >= W * H - 1 and (want[3] != s.sky_0rgb).any()
        for c in ctxs:
            c.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 0)
            c.set_tuning(rt.RT_TUNE_HEAVY_SYNC, 1)
        outs = [[torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in frames] for _ in ctxs]
        torch.cuda.synchronize()
        for rep in range(2):
            for k, f in enumerate(frames):
                for c, o in zip(ctxs, outs):
                    c.render_device(f, o[k].data_ptr())
        for c in ctxs:
            c.sync()
        for ci, o in enumerate(outs):
            for k in range(len(frames)):
                got = o[k].cpu().numpy().view(np.uint32)
                assert np.array_equal(got, want[k]), (ci, k, int((got != want[k]).sum()))
            assert ctxs[ci].debug_status() == 0
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("refine", [0, 1, 2])
def test_cull_refinement_forced(tuned_ctx, oracle_mod, default_scene, water_scene, refine):
    """RT_TUNE_CULL_REFINE: the plain short-list kernel without / with the cull refinement (the box form of the
    wave-level cull applied to chunks where the one-radius form leaves 20 candidates or more) / with the refinement
    and the walk by bundles (waves whose secondary rays left through different faces).  Each kernel paints the debug
    variant's frame (which has neither) and the oracle's: the diorama from four sides, the water variant, a dense
    random scene, 16 lights."""
    ctx = tuned_ctx
    ctx.set_tuning(rt.RT_TUNE_CULL_REFINE, refine)
    s = default_scene
    lights16 = [rt.Light(rt.Vec3(-8.0 + k, 6.0 + 0.5 * k, 9.0 - 1.3 * k), 0x102030 * (k % 7 + 1) & 0xFFFFFF, 0.05 + 0.07 * k) for k in range(16)]
    many = rt.make_scene(list(s.cubes), [], list(s.materials), lights16, s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    cases = [(s, 480, 270, {}), (water_scene, 480, 270, {}), (many, 160, 96, {}), (rt.synthetic_scene(n_cubes=240, n_spheres=6), 320, 180, {})]
    cam = s.camera
    for k in range(3):
        cam = rt.rotate_cam(cam, 1.3, 0.1)
        cases.append((s, 320, 200, {"camera": cam}))
    for sc, w, h, kw in cases:
        f, gfb, gpl, ofb, opl, ocnt = _both(ctx, oracle_mod, sc, w, h, **kw)
        assert_parity(gfb, gpl, ofb, opl, f"cull refinement {refine}: {sc.name} {w}x{h}")
    assert ctx.kernel_name() == f"render_kernel<false,false,{(0, 2, 3)[refine]},false>"
    for seed in range(300, 312):
        sc, label = random_scene(rt, default_scene, seed, 250)
        f, gfb, gpl, ofb, opl, ocnt = _both(ctx, oracle_mod, sc, 96, 64)
        assert_parity(gfb, gpl, ofb, opl, f"cull refinement {refine}: " + label)


def _wavefront_cases():
    s = rt.default_scene()
    water = rt.default_scene(with_water=True)
    lights16 = [rt.Light(rt.Vec3(-8.0 + k, 6.0 + 0.5 * k, 9.0 - 1.3 * k), 0x102030 * (k % 7 + 1) & 0xFFFFFF, 0.05 + 0.07 * k) for k in range(16)]
    three = rt.make_scene(list(s.cubes), [], list(s.materials), lights16[:3], s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    none = rt.make_scene(list(s.cubes), [], list(s.materials), [], s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    syn = rt.synthetic_scene(n_cubes=200, n_spheres=12)
    matte = []
    for m in syn.materials:
        m = rt.Material.from_buffer_copy(m)
        m.reflectivity = 0.0
        m.transparency = 0.0
        matte.append(m)
    syn_matte = rt.make_scene(list(syn.cubes), list(syn.spheres), matte, list(syn.lights), syn.camera, syn.ambient_light, syn.sky_0rgb, syn.atlas)
    syn_dark = rt.make_scene(list(syn.cubes), list(syn.spheres), list(syn.materials), [], syn.camera, syn.ambient_light, syn.sky_0rgb, syn.atlas)
    # two passes (spheres and cubes) whose roots have no children: the merge still has to run
    flat = [("spheres, depth 0", syn, 200, 120, {"max_depth": 0}), ("spheres, matte materials", syn_matte, 200, 120, {}),
            ("spheres, no lights", syn_dark, 200, 120, {})]
    return flat + [("diorama", s, 480, 270, {}), ("water", water, 480, 270, {}), ("water depth 1", water, 200, 120, {"max_depth": 1}),
            ("water depth 0", water, 200, 120, {"max_depth": 0}), ("water depth 2", water, 203, 117, {"max_depth": 2}),
            ("synthetic short list", rt.synthetic_scene(n_cubes=200, n_spheres=12), 320, 180, {}),
            ("synthetic long list", rt.synthetic_scene(n_cubes=1500, n_spheres=12), 320, 180, {}),
            ("three lights", three, 160, 96, {}), ("no lights", none, 160, 96, {}),
            ("eye inside the diorama", s, 256, 160, {"camera": _cam((0.2, 1.0, 0.3), (3.0, 0.0, -3.0))}),
            ("NaN basis", water, 64, 40, {"camera": _cam((0.0, 0.0, 10.0), (0.0, 0.0, 10.0))}),
            ("band", water, 328, 200, {"y0": 37, "y1": 150})]


@pytest.mark.parametrize("cap", [0, 3, 64])
def test_wavefront_launches_paint_the_same_frame(tuned_ctx, oracle_mod, cap):
    """RT_TUNE_WAVEFRONT: the plain frame through one launch per ray-tree level over bundles of coherent rays
    (wf_trace_kernel, wf_combine_kernel) instead of the tile-resident walk.  Every frame equals the debug variant's
    (one wave per tile walks the whole tree) and the oracle's, bit for bit; with the item capacity cut to a few
    bundles (RT_TUNE_WAVEFRONT_CAP) most tiles overflow and are rendered again by render_kernel from the redo
    list, with the same result.  Repeated launches reuse the buffers (stamps tell stale items apart)."""
    ctx = tuned_ctx
    ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
    ctx.set_tuning(rt.RT_TUNE_WAVEFRONT_CAP, cap)
    for name, sc, w, h, kw in _wavefront_cases():
        f, gfb, gpl, ofb, opl, ocnt = _both(ctx, oracle_mod, sc, w, h, **kw)   # _both: plain (wavefront) == debug
        assert_parity(gfb, gpl, ofb, opl, f"wavefront cap {cap}: {name}")
        for _ in range(2):
            got = ctx.render(f)
            assert np.array_equal(got, gfb), (name, int((got != gfb).sum()))


def test_wavefront_random_scenes(tuned_ctx, oracle_mod, default_scene):
    ctx = tuned_ctx
    ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
    for seed in range(100, 124):
        sc, label = random_scene(rt, default_scene, seed)
        f, gfb, gpl, ofb, opl, ocnt = _both(ctx, oracle_mod, sc, 96, 64)
        assert_parity(gfb, gpl, ofb, opl, "wavefront " + label)


def test_wavefront_parts_and_async(tuned_ctx, water_scene):
    """The wavefront launches behind rt_render_device_part: every part of 2/3-way splits, frames queued back to
    back on one stream with a moving camera and changing sizes, equal the tile walk's frames."""
    import torch
    ctx = tuned_ctx
    ctx.upload(water_scene)
    cams = [rt.rotate_cam(water_scene.camera, 0.3 * k, 0.05 * k) for k in range(5)]
    sizes = [(480, 272), (480, 272), (320, 200), (328, 203), (480, 272)]
    frames = [water_scene.frame(w, h, camera=c) for (w, h), c in zip(sizes, cams)]
    ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 0)
    want = [ctx.render(f) for f in frames]
    ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
    outs = [torch.zeros((f.height, f.width), dtype=torch.int32, device="cuda") for f in frames]
    torch.cuda.synchronize()
    for rep in range(2):
        for f, o in zip(frames, outs):
            ctx.render_device(f, o.data_ptr())
    ctx.sync()
    for f, o, wnt in zip(frames, outs, want):
        got = o.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, wnt), (f.width, f.height, int((got != wnt).sum()))
    f = frames[1]
    for n in (2, 3):
        for part in range(n):
            rows = rt.part_rows(f.y1 - f.y0, n, part)
            po = torch.zeros((rows, f.width), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.render_device(f, po.data_ptr(), n_parts=n, part=part)
            ctx.sync()
            idx = rt.part_row_indices(f.y1 - f.y0, n, part)
            assert np.array_equal(po.cpu().numpy().view(np.uint32), want[1][idx]), (n, part)


def test_tuning_keys_are_validated(gpu_ctx):
    for key, bad in ((rt.RT_TUNE_HEAVY_SYNC, 2), (rt.RT_TUNE_HEAVY_SPIN_LIMIT, -1), (rt.RT_TUNE_HEAVY_SPIN_LIMIT, 0.5), (rt.RT_TUNE_WAVEFRONT, 2), (rt.RT_TUNE_CULL_REFINE, 3), (rt.RT_TUNE_WAVEFRONT_CAP, -1), (rt.RT_TUNE_HEAVY_CLASS, 9), (rt.RT_TUNE_HEAVY_CLASS, -2), (rt.RT_TUNE_HEAVY_CLASS, 2.5), (rt.RT_TUNE_COHERENT_SCALE, -1.0), (rt.RT_TUNE_SHADOW_LIST, 2), (rt.RT_TUNE_DARK_KERNEL, 0.5), (rt.RT_TUNE_FILL_MEMSET, -1), (99, 0)):
        with pytest.raises(rt.RtError) as e:
            gpu_ctx.set_tuning(key, bad)
        assert e.value.code == rt._ffi.RT_ERR_INVALID_ARG
    gpu_ctx.reset_tuning()


def test_current_device_is_restored(gpu_ctx, default_scene):
    """Every entry point makes the context's GPU current only for the duration of the call."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    dev = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(dev)) == 0
    before = dev.value
    gpu_ctx.upload(default_scene)
    gpu_ctx.render(default_scene.frame(64, 40))
    assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == before


def test_executed_counters(gpu_ctx, default_scene):
    """rt_count_executed: the reference counters of the same launch equal rt_count_rays; the executed
    events are consistent with them (the cull proves most brute-force tests away, never more than all)."""
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(640, 360)
    c, ex = gpu_ctx.count_executed(f)
    assert c.uniq_dict() == gpu_ctx.count_rays(f).uniq_dict()
    assert 0 < ex.candidate_evals < c.cube_tests and ex.exact_evals < ex.candidate_evals
    assert ex.resolved_hits == c.shaded_hits and ex.light_iterations == c.shaded_hits * len(default_scene.lights)
    assert ex.blockers <= c.shadow_rays and 0 < ex.primary_rays <= 640 * 360
    assert ex.queries >= c.shadow_rays and ex.bvh_steps == 0 and ex.chunk_tests == 0 and ex.sphere_evals == 0
    assert ex.cull_tests > 0 and ex.child_setups > 0
    s = rt.synthetic_scene(n_cubes=2000, n_spheres=16)
    gpu_ctx.upload(s)
    c, ex = gpu_ctx.count_executed(s.frame(320, 180))
    assert ex.chunk_tests > 0 and ex.sphere_evals > 0 and ex.candidate_evals < c.cube_tests


def test_render_multi_on_one_gpu(gpu_ctx, default_scene):
    """rt_render_multi (one host thread, n contexts, peer copies into frame order on ctxs[0]'s GPU) with all
    'peers' on device 0: whole frames and bands, ragged last blocks, views whose extent is empty, whole or
    cut, 1/2/3/8 contexts."""
    import torch
    others = [rt.Context(0) for _ in range(7)]
    try:
        for c in [gpu_ctx] + others:
            c.upload(default_scene)
        for w, h, y0, y1, cam in ((328, 200, 0, None, None), (328, 203, 0, None, None), (200, 150, 37, 149, None), (64, 48, 0, None, ((0.0, 0.0, 10.0), (0.0, 0.0, 20.0))),
                                  (160, 100, 0, None, ((0.0, 0.0, 40.0), (0.0, 120.0, 0.0))), (96, 40, 1, 40, None), (96, 40, 2, 33, None)):
            f = default_scene.frame(w, h, y0=y0, y1=y1, camera=_cam(*cam) if cam else None)
            want = gpu_ctx.render(f)
            rows = f.y1 - f.y0
            for n in (1, 2, 3, 8):
                buf = torch.full((rows, w), 0x55AA55, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                moved = rt.render_multi(([gpu_ctx] + others)[:n], f, buf.data_ptr())
                gpu_ctx.sync()
                got = buf.cpu().numpy().view(np.uint32)
                assert np.array_equal(got, want), (w, h, y0, y1, n, int((got != want).sum()))
                assert moved <= rows * w * 4 and (n > 1 or moved == 0)
        with pytest.raises(rt.RtError):
            rt.render_multi([gpu_ctx, gpu_ctx], default_scene.frame(64, 48), 1)
    finally:
        for c in others:
            c.close()


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_max_depth_below_three(gpu_ctx, oracle_mod, water_scene, depth):
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, water_scene, 192, 128, max_depth=depth)
    assert_parity(gfb, gpl, ofb, opl, f"max_depth {depth}")
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


@pytest.mark.parametrize("n_lights", [0, 1, 16])
def test_light_counts(gpu_ctx, oracle_mod, default_scene, n_lights):
    s = default_scene
    lights = [rt.Light(rt.Vec3(-8.0 + k, 6.0 + 0.5 * k, 9.0 - 1.3 * k), 0x102030 * (k % 7 + 1) & 0xFFFFFF, 0.05 + 0.07 * k) for k in range(n_lights)]
    sc = rt.make_scene(list(s.cubes), [], list(s.materials), lights, s.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, 160, 96)
    assert_parity(gfb, gpl, ofb, opl, f"{n_lights} lights")
    c = gpu_ctx.count_rays(f)
    assert c.uniq_dict() == ocnt.uniq_dict()


@pytest.mark.parametrize("name,eye,center,up", [
    ("eye == center (NaN basis)", (0.0, 0.0, 10.0), (0.0, 0.0, 10.0), (0.0, 1.0, 0.0)),
    ("up parallel to the view direction", (0.0, 12.0, 0.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
])
def test_degenerate_cameras(gpu_ctx, oracle_mod, default_scene, name, eye, center, up):
    """Non-finite camera bases: every ray is NaN; the reference's compares then decide (a NaN ray passes
    every slab compare), and the tile classification must stand aside."""
    f, gfb, gpl, ofb, opl, _ = _both(gpu_ctx, oracle_mod, default_scene, 64, 40, camera=_cam(eye, center, up))
    assert_parity(gfb, gpl, ofb, opl, name)


def test_degenerate_spheres(gpu_ctx, oracle_mod, default_scene):
    """Zero and negative radius, a sphere around the eye, far and near spheres, cubes behind spheres."""
    s = default_scene
    m = list(s.materials)
    V = rt.Vec3
    spheres = [rt.Sphere(100, V(0.0, 0.0, 0.0), 1.5, 0), rt.Sphere(101, V(2.5, 0.5, 1.0), -1.0, 1 % len(m)),
               rt.Sphere(102, V(-2.5, 0.0, 0.0), 0.0, 2 % len(m)), rt.Sphere(103, V(0.0, 0.0, 10.0), 3.0, 3 % len(m)),
               rt.Sphere(100, V(0.5, 2.5, -4.0), 1.25, 4 % len(m)), rt.Sphere(105, V(-300.0, 40.0, -500.0), 60.0, 5 % len(m))]
    cubes = [rt.Cube(7, V(0.0, -2.5, 0.0), 3.0, 0), rt.Cube(8, V(3.0, 0.0, -3.0), 2.0, 6 % len(m))]
    sc = rt.make_scene(cubes, spheres, m, list(s.lights), _cam((0.0, 0.0, 10.0), (0.0, 0.0, 0.0)), s.ambient_light, s.sky_0rgb, s.atlas)
    for w, h in [(160, 120), (65, 33)]:
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"degenerate spheres {w}x{h}")
        assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_sphere_classification(gpu_ctx, oracle_mod, default_scene):
    """classify_tiles projects the spheres as well (the bounding box of the sphere grown as in the wave-level sphere cull):
    a tile that no sphere and no cube projects into is only painted, a tile without spheres skips the sphere pass
    (src/render.rs:248-276 would cast its rays and find nothing).  Spheres small and large, sub-pixel, across the frame's
    edges, half outside, with cubes in front of and behind them (and, in a second scene, one sphere behind the eye's plane
    and a far one that grows the small spheres' boxes past the eye: nothing can be concluded, all is traced); whole frames, bands that do
    not start on a tile row, the parts of a 3-way split, against the oracle with the ray counters (the skipped pass still
    counts the reference's list walk).  The counting variant shows that the classification took effect."""
    import torch
    s = default_scene
    m = list(s.materials)
    V = rt.Vec3
    spheres = [rt.Sphere(100, V(-6.0, 3.0, -2.0), 1.0, 0), rt.Sphere(101, V(5.5, -2.5, 1.0), 0.6, 1 % len(m)),
               rt.Sphere(102, V(0.5, 4.2, 2.0), 0.004, 2 % len(m)), rt.Sphere(103, V(13.0, 0.0, -4.0), 2.5, 3 % len(m)),
               rt.Sphere(104, V(-2.0, -9.5, -6.0), 2.0, 4 % len(m)), rt.Sphere(105, V(-9.0, 6.0, -13.0), 2.0, 5 % len(m)),
               rt.Sphere(106, V(2.0, 1.0, -14.0), 1.5, 8 % len(m)), rt.Sphere(107, V(1.0, -1.0, 6.5), 0.25, 2 % len(m))]
    cubes = [rt.Cube(7, V(2.0, 1.0, -8.0), 2.0, 0), rt.Cube(8, V(-6.0, 3.0, -6.0), 3.0, 6 % len(m)), rt.Cube(9, V(6.0, 5.0, 0.0), 1.0, 5 % len(m))]
    sc = rt.make_scene(cubes, spheres, m, list(s.lights), _cam((0.0, 0.0, 10.0), (0.0, 0.0, 0.0)), s.ambient_light, s.sky_0rgb, s.atlas)
    for w, h in [(320, 200), (131, 77)]:
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, w, h)
        assert_parity(gfb, gpl, ofb, opl, f"sphere classification {w}x{h}")
        assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()
        assert (gpl["hit_sphere"] != rt.RT_NO_HIT).mean() > 0.01
    W, H = 320, 200
    f = sc.frame(W, H)
    full = gpu_ctx.render(f)
    _, ex = gpu_ctx.count_executed(f)
    assert 0 < ex.primary_rays < 0.5 * W * H, ex.primary_rays  # most tiles are sky: painted, no ray generated
    for y0, y1 in ((0, 61), (3, 150), (93, 200), (150, 151)):
        assert np.array_equal(gpu_ctx.render(sc.frame(W, H, y0=y0, y1=y1)), full[y0:y1]), (y0, y1)
    for n in (3,):
        for p_ in range(n):
            part = torch.zeros((rt.part_rows(H, n, p_), W), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            gpu_ctx.render_device(f, part.data_ptr(), n_parts=n, part=p_)
            gpu_ctx.sync()
            assert np.array_equal(part.cpu().numpy().view(np.uint32), full[rt.part_row_indices(H, n, p_)]), (n, p_)
    sc2 = rt.make_scene(cubes, spheres + [rt.Sphere(108, V(0.0, 1.0, 14.0), 1.0, 0), rt.Sphere(109, V(-40.0, 25.0, -90.0), 9.0, 1 % len(m))], m,
                        list(s.lights), sc.camera, s.ambient_light, s.sky_0rgb, s.atlas)
    f2, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc2, 200, 120)
    assert_parity(gfb, gpl, ofb, opl, "spheres that cannot be projected")
    assert gpu_ctx.count_rays(f2).uniq_dict() == ocnt.uniq_dict()
    gpu_ctx.upload(sc)
    # a camera that turns: stamps of earlier frames must not leak into later ones (frame ids), with and without spheres in view
    cam = sc.camera
    for k in range(6):
        cam = rt.rotate_cam(cam, 1.1, 0.05)
        fk = sc.frame(160, 96, camera=cam)
        gk, _ = gpu_ctx.render(fk, debug=True)
        ok, _, _, _ = oracle_mod.OracleScene(sc).render(fk, debug=True, reuse_secondary=True)
        assert np.array_equal(gk, ok) and np.array_equal(gpu_ctx.render(fk), ok), k
    # an eye beyond the bound the spheres' growth was derived for (2^20): no sphere is classified, the frame is the oracle's
    for eye in ((0.0, 0.0, 3.0e6), (40.0, 5.0, 60.0)):
        fk = sc.frame(96, 64, camera=_cam(eye, (0.0, 0.0, 0.0)))
        gk, _ = gpu_ctx.render(fk, debug=True)
        ok, _, _, _ = oracle_mod.OracleScene(sc).render(fk, debug=True, reuse_secondary=True)
        assert np.array_equal(gk, ok) and np.array_equal(gpu_ctx.render(fk), ok), eye


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_scenes(gpu_ctx, oracle_mod, default_scene, seed):
    """Fuzz (parity.random_scene); tests/fuzz_soak.py runs the same generator over hundreds of seeds."""
    sc, label = random_scene(rt, default_scene, seed)
    f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, 96, 64)
    assert_parity(gfb, gpl, ofb, opl, label)
    assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()


def test_rccl_gather_on_the_library_stream(gpu_ctx, default_scene):
    """The collective of the N-GPU path as RCCL sees it, at world size 1 (one GPU here): dist.gather called
    with the context's stream current, receiving into the views FrameGather hands it, ordered behind the
    render on that stream with no host wait in between."""
    import socket
    import torch
    import torch.distributed as dist
    from raytracing_amd import multigpu
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gpu_ctx.upload(default_scene)
        w, h = 200, 120
        f = default_scene.frame(w, h)
        ext = torch.cuda.ExternalStream(gpu_ctx.stream(), device=dev)
        with torch.cuda.stream(ext):
            fg = multigpu.FrameGather(h, w, 1, 0, dev, depth=1)
            recv = torch.zeros((1, fg.max_rows, w), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        y0, y1 = gpu_ctx.row_extent(f)
        a, b, fa, fb = multigpu.message_window(h, 1, (y0, y1))
        gpu_ctx.render_device(f, fg.local.data_ptr())
        with torch.cuda.stream(ext):
            dist.gather(fg.local[a:b], [recv[0, : b - a]], dst=0)
        ext.synchronize()
        want = gpu_ctx.render(f)
        got = recv[0, : b - a].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want[a:b])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,y0,y1", [(328, 200, 0, None), (1080, 720, 0, None), (328, 200, 1, 77), (328, 200, 2, 200), (36, 9, 0, None), (8, 2, 0, None)])
def test_fill_paths_agree(gpu_ctx, default_scene, w, h, y0, y1):
    """Sky tiles painted by build_queue (16-byte stores; aligned buffer, width % 4 == 0) == the runtime's
    fill ahead of the launches (RT_TUNE_FILL_MEMSET forces it) == a buffer the stores do not fit (offset by
    one pixel), for whole frames, bands and the parts of a 3-rank job."""
    import torch
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(w, h, y0=y0, y1=y1)
    rows = f.y1 - f.y0
    dev = torch.device("cuda", 0)
    for n_parts, part in ((1, 0), (3, 0), (3, 2)):
        bufs = []
        for mode in ("paint", "memset", "misaligned"):
            raw = torch.full((rows * w + 8,), 0x55AA55, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            off = 1 if mode == "misaligned" else 0
            gpu_ctx.set_tuning(rt.RT_TUNE_FILL_MEMSET, 1 if mode == "memset" else 0)
            try:
                gpu_ctx.render_device(f, raw.data_ptr() + 4 * off, n_parts=n_parts, part=part)
                gpu_ctx.sync()
            finally:
                gpu_ctx.set_tuning(rt.RT_TUNE_FILL_MEMSET, 0)
            got = raw.cpu().numpy()
            assert (got[:off] == 0x55AA55).all()
            bufs.append(got[off:])
        from raytracing_amd import multigpu
        n = multigpu.part_rows(rows, n_parts, part) * w
        assert np.array_equal(bufs[0][:n], bufs[1][:n]) and np.array_equal(bufs[0][:n], bufs[2][:n]), (n_parts, part)
        # nothing is written past the part's rows
        assert (bufs[0][n:] == 0x55AA55).all() and (bufs[1][n:] == 0x55AA55).all() and (bufs[2][n:n + 7] == 0x55AA55).all()


def test_grid_limit_changes_no_pixel(gpu_ctx, default_scene):
    """rt_set_grid_limit is a tuning knob: any limit (one workgroup striding over the whole queue included)
    renders the same frame and the same counters; 0 restores the default; absurd limits are refused."""
    gpu_ctx.upload(default_scene)
    f = default_scene.frame(328, 200)
    want, want_planes = gpu_ctx.render(f, debug=True)
    want_cnt = gpu_ctx.count_rays(f).uniq_dict()
    try:
        for limit in (1, 7, 64, 12288, 0):
            gpu_ctx.set_grid_limit(limit)
            got, planes = gpu_ctx.render(f, debug=True)
            assert np.array_equal(got, want), limit
            for k in want_planes:
                assert np.array_equal(planes[k], want_planes[k]), (limit, k)
            assert np.array_equal(gpu_ctx.render(f), want), limit
            assert gpu_ctx.count_rays(f).uniq_dict() == want_cnt, limit
        with pytest.raises(rt.RtError):
            gpu_ctx.set_grid_limit(1 << 30)
    finally:
        gpu_ctx.set_grid_limit(0)


def test_specular_term_proved_black_early(gpu_ctx, oracle_mod, default_scene):
    """The kernel skips the reflection normalisation and powf when 255 * albedo.1 * cos^specular < 1 can be
    seen from the unnormalised reflection vector.  Materials and lights around that threshold: albedo.1
    from 'always black' (255 a < 1) to far above 1, exponents below, at and above 1, zero, negative, huge
    and NaN, dim and bright lights placed so that highlights fall on cube faces and on spheres."""
    s = default_scene
    V = rt.Vec3
    base = list(s.materials)
    combos = [(0.003, 1.0), (0.0039, 40.0), (0.05, 1.0), (0.05, 0.25), (1.0, 3.0), (8.0, 1.0), (0.5, 150.0), (0.05, 0.0),
              (0.05, -2.0), (0.3, 1e30), (float("nan"), 2.0), (0.2, float("nan")), (-0.5, 2.0), (float("inf"), 5.0)]
    mats = []
    for k, (a1, sp) in enumerate(combos):
        m = rt.Material.from_buffer_copy(base[k % len(base)])
        m.albedo1, m.specular = a1, sp
        m.reflectivity = 0.25 if k % 4 == 0 else 0.0
        m.transparency = 0.0
        mats.append(m)
    cubes = [rt.Cube(k, V(-6.0 + 1.5 * (k % 8), -1.5 + 1.5 * (k // 8), 0.0), 1.5, k % len(mats)) for k in range(16)]
    spheres = [rt.Sphere(100 + k, V(-5.0 + 2.5 * k, 2.5, 1.0), 0.9, (k + 3) % len(mats)) for k in range(5)]
    lights = [rt.Light(V(0.5, 1.0, 9.0), 0xFFFFFF, 1.3), rt.Light(V(-7.0, 6.0, 4.0), 0x0A40FF, 0.8), rt.Light(V(6.0, -2.0, 7.0), 0x010203, 1.0)]
    for eye in ((0.0, 0.5, 10.0), (7.0, 4.0, 6.0)):
        sc = rt.make_scene(cubes, spheres, mats, lights, _cam(eye, (0.0, 0.0, 0.0)), s.ambient_light, s.sky_0rgb, s.atlas)
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, 320, 200)
        rep = assert_parity(gfb, gpl, ofb, opl, f"specular thresholds, eye {eye}")
        assert rep["exact_fraction"] == 1.0, rep


@pytest.mark.parametrize("force", [0, 1])
def test_dark_material_kernel(tuned_ctx, oracle_mod, default_scene, water_scene, force):
    """Materials whose direct light is exactly black (1 - reflectivity - transparency < 1/255, e.g. the
    reference's water) cast no shadow rays in the plain kernel's DARK variant, which rt_scene_upload picks
    per scene; RT_TUNE_DARK_KERNEL forces either variant for any scene.  k_ds around the threshold, zero,
    negative, NaN; mixed tiles (dark and lit materials side by side); the plain frame must equal the
    oracle and the debug variant's frame (which casts every shadow ray)."""
    gpu_ctx = tuned_ctx
    gpu_ctx.set_tuning(rt.RT_TUNE_DARK_KERNEL, force)
    s = default_scene
    V = rt.Vec3
    base = list(s.materials)
    ks = [(0.1, 0.9), (0.5, 0.4962), (0.5, 0.4960), (0.3, 0.7), (0.6, 0.6), (float("nan"), 0.2), (0.0, 1.0), (0.0, 0.0), (0.25, 0.25)]
    mats = []
    for k, (r, t) in enumerate(ks):
        m = rt.Material.from_buffer_copy(base[k % len(base)])
        m.reflectivity, m.transparency = r, t
        mats.append(m)
    cubes = [rt.Cube(k, V(-5.25 + 1.5 * (k % 8), -0.75 + 1.5 * (k // 8), 0.0), 1.5, k % len(mats)) for k in range(24)]
    spheres = [rt.Sphere(100 + k, V(-4.0 + 2.0 * k, 3.5, 1.5), 0.8, (2 * k) % len(mats)) for k in range(5)]
    custom = rt.make_scene(cubes, spheres, mats, list(s.lights), _cam((1.0, 2.0, 11.0), (0.0, 0.5, 0.0)), s.ambient_light, s.sky_0rgb, s.atlas)
    for name, sc in (("custom", custom), ("water", water_scene), ("default", s)):
        f, gfb, gpl, ofb, opl, ocnt = _both(gpu_ctx, oracle_mod, sc, 240, 160)
        rep = assert_parity(gfb, gpl, ofb, opl, f"dark-material kernel forced={force}, {name}")
        assert rep["exact_fraction"] == 1.0, (name, rep)
        assert np.array_equal(gpu_ctx.render(f), gfb), name  # plain variant == debug variant
        assert gpu_ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict()
