"""CPU tests of the boundary and the C++ host mirror (no GPU): the C-ABI library loads and exports every
symbol include/rtx950.h declares, calls fail loudly without a device, and the host-side scene/camera/
framebuffer/PNG code reproduces the reference's data."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtx950.h")).read()
    declared = set(re.findall(r"^\s*(?:int|uint32_t|const char\*)\s+(rt_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert "rt_kernel_name" in declared and "rt_render_device_noalloc" in declared and "rt_render_multi" in declared
    assert len(declared) >= 15
    lib = C.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"librtx950.so does not export {name}"
    assert declared == set(_ffi.ABI_SYMBOLS), "ctypes table and header disagree"
    assert _ffi.lib().rt_abi_version() == 2


def test_struct_layouts_match_the_header():
    assert C.sizeof(_ffi.Cube) == 24 and C.sizeof(_ffi.Sphere) == 24 and C.sizeof(_ffi.Material) == 32
    assert C.sizeof(_ffi.Light) == 20 and C.sizeof(_ffi.Camera) == 36 and C.sizeof(_ffi.Counters) == 80
    assert C.sizeof(_ffi.Frame) == 80 and _ffi.Frame.lights.offset == 56


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Context(0)
    assert e.value.code == _ffi.RT_ERR_NO_DEVICE and "no CPU path" in str(e.value)
    lib = _ffi.lib()
    assert lib.rt_render(None, None, None, None) == _ffi.RT_ERR_INVALID_ARG
    assert b"ctx is null" in lib.rt_last_error(None)


def test_default_scene_matches_reference_construction(default_scene):
    s = default_scene
    assert s.n_cubes == 97 and s.n_spheres == 0 and len(s.lights) == 2
    c = s.cubes
    # platform (src/main.rs:150-156,318-357): idx = 8*(k+4)+(j+4), centre (1.5j,-1.5,1.5k), id = j+4
    for k in range(-4, 4):
        for j in range(-4, 4):
            q = c[8 * (k + 4) + (j + 4)]
            assert (q.center.x, q.center.y, q.center.z) == (1.5 * j, -1.5, 1.5 * k) and q.id == j + 4 and q.size == 1.5
    # obsidian frame, portal wall, loose cubes: ids 65.., 130.., 212 (src/main.rs:157-218)
    assert [q.id for q in c[64:72]] == [65, 66, 67, 68] * 2 and [q.id for q in c[72:80]] == [65, 66, 67, 68] * 2
    assert [(q.center.x, q.center.y) for q in c[64:68]] == [(-3.0, 0.0), (-1.5, 0.0), (0.0, 0.0), (1.5, 0.0)]
    assert [(q.center.x, q.center.y) for q in c[72:76]] == [(-3.0, 1.5), (-3.0, 3.0), (-3.0, 4.5), (-3.0, 6.0)]
    assert [(q.center.x, q.center.y) for q in c[76:80]] == [(1.5, 1.5), (1.5, 3.0), (1.5, 4.5), (1.5, 6.0)]
    assert [q.id for q in c[80:92]] == [130, 131, 132] * 4 and all(q.center.z == -4.5 for q in c[64:92])
    assert [(q.center.x, q.center.y) for q in c[80:83]] == [(-1.5, 0.0), (0.0, 0.0), (1.5, 0.0)]
    assert [q.id for q in c[92:97]] == [212] * 5
    assert [(q.center.x, q.center.y, q.center.z) for q in c[92:97]] == [(0, 0, 0), (0, 0, -1.5), (-1.5, 0, -1.5), (-1.5, 0, -1.5), (1.5, 0, 1.5)]
    # lights: portal light then top light (src/main.rs:445-450,233-237; order of src/render.rs:241-243)
    l0, l1 = s.lights
    assert (l0.position.x, l0.position.y, l0.position.z, l0.color_0rgb, l0.intensity) == (0.0, 2.25, -3.5, 0x361B6F, 0.25)
    assert (l1.position.x, l1.position.y, l1.position.z, l1.color_0rgb, l1.intensity) == (0.0, 20.0, 0.0, 0xFFFFFF, 1.0)
    assert s.sky_0rgb == 0x87CEEB and s.ambient_light == np.float32(0.15)
    assert (s.camera.eye.x, s.camera.eye.y, s.camera.eye.z) == (0, 0, 10)
    mats = {(m.texture, m.specular, m.reflectivity, m.transparency) for m in s.materials}
    assert (5, 150.0, 0.0, np.float32(0.4)) in mats and (8, 2.0, np.float32(0.1), 0.0) in mats  # PORTAL, GOLD


def test_pinned_materials_are_reproducible_and_seeded():
    a, b, c = rt.default_scene(), rt.default_scene(), rt.default_scene(material_seed=1)
    ma = [a.materials[q.material].texture for q in a.cubes[:64]]
    assert ma == [b.materials[q.material].texture for q in b.cubes[:64]]
    assert ma != [c.materials[q.material].texture for q in c.cubes[:64]]
    assert set(ma) <= {6, 1, 7, 8}  # NETHERRACK, STONE, MAGMA, GOLD (src/main.rs:338-346)
    assert ma.count(6) > 25  # ~60 %


def test_water_variant_and_synthetic_world():
    w = rt.default_scene(with_water=True)
    assert w.n_cubes == 101 and all(w.materials[q.material].texture == 3 for q in w.cubes[97:])
    assert [(q.center.x, q.center.y, q.center.z) for q in w.cubes[97:]] == [(-1.5, 3, -1.5), (0, 3, -1.5), (-1.5, 3, 0), (0, 3, 0)]
    s = rt.synthetic_scene(n_cubes=1000, n_spheres=16)
    assert s.n_cubes == 1000 and s.n_spheres == 16
    assert [q.id for q in s.cubes[:5]] == [0, 1, 2, 3, 4] and s.spheres[0].id == 1000
    assert all(q.size == 1.5 for q in s.cubes) and all(0.5 <= q.radius < 2.0 for q in s.spheres)
    s2 = rt.synthetic_scene(n_cubes=1000, n_spheres=16)
    assert bytes(s.cubes) == bytes(s2.cubes) and bytes(s.spheres) == bytes(s2.spheres)


def test_png_loader_reproduces_the_baked_atlas():
    """raytracing_amd/host/png_loader.hpp vs the PIL-baked blob (tools/bake_textures.py), per-texture sha256
    as recorded in SURVEY.md section 8(c)."""
    h = _ffi.host()
    names = ["dirt", "stone", "moss", "water", "obsidian", "portal", "netherrack", "magma", "gold"]
    blob = open(_ffi.BAKED_ATLAS_PATH, "rb").read()
    pins = dict(line.split()[::-1] for line in open(_ffi.BAKED_ATLAS_PATH.replace(".bin", ".sha256")))
    for i, n in enumerate(names):
        path = os.path.join(ROOT, "tests", "golden", "imgs", n + ".png").encode()
        w, hh = C.c_uint32(), C.c_uint32()
        buf = np.zeros(48 * 64 * 3, dtype=np.uint8)
        assert h.rth_png_decode(path, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(w), C.byref(hh)) == 0, h.rth_last_error()
        assert (w.value, hh.value) == (48, 64)
        assert buf.tobytes() == blob[i * 9216:(i + 1) * 9216]
        assert hashlib.sha256(buf.tobytes()).hexdigest() == pins[n]
    # the reference's asset-dir form ("prefix ending in /", src/texture.rs:48-57) loads the same scene
    s = rt.default_scene(assets=os.path.join(ROOT, "tests", "golden", "imgs") + "/")
    assert np.array_equal(s.atlas, rt.default_scene().atlas)
    assert h.rth_png_decode(b"/nonexistent.png", None, 0, C.byref(w), C.byref(hh)) != 0


def test_framebuffer_paint_point_errors():
    h = _ffi.host()  # src/framebuffer.rs:79-109: Ok, XTooLarge, XTooSmall, YTooLarge, YTooSmall
    assert h.rth_paint_point(8, 4, 7.4, 3.4) == 0
    assert h.rth_paint_point(8, 4, 8.0, 1.0) == 1  # the i == width pixel of src/render.rs:316-322
    assert h.rth_paint_point(8, 4, -1.0, 1.0) == 2
    assert h.rth_paint_point(8, 4, 1.0, 3.6) == 3  # rounds to 4
    assert h.rth_paint_point(8, 4, 1.0, -0.5) == 4


def test_update_messages():
    import ctypes
    s = rt.default_scene()
    h = _ffi.host()
    assert h.rth_model_update(s._model, 4, 0, 0, 0, 0, 0) == 0  # TimeToNight (src/main.rs:299-314)
    n = rt.api._scene_from_model(h, s._model, "night")
    assert n.sky_0rgb == 0x030305 and n.lights[1].color_0rgb == 0 and n.lights[1].intensity == 0.5
    assert h.rth_model_update(s._model, 1, 2.0, 0, 0, 0, 0) == 0  # ZoomCamera(2.0)
    z = rt.api._scene_from_model(h, s._model, "zoom")
    assert (z.camera.eye.x, z.camera.eye.y, z.camera.eye.z) == (0, 0, 8)
    assert h.rth_model_update(s._model, 3, 0, 0, 0, 0, 0) == 0  # TimeToDay
    assert rt.api._scene_from_model(h, s._model, "day").sky_0rgb == 0x87CEEB


def test_part_rows_matches_python_partition():
    for rows in (1, 16, 17, 720, 2160):
        for n in (1, 2, 3, 8):
            for p in range(n):
                assert rt.part_rows(rows, n, p) == len(rt.part_row_indices(rows, n, p))


def test_atlases_of_different_sizes(tmp_path):
    """The reference keeps a width and height per Texture (src/texture.rs:98-103,141-169); the C ABI takes one
    for all nine.  A user asset directory whose PNGs differ in size is padded to a common size (texel
    coordinates unchanged, so every face lookup reads what the reference reads); an atlas smaller than the
    3x4-sprite cross layout is refused instead of read out of bounds."""
    from PIL import Image
    src = os.path.join(ROOT, "tests", "golden", "imgs")
    names = ["dirt", "stone", "moss", "water", "obsidian", "portal", "netherrack", "magma", "gold"]
    d = tmp_path / "assets"
    d.mkdir()
    for n in names:
        img = Image.open(os.path.join(src, n + ".png")).convert("RGBA")
        if n == "gold":  # a larger canvas, the original in its top-left corner
            big = Image.new("RGBA", (64, 80), (9, 9, 9, 255))
            big.paste(img, (0, 0))
            img = big
        img.save(d / (n + ".png"))
    s = rt.default_scene(assets=str(d) + "/")
    base = rt.default_scene()
    assert s.atlas.shape == (9, 80, 64, 3) and (s.tex_w, s.tex_h, s.sprite) == (64, 80, 16)
    assert np.array_equal(s.atlas[:, :64, :48], base.atlas)
    assert (s.atlas[:8, 64:, :] == 0).all() and (s.atlas[8, 70, 60] == 9).all()
    Image.open(os.path.join(src, "magma.png")).convert("RGBA").crop((0, 0, 40, 64)).save(d / "magma.png")
    with pytest.raises(RuntimeError) as e:
        rt.default_scene(assets=str(d) + "/")
    assert "cross layout" in str(e.value)
