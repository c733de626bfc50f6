"""Python face of the host side: scenes built by the C++ host mirror, rendered through the C-ABI.

Names follow the reference crate (Model/scene, Camera, Light, Cube, Sphere, Framebuffer, render).
Everything that produces pixels calls rt_render / rt_render_device in librtx950.so.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _ffi
from ._ffi import (Camera, Counters, Cube, DebugPlanes, ExecCounters, Frame, Light, Material, Sphere, Vec3, RT_MAX_DEPTH,
                   RT_NO_HIT, RT_OK)

DAY_COLOR = 0x87CEEB  # reference src/render.rs:231
NIGHT_COLOR = 0x030305  # reference src/render.rs:232
DEFAULT_MATERIAL_SEED = 0xC0FFEE
SYNTHETIC_SEED = 0x5EED5EED


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtx950 error {code}: {message}")
        self.code = code


@dataclass
class Scene:
    """`Model` (reference src/lib.rs:42-52) lowered to the C-ABI arrays."""
    cubes: C.Array
    spheres: C.Array
    materials: C.Array
    lights: C.Array  # data.lights ++ [top_light], the order of reference src/render.rs:241-243
    camera: Camera
    ambient_light: float
    sky_0rgb: int
    atlas: np.ndarray  # uint8 [n_tex, tex_h, tex_w, 3]
    tex_w: int = 48
    tex_h: int = 64
    sprite: int = 16
    name: str = "scene"
    _model: Optional[int] = field(default=None, repr=False)
    _owns_model: bool = field(default=False, repr=False)

    def __del__(self):
        # the C++ host model behind a scene built by default_scene()/synthetic_scene()
        m, self._model = getattr(self, "_model", None), None
        if m and getattr(self, "_owns_model", False):
            try:
                _ffi.host().rth_model_free(m)
            except Exception:
                pass

    @property
    def n_cubes(self):
        return len(self.cubes)

    @property
    def n_spheres(self):
        return len(self.spheres)

    def frame(self, width, height, y0=0, y1=None, max_depth=RT_MAX_DEPTH, camera=None):
        f = Frame()
        f.camera = camera if camera is not None else self.camera
        f.width, f.height = width, height
        f.y0, f.y1 = y0, (height if y1 is None else y1)
        f.lights = C.cast(self.lights, C.POINTER(Light))
        f.n_lights = len(self.lights)
        f.ambient_light = self.ambient_light
        f.sky_0rgb = self.sky_0rgb
        f.max_depth = max_depth
        return f


def make_scene(cubes, spheres, materials, lights, camera, ambient_light, sky_0rgb, atlas, sprite=16, name="custom"):
    """Build a Scene from Python lists of _ffi structs (used by tests for hand-made scenes)."""
    atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
    assert atlas.ndim == 4 and atlas.shape[3] == 3
    return Scene((Cube * len(cubes))(*cubes), (Sphere * len(spheres))(*spheres), (Material * len(materials))(*materials),
                 (Light * len(lights))(*lights), camera, float(ambient_light), int(sky_0rgb), atlas, atlas.shape[2],
                 atlas.shape[1], sprite, name)


def _scene_from_model(h, handle, name, own=False):
    """Lower a host model to a Scene; own=True hands the model to the Scene (freed with it)."""
    n = [C.c_uint32() for _ in range(4)]
    if h.rth_model_counts(handle, *[C.byref(x) for x in n]) != 0:
        raise RuntimeError(h.rth_last_error().decode())
    nc, ns, nm, nl = [x.value for x in n]
    cubes, spheres = (Cube * nc)(), (Sphere * ns)()
    mats, lights = (Material * nm)(), (Light * nl)()
    cam, amb, sky = Camera(), C.c_float(), C.c_uint32()
    if h.rth_model_lower(handle, cubes, spheres, mats, lights, C.byref(cam), C.byref(amb), C.byref(sky)) != 0:
        raise RuntimeError(h.rth_model_lower.__name__ + ": " + h.rth_last_error().decode())
    tw, th, sp = C.c_uint32(), C.c_uint32(), C.c_uint32()
    nt = C.c_uint32()
    if h.rth_model_atlas(handle, None, 0, C.byref(nt), C.byref(tw), C.byref(th), C.byref(sp)) != 0:
        raise RuntimeError("rth_model_atlas: " + h.rth_last_error().decode())
    atlas = np.zeros((nt.value, th.value, tw.value, 3), dtype=np.uint8)
    if h.rth_model_atlas(handle, atlas.ctypes.data_as(C.c_void_p), atlas.nbytes, C.byref(nt), C.byref(tw), C.byref(th), C.byref(sp)) != 0:
        raise RuntimeError("rth_model_atlas: " + h.rth_last_error().decode())
    sc = Scene(cubes, spheres, mats, lights, cam, amb.value, sky.value, atlas, tw.value, th.value, sp.value, name, handle)
    sc._owns_model = own
    return sc


def default_scene(material_seed=DEFAULT_MATERIAL_SEED, with_water=False, assets=None):
    """`init()` of the reference (src/main.rs:137-260) with the platform materials drawn from
    PCG32(material_seed) instead of the unseeded thread_rng.  `assets` is either the baked atlas
    blob (default) or an asset directory prefix ending in '/', as the reference's CLI argument."""
    h = _ffi.host()
    path = (assets or _ffi.BAKED_ATLAS_PATH).encode()
    handle = h.rth_model_default(path, material_seed, 1 if with_water else 0)
    if not handle:
        raise RuntimeError("rth_model_default: " + h.rth_last_error().decode())
    return _scene_from_model(h, handle, "minecraft+water" if with_water else "minecraft", own=True)


def synthetic_scene(n_cubes=10000, n_spheres=256, seed=SYNTHETIC_SEED, assets=None):
    """Seeded synthetic voxel world (BASELINE.json config 4; spec in raytracing_amd/host/scene.hpp)."""
    h = _ffi.host()
    path = (assets or _ffi.BAKED_ATLAS_PATH).encode()
    handle = h.rth_model_synthetic(path, n_cubes, n_spheres, seed)
    if not handle:
        raise RuntimeError("rth_model_synthetic: " + h.rth_last_error().decode())
    return _scene_from_model(h, handle, f"synthetic_{n_cubes}c_{n_spheres}s", own=True)


def rotate_cam(camera, delta_yaw, delta_pitch):
    """Camera::rotate_cam (reference src/camera.rs:61-86); returns a new Camera."""
    out = Camera.from_buffer_copy(camera)
    _ffi.host().rth_camera_rotate(C.byref(out), delta_yaw, delta_pitch)
    return out


def bmp_encode(buffer_0rgb, width, height):
    """write_bmp_file's byte image (reference src/bmp.rs:19-113) for a 0x00RRGGBB buffer."""
    h = _ffi.host()
    buf = np.ascontiguousarray(buffer_0rgb, dtype=np.uint32).reshape(-1)
    assert buf.size == width * height
    n = h.rth_bmp_encode(buf.ctypes.data_as(C.c_void_p), width, height, None, 0)
    out = np.zeros(n, dtype=np.uint8)
    h.rth_bmp_encode(buf.ctypes.data_as(C.c_void_p), width, height, out.ctypes.data_as(C.c_void_p), n)
    return out.tobytes()


class Context:
    """One rt_ctx: a GPU, its stream, the device-resident scene and framebuffer."""

    def __init__(self, device=0):
        self._lib = _ffi.lib()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_ctx_create(device, C.byref(self._ctx))
        if rc != RT_OK:
            raise RtError(rc, self._lib.rt_last_error(None).decode())
        self.device = device
        self.scene = None

    def close(self):
        if self._ctx:
            self._lib.rt_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def upload(self, scene: Scene):
        self._check(self._lib.rt_scene_upload(
            self._ctx, C.cast(scene.cubes, C.POINTER(Cube)), len(scene.cubes), C.cast(scene.spheres, C.POINTER(Sphere)),
            len(scene.spheres), C.cast(scene.materials, C.POINTER(Material)), len(scene.materials),
            scene.atlas.ctypes.data_as(C.c_void_p), scene.atlas.shape[0], scene.tex_w, scene.tex_h, scene.sprite))
        self.scene = scene

    def render(self, frame: Frame, debug=False, out=None):
        """rt_render: returns the band as uint32 [rows, width] (+ dict of debug planes if debug).
        `out`: a C-contiguous uint32 [rows, width] array to render into (e.g. PinnedFrame.array)."""
        rows = frame.y1 - frame.y0
        if out is None:
            out = np.empty((rows, frame.width), dtype=np.uint32)
        assert out.dtype == np.uint32 and out.shape == (rows, frame.width) and out.flags.c_contiguous
        planes = None
        dbg = None
        if debug:
            planes = {k: np.empty((rows, frame.width), dtype=np.uint32) for k in ("hit_cube", "hit_sphere", "tree_hash")}
            dbg = DebugPlanes(*[planes[k].ctypes.data for k in ("hit_cube", "hit_sphere", "tree_hash")])
        self._check(self._lib.rt_render(self._ctx, C.byref(frame), out.ctypes.data_as(C.c_void_p),
                                        C.byref(dbg) if dbg is not None else None))
        return (out, planes) if debug else out

    def render_device(self, frame: Frame, d_out_ptr: int, d_hit_cube=0, d_hit_sphere=0, d_tree_hash=0, n_parts=1, part=0):
        """rt_render_device[_part]: asynchronous render into device memory (raw pointers)."""
        dbg = None
        if d_hit_cube or d_hit_sphere or d_tree_hash:
            dbg = DebugPlanes(d_hit_cube or None, d_hit_sphere or None, d_tree_hash or None)
        self._check(self._lib.rt_render_device_part(self._ctx, C.byref(frame), n_parts, part, C.c_void_p(d_out_ptr),
                                                    C.byref(dbg) if dbg is not None else None))

    def render_device_noalloc(self, frame: Frame, d_out_ptr: int):
        """rt_render_device_noalloc: the allocation-free variant (fixed work buffer per GPU)."""
        self._check(self._lib.rt_render_device_noalloc(self._ctx, C.byref(frame), C.c_void_p(d_out_ptr)))

    def sync(self):
        self._check(self._lib.rt_sync(self._ctx))

    def count_rays(self, frame: Frame, n_parts=1, part=0) -> Counters:
        c = Counters()
        self._check(self._lib.rt_count_rays_part(self._ctx, C.byref(frame), n_parts, part, C.byref(c)))
        return c

    def count_executed(self, frame: Frame, n_parts=1, part=0):
        """rt_count_executed: (Counters, ExecCounters) of one counting launch."""
        c, e = Counters(), ExecCounters()
        self._check(self._lib.rt_count_executed(self._ctx, C.byref(frame), n_parts, part, C.byref(c), C.byref(e)))
        return c, e

    def set_tuning(self, key, value):
        """rt_set_tuning (keys: _ffi.RT_TUNE_*)."""
        self._check(self._lib.rt_set_tuning(self._ctx, int(key), float(value)))

    # the library's defaults; a test session may override entries (tests/conftest.py) to soak a path
    DEFAULT_TUNING = {_ffi.RT_TUNE_COHERENT_SCALE: 1.0, _ffi.RT_TUNE_SHADOW_LIST: -1, _ffi.RT_TUNE_DARK_KERNEL: -1,
                      _ffi.RT_TUNE_FILL_MEMSET: 0, _ffi.RT_TUNE_BVH_DEPTH_LIMIT: 28, _ffi.RT_TUNE_HEAVY_CLASS: -1, _ffi.RT_TUNE_HEAVY_SYNC: -1,
                      _ffi.RT_TUNE_HEAVY_SPIN_LIMIT: 0, _ffi.RT_TUNE_WAVEFRONT: -1, _ffi.RT_TUNE_WAVEFRONT_CAP: 0, _ffi.RT_TUNE_CULL_REFINE: -1}

    def reset_tuning(self):
        for key, value in self.DEFAULT_TUNING.items():
            self.set_tuning(key, value)

    def selftest_division(self, n_cases, seed=1):
        """(mismatches, cases) of the device's shared-reciprocal division against its IEEE division."""
        bad, n = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rt_selftest_division(self._ctx, seed, n_cases, C.byref(bad), C.byref(n)))
        return bad.value, n.value

    def enable_timing(self, on=True):
        """False/0: off; True/1: every launch; k > 1: every k-th launch."""
        self._check(self._lib.rt_enable_timing(self._ctx, int(on)))

    def set_grid_limit(self, max_workgroups=0):
        """Upper limit of the render kernel's grid; 0 restores the default (rt_set_grid_limit)."""
        self._check(self._lib.rt_set_grid_limit(self._ctx, int(max_workgroups)))

    def kernel_time_ms(self):
        avg, n = C.c_double(), C.c_uint32()
        self._check(self._lib.rt_kernel_time_ms(self._ctx, C.byref(avg), C.byref(n)))
        return avg.value, n.value

    def row_extent(self, frame: Frame):
        """rt_frame_row_extent: (y_first, y_end); rows of the band outside it are sky (host-side bound)."""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rt_frame_row_extent(self._ctx, C.byref(frame), C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_error(self) -> str:
        """rt_last_error of this context (also set, as a warning, by a call that recovered by itself)."""
        return (self._lib.rt_last_error(self._ctx) or b"").decode()

    def debug_status(self) -> int:
        """rt_debug_status: 0 unless a heavy tile's workgroup gave up waiting (never expected)."""
        w = C.c_uint32()
        self._check(self._lib.rt_debug_status(self._ctx, C.byref(w)))
        return w.value

    def tile_costs(self):
        """rt_debug_tile_costs: cycles per 8x8 tile of the last launch (uint32 array, launch tile grid order)."""
        n = C.c_size_t()
        self._check(self._lib.rt_debug_tile_costs(self._ctx, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        self._check(self._lib.rt_debug_tile_costs(self._ctx, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out

    def wavefront_items(self, level):
        """rt_debug_wavefront_items: uint32 [items, 8] records of the last wavefront frame at `level`."""
        n = C.c_size_t()
        self._check(self._lib.rt_debug_wavefront_items(self._ctx, level, None, 0, C.byref(n)))
        out = np.zeros((n.value, 8), dtype=np.uint32)
        if n.value:
            self._check(self._lib.rt_debug_wavefront_items(self._ctx, level, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def wavefront_redo_tiles(self) -> int:
        """Tiles of the last wavefront frame whose items did not fit and that render_kernel traced again."""
        n = C.c_size_t()
        self._check(self._lib.rt_debug_wavefront_items(self._ctx, 4, None, 0, C.byref(n)))
        return n.value

    def kernel_name(self) -> str:
        """rt_kernel_name: the render kernel instantiation the uploaded scene selects."""
        return (self._lib.rt_kernel_name(self._ctx) or b"").decode()

    def stream(self) -> int:
        s = C.c_void_p()
        self._check(self._lib.rt_get_stream(self._ctx, C.byref(s)))
        return s.value or 0


def render_multi(ctxs, frame: Frame, d_frame_ptr: int):
    """rt_render_multi: one frame over several contexts (GPUs), gathered at d_frame_ptr on ctxs[0]'s GPU.
    Returns the bytes that moved between contexts.  Asynchronous: ctxs[0].sync() waits for the frame."""
    arr = (C.c_void_p * len(ctxs))(*[c._ctx for c in ctxs])
    moved = C.c_uint64()
    rc = _ffi.lib().rt_render_multi(arr, len(ctxs), C.byref(frame), C.c_void_p(d_frame_ptr), C.byref(moved))
    if rc != RT_OK:
        raise RtError(rc, _ffi.lib().rt_last_error(ctxs[0]._ctx).decode())
    return moved.value


class PinnedFrame:
    """A frame in page-locked host memory (rt_host_alloc) as a numpy array: the fast target of rt_render."""

    def __init__(self, rows, width):
        self._ptr = C.c_void_p()
        rc = _ffi.lib().rt_host_alloc(rows * width * 4, C.byref(self._ptr))
        if rc != RT_OK:
            raise RtError(rc, _ffi.lib().rt_last_error(None).decode())
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, C.POINTER(C.c_uint32)), shape=(rows, width))

    def close(self):
        if self._ptr:
            self.array = None
            _ffi.lib().rt_host_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def part_rows(rows, n_parts, part):
    """Rows held by interleaved part `part` of `n_parts` (16-row blocks), see rt_render_device_part."""
    return int(_ffi.lib().rt_part_rows(rows, n_parts, part))


def part_row_indices(rows, n_parts, part, block=16):
    """Band-relative row numbers of that part, in the order of its compact buffer."""
    idx = []
    for b in range(part, (rows + block - 1) // block, n_parts):
        idx.extend(range(b * block, min(rows, (b + 1) * block)))
    return np.asarray(idx, dtype=np.int64)


def render(scene: Scene, width, height, device=0, debug=False):
    """render(framebuffer, data) of the reference (src/render.rs:234) for a whole frame."""
    with Context(device) as ctx:
        ctx.upload(scene)
        return ctx.render(scene.frame(width, height), debug=debug)
