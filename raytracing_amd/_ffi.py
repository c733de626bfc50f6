"""ctypes declarations for include/rtx950.h (the C-ABI) and the rth_* host shim.

The structs below are field-for-field the ones in include/rtx950.h.  Loading fails loudly when the
in-tree shared libraries are missing: there is no Python or CPU fallback for rendering.
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RTX950_LIB lets the A/B harness (tools/) load an experimental build of the same C-ABI; default is the in-tree library
LIB_PATH = os.environ.get("RTX950_LIB") or os.path.join(_PKG_DIR, "librtx950.so")
HOST_LIB_PATH = os.path.join(_PKG_DIR, "librtx950_host.so")
BAKED_ATLAS_PATH = os.path.join(_PKG_DIR, "assets", "atlas_rgb8.bin")

RT_OK = 0
RT_ERR_INVALID_ARG = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_NO_SCENE = -4
RT_ERR_UNSUPPORTED = -5
RT_ERR_LIMIT = -6
RT_ERR_DEVICE_WAIT = -7
RT_MAX_DEPTH = 3
RT_MAX_LIGHTS = 16
RT_TEXTURE_NONE = -1
RT_NO_HIT = 0xFFFFFFFF
RT_EXEC_EVENTS = 12
RT_NOALLOC_MAX_TILES = 131072
RT_TUNE_COHERENT_SCALE = 1
RT_TUNE_SHADOW_LIST = 2
RT_TUNE_DARK_KERNEL = 3
RT_TUNE_FILL_MEMSET = 4
RT_TUNE_BVH_DEPTH_LIMIT = 5
RT_TUNE_HEAVY_CLASS = 6
RT_TUNE_HEAVY_SYNC = 7
RT_TUNE_HEAVY_SPIN_LIMIT = 8
RT_TUNE_WAVEFRONT = 9
RT_TUNE_WAVEFRONT_CAP = 10
RT_TUNE_CULL_REFINE = 11


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __repr__(self):
        return f"Vec3({self.x}, {self.y}, {self.z})"


class Cube(C.Structure):
    _fields_ = [("id", C.c_uint32), ("center", Vec3), ("size", C.c_float), ("material", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("id", C.c_uint32), ("center", Vec3), ("radius", C.c_float), ("material", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [
        ("diffuse_0rgb", C.c_uint32),
        ("texture", C.c_int32),
        ("specular", C.c_float),
        ("albedo0", C.c_float),
        ("albedo1", C.c_float),
        ("reflectivity", C.c_float),
        ("transparency", C.c_float),
        ("refractive_index", C.c_float),
    ]


class Light(C.Structure):
    _fields_ = [("position", Vec3), ("color_0rgb", C.c_uint32), ("intensity", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("eye", Vec3), ("center", Vec3), ("up", Vec3)]


class Frame(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("y0", C.c_uint32),
        ("y1", C.c_uint32),
        ("lights", C.POINTER(Light)),
        ("n_lights", C.c_uint32),
        ("ambient_light", C.c_float),
        ("sky_0rgb", C.c_uint32),
        ("max_depth", C.c_uint32),
    ]


class DebugPlanes(C.Structure):
    _fields_ = [("hit_cube", C.c_void_p), ("hit_sphere", C.c_void_p), ("tree_hash", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [
        ("nearest_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("cube_tests", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("ref_nearest_rays", C.c_uint64),
        ("ref_shadow_rays", C.c_uint64),
        ("ref_cube_tests", C.c_uint64),
        ("ref_sphere_tests", C.c_uint64),
        ("ref_shaded_hits", C.c_uint64),
    ]

    _BASE = ("nearest_rays", "shadow_rays", "cube_tests", "sphere_tests", "shaded_hits")

    def uniq_dict(self):
        return {k: int(getattr(self, k)) for k in self._BASE}

    def ref_dict(self):
        return {k: int(getattr(self, "ref_" + k)) for k in self._BASE}

    @property
    def ref_rays(self):
        return int(self.ref_nearest_rays + self.ref_shadow_rays)

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    @property
    def rays(self):
        return int(self.nearest_rays + self.shadow_rays)


class ExecCounters(C.Structure):
    """rt_exec_counters: what the kernels executed (per lane), see include/rtx950.h."""
    _fields_ = [(k, C.c_uint64) for k in (
        "cull_tests", "chunk_tests", "candidate_evals", "exact_evals", "bvh_steps", "sphere_evals", "queries",
        "light_iterations", "resolved_hits", "blockers", "child_setups", "primary_rays")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# every symbol include/rtx950.h declares: name -> (restype, argtypes)
ABI_SYMBOLS = {
    "rt_abi_version": (C.c_uint32, []),
    "rt_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rt_ctx_destroy": (C.c_int, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_scene_upload": (
        C.c_int,
        [C.c_void_p, C.POINTER(Cube), C.c_size_t, C.POINTER(Sphere), C.c_size_t, C.POINTER(Material), C.c_size_t,
         C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32],
    ),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.POINTER(DebugPlanes)]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.POINTER(DebugPlanes)]),
    "rt_render_device_part": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(DebugPlanes)]),
    "rt_part_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_sync": (C.c_int, [C.c_void_p]),
    "rt_count_rays": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.POINTER(Counters)]),
    "rt_count_rays_part": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_uint32, C.c_uint32, C.POINTER(Counters)]),
    "rt_kernel_time_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "rt_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_set_grid_limit": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_selftest_division": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rt_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_frame_row_extent": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_count_executed": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_uint32, C.c_uint32, C.POINTER(Counters), C.POINTER(ExecCounters)]),
    "rt_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "rt_kernel_name": (C.c_char_p, [C.c_void_p]),
    "rt_debug_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rt_debug_tile_costs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rt_debug_wavefront_items": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rt_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rt_host_free": (C.c_int, [C.c_void_p]),
    "rt_render_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_void_p]),
    "rt_render_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(Frame), C.c_void_p, C.POINTER(C.c_uint64)]),
    "rt_gather_bands": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p,
                                  C.POINTER(C.c_uint64)]),
}

HOST_SYMBOLS = {
    "rth_last_error": (C.c_char_p, []),
    "rth_model_default": (C.c_void_p, [C.c_char_p, C.c_uint64, C.c_int]),
    "rth_model_synthetic": (C.c_void_p, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "rth_model_free": (None, [C.c_void_p]),
    "rth_model_counts": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4),
    "rth_model_lower": (
        C.c_int,
        [C.c_void_p, C.POINTER(Cube), C.POINTER(Sphere), C.POINTER(Material), C.POINTER(Light), C.POINTER(Camera),
         C.POINTER(C.c_float), C.POINTER(C.c_uint32)],
    ),
    "rth_model_atlas": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 4),
    "rth_model_update": (C.c_int, [C.c_void_p, C.c_int] + [C.c_float] * 5),
    "rth_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rth_bmp_encode": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]),
    "rth_color_mul": (C.c_uint32, [C.c_uint32, C.c_float]),
    "rth_color_add": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "rth_color_sub": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "rth_camera_rotate": (None, [C.POINTER(Camera), C.c_float, C.c_float]),
    "rth_paint_point": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    "rth_png_decode": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

_lib = None
_host = None


def _bind(lib, table, path):
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"{path} does not export {name}; rebuild with __graft_entry__.build()") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    """The C-ABI library (HIP kernels inside).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no fallback renderer.")
        _lib = _bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL), ABI_SYMBOLS, LIB_PATH)
    return _lib


def host():
    """The C++ host mirror behind its flat C shim."""
    global _host
    if _host is None:
        lib()  # librtx950_host.so links against librtx950.so
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        _host = _bind(C.CDLL(HOST_LIB_PATH), HOST_SYMBOLS, HOST_LIB_PATH)
    return _host
