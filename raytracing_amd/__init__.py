"""raytracing_amd — MI355X-native renderer for the ElrohirGT/Raytracing per-pixel ray-cast path.

Layout: csrc/ (HIP kernels + the C-ABI of include/rtx950.h), host/ (C++ mirror of the reference's
scene/API surface), api.py (Python face used by tests and bench).  Importing this package does not
load the shared libraries; the first call that needs them does, and fails loudly if they are absent.
"""
from . import _ffi  # noqa: F401
from .api import (Context, RtError, Scene, bmp_encode, default_scene, make_scene, render, rotate_cam,  # noqa: F401
                  synthetic_scene, part_rows, part_row_indices, render_multi, PinnedFrame, DAY_COLOR, NIGHT_COLOR, DEFAULT_MATERIAL_SEED, SYNTHETIC_SEED)
from ._ffi import (Camera, Counters, Cube, DebugPlanes, ExecCounters, Frame, Light, Material, Sphere, Vec3,  # noqa: F401
                   RT_MAX_DEPTH, RT_MAX_LIGHTS, RT_NO_HIT, RT_TEXTURE_NONE, RT_TUNE_COHERENT_SCALE, RT_TUNE_SHADOW_LIST,
                   RT_TUNE_DARK_KERNEL, RT_TUNE_FILL_MEMSET, RT_TUNE_BVH_DEPTH_LIMIT, RT_TUNE_HEAVY_CLASS, RT_TUNE_HEAVY_SYNC, RT_TUNE_HEAVY_SPIN_LIMIT, RT_TUNE_WAVEFRONT, RT_TUNE_WAVEFRONT_CAP, RT_TUNE_CULL_REFINE,
                   RT_NOALLOC_MAX_TILES)

__all__ = [n for n in dir() if not n.startswith("_")]
