#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (run in the authoring container).

The reference holds no golden images and cannot be built here (no Rust toolchain), so these vectors are
outputs of oracle/rt_oracle.c — "parity unpinned" fixtures: they pin the oracle against drift and let
GPU tests compare against committed data.  Scene = reference default diorama, materials PCG32(0xC0FFEE).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
import raytracing_amd as rt  # noqa: E402


def dump(name, scene, w, h):
    fb, planes, cnt, _ = orc.OracleScene(scene).render(scene.frame(w, h), debug=True, reuse_secondary=True)
    _, _, cref, _ = orc.OracleScene(scene).render(scene.frame(w, h), reuse_secondary=False)
    bmp = rt.bmp_encode(fb, w, h)
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, fb=fb, bmp_sha256=hashlib.sha256(bmp).hexdigest(),
                        uniq=np.array(list(cnt.uniq_dict().values()), dtype=np.uint64),
                        ref=np.array(list(cref.uniq_dict().values()), dtype=np.uint64), **planes)
    print(path, os.path.getsize(path + ("" if path.endswith(".npz") else ".npz")), cnt.uniq_dict())


if __name__ == "__main__":
    dump("minecraft_270x180.npz", rt.default_scene(), 270, 180)
    dump("minecraft_water_240x135.npz", rt.default_scene(with_water=True), 240, 135)
    dump("synthetic_600c_24s_160x90.npz", rt.synthetic_scene(n_cubes=600, n_spheres=24), 160, 90)
