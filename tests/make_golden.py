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


def dump_default_frame(name, scene, w, h):
    """Config 1 (the default diorama at main.rs's framebuffer size): the BMP's and the frame's sha256, the
    depth-0 hit map and a checksum of the ray-tree hash plane (the plane itself is incompressible)."""
    fb, planes, cnt, _ = orc.OracleScene(scene).render(scene.frame(w, h), debug=True, reuse_secondary=True)
    bmp = orc.bmp_encode(fb, w, h)
    assert bmp == rt.bmp_encode(fb, w, h)  # C oracle writer == C++ host mirror writer
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, bmp_sha256=hashlib.sha256(bmp).hexdigest(), fb_sha256=hashlib.sha256(fb.tobytes()).hexdigest(),
                        hit_cube=planes["hit_cube"], tree_hash_xor=np.uint32(np.bitwise_xor.reduce(planes["tree_hash"].ravel())),
                        uniq=np.array(list(cnt.uniq_dict().values()), dtype=np.uint64))
    print(path, os.path.getsize(path), cnt.uniq_dict())


if __name__ == "__main__":
    dump("minecraft_270x180.npz", rt.default_scene(), 270, 180)
    dump("minecraft_water_240x135.npz", rt.default_scene(with_water=True), 240, 135)
    dump("synthetic_600c_24s_160x90.npz", rt.synthetic_scene(n_cubes=600, n_spheres=24), 160, 90)
    dump("synthetic_10000c_256s_320x180.npz", rt.synthetic_scene(), 320, 180)  # BASELINE.json config 4 at its size
    dump_default_frame("minecraft_1080x720.npz", rt.default_scene(), 1080, 720)  # BASELINE.json config 1
