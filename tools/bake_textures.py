#!/usr/bin/env python3
"""Bake the reference's 9 texture PNGs into one raw RGB8 blob + copy the PNGs as test fixtures.

Run once in the authoring container (needs PIL and /root/reference):
    python tools/bake_textures.py [/root/reference/imgs]
Output:
    raytracing_amd/assets/atlas_rgb8.bin   9 x 48 x 64 x 3 bytes, `enum Textures` order
                                           (reference src/texture.rs:35-45), each image row-major
                                           RGB exactly as `Texture::new` stores it (texture.rs:151-161)
    raytracing_amd/assets/atlas_rgb8.sha256  per-texture sha256 of the 9,216 RGB bytes
    tests/golden/imgs/*.png                the source PNGs (data fixtures for the PNG decoder test)
PIL's Image.open(f).convert('RGB') is the equivalent of image-0.25 decode() + to_rgb() for these
files (8-bit gray -> replicate, palette -> PLTE entry, RGBA -> drop alpha).
"""
import hashlib
import os
import shutil
import sys

from PIL import Image

NAMES = ["dirt", "stone", "moss", "water", "obsidian", "portal", "netherrack", "magma", "gold"]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/imgs"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "raytracing_amd", "assets")
    fix_dir = os.path.join(root, "tests", "golden", "imgs")
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(fix_dir, exist_ok=True)
    blob = bytearray()
    lines = []
    for n in NAMES:
        path = os.path.join(src, n + ".png")
        im = Image.open(path).convert("RGB")
        assert im.size == (48, 64), (n, im.size)
        raw = im.tobytes()
        assert len(raw) == 48 * 64 * 3
        blob += raw
        lines.append(f"{hashlib.sha256(raw).hexdigest()}  {n}")
        shutil.copyfile(path, os.path.join(fix_dir, n + ".png"))
    with open(os.path.join(out_dir, "atlas_rgb8.bin"), "wb") as f:
        f.write(blob)
    with open(os.path.join(out_dir, "atlas_rgb8.sha256"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", len(blob), "bytes")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
