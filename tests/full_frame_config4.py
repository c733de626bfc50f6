#!/usr/bin/env python3
"""tests/full_frame_config4.py [width height [band_rows]]: BASELINE.json configuration 4 (the 10k-cube + 256-sphere
world) at its exact size against the oracle on EVERY pixel.  The whole 3840x2160 frame is minutes of brute force
for the oracle, which is why the test suite compares two 32-row bands (test_config4_bands_of_the_4k_frame_against_
the_oracle); this script walks all of them: the GPU renders the frame once with the debug variant (hit planes,
ray-tree hashes) and once with the plain kernel that ships, the oracle renders it band by band (the same rays
as in the whole frame), and every band must agree bit for bit, ray counters included.  Prints one line per
band (a long run keeps talking) and a summary; exit status 1 on any difference.  Not collected by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracing_amd as rt  # noqa: E402
import oracle as orc  # noqa: E402
from parity import assert_parity  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
ROWS = int(sys.argv[3]) if len(sys.argv) > 3 else 40

scene = rt.synthetic_scene()
ctx = rt.Context(0)
ctx.upload(scene)
full = scene.frame(W, H)
gfb, gpl = ctx.render(full, debug=True)
plain = ctx.render(full)
if not np.array_equal(plain, gfb):
    print(f"plain and debug kernels differ on {int((plain != gfb).sum())} pixels")
    sys.exit(1)
O = orc.OracleScene(scene)
bad = 0
pixels = 0
hits = 0
t0 = time.time()
for y0 in range(0, H, ROWS):
    y1 = min(H, y0 + ROWS)
    f = scene.frame(W, H, y0=y0, y1=y1)
    ofb, opl, ocnt, _ = O.render(f, debug=True, reuse_secondary=True)
    band_planes = {k: v[y0:y1] for k, v in gpl.items()}
    try:
        rep = assert_parity(gfb[y0:y1], band_planes, ofb, opl, f"config 4: rows {y0}..{y1} of {W}x{H}")
        assert rep["exact_fraction"] == 1.0, rep
        assert ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict(), "ray counters differ"
        status = "equal"
    except AssertionError as e:
        bad += 1
        status = f"DIFFERENT: {str(e)[:200]}"
    pixels += (y1 - y0) * W
    hits += int((band_planes["hit_cube"] != rt.RT_NO_HIT).sum())
    print(f"rows {y0:4d}..{y1:4d}: {status}  ({time.time() - t0:.0f} s)", flush=True)
print(f"full_frame_config4: {W}x{H}, {scene.n_cubes} cubes + {scene.n_spheres} spheres: {pixels} pixels in {(H + ROWS - 1) // ROWS} bands, "
      f"{hits} with a cube hit, {bad} bands differing; oracle time {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
