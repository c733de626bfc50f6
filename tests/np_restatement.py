"""Independent numpy (float32) restatement of the reference's depth-0 nearest-cube selection.

Written separately from oracle/rt_oracle.c, straight from reference src/render.rs:253-263 (ray
generation), src/camera.rs:33-41, src/cube.rs:90-145 and src/render.rs:87-101, vectorised over the
pixels of a frame.  Used to cross-check the C oracle's hit-index plane (tests/test_oracle_kat.py).
Every operation is a single float32 numpy op, so there is no fused multiply-add.
"""
import numpy as np

f32 = np.float32


def _normalize(x, y, z):
    n = np.sqrt((x * x + y * y) + z * z)
    return x / n, y / n, z / n


def primary_rays(camera, width, height):
    W, H = f32(width), f32(height)
    xs = np.arange(width, dtype=np.float32)[None, :].repeat(height, 0)
    ys = np.arange(height, dtype=np.float32)[:, None].repeat(width, 1)
    sx = (f32(2.0) * xs) / W - f32(1.0)
    sy = -(f32(2.0) * ys) / H + f32(1.0)
    sx = sx * (W / H)
    vx, vy, vz = _normalize(sx, sy, np.full_like(sx, -1.0))
    eye = np.array([camera.eye.x, camera.eye.y, camera.eye.z], dtype=np.float32)
    cen = np.array([camera.center.x, camera.center.y, camera.center.z], dtype=np.float32)
    up = np.array([camera.up.x, camera.up.y, camera.up.z], dtype=np.float32)

    def nrm(v):
        n = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
        return v / n

    def cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=np.float32)

    fwd = nrm(cen - eye)
    right = nrm(cross(fwd, up))
    upv = nrm(cross(right, fwd))
    cx = (vx * right[0] + vy * upv[0]) - vz * fwd[0]
    cy = (vx * right[1] + vy * upv[1]) - vz * fwd[1]
    cz = (vx * right[2] + vy * upv[2]) - vz * fwd[2]
    dx, dy, dz = _normalize(cx, cy, cz)
    return eye, (dx, dy, dz)


def _minmax(a, b):
    lt = a < b
    return np.where(lt, a, b), np.where(lt, b, a)


def nearest_cube(scene, width, height):
    """uint32 [height,width]: index of the nearest cube per pixel (0xFFFFFFFF = none)."""
    eye, (dx, dy, dz) = primary_rays(scene.camera, width, height)
    best = np.full((height, width), np.inf, dtype=np.float32)
    idx = np.full((height, width), 0xFFFFFFFF, dtype=np.uint32)
    with np.errstate(all="ignore"):
        for i, c in enumerate(scene.cubes):
            half = f32(c.size) / f32(2.0)
            lo = [f32(c.center.x) - half, f32(c.center.y) - half, f32(c.center.z) - half]
            hi = [f32(c.center.x) + half, f32(c.center.y) + half, f32(c.center.z) + half]
            tmin, tmax = _minmax((lo[0] - eye[0]) / dx, (hi[0] - eye[0]) / dx)
            tymin, tymax = _minmax((lo[1] - eye[1]) / dy, (hi[1] - eye[1]) / dy)
            ok = ~((tmin > tymax) | (tymin > tmax))
            tmin = np.fmax(tmin, tymin)
            tmax = np.fmin(tmax, tymax)
            tzmin, tzmax = _minmax((lo[2] - eye[2]) / dz, (hi[2] - eye[2]) / dz)
            ok &= ~((tmin > tzmax) | (tzmin > tmax))
            tmin = np.fmax(tmin, tzmin)
            tmax = np.fmin(tmax, tzmax)
            dist = np.where(tmin < 0, tmax, tmin)
            win = ok & (dist < best)
            best = np.where(win, dist, best)
            idx = np.where(win, np.uint32(i), idx)
    return idx
