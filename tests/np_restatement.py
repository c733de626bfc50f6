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


# ---------------------------------------------------------------------------------------------------------
# Whole frames.  A second, independent restatement of the reference's render path, written from the Rust
# sources - NOT from oracle/rt_oracle.c - and literal about it: secondary rays are re-cast inside the light fold
# exactly as the reference does (src/render.rs:154-193 inside :104-106; no once-per-hit reuse), shadows take the
# first object in list order (:44-70), every colour operation goes through u8 (src/color.rs:99-130).  Vectorised
# over the rays of one cast_ray call; one float32 numpy op per reference op, so nothing is fused.
#   render()            src/render.rs:234-324        render_frame
#   cast_ray()          src/render.rs:73-229         _Scene.cast_ray
#   cast_shadow()       src/render.rs:44-70          _Scene.cast_shadow
#   reflect / refract   src/render.rs:17-42          _reflect / _refract
#   Cube::ray_intersect src/cube.rs:77-218           _Scene.cube_slab / cube_resolve
#   Sphere::ray_intersect src/sphere.rs:25-56        _Scene.sphere_hit
#   get_color_of_face   src/texture.rs:171-196       _Scene.texel
#   Color + / * f32     src/color.rs:99-130          _cadd / _cmul
# powf is the C library's (Rust's f32::powf lowers to it), called through ctypes.
import ctypes
import ctypes.util

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
_powf = np.frompyfunc(lambda x, y: _libm.powf(float(x), float(y)), 2, 1)

FACE_NONE, FACE_TOP, FACE_BOTTOM, FACE_FORWARDS, FACE_BACKWARDS, FACE_LEFT, FACE_RIGHT = range(7)  # src/texture.rs:11-20


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _norm3(a):
    n = np.sqrt(_dot(a, a))
    return [a[0] / n, a[1] / n, a[2] / n]


def _rs_clamp(x, lo, hi):  # f32::clamp: NaN passes through
    return np.where(x < lo, lo, np.where(x > hi, hi, x)).astype(np.float32)


def _reflect(i, n):  # incident - 2.0 * incident.dot(normal) * normal
    k = f32(2.0) * _dot(i, n)
    return [i[0] - k * n[0], i[1] - k * n[1], i[2] - k * n[2]]


def _refract(i, n, eta_t):
    cosi = -_rs_clamp(_dot(i, n), f32(-1.0), f32(1.0))
    enter = cosi < 0
    n_cosi = np.where(enter, -cosi, cosi)
    eta = np.where(enter, f32(1.0) / eta_t, eta_t).astype(np.float32)
    nn = [np.where(enter, -c, c) for c in n]
    k = f32(1.0) - eta * eta * (f32(1.0) - n_cosi * n_cosi)
    refl = _reflect(i, nn)
    s = eta * n_cosi - np.sqrt(k)
    refr = [eta * i[c] + s * nn[c] for c in range(3)]
    return [np.where(k < 0, refl[c], refr[c]).astype(np.float32) for c in range(3)]


def _cmul(c, f):  # Color * f32: (c as f32 * f).clamp(0, 255) as u8 (NaN as u8 = 0; `as` truncates)
    f = np.asarray(f, dtype=np.float32)
    v = c.astype(np.float32) * (f[:, None] if f.ndim == 1 else f)
    v = np.where(v < 0, f32(0.0), np.where(v > 255, f32(255.0), v))
    return np.where(np.isnan(v), f32(0.0), v).astype(np.uint8)


def _cadd(a, b):  # saturating_add per channel
    return np.minimum(a.astype(np.uint16) + b.astype(np.uint16), 255).astype(np.uint8)


def _hex(v):
    return np.array([(v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF], dtype=np.uint8)


class _Scene:
    def __init__(self, scene, frame):
        c = scene.cubes
        half = np.array([x.size for x in c], dtype=np.float32) / f32(2.0)  # cube.rs:52
        ctr = np.array([[x.center.x, x.center.y, x.center.z] for x in c], dtype=np.float32).reshape(-1, 3)
        self.lo = ctr - half[:, None]
        self.hi = ctr + half[:, None]
        self.size = np.array([x.size for x in c], dtype=np.float32)
        self.cube_id = np.array([x.id for x in c], dtype=np.int64)
        self.cube_mat = np.array([x.material for x in c], dtype=np.int64)
        s = scene.spheres
        self.sc = np.array([[x.center.x, x.center.y, x.center.z] for x in s], dtype=np.float32).reshape(-1, 3)
        self.sr = np.array([x.radius for x in s], dtype=np.float32)
        self.sphere_id = np.array([x.id for x in s], dtype=np.int64)
        self.sphere_mat = np.array([x.material for x in s], dtype=np.int64)
        m = scene.materials
        g = lambda name, dt=np.float32: np.array([getattr(x, name) for x in m], dtype=dt)
        self.m_spec, self.m_a0, self.m_a1 = g("specular"), g("albedo0"), g("albedo1")
        self.m_refl, self.m_tr, self.m_ior = g("reflectivity"), g("transparency"), g("refractive_index")
        self.m_tex = g("texture", np.int64)
        self.m_diff = np.stack([_hex(int(x.diffuse_0rgb)) for x in m])
        self.atlas = scene.atlas  # [n_tex, tex_h, tex_w, 3]: colors[y * width + x], texture.rs:151-161
        self.sprite = f32(scene.sprite)
        self.lights = [(np.array([frame.lights[i].position.x, frame.lights[i].position.y, frame.lights[i].position.z], dtype=np.float32),
                        _hex(int(frame.lights[i].color_0rgb)), f32(frame.lights[i].intensity)) for i in range(frame.n_lights)]
        self.sky = _hex(int(frame.sky_0rgb))
        self.max_depth = int(frame.max_depth)

    # ---- Cube::ray_intersect, slab part (cube.rs:90-145): Some?, distance
    def cube_slab(self, lo, hi, o, d):
        with np.errstate(all="ignore"):
            tmin, tmax = _minmax((lo[0] - o[0]) / d[0], (hi[0] - o[0]) / d[0])
            tymin, tymax = _minmax((lo[1] - o[1]) / d[1], (hi[1] - o[1]) / d[1])
            some = ~((tmin > tymax) | (tymin > tmax))
            tmin = np.fmax(tmin, tymin)
            tmax = np.fmin(tmax, tymax)
            tzmin, tzmax = _minmax((lo[2] - o[2]) / d[2], (hi[2] - o[2]) / d[2])
            some &= ~((tmin > tzmax) | (tzmin > tmax))
            tmin = np.fmax(tmin, tzmin)
            tmax = np.fmin(tmax, tzmax)
            return some, np.where(tmin < 0, tmax, tmin).astype(np.float32)

    # ---- the rest of it for the cube idx[r] of ray r (cube.rs:146-204): point, normal, face, texture_cords
    def cube_resolve(self, idx, o, d):
        lo = [self.lo[idx, c] for c in range(3)]
        hi = [self.hi[idx, c] for c in range(3)]
        size = self.size[idx]
        _, dist = self.cube_slab(lo, hi, o, d)
        with np.errstate(all="ignore"):
            p = [o[c] + d[c] * dist for c in range(3)]
            limit = f32(1e-3)
            n = len(idx)
            face = np.zeros(n, dtype=np.int64)
            nrm = [np.zeros(n, dtype=np.float32) for _ in range(3)]
            u = np.zeros(n, dtype=np.float32)
            v = np.zeros(n, dtype=np.float32)
            px, py, pz = [(p[c] - lo[c]) / size for c in range(3)]
            one = f32(1.0)
            table = [(np.abs(p[0] - lo[0]) < limit, FACE_LEFT, (-1, 0, 0), py, pz),
                     (np.abs(p[0] - hi[0]) < limit, FACE_RIGHT, (1, 0, 0), one - py, pz),
                     (np.abs(p[1] - lo[1]) < limit, FACE_BOTTOM, (0, -1, 0), px, pz),
                     (np.abs(p[1] - hi[1]) < limit, FACE_TOP, (0, 1, 0), px, pz),
                     (np.abs(p[2] - lo[2]) < limit, FACE_BACKWARDS, (0, 0, -1), px, one - py),
                     (np.abs(p[2] - hi[2]) < limit, FACE_FORWARDS, (0, 0, 1), px, py)]
            open_ = np.ones(n, dtype=bool)
            for cond, fc, nv, uu, vv in table:  # the else-if chain: first match wins
                take = open_ & cond
                face = np.where(take, fc, face)
                for c in range(3):
                    nrm[c] = np.where(take, f32(nv[c]), nrm[c])
                u = np.where(take, uu, u).astype(np.float32)
                v = np.where(take, vv, v).astype(np.float32)
                open_ &= ~cond
        return dist, p, nrm, face, u, v

    # ---- Sphere::ray_intersect (sphere.rs:25-56)
    def sphere_hit(self, ctr, rad, o, d):
        with np.errstate(all="ignore"):
            oc = [o[c] - ctr[c] for c in range(3)]
            a = _dot(d, d)
            b = f32(2.0) * _dot(oc, d)
            cc = _dot(oc, oc) - rad * rad
            disc = b * b - f32(4.0) * a * cc
            t = (-b - np.sqrt(disc)) / (f32(2.0) * a)
            some = (disc > 0) & (t > 0)
            return some, t.astype(np.float32)

    def objects(self, kind):
        return range(len(self.cube_id) if kind == "cubes" else len(self.sphere_id))

    def hit_test(self, kind, i, o, d):
        if kind == "cubes":
            return self.cube_slab(self.lo[i], self.hi[i], o, d)
        return self.sphere_hit(self.sc[i], self.sr[i], o, d)

    # ---- Texture::get_color_of_face (texture.rs:178-196)
    def texel(self, tex, face, x, y):
        sp = self.sprite
        ox = np.select([face == FACE_TOP, face == FACE_BOTTOM, face == FACE_FORWARDS, face == FACE_BACKWARDS, face == FACE_LEFT, face == FACE_RIGHT],
                       [sp, sp, sp, sp, f32(0.0), sp * f32(2.0)], f32(0.0)).astype(np.float32)
        oy = np.select([face == FACE_TOP, face == FACE_BOTTOM, face == FACE_FORWARDS, face == FACE_BACKWARDS, face == FACE_LEFT, face == FACE_RIGHT],
                       [sp, sp * f32(3.0), sp * f32(0.0), sp * f32(2.0), sp, sp], f32(0.0)).astype(np.float32)
        with np.errstate(all="ignore"):
            cx = _rs_clamp(ox + x, ox, ox + sp - f32(1.0))
            cy = _rs_clamp(oy + y, oy, oy + sp - f32(1.0))
            xi = np.where(np.isnan(cx), 0, np.maximum(cx, 0)).astype(np.int64)  # `as u32`
            yi = np.where(np.isnan(cy), 0, np.maximum(cy, 0)).astype(np.int64)
        col = self.atlas[np.maximum(tex, 0), yi, xi]
        return np.where((face == FACE_NONE)[:, None], _hex(0xFF00FF)[None, :], col).astype(np.uint8)

    # ---- cast_shadow (render.rs:44-70); impact_id: Eq by id (cube.rs:69-74, sphere.rs:17-22)
    def cast_shadow(self, kind, p, light_pos, impact_id):
        n = len(impact_id)
        with np.errstate(all="ignore"):
            ld = _norm3([light_pos[c] - p[c] for c in range(3)])
        shadow = np.zeros(n, dtype=np.float32)
        pending = np.ones(n, dtype=bool)
        ids = self.cube_id if kind == "cubes" else self.sphere_id
        mats = self.cube_mat if kind == "cubes" else self.sphere_mat
        for i in self.objects(kind):
            consider = pending & (impact_id != ids[i])
            if not consider.any():
                continue
            some, dist = self.hit_test(kind, i, p, ld)
            first = consider & some
            if not first.any():
                continue
            with np.errstate(all="ignore"):
                hp = [p[c] + ld[c] * dist for c in range(3)]
                dv = [hp[c] - light_pos[c] for c in range(3)]  # distance2(a, b) = (b - a).norm_squared()
                val = dist / _dot(dv, dv) * (f32(1.0) - self.m_tr[mats[i]])
            shadow = np.where(first & ~(dist < 0), val, shadow).astype(np.float32)  # distance < 0: break with 0
            pending &= ~first
        return shadow

    # ---- cast_ray (render.rs:73-229); returns (colours u8 [n,3], index of the object hit or -1)
    def cast_ray(self, kind, o, d, ambient, depth):
        n = len(o[0])
        out = np.repeat(self.sky[None, :], n, axis=0)
        hit_idx = np.full(n, -1, dtype=np.int64)
        if depth > self.max_depth or n == 0:
            return out, hit_idx
        best = np.full(n, np.inf, dtype=np.float32)
        for i in self.objects(kind):
            some, dist = self.hit_test(kind, i, o, d)
            with np.errstate(invalid="ignore"):
                win = some & (dist < best)  # strict <: the first of equals keeps the hit
            best = np.where(win, dist, best)
            hit_idx = np.where(win, i, hit_idx)
        sel = np.nonzero(hit_idx >= 0)[0]
        if len(sel) == 0:
            return out, hit_idx
        idx = hit_idx[sel]
        oo = [o[c][sel] for c in range(3)]
        dd = [d[c][sel] for c in range(3)]
        m = len(sel)
        if kind == "cubes":
            dist, p, nrm, face, u, v = self.cube_resolve(idx, oo, dd)
            mat = self.cube_mat[idx]
            obj_id = self.cube_id[idx]
        else:
            with np.errstate(all="ignore"):
                _, dist = self.sphere_hit([self.sc[idx, c] for c in range(3)], self.sr[idx], oo, dd)
                p = [oo[c] + dd[c] * dist for c in range(3)]
                nrm = _norm3([p[c] - self.sc[idx, c] for c in range(3)])
            face = np.zeros(m, dtype=np.int64)
            u = np.zeros(m, dtype=np.float32)
            v = np.zeros(m, dtype=np.float32)
            mat = self.sphere_mat[idx]
            obj_id = self.sphere_id[idx]
        tex = self.m_tex[mat]
        tx = np.where((tex >= 0)[:, None], self.texel(tex, face, u * self.sprite, v * self.sprite), self.m_diff[mat])
        k_r, k_t = self.m_refl[mat], self.m_tr[mat]
        acc = np.zeros((m, 3), dtype=np.uint8)  # Color::default()
        for lp, lcol, lint in self.lights:
            with np.errstate(all="ignore"):
                light_dir = _norm3([lp[c] - p[c] for c in range(3)])
                view_dir = _norm3([oo[c] - p[c] for c in range(3)])
                reflect_dir = _norm3(_reflect([-x for x in light_dir], nrm))
                shadow = self.cast_shadow(kind, p, lp, obj_id)
                light_intensity = lint * (f32(1.0) - shadow)
                diffuse_intensity = _rs_clamp(_dot(nrm, light_dir), f32(0.0), f32(1.0)) + f32(ambient)
                diffuse = _cmul(_cmul(_cmul(tx, self.m_a0[mat]), diffuse_intensity), light_intensity)
                base = _rs_clamp(_dot(view_dir, reflect_dir), f32(0.0), f32(1.0))
                spec_i = _powf(base, self.m_spec[mat]).astype(np.float32)
                specular = _cmul(_cmul(_cmul(np.repeat(lcol[None, :], m, axis=0), self.m_a1[mat]), spec_i), light_intensity)
                reflect_color = np.zeros((m, 3), dtype=np.uint8)
                r_sel = np.nonzero(k_r > 0)[0]
                if len(r_sel):
                    rd = _norm3(_reflect([-dd[c][r_sel] for c in range(3)], [nrm[c][r_sel] for c in range(3)]))
                    ro = [p[c][r_sel] + f32(1e-2) * nrm[c][r_sel] for c in range(3)]
                    reflect_color[r_sel] = self.cast_ray(kind, ro, rd, ambient, depth + 1)[0]
                refract_color = np.zeros((m, 3), dtype=np.uint8)
                t_sel = np.nonzero(k_t > 0)[0]
                if len(t_sel):
                    td = _refract([dd[c][t_sel] for c in range(3)], [nrm[c][t_sel] for c in range(3)], self.m_ior[mat][t_sel])
                    to = [p[c][t_sel] + f32(1e-3) * nrm[c][t_sel] for c in range(3)]
                    refract_color[t_sel] = self.cast_ray(kind, to, td, ambient, depth + 1)[0]
                acc = _cadd(_cadd(_cadd(acc, _cmul(_cadd(diffuse, specular), f32(1.0) - k_r - k_t)), _cmul(reflect_color, k_r)), _cmul(refract_color, k_t))
        out[sel] = acc
        return out, hit_idx


def render_frame(scene, frame):
    """render() of the reference (src/render.rs:234-324) for the whole frame described by `frame` (rt_frame):
    returns (framebuffer uint32 [H, W] 0x00RRGGBB, hit_cube, hit_sphere uint32 [H, W], 0xFFFFFFFF = none)."""
    W, H = frame.width, frame.height
    S = _Scene(scene, frame)
    eye, (dx, dy, dz) = primary_rays(frame.camera, W, H)
    n = W * H
    o = [np.full(n, eye[c], dtype=np.float32) for c in range(3)]
    d = [dx.reshape(-1), dy.reshape(-1), dz.reshape(-1)]
    sphere_colors, hs = S.cast_ray("spheres", o, d, 1.0, 0)                       # render.rs:248-276
    cube_colors, hc = S.cast_ray("cubes", o, d, frame.ambient_light, 0)           # :278-306
    is_sky = (sphere_colors == S.sky[None, :]).all(axis=1)                        # Color == by hex value
    px = np.where(is_sky[:, None], cube_colors, sphere_colors).astype(np.uint32)  # :308-312
    fb = np.zeros(n, dtype=np.uint32)                                             # framebuffer.clear(): black
    packed = (px[:, 0] << 16) | (px[:, 1] << 8) | px[:, 2]
    i = np.arange(n)
    x = np.where(i > W, i % W, i)                                                 # :317-321
    y = i // W
    ok = x < W                                                                    # paint_point: XTooLarge is ignored (:322)
    fb[(y * W + x)[ok]] = packed[ok]
    as_u32 = lambda h: np.where(h < 0, 0xFFFFFFFF, h).astype(np.uint32).reshape(H, W)
    return fb.reshape(H, W), as_u32(hc), as_u32(hs)
