"""CPU analysis (numpy restatement) of one 8x8 tile of the 4K diorama: per ray-tree level the cubes hit, the faces, how many cubes the lines really meet, and the spread of directions / origins / moments of the level's rays and of its children per face.  Usage: python tools/tile_rays.py <tile x> <tile y>"""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np
import raytracing_amd as rt, np_restatement as npr
s = rt.default_scene()
W, H = 3840, 2160
f = s.frame(W, H)
S = npr._Scene(s, f)
eye, (dx, dy, dz) = npr.primary_rays(f.camera, W, H)
tx, ty = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (135, 156)
ys, xs = np.mgrid[ty*8:ty*8+8, tx*8:tx*8+8]
d = [dx[ys, xs].ravel(), dy[ys, xs].ravel(), dz[ys, xs].ravel()]
o = [np.full(64, eye[c], dtype=np.float32) for c in range(3)]
def nearest(o, d):
    best = np.full(len(o[0]), np.inf, dtype=np.float32); idx = np.full(len(o[0]), -1)
    for i in S.objects("cubes"):
        some, dist = S.hit_test("cubes", i, o, d)
        with np.errstate(invalid="ignore"):
            win = some & (dist < best)
        best = np.where(win, dist, best); idx = np.where(win, i, idx)
    return idx, best
def spread(o, d, act, label):
    oo = np.stack(o, 1)[act]; dd = np.stack(d, 1)[act]
    Wm = np.cross(oo, dd)
    print(f"  {label}: lanes {act.sum()} dir spread {np.ptp(dd, axis=0)} origin spread {np.ptp(oo, axis=0)} moment spread {np.ptp(Wm, axis=0)}")
level_o, level_d, act = o, d, np.ones(64, bool)
for depth in range(4):
    idx, best = nearest(level_o, level_d)
    hit = act & (idx >= 0)
    sel = np.nonzero(hit)[0]
    if len(sel) == 0: break
    oo = [level_o[c][sel] for c in range(3)]; dd = [level_d[c][sel] for c in range(3)]
    dist, p, nrm, face, u, v = S.cube_resolve(idx[sel], oo, dd)
    mat = S.cube_mat[idx[sel]]
    kr, kt = S.m_refl[mat], S.m_tr[mat]
    print(f"depth {depth}: hits {len(sel)} cubes {sorted(set(idx[sel].tolist()))} faces {sorted(set(face.tolist()))} mats {sorted(set(mat.tolist()))} dist range {dist.min():.2f}..{dist.max():.2f} k_r>0 {int((kr>0).sum())} k_t>0 {int((kt>0).sum())}")
    # true candidates per ray (line test) for the nearest query of this level
    cnt = np.zeros(len(sel), int)
    for i in S.objects("cubes"):
        some, _ = S.hit_test("cubes", i, oo, dd); cnt += some
    print(f"  cubes the lines really meet: avg {cnt.mean():.1f} max {cnt.max()}; union over the wave {sum(S.hit_test('cubes', i, oo, dd)[0].any() for i in S.objects('cubes'))}")
    spread(level_o, level_d, hit, "this level's rays")
    # children: reflection first if any, else refraction (the walk's slot 0)
    has_r, has_t = kr > 0, kt > 0
    if not (has_r | has_t).any(): break
    with np.errstate(all="ignore"):
        rd = npr._norm3(npr._reflect([-x for x in dd], nrm)); ro = [p[c] + np.float32(1e-2) * nrm[c] for c in range(3)]
        td = npr._refract(dd, nrm, S.m_ior[mat]); to = [p[c] + np.float32(1e-3) * nrm[c] for c in range(3)]
    no = [np.zeros(64, np.float32) for _ in range(3)]; nd = [np.zeros(64, np.float32) for _ in range(3)]
    for c in range(3):
        no[c][sel] = np.where(has_r, ro[c], to[c]); nd[c][sel] = np.where(has_r, rd[c], td[c])
    act = np.zeros(64, bool); act[sel] = has_r | has_t
    for fc in sorted(set(face.tolist())):
        m = np.zeros(64, bool); m[sel] = (has_r | has_t) & (face == fc)
        if m.any(): spread(no, nd, m, f"children off face {fc}")
    level_o, level_d = no, nd
