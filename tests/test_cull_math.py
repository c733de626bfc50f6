"""The wave-level culls of raytracing_amd/csrc/rtx950_kernels.hip restated in numpy float32 (cull_pre + cull_test,
the one-radius form every tile uses; cull_pre_g + cull_test(CullPreG), the box form of the side kernel), held
to the property the kernels rely on: a cube that ANY lane's line meets (the reference's slab test is a line
test, src/cube.rs:90-145) is never culled.  CPU only: this checks the inequalities and their rounding slack,
not the device code; the GPU parity tests check that."""
import numpy as np
import pytest

f32 = np.float32


def round_up_bf16(v):
    """wave_max2: a non-negative float rounded up to its top 16 bits (+1 unit there)."""
    u = (np.asarray(v, dtype=f32).view(np.uint32) >> 16) + 1
    return (u.astype(np.uint32) << 16).view(f32)


def slab_hits(o, d, k, h):
    """exact (f64) line-box test per lane and cube: [lanes, cubes] bool"""
    o = o.astype(np.float64)[:, None, :]; d = d.astype(np.float64)[:, None, :]
    k = k.astype(np.float64)[None, :, :]; h = h.astype(np.float64)[None, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (k - h - o) / d; t1 = (k + h - o) / d
    lo = np.fmax.reduce(np.minimum(t0, t1), axis=-1); hi = np.fmin.reduce(np.maximum(t0, t1), axis=-1)
    return lo <= hi


def moments(o, d):
    return np.stack([o[:, 0] * d[:, 1] - o[:, 1] * d[:, 0], o[:, 1] * d[:, 2] - o[:, 2] * d[:, 1], o[:, 2] * d[:, 0] - o[:, 0] * d[:, 2]], 1).astype(f32)


def slack(bound_max, origin_bound, a):
    return f32(2.0 ** -17) * f32(bound_max + origin_bound + 1.0) * max(f32(a.sum()), f32(1.0))


def cull_one_radius(o, d, k, h, ref, bound_max, origin_bound):
    w = moments(o, d)
    dc, wc = d[ref], w[ref]
    dr = round_up_bf16(np.abs(d - dc).max()); wr = round_up_bf16(np.abs(w - wc).max())
    a = np.abs(dc) + dr
    wr = f32(wr + slack(bound_max, origin_bound, a))
    keep = np.ones(len(k), bool)
    for p, (i, j) in enumerate(((0, 1), (1, 2), (2, 0))):
        g = k[:, i] * dc[j] - k[:, j] * dc[i] - wc[p]
        l = h * (a[i] + a[j]) + (np.abs(k[:, i]) + np.abs(k[:, j])) * dr + wr
        keep &= ~(np.abs(g) > l)
    return keep


def cull_box(o, d, k, h, ref, bound_max, origin_bound):
    c = o[ref]
    w = moments((o - c).astype(f32), d)
    def box(v, r):
        dl = (v - r).astype(f32)
        up = round_up_bf16(np.where(dl > 0, dl, 0).max()); dn = round_up_bf16(np.where(dl < 0, -dl, 0).max())
        return f32(r + f32(0.5) * (up - dn)), f32(f32(0.5) * (up + dn))
    dc = np.zeros(3, f32); rd = np.zeros(3, f32); wc = np.zeros(3, f32); rw = np.zeros(3, f32)
    for q in range(3):
        dc[q], rd[q] = box(d[:, q], d[ref, q])
        wc[q], rw[q] = box(w[:, q], f32(0))
    a = np.abs(dc) + rd
    rw = rw + slack(bound_max, origin_bound, a)
    kk = (k - c).astype(f32)
    keep = np.ones(len(k), bool)
    for p, (i, j) in enumerate(((0, 1), (1, 2), (2, 0))):
        g = kk[:, i] * dc[j] - kk[:, j] * dc[i] - wc[p]
        l = h * (a[i] + a[j]) + np.abs(kk[:, i]) * rd[j] + np.abs(kk[:, j]) * rd[i] + rw[p]
        keep &= ~(np.abs(g) > l)
    return keep


def bundles(rng, n):
    """ray bundles of the kinds the kernels meet: coherent primaries, a pencil towards a nearby point, two
    bundles in one wave, axis-parallel directions, widely scattered lines"""
    for t in range(n):
        kind = t % 5
        base_o = rng.uniform(-8, 8, 3)
        if kind == 0:
            o = np.tile(base_o, (64, 1)); d = rng.normal(size=3) + rng.normal(scale=0.01, size=(64, 3))
        elif kind == 1:
            L = base_o + rng.normal(scale=0.3, size=3)
            o = base_o + rng.uniform(-0.1, 0.1, (64, 3)); d = L - o
        elif kind == 2:
            o = base_o + rng.uniform(-0.1, 0.1, (64, 3))
            d = np.where(np.arange(64)[:, None] < 32, rng.normal(size=3), rng.normal(size=3)) + rng.normal(scale=0.02, size=(64, 3))
        elif kind == 3:
            o = base_o + rng.uniform(-0.5, 0.5, (64, 3)); d = np.zeros((64, 3)); d[:, rng.integers(3)] = rng.choice([-1.0, 1.0]); d[:, rng.integers(3)] += rng.choice([0.0, 0.3])
        else:
            o = rng.uniform(-8, 8, (64, 3)); d = rng.normal(size=(64, 3))
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        yield o.astype(f32), d.astype(f32)


@pytest.mark.parametrize("form", ["one_radius", "box"])
def test_wave_cull_never_drops_a_cube_some_lane_meets(form):
    rng = np.random.default_rng(20261111)
    k = (np.round(rng.uniform(-6, 6, (400, 3)) / 1.5) * 1.5).astype(f32)
    h = np.where(rng.random(400) < 0.1, rng.uniform(0.01, 3.0, 400), 0.75).astype(f32)
    bound_max = float(np.abs(k).max() + h.max()); origin_bound = 10.0
    kept = met = 0
    for o, d in bundles(rng, 400):
        hits = slab_hits(o, d, k, h).any(axis=0)
        for ref in (27, 0, 63):
            keep = (cull_one_radius if form == "one_radius" else cull_box)(o, d, k, h, ref, bound_max, origin_bound)
            assert not (hits & ~keep).any(), f"{form}: culled a cube that a lane's line meets (reference lane {ref})"
        kept += keep.sum(); met += hits.sum()
    assert met > 1000 and kept >= met  # the bundles do meet cubes, and the cull is a superset


def test_box_cull_is_tighter_on_a_pencil_towards_a_nearby_light():
    """the case the side kernel's general cull was written for: hit points on a wall, a light a quarter of a cube in front of it"""
    rng = np.random.default_rng(7)
    gx, gy = np.meshgrid(np.arange(-4, 5) * 1.5, np.arange(0, 5) * 1.5)
    k = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, -4.5)], 1).astype(f32); h = np.full(len(k), 0.75, f32)
    L = np.array([0.0, 2.25, -3.5])
    px, py = np.meshgrid(np.linspace(-0.05, 0.05, 8), np.linspace(1.9, 2.0, 8))
    o = np.stack([px.ravel(), py.ravel(), np.full(64, -3.75)], 1).astype(f32)
    d = L - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    hits = slab_hits(o, d, k, h).any(axis=0)
    one = cull_one_radius(o, d, k, h, 0, 8.0, 10.0); box = cull_box(o, d, k, h, 27, 8.0, 10.0)
    assert not (hits & ~one).any() and not (hits & ~box).any()
    assert box.sum() < one.sum()
