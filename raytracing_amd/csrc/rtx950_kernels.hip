// rtx950_kernels.hip — the per-pixel ray-cast hot path as one gfx950 kernel.
//
// Replaces (reference = ElrohirGT/Raytracing, paths relative to its root):
//   render()            src/render.rs:234-324   -> render_kernel (one lane per pixel, 8x8 px per wave)
//   cast_ray()          src/render.rs:73-229    -> cast_ray<PASS,DEPTH> (recursion unrolled by template
//                                                 depth; per-level state lives in registers)
//   cast_shadow()       src/render.rs:44-70     -> shadow_query (inlined, list order, early exit)
//   Cube::ray_intersect src/cube.rs:77-218      -> cube_filter (conservative, FMA) + cube_exact (IEEE)
//                                                 + cube_resolve (face/uv for the winning hit only)
//   Sphere::ray_intersect src/sphere.rs:25-56   -> sphere_exact
//   Texture::get_color_of_face src/texture.rs:178-196 -> texel_of_face (packed atlas gather)
//   Color ops           src/color.rs:99-149     -> col_* on floats that hold exact u8 values
//
// Exactness contract.  This file is compiled with -ffp-contract=off, and HIP's default correctly
// rounded f32 divide/sqrt stay on, so every expression below that restates a reference
// expression rounds exactly like the reference's f32 arithmetic.  FMA is used ONLY in
// cube_filter, whose result never reaches an output: it only proves "this cube cannot pass the
// slab test" with an error bound, and everything it cannot prove goes through cube_exact.
//
// Wave-uniform data.  Object records are indexed by the loop counter only, so every lane of a wave
// reads the same record: they are loaded through the constant address space (s_load_dwordx8 into
// SGPRs) and feed VALU instructions as scalar operands.  No LDS round trip, no VGPR copies.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx950_device.h"

namespace rtx950 {
namespace {

#define DEV __device__ __forceinline__
#define KCONST const __attribute__((address_space(4)))

constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr uint32_t kHashBasis = 2166136261u;
constexpr uint32_t kHashPrime = 16777619u;
constexpr uint32_t kHashCut = 0x9E3779B9u;

enum { PASS_SPHERES = 0, PASS_CUBES = 1 };
enum { FACE_NONE = 0, FACE_TOP, FACE_BOTTOM, FACE_FORWARDS, FACE_BACKWARDS, FACE_LEFT, FACE_RIGHT };

DEV uint32_t hmix(uint32_t h, uint32_t v) { return (h ^ v) * kHashPrime; }

// Wave-uniform record loads: reading through the constant address space with an index that is the
// same in every lane makes hipcc emit s_load_dwordx8 / x4 / x2 (SGPR destination).
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
// NB: never apply __builtin_bit_cast directly to a vector element (`bit_cast(float, v[k])`): hipcc
// (ROCm 7.2) then reads element 0 for every k.  Copy the element to a scalar first.
DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
DEV CubeRec load_cube_uniform(const CubeRec* base, uint32_t i)
{
    const u32x8 v = ((KCONST u32x8*)base)[i];
    const uint32_t e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3], e4 = v[4], e5 = v[5], e6 = v[6], e7 = v[7];
    CubeRec c;
    c.minx = u2f(e0); c.maxx = u2f(e1);
    c.miny = u2f(e2); c.maxy = u2f(e3);
    c.minz = u2f(e4); c.maxz = u2f(e5);
    c.id = e6;
    c.size = u2f(e7);
    return c;
}
// 4 consecutive filter records (centre.xyz, half size), 64-byte aligned -> one s_load_dwordx16
DEV f32x16 load_filt4_uniform(const float* base, uint32_t first)
{
    return *(KCONST f32x16*)(base + (size_t)first * 4u);
}
DEV SphereRec load_sphere_uniform(const SphereRec* base, uint32_t i)
{
    const f32x4 v = ((KCONST f32x4*)base)[i];
    return SphereRec{v[0], v[1], v[2], v[3]};
}
DEV SphereMeta load_sphere_meta_uniform(const SphereMeta* base, uint32_t i)
{
    const u32x2 v = ((KCONST u32x2*)base)[i];
    return SphereMeta{v[0], v[1]};
}

// ---------------------------------------------------------------- Vec3 (nalgebra-glm semantics)
struct V3 {
    float x, y, z;
};
DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DEV float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV V3 vnormalize(V3 a)
{
    float n = __builtin_sqrtf(vdot(a, a));
    return v3(a.x / n, a.y / n, a.z / n);
}
// f32::clamp: NaN passes through
DEV float rs_clamp(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// ---------------------------------------------------------------- Color: 3 floats holding exact u8 values
struct Col {
    float r, g, b;
};
DEV Col col_from_hex(uint32_t v)
{
    return Col{(float)((v >> 16) & 0xFFu), (float)((v >> 8) & 0xFFu), (float)(v & 0xFFu)};
}
DEV uint32_t col_to_hex(Col c) { return ((uint32_t)c.r << 16) | ((uint32_t)c.g << 8) | (uint32_t)c.b; }
// color.rs:118-130: (c as f32 * f).clamp(0,255) as u8.  med3 returns min3 when an operand is NaN,
// i.e. 0, which is what `NaN as u8` gives.
DEV float chan_mul(float c, float f) { return __builtin_truncf(__builtin_amdgcn_fmed3f(c * f, 0.0f, 255.0f)); }
DEV Col col_mul(Col a, float f) { return Col{chan_mul(a.r, f), chan_mul(a.g, f), chan_mul(a.b, f)}; }
// color.rs:99-116: saturating add (all operands are exact integers in [0,255])
DEV Col col_add(Col a, Col b)
{
    return Col{__builtin_fminf(a.r + b.r, 255.0f), __builtin_fminf(a.g + b.g, 255.0f), __builtin_fminf(a.b + b.b, 255.0f)};
}

// ---------------------------------------------------------------- counters (count variant only)
struct Counters {  // per ray-tree node
    uint32_t nearest_rays, shadow_rays, cube_tests, sphere_tests, shaded_hits;
};
struct Totals {  // per lane: [0..4] unique work, [5..9] the reference's n_lights^depth-weighted work
    unsigned long long v[10];
};
DEV void totals_add(Totals& t, const Counters& c, unsigned long long w)
{
    const uint32_t a[5] = {c.nearest_rays, c.shadow_rays, c.cube_tests, c.sphere_tests, c.shaded_hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        t.v[k] += a[k];
        t.v[5 + k] += (unsigned long long)a[k] * w;
    }
}

// ---------------------------------------------------------------- conservative cube filter
// The reference tests every ray against every cube with 6 true divisions (cube.rs:90-109).  Almost
// all of those tests fail, and a failing test contributes nothing to the output.  The filter proves
// "fails" cheaply and sends everything it cannot prove to cube_exact.
//
// Per ray (once):   r_c = RN(1/d_c), c_c = -RN(o_c r_c), a_c = |r_c|, tol.
// Per cube (centre k, half-size h; 9 VALU instructions, 4 of them packed):
//   m_c   = fma(k_c, r_c, c_c)                      ~ (k_c - o_c)/d_c
//   near_c = fma(h, -a_c, m_c), far_c = fma(h, a_c, m_c)   ~ the slab interval on axis c
//   candidate  <=>  !(max3(near) - min3(far) > tol)
// Error bound, u = 2^-24, no overflow/underflow (dir_ok/org_ok/fast_ok guarantee the ranges):
//   reference  t = RN(RN(RN(k -+ h) - o)/d)            |t  - T| <= 3u (B+|o|)|r| (1+O(u)),  T exact
//   filter     t~ = RN(fma(h,-+a, RN(fma(k,r,-RN(o r)))))  |t~ - T| <= 5u (B+|o|)|r| (1+O(u))
// with B >= |k|+h the largest bound magnitude in the scene.  lo = max3 and hi = min3 each move by at
// most the largest per-axis error E = 8u max_c (B+|o_c|)|r_c|, so lo~ - hi~ > 2E implies lo > hi, which
// is exactly when the reference's staged slab test (cube.rs:101,113) returns None.  tol = 32u max_c(...)
// = 2E: twice the requirement.  Rays outside the safe ranges get tol = +inf (every cube is a
// candidate), and the negated compare makes NaN a candidate too.
struct RayPre {
    f32x2 r_xy, c_xy, na_xy;  // (r_x,r_y), (c_x,c_y), (-a_x,-a_y)
    f32x2 rc_z;               // (r_z, c_z)
    f32x2 a_z_pm;             // (-a_z, +a_z)
    float tol;
};

DEV bool dir_ok(float v)
{
    float a = __builtin_fabsf(v);
    return a >= 0x1p-40f && a <= 0x1p40f;
}
DEV bool org_ok(float v)
{
    float a = __builtin_fabsf(v);
    return a == 0.0f || (a >= 0x1p-30f && a <= 0x1p20f);
}

DEV RayPre ray_pre(const Params& P, V3 o, V3 d)
{
    RayPre q;
    const bool safe = (P.fast_ok != 0u) && dir_ok(d.x) && dir_ok(d.y) && dir_ok(d.z) && org_ok(o.x) && org_ok(o.y) && org_ok(o.z);
    const float rx = 1.0f / d.x, ry = 1.0f / d.y, rz = 1.0f / d.z;
    const float ax = __builtin_fabsf(rx), ay = __builtin_fabsf(ry), az = __builtin_fabsf(rz);
    q.r_xy = f32x2{rx, ry};
    q.c_xy = f32x2{-(o.x * rx), -(o.y * ry)};
    q.na_xy = f32x2{-ax, -ay};
    q.rc_z = f32x2{rz, -(o.z * rz)};
    q.a_z_pm = f32x2{-az, az};
    const float ex = (P.bound_max + __builtin_fabsf(o.x)) * ax;
    const float ey = (P.bound_max + __builtin_fabsf(o.y)) * ay;
    const float ez = (P.bound_max + __builtin_fabsf(o.z)) * az;
    q.tol = safe ? 0x1p-19f * __builtin_fmaxf(ex, __builtin_fmaxf(ey, ez)) : __builtin_inff();
    return q;
}

// rec = (centre.x, centre.y, centre.z, half size), wave-uniform.
// true -> the cube may pass the reference slab test (go to cube_exact); false -> proven miss
DEV bool cube_filter(f32x4 rec, const RayPre& q)
{
    const f32x2 k_xy = f32x2{rec[0], rec[1]};
    const f32x2 hh = f32x2{rec[3], rec[3]};
    const f32x2 m_xy = __builtin_elementwise_fma(k_xy, q.r_xy, q.c_xy);
    const float m_z = __builtin_fmaf(rec[2], q.rc_z[0], q.rc_z[1]);
    const f32x2 near_xy = __builtin_elementwise_fma(hh, q.na_xy, m_xy);
    const f32x2 far_xy = __builtin_elementwise_fma(hh, -q.na_xy, m_xy);
    const f32x2 nf_z = __builtin_elementwise_fma(hh, q.a_z_pm, f32x2{m_z, m_z});
    const float lo = __builtin_fmaxf(__builtin_fmaxf(near_xy[0], near_xy[1]), nf_z[0]);
    const float hi = __builtin_fminf(__builtin_fminf(far_xy[0], far_xy[1]), nf_z[1]);
    return !(lo - hi > q.tol);
}

// reference src/cube.rs:90-145 (slab test and distance), src/lib.rs:29-35 (minmax)
DEV bool cube_exact(const CubeRec& c, V3 o, V3 d, float& distance)
{
    float t0x = (c.minx - o.x) / d.x;
    float t1x = (c.maxx - o.x) / d.x;
    bool sx = t0x < t1x;
    float tmin = sx ? t0x : t1x, tmax = sx ? t1x : t0x;

    float t0y = (c.miny - o.y) / d.y;
    float t1y = (c.maxy - o.y) / d.y;
    bool sy = t0y < t1y;
    float tymin = sy ? t0y : t1y, tymax = sy ? t1y : t0y;

    if (tmin > tymax || tymin > tmax) return false;
    tmin = __builtin_fmaxf(tmin, tymin);  // f32::max: NaN-ignoring, = llvm.maxnum
    tmax = __builtin_fminf(tmax, tymax);

    float t0z = (c.minz - o.z) / d.z;
    float t1z = (c.maxz - o.z) / d.z;
    bool sz = t0z < t1z;
    float tzmin = sz ? t0z : t1z, tzmax = sz ? t1z : t0z;

    if (tmin > tzmax || tzmin > tmax) return false;
    tmin = __builtin_fmaxf(tmin, tzmin);
    tmax = __builtin_fminf(tmax, tzmax);

    distance = (tmin < 0.0f) ? tmax : tmin;
    return true;
}

// reference src/sphere.rs:25-39 (hit test and distance)
DEV bool sphere_exact(const SphereRec& s, V3 o, V3 d, float& distance)
{
    V3 oc = vsub(o, v3(s.cx, s.cy, s.cz));
    float a = vdot(d, d);
    float b = 2.0f * vdot(oc, d);
    float c = vdot(oc, oc) - s.radius * s.radius;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float t = (-b - __builtin_sqrtf(disc)) / (2.0f * a);
        if (t > 0.0f) {
            distance = t;
            return true;
        }
    }
    return false;
}

struct Hit {
    V3 p, n;
    float u, v;
    int face;
    uint32_t material;
    uint32_t id;
};

// reference src/cube.rs:146-204, evaluated for the winning cube only (the reference evaluates it for
// every slab-passing candidate and throws the losers away)
DEV Hit cube_resolve(const Params& P, uint32_t idx, float distance, V3 o, V3 d)
{
    const CubeRec c = P.cubes[idx];  // per-lane gather: lanes may have different winners
    Hit h;
    h.p = vadd(o, vscale(d, distance));
    h.n = v3(0.0f, 0.0f, 0.0f);
    h.face = FACE_NONE;
    h.u = 0.0f;
    h.v = 0.0f;
    const float limit = 1e-3f;
    if (__builtin_fabsf(h.p.x - c.minx) < limit) {
        h.n = v3(-1.0f, 0.0f, 0.0f); h.face = FACE_LEFT;
        h.u = (h.p.y - c.miny) / c.size; h.v = (h.p.z - c.minz) / c.size;
    } else if (__builtin_fabsf(h.p.x - c.maxx) < limit) {
        h.n = v3(1.0f, 0.0f, 0.0f); h.face = FACE_RIGHT;
        h.u = 1.0f - (h.p.y - c.miny) / c.size; h.v = (h.p.z - c.minz) / c.size;
    } else if (__builtin_fabsf(h.p.y - c.miny) < limit) {
        h.n = v3(0.0f, -1.0f, 0.0f); h.face = FACE_BOTTOM;
        h.u = (h.p.x - c.minx) / c.size; h.v = (h.p.z - c.minz) / c.size;
    } else if (__builtin_fabsf(h.p.y - c.maxy) < limit) {
        h.n = v3(0.0f, 1.0f, 0.0f); h.face = FACE_TOP;
        h.u = (h.p.x - c.minx) / c.size; h.v = (h.p.z - c.minz) / c.size;
    } else if (__builtin_fabsf(h.p.z - c.minz) < limit) {
        h.n = v3(0.0f, 0.0f, -1.0f); h.face = FACE_BACKWARDS;
        h.u = (h.p.x - c.minx) / c.size; h.v = 1.0f - (h.p.y - c.miny) / c.size;
    } else if (__builtin_fabsf(h.p.z - c.maxz) < limit) {
        h.n = v3(0.0f, 0.0f, 1.0f); h.face = FACE_FORWARDS;
        h.u = (h.p.x - c.minx) / c.size; h.v = (h.p.y - c.miny) / c.size;
    }
    h.material = P.cube_mat[idx];
    h.id = c.id;
    return h;
}

// reference src/sphere.rs:39-49
DEV Hit sphere_resolve(const Params& P, uint32_t idx, float distance, V3 o, V3 d)
{
    const SphereRec s = P.spheres[idx];
    const SphereMeta m = P.sphere_meta[idx];
    Hit h;
    h.p = vadd(o, vscale(d, distance));
    h.n = vnormalize(vsub(h.p, v3(s.cx, s.cy, s.cz)));
    h.u = 0.0f;
    h.v = 0.0f;
    h.face = FACE_NONE;
    h.material = m.material;
    h.id = m.id;
    return h;
}

// reference src/texture.rs:178-196 + :171-174
DEV Col texel_of_face(const Params& P, int tex, int face, float x, float y)
{
    float sp = P.sprite;
    float ox, oy;
    switch (face) {
    case FACE_TOP: ox = sp; oy = sp; break;
    case FACE_BOTTOM: ox = sp; oy = sp * 3.0f; break;
    case FACE_FORWARDS: ox = sp; oy = sp * 0.0f; break;
    case FACE_BACKWARDS: ox = sp; oy = sp * 2.0f; break;
    case FACE_LEFT: ox = 0.0f; oy = sp; break;
    case FACE_RIGHT: ox = sp * 2.0f; oy = sp; break;
    default: return col_from_hex(0xff00ffu);
    }
    float px = ox + x, py = oy + y;
    float cx = rs_clamp(px, ox, ox + sp - 1.0f);
    float cy = rs_clamp(py, oy, oy + sp - 1.0f);
    // `as u32`: NaN -> 0, negative -> 0, truncation; the clamp keeps real values inside the sprite
    uint32_t xi = (cx == cx) ? (uint32_t)__builtin_fmaxf(cx, 0.0f) : 0u;
    uint32_t yi = (cy == cy) ? (uint32_t)__builtin_fmaxf(cy, 0.0f) : 0u;
    xi = xi < P.tex_w ? xi : P.tex_w - 1u;  // memory safety only; unreachable for sprite*3 <= tex_w
    yi = yi < P.tex_h ? yi : P.tex_h - 1u;
    uint32_t idx = (uint32_t)tex * (P.tex_w * P.tex_h) + yi * P.tex_w + xi;
    return col_from_hex(P.atlas[idx]);
}

// reference src/render.rs:17-19
DEV V3 reflect(V3 i, V3 n) { return vsub(i, vscale(n, 2.0f * vdot(i, n))); }

// reference src/render.rs:21-42
DEV V3 refract(V3 i, V3 n, float eta_t)
{
    float cosi = -rs_clamp(vdot(i, n), -1.0f, 1.0f);
    float n_cosi = cosi, eta = eta_t;
    V3 nn = n;
    if (cosi < 0.0f) {
        n_cosi = -cosi;
        eta = 1.0f / eta_t;
        nn = vneg(n);
    }
    float k = 1.0f - eta * eta * (1.0f - n_cosi * n_cosi);
    if (k < 0.0f) return reflect(i, nn);
    return vadd(vscale(i, eta), vscale(nn, eta * n_cosi - __builtin_sqrtf(k)));
}

// f32::powf.  The reference calls glibc powf (error < 1 ulp, correctly rounded in all but a
// vanishing fraction of cases); evaluating in f64 and rounding once reproduces that far more
// closely than the 1-2 ulp f32 device routine.  Two calls per shaded hit: not a hot spot.
DEV float powf_like_libm(float x, float y) { return (float)pow((double)x, (double)y); }

// ---------------------------------------------------------------- nearest hit (render.rs:87-101)
template <int PASS, bool COUNT>
DEV void nearest_query(const Params& P, V3 o, V3 d, uint32_t& best_idx, float& best_dist, Counters& cnt)
{
    best_idx = kNoHit;
    best_dist = __builtin_inff();
    if constexpr (PASS == PASS_CUBES) {
        const RayPre q = ray_pre(P, o, d);
        const uint32_t n = P.n_cubes;
        // 4 filter records (one 64-byte line, one s_load_dwordx16) per iteration; the array is padded
        // to a multiple of 4 with far-away zero-size boxes.  One branch per batch: the exact path
        // runs only when some lane's filter could not prove a miss.
        f32x16 cur = load_filt4_uniform(P.cube_filt, 0u);
        for (uint32_t base = 0; base < n; base += 4u) {
            const uint32_t nb = base + 4u;
            const f32x16 nxt = load_filt4_uniform(P.cube_filt, nb < n ? nb : base);  // prefetch
            const bool c0 = cube_filter(f32x4{cur[0], cur[1], cur[2], cur[3]}, q);
            const bool c1 = cube_filter(f32x4{cur[4], cur[5], cur[6], cur[7]}, q);
            const bool c2 = cube_filter(f32x4{cur[8], cur[9], cur[10], cur[11]}, q);
            const bool c3 = cube_filter(f32x4{cur[12], cur[13], cur[14], cur[15]}, q);
            if (c0 | c1 | c2 | c3) {
                const bool cand[4] = {c0, c1, c2, c3};
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t i = base + k;
                    if (i < n && cand[k]) {
                        const CubeRec c = load_cube_uniform(P.cubes, i);
                        float dist;
                        if (cube_exact(c, o, d, dist)) {
                            if (dist < best_dist) {  // strict <: lowest index wins ties, NaN never wins
                                best_dist = dist;
                                best_idx = i;
                            }
                        }
                    }
                }
            }
            cur = nxt;
        }
        if constexpr (COUNT) cnt.cube_tests += n;
    } else {
        const uint32_t n = P.n_spheres;
        for (uint32_t i = 0; i < n; ++i) {
            const SphereRec s = load_sphere_uniform(P.spheres, i);
            float dist;
            if (sphere_exact(s, o, d, dist)) {
                if (dist < best_dist) {
                    best_dist = dist;
                    best_idx = i;
                }
            }
        }
        if constexpr (COUNT) cnt.sphere_tests += n;
    }
    if constexpr (COUNT) cnt.nearest_rays += 1;
}

// ---------------------------------------------------------------- shadow (render.rs:44-70)
// First object in array order, among those whose id differs from the impact object's
// (render.rs:113), whose ray_intersect returns Some.  Not the nearest one.
template <int PASS, bool COUNT>
DEV float shadow_query(const Params& P, V3 origin, V3 light_pos, uint32_t impact_id, uint32_t& blocker, Counters& cnt)
{
    const V3 dir = vnormalize(vsub(light_pos, origin));
    blocker = kNoHit;
    float sdist = 0.0f;
    bool done = false;
    if constexpr (COUNT) cnt.shadow_rays += 1;
    if constexpr (PASS == PASS_CUBES) {
        const RayPre q = ray_pre(P, origin, dir);
        const uint32_t n = P.n_cubes;
        // Same batched filter as nearest_query.  The id exclusion (render.rs:113) is applied in the
        // rare exact path only: excluded cubes that the filter happens to pass cost one compare.
        f32x16 cur = load_filt4_uniform(P.cube_filt, 0u);
        for (uint32_t base = 0; base < n; base += 4u) {
            const uint32_t nb = base + 4u;
            const f32x16 nxt = load_filt4_uniform(P.cube_filt, nb < n ? nb : base);
            const bool c0 = !done && cube_filter(f32x4{cur[0], cur[1], cur[2], cur[3]}, q);
            const bool c1 = !done && cube_filter(f32x4{cur[4], cur[5], cur[6], cur[7]}, q);
            const bool c2 = !done && cube_filter(f32x4{cur[8], cur[9], cur[10], cur[11]}, q);
            const bool c3 = !done && cube_filter(f32x4{cur[12], cur[13], cur[14], cur[15]}, q);
            if (c0 | c1 | c2 | c3) {
                const bool cand[4] = {c0, c1, c2, c3};
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t i = base + k;
                    if (i < n && cand[k] && !done) {
                        const CubeRec c = load_cube_uniform(P.cubes, i);
                        float dist;
                        if (c.id != impact_id && cube_exact(c, origin, dir, dist)) {
                            done = true;
                            blocker = i;
                            sdist = dist;
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            }
            cur = nxt;
        }
        if constexpr (COUNT) {
            // executed tests = cubes before (and including) the blocker whose id differs from the impact id
            const uint32_t limit = done ? blocker + 1u : n;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t id = load_cube_uniform(P.cubes, i).id;
                cnt.cube_tests += (i < limit && id != impact_id) ? 1u : 0u;
            }
        }
    } else {
        const uint32_t n = P.n_spheres;
        for (uint32_t i = 0; i < n; ++i) {
            const SphereRec s = load_sphere_uniform(P.spheres, i);
            const bool tested = !done && (load_sphere_meta_uniform(P.sphere_meta, i).id != impact_id);
            if constexpr (COUNT) cnt.sphere_tests += tested ? 1u : 0u;
            if (tested) {
                float dist;
                if (sphere_exact(s, origin, dir, dist)) {
                    done = true;
                    blocker = i;
                    sdist = dist;
                }
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        }
    }
    float shadow = 0.0f;
    if (blocker != kNoHit && !(sdist < 0.0f)) {  // render.rs:56-58: negative distance -> 0
        const V3 sp = vadd(origin, vscale(dir, sdist));
        const V3 dv = vsub(sp, light_pos);  // distance2(a,b) = norm_squared(b-a)
        const float d2 = vdot(dv, dv);
        uint32_t bm;
        if constexpr (PASS == PASS_CUBES) bm = P.cube_mat[blocker];
        else bm = P.sphere_meta[blocker].material;
        shadow = sdist / d2 * (1.0f - P.mats[bm].transparency);
    }
    return shadow;
}

// ---------------------------------------------------------------- cast_ray (render.rs:73-229)
// Secondary rays are traced once per hit and reused by every light iteration; the reference
// re-traces them per light with identical results.  Because every operand of the light fold is a
// non-negative u8 and `+` saturates, ((acc + X) + Y) + Z == min(255, acc+X+Y+Z), so the fold
// equals min(255, sum_l DS_l + L*(refl*k_r) + L*(refr*k_t)) per channel and can be evaluated
// after the children return.
template <int PASS, int DEPTH, bool DBG, bool COUNT>
DEV Col cast_ray(const Params& P, V3 o, V3 d, float ambient, uint32_t& hash, uint32_t& hit_idx, Totals& tot)
{
    const Col sky = col_from_hex(P.sky);
    hit_idx = kNoHit;
    if constexpr (DEPTH > 3) {
        hash = kHashCut;
        return sky;
    } else {
        if ((uint32_t)DEPTH > P.max_depth) {  // wave-uniform
            hash = kHashCut;
            return sky;
        }
        uint32_t idx;
        float dist;
        Counters cnt = {0u, 0u, 0u, 0u, 0u};
        nearest_query<PASS, COUNT>(P, o, d, idx, dist, cnt);
        hit_idx = idx;
        uint32_t h = hmix(kHashBasis, idx);
        Col result = sky;
        if (idx != kNoHit) {
            if constexpr (COUNT) cnt.shaded_hits += 1;
            Hit hit;
            if constexpr (PASS == PASS_CUBES) hit = cube_resolve(P, idx, dist, o, d);
            else hit = sphere_resolve(P, idx, dist, o, d);
            const MatRec m = P.mats[hit.material];

            Col tx;
            if (m.texture >= 0) tx = texel_of_face(P, m.texture, hit.face, hit.u * P.sprite, hit.v * P.sprite);
            else tx = col_from_hex(m.diffuse_0rgb);
            const Col tx_a = col_mul(tx, m.albedo0);
            const V3 view_dir = vnormalize(vsub(o, hit.p));
            const float k_ds = 1.0f - m.reflectivity - m.transparency;

            Col sum = Col{0.0f, 0.0f, 0.0f};
            const uint32_t nl = P.n_lights;
            for (uint32_t li = 0; li < nl; ++li) {
                const LightRec L = P.lights[li];
                const V3 lp = v3(L.px, L.py, L.pz);
                const V3 light_dir = vnormalize(vsub(lp, hit.p));
                const V3 reflect_dir = vnormalize(reflect(vneg(light_dir), hit.n));
                uint32_t blocker;
                const float shadow = shadow_query<PASS, COUNT>(P, hit.p, lp, hit.id, blocker, cnt);
                if constexpr (DBG) h = hmix(h, blocker);
                const float light_intensity = L.intensity * (1.0f - shadow);
                const float diffuse_intensity = rs_clamp(vdot(hit.n, light_dir), 0.0f, 1.0f) + ambient;
                const Col diffuse = col_mul(col_mul(tx_a, diffuse_intensity), light_intensity);
                const float spec_i = powf_like_libm(rs_clamp(vdot(view_dir, reflect_dir), 0.0f, 1.0f), m.specular);
                const Col specular = col_mul(col_mul(col_mul(col_from_hex(L.color_0rgb), m.albedo1), spec_i), light_intensity);
                sum = col_add(sum, col_mul(col_add(diffuse, specular), k_ds));
            }

            // children (render.rs:154-193)
            const bool has_refl = m.reflectivity > 0.0f;
            const bool has_refr = m.transparency > 0.0f;
            Col c_refl = Col{0.0f, 0.0f, 0.0f}, c_refr = Col{0.0f, 0.0f, 0.0f};
            uint32_t h_refl = 0u, h_refr = 0u;
            if (has_refl || has_refr) {
                const V3 refl_d = vnormalize(reflect(vneg(d), hit.n));
                const V3 refl_o = vadd(hit.p, vscale(hit.n, 1e-2f));
                const V3 refr_d = refract(d, hit.n, m.ior);
                const V3 refr_o = vadd(hit.p, vscale(hit.n, 1e-3f));
                const int n_child = (has_refl ? 1 : 0) + (has_refr ? 1 : 0);
                // slot 0 = the lane's first pending child, slot 1 = its second: lanes whose only
                // child is a refraction trace it together with other lanes' reflections.
                for (int slot = 0; slot < 2; ++slot) {
                    if (slot < n_child) {
                        const bool is_refl = (slot == 0) && has_refl;
                        const V3 co = is_refl ? refl_o : refr_o;
                        const V3 cd = is_refl ? refl_d : refr_d;
                        uint32_t ch, cidx;
                        const Col cc = cast_ray<PASS, DEPTH + 1, DBG, COUNT>(P, co, cd, ambient, ch, cidx, tot);
                        if (is_refl) { c_refl = cc; h_refl = ch; }
                        else { c_refr = cc; h_refr = ch; }
                    }
                }
            }
            const Col rl = col_mul(c_refl, m.reflectivity);
            const Col rr = col_mul(c_refr, m.transparency);
            const float fl = (float)nl;
            result.r = __builtin_fminf(sum.r + fl * rl.r + fl * rr.r, 255.0f);
            result.g = __builtin_fminf(sum.g + fl * rl.g + fl * rr.g, 255.0f);
            result.b = __builtin_fminf(sum.b + fl * rl.b + fl * rr.b, 255.0f);
            if constexpr (DBG) {
                h = hmix(h, h_refl);
                h = hmix(h, h_refr);
            }
        }
        if constexpr (COUNT) {
            unsigned long long w = 1ull;  // n_lights^DEPTH: how often the reference runs this node
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) w *= P.n_lights;
            totals_add(tot, cnt, w);
        }
        hash = h;
        return result;
    }
}

// ---------------------------------------------------------------- render (render.rs:234-324)
// One lane per pixel; a wave covers an 8x8 pixel tile (coherent rays -> few divergent exact tests),
// a 256-thread workgroup covers 16x16 pixels.
template <bool DBG, bool COUNT>
__global__ __launch_bounds__(256) void render_kernel(const Params P)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t ly = (wave >> 1) * 8u + (lane >> 3);
    // interleaved parts: this launch renders 16-row blocks part, part+n_parts, ... of [y0,y1)
    const uint32_t y = P.y0 + (blockIdx.y * P.n_parts + P.part) * kRowBlock + ly;
    const uint32_t out_row = blockIdx.y * kRowBlock + ly;
    Totals cnt;
#pragma unroll
    for (int k = 0; k < 10; ++k) cnt.v[k] = 0ull;
    if (x < P.width && y < P.y1) {
        const float width = (float)P.width, height = (float)P.height;
        const float aspect = width / height;
        float sx = (2.0f * (float)x) / width - 1.0f;
        const float sy = -(2.0f * (float)y) / height + 1.0f;
        sx = sx * aspect;
        const V3 v = vnormalize(v3(sx, sy, -1.0f));
        // camera.rs:38-40: v.x*right + v.y*up - v.z*forward, normalised
        const V3 right = v3(P.right[0], P.right[1], P.right[2]);
        const V3 up = v3(P.up[0], P.up[1], P.up[2]);
        const V3 fwd = v3(P.fwd[0], P.fwd[1], P.fwd[2]);
        const V3 dir = vnormalize(vsub(vadd(vscale(right, v.x), vscale(up, v.y)), vscale(fwd, v.z)));
        const V3 eye = v3(P.eye[0], P.eye[1], P.eye[2]);

        uint32_t hs, hc, hit_s, hit_c;
        const Col cs = cast_ray<PASS_SPHERES, 0, DBG, COUNT>(P, eye, dir, 1.0f, hs, hit_s, cnt);
        const Col cc = cast_ray<PASS_CUBES, 0, DBG, COUNT>(P, eye, dir, P.ambient, hc, hit_c, cnt);
        const uint32_t s_hex = col_to_hex(cs);
        uint32_t px = (s_hex == P.sky) ? col_to_hex(cc) : s_hex;  // render.rs:308-312
        const size_t i_frame = (size_t)y * P.width + x;
        if (i_frame == (size_t)P.width) px = 0u;  // render.rs:316-322: pixel i == width is never painted
        const size_t i_band = (size_t)out_row * P.width + x;
        if constexpr (!COUNT) {
            if (P.out) P.out[i_band] = px;
        }
        if constexpr (DBG) {
            if (P.hit_cube) P.hit_cube[i_band] = hit_c;
            if (P.hit_sphere) P.hit_sphere[i_band] = hit_s;
            if (P.tree_hash) P.tree_hash[i_band] = hmix(hmix(kHashBasis, hs), hc);
        }
    }
    if constexpr (COUNT) {
        // wave reduction, then one atomic per wave and counter
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            unsigned long long v = cnt.v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0u && v != 0ull) atomicAdd(&P.counters[k], v);
        }
    }
}

}  // namespace

int launch_render(const Params& p, int variant, void* stream)
{
    const uint32_t rows = p.y1 - p.y0;
    const uint32_t blocks = (rows + kRowBlock - 1u) / kRowBlock;
    if (p.n_parts == 0u || p.part >= p.n_parts) return (int)hipErrorInvalidValue;
    if (p.part >= blocks) return (int)hipSuccess;  // this part owns no rows
    const uint32_t my_blocks = (blocks - p.part + p.n_parts - 1u) / p.n_parts;
    dim3 grid((p.width + 15u) / 16u, my_blocks, 1u);
    dim3 block(256u, 1u, 1u);
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
    case kVariantDebug: hipLaunchKernelGGL((render_kernel<true, false>), grid, block, 0, s, p); break;
    case kVariantCount: hipLaunchKernelGGL((render_kernel<false, true>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((render_kernel<false, false>), grid, block, 0, s, p); break;
    }
    return (int)hipGetLastError();
}

}  // namespace rtx950
