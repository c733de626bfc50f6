// rtx950_kernels.hip — the per-pixel ray-cast hot path as one gfx950 kernel.
//
// Replaces (reference = ElrohirGT/Raytracing, paths relative to its root):
//   render()            src/render.rs:234-324   -> render_kernel (one lane per pixel, 8x8 px per wave)
//   cast_ray()          src/render.rs:73-229    -> trace_tree (explicit depth-first walk, one copy of the node
//                                                 code, per-lane frames of waiting nodes in LDS) + light_term
//   cast_shadow()       src/render.rs:44-70     -> shadow_query (inlined, list order, early exit)
//   the slowest tiles of a frame                -> render_heavy_kernel (three waves per tile on a second stream:
//                                                 same functions, same arithmetic; see "heavy tiles")
//   cast_ray()'s recursion, level by level      -> wf_trace_kernel / wf_combine_kernel (opt-in wavefront path: one launch
//                                                 per ray-tree level over bundles of coherent rays; same functions)
//   Cube::ray_intersect src/cube.rs:77-218      -> cull_test (one conservative test per cube for all 64
//                                                 rays of a wave) + cube_slab_approx (per-ray two-sided
//                                                 filter, FMA) + cube_exact (IEEE, only where needed)
//                                                 + cube_resolve (face/uv for the winning hit only)
//   Sphere::ray_intersect src/sphere.rs:25-56   -> sphere_exact
//   Texture::get_color_of_face src/texture.rs:178-196 -> texel_of_face (packed atlas gather)
//   Color ops           src/color.rs:99-149     -> col_* on floats that hold exact u8 values
//
// Exactness contract.  This file is compiled with -ffp-contract=off, and HIP's default correctly
// rounded f32 divide/sqrt stay on, so every expression below that restates a reference
// expression rounds exactly like the reference's f32 arithmetic.  FMA is used ONLY in cull_test
// and cube_slab_approx, whose values never reach an output: with explicit error bounds they only
// prove "this cube cannot pass the slab test" or "it passes, and it is / is not the nearest by a
// margin"; whatever they cannot prove goes through cube_exact, and every distance that is used
// (hit point, shadow attenuation) is recomputed by cube_exact for the selected cube.
//
// Wave-cooperative traversal.  A wave owns an 8x8 pixel tile.  For each object-list traversal lane j
// tests cube base+j against bounds of ALL the wave's rays (one coalesced 1 KB load + 18 FMAs per 64
// cubes), a ballot yields the candidate bitmask, and only candidates are examined per ray, with the
// cube broadcast from its lane by v_readlane (SGPR operands, no memory round trip, no LDS).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#ifndef RTX_WF_GRID
#define RTX_WF_GRID 8192u
#endif

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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
// ---- correctly rounded a / d for several numerators over one denominator
// hipcc expands an IEEE f32 division into v_div_scale x2, v_rcp_f32, this chain and v_div_fixup:
//     e = fma(-d, r0, 1)   r = fma(e, r0, r0)              (one Newton step on r0 = v_rcp_f32(d))
//     q0 = a r   q1 = fma(fma(-d, q0, a), r, q0)   q = fma(fma(-d, q1, a), r, q1)
// v_div_scale / v_div_fixup only act on operands near the ends of the exponent range, on zeros,
// infinities and NaN.  For d and a of magnitude in [2^-60, 2^60] (a may also be +-0) the chain alone
// IS the compiled division, bit for bit (rt_selftest_division checks that on the device), and r can
// be shared by every numerator over the same d.  Operands outside that range take the plain `/`.
DEV bool div_in_range(float v)
{
    const float a = __builtin_fabsf(v);
    return a >= 0x1p-60f && a <= 0x1p60f;
}
DEV bool num_in_range(float v) { return v == 0.0f || div_in_range(v); }
DEV float rcp_refined(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
DEV float div_shared(float a, float d, float r)
{
    const float q0 = a * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, a), r, q0);
    const float q2 = __builtin_fmaf(__builtin_fmaf(-d, q1, a), r, q1);
    return (a == 0.0f) ? q0 : q2;  // +-0 / d keeps the sign rule of the product
}
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
// Executed-work events (count variant only): what the kernel actually evaluates, per lane, as opposed to
// the reference's brute-force tests above.  bench.py turns them into executed FP32 operations with the
// per-event costs listed in DESIGN.md ("Roofline").
enum { EV_CULL = 0, EV_CHUNK, EV_CAND, EV_EXACT, EV_BVH, EV_SPHERE, EV_QUERY, EV_LIGHT, EV_RESOLVE, EV_BLOCKER, EV_CHILD, EV_PRIMARY, EV_COUNT };
struct Counters {  // per ray-tree node
    uint32_t nearest_rays, shadow_rays, cube_tests, sphere_tests, shaded_hits;
    uint32_t ev[EV_COUNT];
};
struct Totals {  // per lane: [0..4] unique work, [5..9] the reference's n_lights^depth-weighted work, [10..] executed events
    unsigned long long v[10 + EV_COUNT];
};
DEV void totals_add(Totals& t, const Counters& c, unsigned long long w)
{
    const uint32_t a[5] = {c.nearest_rays, c.shadow_rays, c.cube_tests, c.sphere_tests, c.shaded_hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        t.v[k] += a[k];
        t.v[5 + k] += (unsigned long long)a[k] * w;
    }
#pragma unroll
    for (int k = 0; k < EV_COUNT; ++k) t.v[10 + k] += c.ev[k];
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
//   lo = max3(near), hi = min3(far); decisions on lo - hi against +-tol (cube_slab_approx)
// Error bound, u = 2^-24, no overflow/underflow (dir_ok/org_ok/fast_ok guarantee the ranges):
//   reference  t = RN(RN(RN(k -+ h) - o)/d)            |t  - T| <= 3u (B+|o|)|r| (1+O(u)),  T exact
//   filter     t~ = RN(fma(h,-+a, RN(fma(k,r,-RN(o r)))))  |t~ - T| <= 5u (B+|o|)|r| (1+O(u))
//              with r = v_rcp_f32(d), 1 ulp = 2u relative                +2u (B+|o|)|r|
// with B >= |k|+h the largest bound magnitude in the scene.  lo = max3 and hi = min3 each move by at
// most the largest per-axis error E = 10u max_c (B+|o_c|)|r_c|, so lo~ - hi~ > 2E implies lo > hi, which
// is exactly when the reference's staged slab test (cube.rs:101,113) returns None.  tol = 32u max_c(...)
// = 2E: twice the requirement.  Rays outside the safe ranges get tol = +inf (every cube is a
// candidate), and the negated compare makes NaN a candidate too.
struct RayPre {
    f32x2 r_xy, c_xy, na_xy;  // (r_x,r_y), (c_x,c_y), (-a_x,-a_y)
    f32x2 rc_z;               // (r_z, c_z)
    f32x2 a_z_pm;             // (-a_z, +a_z)
    float tol;                // 2E: decision threshold of the two-sided filter (+inf for unsafe rays)
    float err;                // E: bound on |approximate - reference| for tmin, tmax and the distance
    bool unsafe;              // outside the ranges the error analysis assumes: always evaluated exactly
    uint32_t zbits;           // safe rays only: bit c set = direction component c is exactly zero
    float zdelta;             // margin of zero_axis_check
    bool div_ok;              // operands of the slab quotients are in div_shared's range (cube_distance)
    bool wild;                // non-finite or huge components: even the wave-level cull keeps every cube
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
    // A direction component that is exactly +-0 (the image's centre column/row, axis-parallel
    // reflections) is handled apart: the reference's two t of that axis are +-inf (cube.rs:90-109), so
    // the axis either does not constrain the interval (origin strictly inside the slab) or makes the
    // staged compares fail (strictly outside).  The filter takes "does not constrain" (r = c = 0,
    // a = 2^120: near = -huge, far = +huge) and cube_query decides inside / outside / too close to call
    // from |k - o| against h -+ zdelta (zero_axis_check).
    const bool zx = (d.x == 0.0f), zy = (d.y == 0.0f), zz = (d.z == 0.0f);
    const bool safe = (P.fast_ok != 0u) && (dir_ok(d.x) || zx) && (dir_ok(d.y) || zy) && (dir_ok(d.z) || zz) && !(zx && zy && zz) &&
                      org_ok(o.x) && org_ok(o.y) && org_ok(o.z);
    // v_rcp_f32: 1 ulp (2u relative) instead of the correctly rounded quotient; the bound below budgets for it
    const float big = 0x1p120f;
    const float rx = zx ? 0.0f : __builtin_amdgcn_rcpf(d.x), ry = zy ? 0.0f : __builtin_amdgcn_rcpf(d.y), rz = zz ? 0.0f : __builtin_amdgcn_rcpf(d.z);
    const float ax = zx ? big : __builtin_fabsf(rx), ay = zy ? big : __builtin_fabsf(ry), az = zz ? big : __builtin_fabsf(rz);
    q.r_xy = f32x2{rx, ry};
    q.c_xy = f32x2{-(o.x * rx), -(o.y * ry)};
    q.na_xy = f32x2{-ax, -ay};
    q.rc_z = f32x2{rz, -(o.z * rz)};
    q.a_z_pm = f32x2{-az, az};
    const float ex = zx ? 0.0f : (P.bound_max + __builtin_fabsf(o.x)) * ax;
    const float ey = zy ? 0.0f : (P.bound_max + __builtin_fabsf(o.y)) * ay;
    const float ez = zz ? 0.0f : (P.bound_max + __builtin_fabsf(o.z)) * az;
    q.tol = safe ? 0x1p-19f * __builtin_fmaxf(ex, __builtin_fmaxf(ey, ez)) : __builtin_inff();
    q.err = 0.5f * q.tol;
    q.unsafe = !safe;
    q.zbits = safe ? ((zx ? 1u : 0u) | (zy ? 2u : 0u) | (zz ? 4u : 0u)) : 0u;
    const float bigo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
    // 2^-20 (B + |o|) = 16 ulp of the largest magnitude involved: RN(k -+ h), RN(k - o) and the
    // reference's RN(bound - o) each err by at most one
    q.zdelta = 0x1p-20f * (P.bound_max + bigo);
    q.div_ok = safe && !(zx || zy || zz);
    // The cull (cull_test) has no division, so zero or tiny direction components are fine there: a ray
    // whose exact line misses a box by the slack margin is rejected by the reference too, also through
    // its +-inf arithmetic (cube.rs:90-113 with d_c = 0: both t of that axis are the same infinity, and
    // one of the staged compares fires).  Only non-finite or huge inputs need the keep-everything path
    // (a NaN ray passes every slab compare of the reference, which matters for shadows).
    const float bigd = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const bool finite = (d.x == d.x) && (d.y == d.y) && (d.z == d.z) && (o.x == o.x) && (o.y == o.y) && (o.z == o.z);
    q.wild = !(finite && bigd <= 0x1p40f && bigo <= P.origin_bound) || (P.fast_ok == 0u);
    return q;
}

// Zero direction components of a safe ray against cube (k, h).  miss: the origin is outside the slab
// of such an axis by more than zdelta, so the reference's slab test fails (both t are the same
// infinity).  unsure: within zdelta of a slab face (includes h <= zdelta): evaluate exactly.
// Otherwise the origin is strictly inside every such slab and the axis does not constrain.
DEV void zero_axis_check(const RayPre& q, V3 o, float kx, float ky, float kz, float h, bool& miss, bool& unsure)
{
    const float sx = __builtin_fabsf(kx - o.x), sy = __builtin_fabsf(ky - o.y), sz = __builtin_fabsf(kz - o.z);
    const float hin = h - q.zdelta, hout = h + q.zdelta;
    const bool zx = (q.zbits & 1u) != 0u, zy = (q.zbits & 2u) != 0u, zz = (q.zbits & 4u) != 0u;
    miss = (zx && sx > hout) || (zy && sy > hout) || (zz && sz > hout);
    const bool inside = (!zx || sx < hin) && (!zy || sy < hin) && (!zz || sz < hin);
    unsure = !miss && !inside;
}

// (kx,ky,kz,h) = cube centre and half size, wave-uniform.  lo ~ tmin and hi ~ tmax of cube.rs:142-143,
// each within q.err of the reference's value when the ray is safe.  Two-sided use:
//   lo - hi >  tol  -> the reference's slab test fails (proven miss)
//   lo - hi < -tol  -> it passes, and distance = (lo<0 ? hi : lo) is within err if also |lo| > err
//   otherwise       -> undecided: evaluate cube_exact
DEV void cube_slab_approx(float kx, float ky, float kz, float h, const RayPre& q, float& lo, float& hi)
{
    const f32x2 k_xy = f32x2{kx, ky};
    const f32x2 hh = f32x2{h, h};
    const f32x2 m_xy = __builtin_elementwise_fma(k_xy, q.r_xy, q.c_xy);
    const float m_z = __builtin_fmaf(kz, q.rc_z[0], q.rc_z[1]);
    const f32x2 near_xy = __builtin_elementwise_fma(hh, q.na_xy, m_xy);
    const f32x2 far_xy = __builtin_elementwise_fma(hh, -q.na_xy, m_xy);
    const f32x2 nf_z = __builtin_elementwise_fma(hh, q.a_z_pm, f32x2{m_z, m_z});
    lo = __builtin_fmaxf(__builtin_fmaxf(near_xy[0], near_xy[1]), nf_z[0]);
    hi = __builtin_fminf(__builtin_fminf(far_xy[0], far_xy[1]), nf_z[1]);
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

// The distance of cube_exact for a cube that is known to pass the slab test, for a ray whose operands
// are in the range of div_shared (div_ok: a "safe" ray, see ray_pre, without zero direction components:
// |d_c| in [2^-40, 2^40], and every bound - o_c is 0 or of magnitude in [2^-53, 2^21], being a difference
// of two floats that are 0 or in [2^-30, 2^20]).  The same quotients in the same staging (three of the six
// unless some lane needs tmax); the two quotients of an axis share one refined reciprocal.  Other lanes
// take cube_exact.
DEV float cube_distance(const CubeRec& c, V3 o, V3 d, bool div_ok)
{
    const float rx = rcp_refined(d.x), ry = rcp_refined(d.y), rz = rcp_refined(d.z);
    // Per axis the reference's smaller quotient is the one whose numerator lies on the near side: for
    // d > 0 the smaller numerator, for d < 0 the larger (division by d is monotone, and equal numerators
    // give equal quotients, so which of two equal ones is taken changes no bit).  The three far quotients
    // are only needed by rays that start inside the slab intersection (tmin < 0): wave-uniformly rare.
    const float n0x = c.minx - o.x, n1x = c.maxx - o.x, n0y = c.miny - o.y, n1y = c.maxy - o.y, n0z = c.minz - o.z, n1z = c.maxz - o.z;
    const bool lx = (n0x < n1x) == (d.x > 0.0f), ly = (n0y < n1y) == (d.y > 0.0f), lz = (n0z < n1z) == (d.z > 0.0f);
    float tmin = div_shared(lx ? n0x : n1x, d.x, rx);
    tmin = __builtin_fmaxf(tmin, div_shared(ly ? n0y : n1y, d.y, ry));
    tmin = __builtin_fmaxf(tmin, div_shared(lz ? n0z : n1z, d.z, rz));
    float distance = tmin;
    if (__builtin_amdgcn_ballot_w64(tmin < 0.0f) != 0ull) {
        float tmax = div_shared(lx ? n1x : n0x, d.x, rx);
        tmax = __builtin_fminf(tmax, div_shared(ly ? n1y : n0y, d.y, ry));
        tmax = __builtin_fminf(tmax, div_shared(lz ? n1z : n0z, d.z, rz));
        distance = (tmin < 0.0f) ? tmax : tmin;
    }
    if (__builtin_amdgcn_ballot_w64(!div_ok) != 0ull) {
        float e = 0.0f;
        (void)cube_exact(c, o, d, e);
        distance = div_ok ? distance : e;
    }
    return distance;
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
DEV Hit cube_resolve(const Params& P, uint32_t idx, V3 o, V3 d, bool div_ok)
{
    const CubeRec c = P.cubes[idx];  // per-lane gather: lanes may have different winners
    const float distance = cube_distance(c, o, d, div_ok);  // the winner's reference distance (cube.rs:90-145), once per ray
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
DEV Hit sphere_resolve(const Params& P, uint32_t idx, V3 o, V3 d)
{
    const SphereRec s = P.spheres[idx];
    float distance = 0.0f;
    (void)sphere_exact(s, o, d, distance);
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
    case FACE_BOTTOM: ox = sp; oy = P.sprite_x3; break;
    case FACE_FORWARDS: ox = sp; oy = P.sprite_x0; break;
    case FACE_BACKWARDS: ox = sp; oy = P.sprite_x2; break;
    case FACE_LEFT: ox = 0.0f; oy = sp; break;
    case FACE_RIGHT: ox = P.sprite_x2; oy = sp; break;
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
DEV bool wave_any_early(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// Out of line on purpose.  Inlined, the f64 routine's thirty-five 32-bit halves of polynomial coefficients (f64
// operands cannot be literals) are hoisted out of every loop into the kernel's prologue and hold 35 of its 128
// vector registers from the first instruction to the last, for a path that one light iteration in ten takes.
__attribute__((noinline)) __device__ float pow_f64_rounded(float x, float y) { return (float)pow((double)x, (double)y); }
DEV float powf_like_libm(float x, float y)
{
    // exact special cases of powf first (most shaded points have a clamped base of exactly 0):
    //   powf(+0, y>0) = +0, powf(1, y) = 1, powf(x, 1) = x (also for NaN x)
    const bool zero = (x == 0.0f) && (y > 0.0f);
    const bool one = (x == 1.0f);
    const bool ident = (y == 1.0f);
    float r = zero ? 0.0f : (one ? 1.0f : x);
    bool full = !(zero || one || ident);
    // The only consumer is (light.color * albedo.1) * spec * intensity with u8 truncation after every
    // product (render.rs:149-152, color.rs:118-130): any spec < 1/255 turns a channel <= 255 into 0.
    // A 3-instruction estimate exp2(y*log2 x) (v_log_f32/v_exp_f32, 1 ulp each; relative error of the
    // result <= ~2^-22 * (1 + |y log2 x|) < 1e-3 for |y log2 x| < 2^12) proves that for most pixels, and
    // the estimate itself (< 2^-9) is then returned: it yields the same zeros as the exact value.
    const bool est_ok = full && x > 0.0f && x < 1.0f && y > 0.0f;
    if (wave_any_early(est_ok) && est_ok) {
        const float e = y * __builtin_amdgcn_logf(x);
        const float est = __builtin_amdgcn_exp2f(e);
        if (est < 0x1p-9f && e > -4096.0f) {
            r = est;
            full = false;
        } else if (!(e > -4096.0f)) {  // far below the smallest subnormal: powf underflows to +0 as well
            r = 0.0f;
            full = false;
        }
    }
    if (__builtin_amdgcn_ballot_w64(full) != 0ull) {
        if (full) r = pow_f64_rounded(x, y);
    }
    return r;
}

// ---------------------------------------------------------------- diagnostic build: per-phase cycle shares
// -DRTX_PROFILE adds s_memtime stamps around the phases of a traversal; the per-wave sums go to
// P.counters[16..] (memory no output depends on).  Never enabled in the shipped library.
#ifdef RTX_PROFILE
struct Prof { uint32_t t[10]; uint32_t last; uint32_t c[8]; uint32_t kind; uint32_t qk[8]; uint32_t ck[8]; };
__device__ Prof* g_prof_unused;
#define PROF_DECL Prof prof; for (int k_ = 0; k_ < 10; ++k_) prof.t[k_] = 0u; for (int k_ = 0; k_ < 8; ++k_) { prof.c[k_] = 0u; prof.qk[k_] = 0u; prof.ck[k_] = 0u; } prof.kind = 0u; prof.last = (uint32_t)__builtin_readcyclecounter(); const uint32_t prof_t0 = prof.last; const uint32_t prof_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#define PROF_ARG , Prof& prof
#define PROF_PASS , prof
#define PROF_MARK(k) do { uint32_t n_ = (uint32_t)__builtin_readcyclecounter(); prof.t[k] += n_ - prof.last; prof.last = n_; } while (0)
#define PROF_COUNT(k, n) do { prof.c[k] += (uint32_t)(n); } while (0)
#define PROF_KIND(k) do { prof.kind = (k); } while (0)
#define PROF_KCOUNT(arr, n) do { prof.arr[prof.kind & 7u] += (uint32_t)(n); } while (0)
#else
#define PROF_KIND(k) do { } while (0)
#define PROF_KCOUNT(arr, n) do { } while (0)
#define PROF_COUNT(k, n) do { } while (0)
#define PROF_DECL
#define PROF_ARG
#define PROF_PASS
#define PROF_MARK(k) do { } while (0)
#endif
enum { PH_RAYGEN = 0, PH_PRE = 1, PH_CULL = 2, PH_CAND = 3, PH_CONTESTED = 4, PH_RESOLVE = 5, PH_SHADE = 6, PH_SHADOW_POST = 7, PH_CHILDSETUP = 8, PH_OTHER = 9 };

// ---------------------------------------------------------------- wave-level helpers
// Everything from here on runs with all 64 lanes enabled: control flow depends only on wave-uniform
// conditions (ballots), per-lane state is carried in predicates (`active`, `hit`, `done`).  That is
// what lets one wave cull the cube list cooperatively (lane j examines cube base+j).
DEV unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEV bool wave_any(bool p) { return wave_ballot(p) != 0ull; }

template <int CTRL, int ROW_MASK>
DEV float dpp_f(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
DEV uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
DEV float readlane_f(float v, uint32_t lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)lane));
}
// max over the 64 lanes (all enabled), result wave-uniform.  6 DPP-modified v_max_f32: an inclusive
// scan inside each row of 16 (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals
// into lane 63.
DEV float wave_max(float v)
{
    const float id = -__builtin_inff();
    v = __builtin_fmaxf(v, dpp_f<0x111, 0xf>(id, v));
    v = __builtin_fmaxf(v, dpp_f<0x112, 0xf>(id, v));
    v = __builtin_fmaxf(v, dpp_f<0x114, 0xf>(id, v));
    v = __builtin_fmaxf(v, dpp_f<0x118, 0xf>(id, v));
    v = __builtin_fmaxf(v, dpp_f<0x142, 0xa>(id, v));
    v = __builtin_fmaxf(v, dpp_f<0x143, 0xc>(id, v));
    return readlane_f(v, 63u);
}

// max over the 64 lanes of an unsigned value (same sweep with v_max_u32; identity 0)
DEV uint32_t wave_max_u32(uint32_t v)
{
    v = __builtin_elementwise_max(v, dpp_u32<0x111, 0xf>(v));
    v = __builtin_elementwise_max(v, dpp_u32<0x112, 0xf>(v));
    v = __builtin_elementwise_max(v, dpp_u32<0x114, 0xf>(v));
    v = __builtin_elementwise_max(v, dpp_u32<0x118, 0xf>(v));
    v = __builtin_elementwise_max(v, dpp_u32<0x142, 0xa>(v));
    v = __builtin_elementwise_max(v, dpp_u32<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---------------------------------------------------------------- wave-cooperative cull
// A line o + t d misses the box (centre k, half size h) iff one of the three coordinate-plane
// separating-axis tests holds:  |(k_i-o_i) d_j - (k_j-o_j) d_i| > h (|d_i| + |d_j|)   (i,j) = xy, yz, zx.
// (This is the slab test of cube.rs:90-113 with the divisions multiplied out.)  With the line moment
// W_ij = o_i d_j - o_j d_i the left side is k_i d_j - k_j d_i - W_ij: linear in the cube centre with
// per-ray coefficients.  For a WAVE of rays, d and W are bounded by centre +- radius over the active
// lanes, which gives one conservative test per cube for all 64 rays; lane j evaluates it for cube
// base+j and a ballot turns the answers into a 64-bit candidate mask.
struct CullPre {  // wave-uniform
    float dxc, dyc, dzc;        // direction of the reference lane
    float wxyc, wyzc, wzxc;     // its moment
    float dr;                   // max over active lanes and axes of |d - dc|
    float wr;                   // same for the moment, plus rounding slack
    float wr_raw;               // ... without the slack
    float sxy, syz, szx;        // upper bounds of |d_i| + |d_j|
    float ax, ay, az;           // upper bounds of |d_i|
    bool all;                   // some active ray is non-finite or huge: keep every cube
};

// max over the 64 lanes of two non-negative floats.  (Rounds 1-2 packed the pair as two bfloat16 halves of a dword and
// reduced both with v_pk_max_u16: twelve instructions for the sweep plus six to pack, results up to 2^-7 above the
// true maxima.  VOP3P takes no DPP modifier, v_max_u32 does: two exact sweeps are shorter.)
DEV void wave_max2(float a, float b, float& out_a, float& out_b)
{
    // non-negative floats order like their bit patterns: two v_max_u32 sweeps (the DPP modifier folds into the
    // instruction), exact maxima.  A NaN's pattern is above every number's: it wins, and compares with it are false.
    out_a = __builtin_bit_cast(float, wave_max_u32(__builtin_bit_cast(uint32_t, a)));
    out_b = __builtin_bit_cast(float, wave_max_u32(__builtin_bit_cast(uint32_t, b)));
}

// origin_bound: an upper bound of every active lane's origin coordinates (Params::origin_bound; lanes
// beyond it are `wild`), used where the wave's largest origin magnitude used to be reduced.
DEV CullPre cull_pre(bool active, V3 o, V3 d, bool wild, float slack_k)
{
    CullPre c;
    const unsigned long long am = wave_ballot(active);
    // The reference lane: any active lane will do; one near the middle of the 8x8 tile (pixel (3,3), else (4,4),
    // else the first active lane from row 2 on) halves the radii of a full tile against its corner lane.
    const unsigned long long am_mid = am >> 20;
    const uint32_t ref = ((am >> 27) & 1ull) ? 27u : (((am >> 36) & 1ull) ? 36u : (am_mid ? 20u + (uint32_t)__builtin_ctzll(am_mid) : (am ? (uint32_t)__builtin_ctzll(am) : 0u)));
    c.all = wave_any(active && wild);
    // moments (only ever compared against bounds with slack: FMA is fine here)
    const float wxy = __builtin_fmaf(o.x, d.y, -(o.y * d.x)), wyz = __builtin_fmaf(o.y, d.z, -(o.z * d.y)), wzx = __builtin_fmaf(o.z, d.x, -(o.x * d.z));
    c.dxc = readlane_f(d.x, ref); c.dyc = readlane_f(d.y, ref); c.dzc = readlane_f(d.z, ref);
    c.wxyc = readlane_f(wxy, ref); c.wyzc = readlane_f(wyz, ref); c.wzxc = readlane_f(wzx, ref);
    const float z = 0.0f;
    // one sweep: radius of the directions and of the moments around the reference lane
    const float dr = __builtin_fmaxf(__builtin_fabsf(d.x - c.dxc), __builtin_fmaxf(__builtin_fabsf(d.y - c.dyc), __builtin_fabsf(d.z - c.dzc)));
    const float wr = __builtin_fmaxf(__builtin_fabsf(wxy - c.wxyc), __builtin_fmaxf(__builtin_fabsf(wyz - c.wyzc), __builtin_fabsf(wzx - c.wzxc)));
    float wrm;
    wave_max2(active ? dr : z, active ? wr : z, c.dr, wrm);
    const float ax = __builtin_fabsf(c.dxc) + c.dr, ay = __builtin_fabsf(c.dyc) + c.dr, az = __builtin_fabsf(c.dzc) + c.dr;
    c.sxy = ax + ay; c.syz = ay + az; c.szx = az + ax;
    c.ax = ax; c.ay = ay; c.az = az;
    // Rounding slack.  Sources (u = 2^-24, B = largest bound magnitude, O = Params::origin_bound, D = ax+ay+az >= |d_c|):
    //   this test's own fma chains and the per-lane moments      <= ~12 u (B + O) D
    //   the reference's rounded t values vs exact arithmetic      <= ~12 u (B + O) D   (see cube_slab_approx)
    // 2^-17 (B + O + 1) max(D, 1) = 128 u (...) covers both several times over.
    const float slack = slack_k * __builtin_fmaxf(ax + ay + az, 1.0f);  // slack_k = 2^-17 (B + O + 1): Params::cull_slack / sphere_cull_slack
    c.wr = wrm + slack;
    c.wr_raw = wrm;
    return c;
}

// lane-private cube (kx,ky,kz,h): true -> some active ray of the wave may pass the slab test
DEV bool cull_test(f32x4 k, const CullPre& c)
{
    const float akx = __builtin_fabsf(k[0]), aky = __builtin_fabsf(k[1]), akz = __builtin_fabsf(k[2]);
    const float gxy = __builtin_fmaf(k[0], c.dyc, __builtin_fmaf(-k[1], c.dxc, -c.wxyc));
    const float lxy = __builtin_fmaf(k[3], c.sxy, __builtin_fmaf(akx + aky, c.dr, c.wr));
    const float gyz = __builtin_fmaf(k[1], c.dzc, __builtin_fmaf(-k[2], c.dyc, -c.wyzc));
    const float lyz = __builtin_fmaf(k[3], c.syz, __builtin_fmaf(aky + akz, c.dr, c.wr));
    const float gzx = __builtin_fmaf(k[2], c.dxc, __builtin_fmaf(-k[0], c.dzc, -c.wzxc));
    const float lzx = __builtin_fmaf(k[3], c.szx, __builtin_fmaf(akz + akx, c.dr, c.wr));
    const bool sep = (__builtin_fabsf(gxy) > lxy) || (__builtin_fabsf(gyz) > lyz) || (__builtin_fabsf(gzx) > lzx);
    return c.all || !sep;  // NaN compares false -> candidate
}

// ---- the general form of the wave-level cull (render_heavy_kernel only)
// cull_pre bounds the wave's directions and moments by one radius each around one lane's values, which
// is as tight as it needs to be for a coherent bundle and costs one DPP sweep.  The heavy tiles' bundles are
// not coherent: shadow rays towards a light that floats a quarter of a cube in front of the surface fan out
// over half a radian within one tile, and with a corner lane as the centre and one radius for all axes the
// bound keeps 80 of the diorama's 97 cubes where the 64 lines meet 10.  This form takes the BOX of the
// lanes' values - per component, centre and half width from the minimum and the maximum (six sweeps) - and
// the moments about the reference lane's origin instead of the world's (W' = (o - c) x d, small when the
// origins are close together whatever the directions do; the cube centres move by -c with them):
//   |k'_i dc_j - k'_j dc_i - wc_ij|  >  h (a_i + a_j) + |k'_i| rd_j + |k'_j| rd_i + rw_ij   =>  every lane's line misses
// (the same coordinate-plane tests with d = dc + e, |e_i| <= rd_i, W' = wc + f, |f_ij| <= rw_ij, a_i = |dc_i| +
// rd_i).  Rounding: the half widths come out of wave_max2 rounded up; recentring adds errors of u (B + O) to
// the cube centres and the origins, which the slack of cull_pre covers many times over, as it does the
// rounding of the box centres.
struct CullPreG {  // wave-uniform
    float dxc, dyc, dzc;        // centre of the box of directions
    float wxyc, wyzc, wzxc;     // centre of the box of moments about (cx, cy, cz)
    float rdx, rdy, rdz;        // half widths of the direction box
    float rwxy, rwyz, rwzx;     // half widths of the moment box, plus rounding slack
    float ax, ay, az;           // upper bounds of |d_i|
    float cx, cy, cz;           // the reference lane's origin: cube centres and moments are taken about it
    float dr, wr_raw;           // largest half widths (traversal choice of long lists, SPLIT trigger)
    bool all;                   // some active ray is non-finite or huge: keep every cube
};
// centre and half width of the active lanes' values of v, as offsets from a reference value
DEV void wave_box(bool active, float v, float ref, float& centre, float& half)
{
    const float dlt = v - ref;
    float up, dn;
    wave_max2((active && dlt > 0.0f) ? dlt : 0.0f, (active && dlt < 0.0f) ? -dlt : 0.0f, up, dn);
    centre = ref + 0.5f * (up - dn);
    half = 0.5f * (up + dn);
}
DEV CullPreG cull_pre_g(bool active, V3 o, V3 d, bool wild, float slack_k)
{
    CullPreG c;
    const unsigned long long am = wave_ballot(active);
    const unsigned long long am_mid = am >> 20;
    const uint32_t ref = ((am >> 27) & 1ull) ? 27u : (((am >> 36) & 1ull) ? 36u : (am_mid ? 20u + (uint32_t)__builtin_ctzll(am_mid) : (am ? (uint32_t)__builtin_ctzll(am) : 0u)));
    c.all = wave_any(active && wild);
    c.cx = readlane_f(o.x, ref); c.cy = readlane_f(o.y, ref); c.cz = readlane_f(o.z, ref);
    const float ox = o.x - c.cx, oy = o.y - c.cy, oz = o.z - c.cz;
    const float wxy = __builtin_fmaf(ox, d.y, -(oy * d.x)), wyz = __builtin_fmaf(oy, d.z, -(oz * d.y)), wzx = __builtin_fmaf(oz, d.x, -(ox * d.z));
    wave_box(active, d.x, readlane_f(d.x, ref), c.dxc, c.rdx);
    wave_box(active, d.y, readlane_f(d.y, ref), c.dyc, c.rdy);
    wave_box(active, d.z, readlane_f(d.z, ref), c.dzc, c.rdz);
    float rxy, ryz, rzx;
    wave_box(active, wxy, 0.0f, c.wxyc, rxy);  // the reference lane's own moment about its origin is 0
    wave_box(active, wyz, 0.0f, c.wyzc, ryz);
    wave_box(active, wzx, 0.0f, c.wzxc, rzx);
    c.ax = __builtin_fabsf(c.dxc) + c.rdx; c.ay = __builtin_fabsf(c.dyc) + c.rdy; c.az = __builtin_fabsf(c.dzc) + c.rdz;
    const float slack = slack_k * __builtin_fmaxf(c.ax + c.ay + c.az, 1.0f);  // see cull_pre
    c.rwxy = rxy + slack; c.rwyz = ryz + slack; c.rwzx = rzx + slack;
    c.dr = __builtin_fmaxf(c.rdx, __builtin_fmaxf(c.rdy, c.rdz));
    c.wr_raw = __builtin_fmaxf(rxy, __builtin_fmaxf(ryz, rzx));
    return c;
}
DEV bool cull_test(f32x4 k, const CullPreG& c)
{
    const float kx = k[0] - c.cx, ky = k[1] - c.cy, kz = k[2] - c.cz;
    const float akx = __builtin_fabsf(kx), aky = __builtin_fabsf(ky), akz = __builtin_fabsf(kz);
    const float gxy = __builtin_fmaf(kx, c.dyc, __builtin_fmaf(-ky, c.dxc, -c.wxyc));
    const float lxy = __builtin_fmaf(k[3], c.ax + c.ay, __builtin_fmaf(akx, c.rdy, __builtin_fmaf(aky, c.rdx, c.rwxy)));
    const float gyz = __builtin_fmaf(ky, c.dzc, __builtin_fmaf(-kz, c.dyc, -c.wyzc));
    const float lyz = __builtin_fmaf(k[3], c.ay + c.az, __builtin_fmaf(aky, c.rdz, __builtin_fmaf(akz, c.rdy, c.rwyz)));
    const float gzx = __builtin_fmaf(kz, c.dxc, __builtin_fmaf(-kx, c.dzc, -c.wzxc));
    const float lzx = __builtin_fmaf(k[3], c.az + c.ax, __builtin_fmaf(akz, c.rdx, __builtin_fmaf(akx, c.rdz, c.rwzx)));
    const bool sep = (__builtin_fabsf(gxy) > lxy) || (__builtin_fabsf(gyz) > lyz) || (__builtin_fabsf(gzx) > lzx);
    return c.all || !sep;  // NaN compares false -> candidate
}

// ---------------------------------------------------------------- cube queries
enum { Q_NEAREST = 0, Q_NEAREST_EXACT = 1, Q_SHADOW = 2 };

struct QState {
    float b1, b2;       // Q_NEAREST: the two smallest approximate distances ...
    uint32_t i1;        //            ... and the cube index of the smallest
    float best;         // Q_NEAREST_EXACT: the reference's fold (render.rs:94-101) on exact distances
    uint32_t best_i;
    bool done;          // Q_SHADOW: a passing cube was found ...
    uint32_t blocker;   //           ... lowest list index so far (kNoHit: none)
    float tol;          // the ray's 2E (out)
    bool div_ok;        // the ray's RayPre::div_ok (out)
    // executed-work events of this query, per lane (read by the count variant only; dead code elsewhere)
    uint32_t ev_cull, ev_chunk, ev_cand, ev_exact, ev_bvh;
};

// One candidate cube (list index i, id cid, centre k, half size kh) against the wave's rays.  UNIFORM:
// the same cube for every lane (wave-cooperative passes); otherwise each lane has its own (BVH walk).
//   Q_NEAREST        approximate distances with error bound E for every cube the two-sided filter
//                    proves to pass; cubes it can neither prove nor refute are evaluated exactly on the
//                    spot (rare).  The caller decides the winner from (b1, b2, i1) and falls back to
//                    Q_NEAREST_EXACT for lanes where b2 - b1 <= 2E ("contested": duplicates, ties).
//   Q_NEAREST_EXACT  exact evaluation of every candidate for the lanes in `active`; the reference's
//                    strict-< fold over the list (render.rs:94-101) = smallest distance, lowest index
//                    among equals, which is how it is evaluated here in any visiting order.
//   Q_SHADOW         render.rs:44-70: the passing cube with the lowest list index among those whose id
//                    differs from the impact object's (render.rs:113); again in any visiting order.
// SLAB: the caller has the cube's approximate interval already (cube_query_bvh: a leaf's box IS its cube, and
// box_slab2's interval of it is cube_slab_approx's, operation for operation).
template <int MODE, bool UNIFORM, bool SLAB = false>
DEV void cube_candidate(const Params& P, const RayPre& q, bool active, bool any_zero, V3 o, V3 d, uint32_t impact_id, uint32_t i, uint32_t cid,
                        float kx, float ky, float kz, float kh, QState& st PROF_ARG, float slab_lo = 0.0f, float slab_hi = 0.0f)
{
    float lo = slab_lo, hi = slab_hi;
    if constexpr (!SLAB) cube_slab_approx(kx, ky, kz, kh, q, lo, hi);
    const float gap = lo - hi;
    bool live = active && !(gap > q.tol);  // not proven to miss
    if constexpr (MODE == Q_SHADOW) live = live && (i < st.blocker) && (cid != impact_id);
    bool sure = gap < -q.tol;  // proven to pass
    if constexpr (MODE == Q_NEAREST) sure = sure && (__builtin_fabsf(lo) > q.err);  // and the sign of tmin is certain
    if constexpr (MODE == Q_NEAREST_EXACT) sure = false;
    if (any_zero) {
        bool zmiss, zunsure;
        zero_axis_check(q, o, kx, ky, kz, kh, zmiss, zunsure);
        live = live && !zmiss;
        sure = sure && !zunsure;
    }
    const bool amb = live && !sure;
    float dist = (lo < 0.0f) ? hi : lo;  // cube.rs:145 on the approximate interval
    st.ev_cand += active ? 1u : 0u;
    st.ev_exact += amb ? 1u : 0u;
    PROF_COUNT(0, 1);
    PROF_KCOUNT(ck, 1);
    PROF_COUNT(1, wave_any(amb) ? 1 : 0);
    PROF_COUNT(2, __builtin_popcountll(wave_ballot(amb && q.unsafe)));
    PROF_COUNT(3, __builtin_popcountll(wave_ballot(amb && !q.unsafe && !(gap < -q.tol))));
    PROF_COUNT(4, __builtin_popcountll(wave_ballot(amb && !q.unsafe && (gap < -q.tol))));
    PROF_COUNT(5, __builtin_popcountll(wave_ballot(live)));
    if (wave_any(amb)) {
        if (amb) {
            CubeRec c;
            if constexpr (UNIFORM) c = load_cube_uniform(P.cubes, i);
            else c = P.cubes[i];
            float ed;
            const bool pass = cube_exact(c, o, d, ed);
            live = pass && (ed == ed);  // a NaN distance is Some but never wins (render.rs:95)
            if constexpr (MODE == Q_SHADOW) live = pass;
            dist = ed;
        }
    }
    if constexpr (MODE == Q_NEAREST) {
        const bool first = live && (dist < st.b1);
        const float nb2 = first ? st.b1 : __builtin_fminf(st.b2, dist);
        st.b2 = live ? nb2 : st.b2;
        st.b1 = first ? dist : st.b1;
        st.i1 = first ? i : st.i1;
    } else if constexpr (MODE == Q_NEAREST_EXACT) {
        const bool win = live && ((dist < st.best) || (dist == st.best && i < st.best_i));
        st.best = win ? dist : st.best;
        st.best_i = win ? i : st.best_i;
    } else {
        st.blocker = live ? i : st.blocker;  // live implies i < blocker
        st.done = st.done || live;
    }
}

// One pass over a short cube list (at most kSmallChunks chunks of 64, in list order) for the whole wave.
constexpr uint32_t kRefineMin = 20u;  // REFINE: candidates of a chunk from which on the box form of the cull is applied as well (12..32 measured equal)
template <int MODE, typename CP, bool REFINE>
DEV void cube_query_small(const Params& P, const RayPre& q, const CP& cp, bool active, V3 o, V3 d, uint32_t impact_id, QState& st PROF_ARG)
{
    const uint32_t n = P.n_cubes;
    const bool any_zero = wave_any(active && q.zbits != 0u);
    PROF_COUNT(6, 1);
    PROF_KCOUNT(qk, 1);
    PROF_COUNT(7, __builtin_popcountll(wave_ballot(active)));
    PROF_MARK(PH_PRE);
    const uint32_t lane = threadIdx.x & 63u;
    const f32x4* filt = (const f32x4*)P.cube_filt;
    f32x4 rec = filt[lane];
    bool all_done = false;
    for (uint32_t base = 0; base < n && !all_done; base += 64u) {
        const uint32_t nb = base + 64u;
        const f32x4 nxt = filt[(nb < n ? nb : base) + lane];  // prefetch (array is padded to a multiple of 64)
        uint32_t ids = 0u;
        if constexpr (MODE == Q_SHADOW) ids = P.cube_id[base + lane];
        unsigned long long M = wave_ballot(cull_test(rec, cp) && (base + lane < n));
        st.ev_cull += (base + lane < n) ? 1u : 0u;
        // The one-radius cull is loose for a fan of rays (shadow rays towards a light close to the surface, a tile
        // across two faces): where it leaves many candidates, the box form of the cull (cull_pre_g: six sweeps
        // instead of one) is worth its price and is applied to the same records.  Its values die here: the candidate
        // loop below sees a smaller mask and nothing else.
        if constexpr (REFINE && !std::is_same<CP, CullPreG>::value) {
            if ((uint32_t)__builtin_popcountll(M) >= kRefineMin && !cp.all) {
                const CullPreG cg = cull_pre_g(active, o, d, q.wild, P.cull_slack);
                M &= wave_ballot(cull_test(rec, cg));
                st.ev_cull += (base + lane < n) ? 1u : 0u;
            }
        }
        PROF_MARK(PH_CULL);
        while (M != 0ull) {
            const uint32_t b = (uint32_t)__builtin_ctzll(M);
            M &= M - 1ull;
            const float kx = readlane_f(rec[0], b), ky = readlane_f(rec[1], b), kz = readlane_f(rec[2], b), kh = readlane_f(rec[3], b);
            uint32_t cid = 0u;
            if constexpr (MODE == Q_SHADOW) cid = (uint32_t)__builtin_amdgcn_readlane((int)ids, (int)b);
            cube_candidate<MODE, true>(P, q, active, any_zero, o, d, impact_id, base + b, cid, kx, ky, kz, kh, st PROF_PASS);
            if constexpr (MODE == Q_SHADOW) {
                // list order: the first passing cube is the answer; stop when every ray has one
                if (!wave_any(active && !st.done)) {
                    all_done = true;
                    M = 0ull;
                }
            }
        }
        PROF_MARK(PH_CAND);
        rec = nxt;
    }
}

// lane-private chunk box (centre, three half extents): may some active ray of the wave meet it?
DEV bool chunk_test(f32x4 k, f32x4 h, const CullPre& c)
{
    const float akx = __builtin_fabsf(k[0]), aky = __builtin_fabsf(k[1]), akz = __builtin_fabsf(k[2]);
    const float gxy = __builtin_fmaf(k[0], c.dyc, __builtin_fmaf(-k[1], c.dxc, -c.wxyc));
    const float lxy = __builtin_fmaf(h[0], c.ay, __builtin_fmaf(h[1], c.ax, __builtin_fmaf(akx + aky, c.dr, c.wr)));
    const float gyz = __builtin_fmaf(k[1], c.dzc, __builtin_fmaf(-k[2], c.dyc, -c.wyzc));
    const float lyz = __builtin_fmaf(h[1], c.az, __builtin_fmaf(h[2], c.ay, __builtin_fmaf(aky + akz, c.dr, c.wr)));
    const float gzx = __builtin_fmaf(k[2], c.dxc, __builtin_fmaf(-k[0], c.dzc, -c.wzxc));
    const float lzx = __builtin_fmaf(h[2], c.ax, __builtin_fmaf(h[0], c.az, __builtin_fmaf(akz + akx, c.dr, c.wr)));
    const bool sep = (__builtin_fabsf(gxy) > lxy) || (__builtin_fabsf(gyz) > lyz) || (__builtin_fabsf(gzx) > lzx);
    return c.all || !sep;
}

DEV bool chunk_test(f32x4 k, f32x4 h, const CullPreG& c)
{
    const float kx = k[0] - c.cx, ky = k[1] - c.cy, kz = k[2] - c.cz;
    const float akx = __builtin_fabsf(kx), aky = __builtin_fabsf(ky), akz = __builtin_fabsf(kz);
    const float gxy = __builtin_fmaf(kx, c.dyc, __builtin_fmaf(-ky, c.dxc, -c.wxyc));
    const float lxy = __builtin_fmaf(h[0], c.ay, __builtin_fmaf(h[1], c.ax, __builtin_fmaf(akx, c.rdy, __builtin_fmaf(aky, c.rdx, c.rwxy))));
    const float gyz = __builtin_fmaf(ky, c.dzc, __builtin_fmaf(-kz, c.dyc, -c.wyzc));
    const float lyz = __builtin_fmaf(h[1], c.az, __builtin_fmaf(h[2], c.ay, __builtin_fmaf(aky, c.rdz, __builtin_fmaf(akz, c.rdy, c.rwyz))));
    const float gzx = __builtin_fmaf(kz, c.dxc, __builtin_fmaf(-kx, c.dzc, -c.wzxc));
    const float lzx = __builtin_fmaf(h[2], c.ax, __builtin_fmaf(h[0], c.az, __builtin_fmaf(akz, c.rdx, __builtin_fmaf(akx, c.rdz, c.rwzx))));
    const bool sep = (__builtin_fabsf(gxy) > lxy) || (__builtin_fabsf(gyz) > lyz) || (__builtin_fabsf(gzx) > lzx);
    return c.all || !sep;
}

// One pass over a long cube list.  The cull stream of such a scene is in Morton order of the cube centres
// (rt_scene_upload), so a chunk of 64 consecutive records is spatially compact and has a tight bounding
// box: lane j first tests chunk j's box against the wave's rays (same separating-axis test, three half
// extents), and only chunks that pass are loaded and culled cube by cube.  Candidates therefore arrive in
// stream order, not list order; cube_perm gives each its list index, and cube_candidate's folds do not
// depend on the visiting order.  (The reference walks all N cubes per ray; 10k cubes = 157 chunks, of
// which a coherent bundle meets a handful.)
template <int MODE, typename CP>
DEV void cube_query_large(const Params& P, const RayPre& q, const CP& cp, bool active, V3 o, V3 d, uint32_t impact_id, QState& st PROF_ARG)
{
    const uint32_t n = P.n_cubes, n_chunks = P.n_chunks;
    const bool any_zero = wave_any(active && q.zbits != 0u);
    PROF_COUNT(6, 1);
    PROF_KCOUNT(qk, 1);
    PROF_MARK(PH_PRE);
    const uint32_t lane = threadIdx.x & 63u;
    // Shadow queries of a list whose order is spatially compact walk the LIST-ordered stream: chunks and
    // records then come in ascending list index, and the walk ends with the first chunk after which every
    // active ray has a blocker (a later chunk only holds higher indices).
    bool in_list_order = false;
    if constexpr (MODE == Q_SHADOW) in_list_order = P.shadow_list != 0u;
    const f32x4* filt = (const f32x4*)(in_list_order ? P.cube_filt_list : P.cube_filt);
    const f32x4* boxes = (const f32x4*)(in_list_order ? P.chunk_box_list : P.chunk_box);
    const uint32_t* id_stream = in_list_order ? P.cube_id_list : P.cube_id;
    for (uint32_t cg = 0; cg < n_chunks; cg += 64u) {
        const f32x4 bk = boxes[(size_t)(cg + lane) * 2u], bh = boxes[(size_t)(cg + lane) * 2u + 1u];  // padded to a multiple of 64 chunks
        unsigned long long CM = wave_ballot(chunk_test(bk, bh, cp) && (cg + lane < n_chunks));
        st.ev_chunk += (cg + lane < n_chunks) ? 1u : 0u;
        const float bk3 = bk[3];  // scalar copy first: see the note at u2f
        const uint32_t chunk_first = __builtin_bit_cast(uint32_t, bk3);  // lowest list index in lane's chunk
        bool refresh = true;  // Q_SHADOW: some lane's blocker may have changed since the chunks were last screened
        while (CM != 0ull) {
            // Chunks that cannot change any lane's result are not loaded.  Shadow: only the lowest list index
            // counts, so a chunk whose lowest index is not below every active lane's blocker is done with.
            // Nearest: every cube of a chunk has distance >= the chunk box's entry parameter (negative ones
            // included), as in the BVH walk below.
            if constexpr (MODE == Q_SHADOW) {
                if (refresh) {
                    const uint32_t need = wave_max_u32(active ? st.blocker : 0u);  // kNoHit while a lane has none
                    CM &= wave_ballot(chunk_first < need);
                    refresh = false;
                    if (CM == 0ull) break;
                }
            }
            const uint32_t c = (uint32_t)__builtin_ctzll(CM);
            CM &= CM - 1ull;
            if constexpr (MODE != Q_SHADOW) {
                const float cx = readlane_f(bk[0], c), cy = readlane_f(bk[1], c), cz = readlane_f(bk[2], c);
                const float hx = readlane_f(bh[0], c), hy = readlane_f(bh[1], c), hz = readlane_f(bh[2], c);
                const float nx = __builtin_fmaf(hx, q.na_xy[0], __builtin_fmaf(cx, q.r_xy[0], q.c_xy[0]));
                const float ny = __builtin_fmaf(hy, q.na_xy[1], __builtin_fmaf(cy, q.r_xy[1], q.c_xy[1]));
                const float nz = __builtin_fmaf(hz, q.a_z_pm[0], __builtin_fmaf(cz, q.rc_z[0], q.rc_z[1]));
                const float lo = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
                bool open;  // NaN and unsafe rays (tol = inf) compare false: not pruned
                if constexpr (MODE == Q_NEAREST) open = !(lo > st.b1 + 2.0f * q.tol);
                else open = !(lo - q.tol > st.best);
                if (!wave_any(active && open)) continue;
            }
            const uint32_t base = (cg + c) * 64u;
            const f32x4 rec = filt[base + lane];
            const uint32_t perm = in_list_order ? base + lane : P.cube_perm[base + lane];
            uint32_t ids = 0u;
            if constexpr (MODE == Q_SHADOW) ids = id_stream[base + lane];
            unsigned long long M = wave_ballot(cull_test(rec, cp) && (base + lane < n));
            st.ev_cull += (base + lane < n) ? 1u : 0u;
            PROF_MARK(PH_CULL);
            while (M != 0ull) {
                const uint32_t b = (uint32_t)__builtin_ctzll(M);
                M &= M - 1ull;
                const float kx = readlane_f(rec[0], b), ky = readlane_f(rec[1], b), kz = readlane_f(rec[2], b), kh = readlane_f(rec[3], b);
                const uint32_t i = (uint32_t)__builtin_amdgcn_readlane((int)perm, (int)b);
                uint32_t cid = 0u;
                if constexpr (MODE == Q_SHADOW) cid = (uint32_t)__builtin_amdgcn_readlane((int)ids, (int)b);
                cube_candidate<MODE, true>(P, q, active, any_zero, o, d, impact_id, i, cid, kx, ky, kz, kh, st PROF_PASS);
                refresh = true;
                if constexpr (MODE == Q_SHADOW) {
                    if (in_list_order && !wave_any(active && !st.done)) return;  // ascending indices: nothing later can be lower
                }
            }
            PROF_MARK(PH_CAND);
        }
    }
}

// One pass for an INCOHERENT bundle (an 8x8 tile across cube edges after a bounce: lanes going every
// way).  The wave-level cull keeps nearly every cube for such a wave (measured: all 97 cubes of the
// diorama, thousands of the 10k-cube world, and these waves are the frame's critical path), so each
// lane walks the BVH with its own line instead.  (Splitting the wave into a few coherent sub-bundles was
// tried first: a fifth fewer candidates, 8 % slower.)
//
// A lane's walk is a chain of dependent loads, 200-300 of them per walk in a stackless depth-first
// order over single nodes, and it cannot stop early: the reference's slab test is a LINE test, a line
// through solid terrain meets a hundred cubes, and only one of them is the answer.  So the walk is
// ordered: a step fetches a PAIR of sibling boxes (one 64-byte record), tests both with the slab filter
// (box half extents, the ray's error bound: proven miss only), handles the leaves among them (a leaf is one
// cube and carries its record: centre, half size, list index, id), and goes
// on with the child that can hold the answer sooner - the smaller entry parameter for a nearest hit,
// the lower list index for a shadow - deferring the other on a per-lane stack in LDS.  What the answer
// so far rules out is dropped when fetched: boxes whose entry parameter lies beyond the best distance
// (every cube in a box has distance >= the box's entry parameter, negative ones included; E = tol/2 as
// for cubes, the box lies within the scene bounds), subtrees whose lowest list index is not below the
// blocker.  Unsafe rays have tol = inf and NaN compares false: they visit everything.
constexpr uint32_t kWgWaves = 3;  // waves of a render_heavy_kernel workgroup (one tracer, two shaders for a heavy tile)
extern __shared__ uint32_t dyn_lds[];  // [wave][level][lane], Params::bvh_levels levels: launch_render / launch_heavy size it
template <uint32_t WAVES>
DEV uint32_t* bvh_stack(const Params& P)
{
    if constexpr (WAVES == 1u) return dyn_lds;
    else return dyn_lds + ((threadIdx.x >> 6) % WAVES) * (P.bvh_levels * 64u);
}

// The slab filter of the two sibling boxes of a pair at once, in nine packed FMAs: x and y of a box are a register
// pair like the ray's per-axis coefficients, the z components of the two boxes are one (BvhPair).  Per box this is
// cube_slab_approx's interval, operation for operation, with three half extents.
struct BoxTest2 { float lo_a, hi_a, lo_b, hi_b; bool hit_a, hit_b; };
DEV BoxTest2 box_slab2(const RayPre& q, f32x2 a_xy, f32x2 a_hxy, f32x2 b_xy, f32x2 b_hxy, f32x2 cz, f32x2 hz)
{
    const f32x2 ma = __builtin_elementwise_fma(a_xy, q.r_xy, q.c_xy), mb = __builtin_elementwise_fma(b_xy, q.r_xy, q.c_xy);
    const f32x2 mz = __builtin_elementwise_fma(cz, f32x2{q.rc_z[0], q.rc_z[0]}, f32x2{q.rc_z[1], q.rc_z[1]});
    const f32x2 na = __builtin_elementwise_fma(a_hxy, q.na_xy, ma), fa = __builtin_elementwise_fma(a_hxy, -q.na_xy, ma);
    const f32x2 nb = __builtin_elementwise_fma(b_hxy, q.na_xy, mb), fb = __builtin_elementwise_fma(b_hxy, -q.na_xy, mb);
    const f32x2 nz = __builtin_elementwise_fma(hz, f32x2{q.a_z_pm[0], q.a_z_pm[0]}, mz);
    const f32x2 fz = __builtin_elementwise_fma(hz, f32x2{q.a_z_pm[1], q.a_z_pm[1]}, mz);
    BoxTest2 t;
    t.lo_a = __builtin_fmaxf(__builtin_fmaxf(na[0], na[1]), nz[0]);
    t.hi_a = __builtin_fminf(__builtin_fminf(fa[0], fa[1]), fz[0]);
    t.lo_b = __builtin_fmaxf(__builtin_fmaxf(nb[0], nb[1]), nz[1]);
    t.hi_b = __builtin_fminf(__builtin_fminf(fb[0], fb[1]), fz[1]);
    t.hit_a = !(t.lo_a - t.hi_a > q.tol);  // NaN -> hit
    t.hit_b = !(t.lo_b - t.hi_b > q.tol);
    return t;
}
// A direction component that is exactly zero does not constrain the filter's interval (ray_pre), but the
// reference's test fails for every cube whose slab on that axis does not hold the origin (zero_axis_check):
// a box whose slab the origin misses by more than zdelta holds no cube that passes.
DEV bool box_zero_axis_miss(const RayPre& q, V3 o, float cx, float cy, float cz, float hx, float hy, float hz)
{
    const bool zx = (q.zbits & 1u) != 0u, zy = (q.zbits & 2u) != 0u, zz = (q.zbits & 4u) != 0u;
    return (zx && __builtin_fabsf(cx - o.x) > hx + q.zdelta) || (zy && __builtin_fabsf(cy - o.y) > hy + q.zdelta) ||
           (zz && __builtin_fabsf(cz - o.z) > hz + q.zdelta);
}

template <int MODE, uint32_t WAVES>
DEV void cube_query_bvh(const Params& P, const RayPre& q, bool active, V3 o, V3 d, uint32_t impact_id, QState& st PROF_ARG)
{
    const bool any_zero = wave_any(active && q.zbits != 0u);
    PROF_COUNT(6, 1);
    PROF_COUNT(2, 65536);  // profile builds: BVH walks in the high half of counter 2
    PROF_KCOUNT(qk, 1);
    PROF_MARK(PH_PRE);
    const uint32_t lane = threadIdx.x & 63u;
    const f32x4* nodes = (const f32x4*)P.bvh;
    uint32_t* stack = bvh_stack<WAVES>(P) + lane;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t pair = active ? 0u : kNone;  // pair 0: the children of the root
    uint32_t sp = 0u;
    while (wave_any(pair != kNone)) {
        PROF_COUNT(MODE == Q_SHADOW ? 4 : 3, 1);  // profile builds: walk steps (drown the few amb counts of these slots)
        const bool on = pair != kNone;
        st.ev_bvh += on ? 1u : 0u;
        // a 32-bit byte offset from the uniform base (pairs are 64 bytes, rt_scene_upload takes at most 2^26 cubes): one
        // address register and no 64-bit arithmetic per step (10k-cube world -1.2 %)
        const uint32_t off = on ? pair * 64u : 0u;
        const f32x4* pn = (const f32x4*)((const char*)nodes + off);
        const f32x4 n0 = pn[0], n1 = pn[1], n2 = pn[2], n3 = pn[3];  // BvhPair: the two boxes interleaved by component
        const float w0 = n3[0], w1 = n3[1], w2 = n3[2], w3 = n3[3];  // scalar copies first: see the note at u2f
        const uint32_t fca = __builtin_bit_cast(uint32_t, w0), fcb = __builtin_bit_cast(uint32_t, w1);
        const uint32_t link_a = __builtin_bit_cast(uint32_t, w2), link_b = __builtin_bit_cast(uint32_t, w3);
        const uint32_t cnt_a = fca & 7u, cnt_b = fcb & 7u, first_a = fca >> 3, first_b = fcb >> 3;
        const f32x2 a_xy = f32x2{n0[0], n0[1]}, a_hxy = f32x2{n0[2], n0[3]}, b_xy = f32x2{n1[0], n1[1]}, b_hxy = f32x2{n1[2], n1[3]};
        const f32x2 cz = f32x2{n2[0], n2[1]}, hz = f32x2{n2[2], n2[3]};
        const BoxTest2 t = box_slab2(q, a_xy, a_hxy, b_xy, b_hxy, cz, hz);
        bool meet_a = on && t.hit_a, meet_b = on && t.hit_b;
        if (any_zero) {
            meet_a = meet_a && !box_zero_axis_miss(q, o, a_xy[0], a_xy[1], cz[0], a_hxy[0], a_hxy[1], hz[0]);
            meet_b = meet_b && !box_zero_axis_miss(q, o, b_xy[0], b_xy[1], cz[1], b_hxy[0], b_hxy[1], hz[1]);
        }
        if constexpr (MODE == Q_SHADOW) {
            meet_a = meet_a && !(cnt_a == 0u && first_a >= st.blocker);
            meet_b = meet_b && !(cnt_b == 0u && first_b >= st.blocker);
        } else if constexpr (MODE == Q_NEAREST) {
            const float limit = st.b1 + 2.0f * q.tol;
            meet_a = meet_a && !(t.lo_a > limit);
            meet_b = meet_b && !(t.lo_b > limit);
        } else {
            meet_a = meet_a && !(t.lo_a - q.tol > st.best);
            meet_b = meet_b && !(t.lo_b - q.tol > st.best);
        }
        const bool leaf_a = meet_a && (cnt_a != 0u), leaf_b = meet_b && (cnt_b != 0u);
        PROF_MARK(PH_CULL);
        // a leaf IS its cube (centre, half size, list index and id in the node): no further fetch
        // lanes with one leaf among the two share one evaluation (each takes its own leaf's values); a second one
        // only when some lane has two
        if (wave_any(leaf_a || leaf_b)) {
            const bool ua = leaf_a;
            cube_candidate<MODE, false, true>(P, q, leaf_a || leaf_b, any_zero, o, d, impact_id, ua ? first_a : first_b, ua ? link_a : link_b,
                                              ua ? a_xy[0] : b_xy[0], ua ? a_xy[1] : b_xy[1], ua ? cz[0] : cz[1], ua ? a_hxy[0] : b_hxy[0], st PROF_PASS,
                                              ua ? t.lo_a : t.lo_b, ua ? t.hi_a : t.hi_b);
            if (wave_any(leaf_a && leaf_b))
                cube_candidate<MODE, false, true>(P, q, leaf_a && leaf_b, any_zero, o, d, impact_id, first_b, link_b, b_xy[0], b_xy[1], cz[1], b_hxy[0], st PROF_PASS, t.lo_b, t.hi_b);
        }
        PROF_MARK(PH_CAND);
        // inner children that are still worth a visit: the nearer / lower-indexed one next, the other deferred
        const bool in_a = meet_a && (cnt_a == 0u), in_b = meet_b && (cnt_b == 0u);
        bool a_first;
        if constexpr (MODE == Q_SHADOW) a_first = !(first_b < first_a);
        else a_first = !(t.lo_b < t.lo_a);
        const bool both = in_a && in_b;
        if (both) stack[sp * 64u] = a_first ? link_b : link_a;
        const bool pop = on && !(in_a || in_b) && (sp != 0u);
        const uint32_t popped = stack[(pop ? sp - 1u : 0u) * 64u];
        uint32_t next = kNone;
        next = pop ? popped : next;
        next = in_b ? link_b : next;
        next = (in_a && (a_first || !in_b)) ? link_a : next;
        sp = both ? sp + 1u : (pop ? sp - 1u : sp);
        pair = on ? next : kNone;
    }
}

// LARGE (a kernel template parameter chosen by the length of the cube list: 0 short, 1 long) keeps the long-list
// code out of the kernel that renders short lists: compiled in, it cost the 97-cube benchmark scene 6 %.  LARGE >= 2:
// a short list with the cull refinement of cube_query_small (kRefineMin); LARGE == 3: and with bundle keys (render_kernel,
// cube_query's SPLIT == 2); chosen per launch (Params::refine).
template <int MODE, int LARGE, uint32_t WAVES, typename CP>
DEV void cube_query_run(const Params& P, const RayPre& q, const CP& cp, bool active, V3 o, V3 d, uint32_t impact_id, QState& st PROF_ARG)
{
    if constexpr (LARGE == 1) {
        // how much the wave's spread inflates every cube in cull_test, against the cubes' own size (wave-uniform)
        const bool coherent = cp.all || !(__builtin_fmaf(2.0f * P.bound_max, cp.dr, cp.wr_raw) > P.coherent_thr);
        if (coherent) cube_query_large<MODE, CP>(P, q, cp, active, o, d, impact_id, st PROF_PASS);
        else cube_query_bvh<MODE, WAVES>(P, q, active, o, d, impact_id, st PROF_PASS);
    } else {
        cube_query_small<MODE, CP, LARGE >= 2>(P, q, cp, active, o, d, impact_id, st PROF_PASS);
    }
}

// SPLIT (1: heavy tiles, render_heavy_kernel's tracer and shaders, with the general cull; 2: render_kernel<.., 3, ..>,
// with the one-radius cull): a wave whose rays come in several
// bundles - an 8x8 tile across the edge between two faces sends its secondary rays two ways, and the
// reference's negative-distance rule can put the hit points of one bundle on the far side of the scene -
// has a wave-level bound that keeps every cube.  `key` names a lane's bundle (the faces its ray tree has
// come through, see trace_tree); when the wave's bound is loose and the keys differ, the lanes are walked
// key by key, each bundle with its own bound (at most kMaxBundles passes, the last takes all the rest).
// The folds are per lane, so the order and grouping of the passes change no result.
constexpr uint32_t kMaxBundles = 3u;
template <int MODE, int LARGE, int SPLIT = 0>
DEV void cube_query(const Params& P, bool active, V3 o, V3 d, uint32_t impact_id, QState& st, uint32_t key PROF_ARG)
{
    if (P.n_cubes == 0u) return;
    const RayPre q = ray_pre(P, o, d);
    st.tol = q.tol;
    st.div_ok = q.div_ok;
    if constexpr (SPLIT == 2) {
        // render_kernel: the one-radius cull, and bundle by bundle where it is loose and the wave holds several bundles
        const CullPre cp = cull_pre(active, o, d, q.wild, P.cull_slack);
        const unsigned long long am = wave_ballot(active);
        const uint32_t key0 = (uint32_t)__builtin_amdgcn_readlane((int)key, am ? (int)__builtin_ctzll(am) : 0);
        if (!cp.all && (__builtin_fmaf(2.0f * P.bound_max, cp.dr, cp.wr_raw) > P.split_thr) && wave_any(active && key != key0)) {
            bool rem = active;
            for (uint32_t b = 0; b < kMaxBundles; ++b) {
                const unsigned long long rm = wave_ballot(rem);
                if (rm == 0ull) break;
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(rm));
                const bool sel = rem && ((b + 1u == kMaxBundles) || (key == kb));
                const CullPre cpb = cull_pre(sel, o, d, q.wild, P.cull_slack);
                cube_query_run<MODE, LARGE, 1u, CullPre>(P, q, cpb, sel, o, d, impact_id, st PROF_PASS);
                rem = rem && !sel;
            }
            return;
        }
        cube_query_run<MODE, LARGE, 1u, CullPre>(P, q, cp, active, o, d, impact_id, st PROF_PASS);
    } else if constexpr (SPLIT == 1) {
        // heavy tiles: the general cull (cull_pre_g), and bundle by bundle where that is still loose
        const CullPreG cp = cull_pre_g(active, o, d, q.wild, P.cull_slack);
        const unsigned long long am = wave_ballot(active);
        const uint32_t key0 = (uint32_t)__builtin_amdgcn_readlane((int)key, am ? (int)__builtin_ctzll(am) : 0);
        if (!cp.all && (__builtin_fmaf(2.0f * P.bound_max, cp.dr, cp.wr_raw) > P.split_thr) && wave_any(active && key != key0)) {
            bool rem = active;
            for (uint32_t b = 0; b < kMaxBundles; ++b) {
                const unsigned long long rm = wave_ballot(rem);
                if (rm == 0ull) break;
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(rm));
                const bool sel = rem && ((b + 1u == kMaxBundles) || (key == kb));
                const CullPreG cpb = cull_pre_g(sel, o, d, q.wild, P.cull_slack);
                cube_query_run<MODE, LARGE, kWgWaves, CullPreG>(P, q, cpb, sel, o, d, impact_id, st PROF_PASS);
                rem = rem && !sel;
            }
            return;
        }
        cube_query_run<MODE, LARGE, kWgWaves, CullPreG>(P, q, cp, active, o, d, impact_id, st PROF_PASS);
    } else {
        const CullPre cp = cull_pre(active, o, d, q.wild, P.cull_slack);
        cube_query_run<MODE, LARGE, 1u, CullPre>(P, q, cp, active, o, d, impact_id, st PROF_PASS);
    }
}

// ---------------------------------------------------------------- sphere queries
// The same wave-level cull with the sphere's bounding box (centre, half = radius), then the reference's
// test (sphere_exact) for the candidates, in list order.  One difference from the cubes: the reference's
// discriminant b^2 - 4ac is computed in f32 with cancellation and can come out positive for a line that
// misses the sphere.  With u = 2^-24, |error| <= (60 |oc|^2 + 24 r^2) u |d|^2, so a miss by more than
// 7.5 u |oc|^2 / r + 3 u r cannot be reported as a hit; |oc|^2 <= 3 (O + B)^2 with O, B the largest origin
// and sphere coordinate.  The box is therefore inflated by 2^-17 (O + B + 1)^2 / r (6x that bound; r -> 0
// gives an infinite box: always a candidate).
DEV bool ray_wild(const Params& P, V3 o, V3 d)
{
    const float bigd = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float bigo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
    const bool finite = (d.x == d.x) && (d.y == d.y) && (d.z == d.z) && (o.x == o.x) && (o.y == o.y) && (o.z == o.z);
    return !(finite && bigd <= 0x1p40f && bigo <= P.origin_bound);
}

// NEAREST: render.rs:87-101 over the sphere list (strict <, list order).  SHADOW: render.rs:44-70, the
// first sphere in list order whose id differs from the impact object's and that is hit.
template <bool SHADOW>
DEV void sphere_query(const Params& P, bool active, V3 o, V3 d, uint32_t impact_id, uint32_t& best_idx, float& best_dist, uint32_t& ev_cull, uint32_t& ev_sphere)
{
    const uint32_t n = P.n_spheres;
    if (n == 0u) return;
    const CullPre cp = cull_pre(active, o, d, ray_wild(P, o, d) || (P.sphere_ok == 0u), P.sphere_cull_slack);
    const float grow = P.sphere_grow;  // 2^-17 (O + B + 1)^2
    const uint32_t lane = threadIdx.x & 63u;
    const f32x4* filt = (const f32x4*)P.sphere_filt;
    bool done = false;
    for (uint32_t base = 0; base < n; base += 64u) {
        f32x4 rec = filt[base + lane];  // (centre, |radius|), padded to a multiple of 64
        rec[3] = __builtin_fmaf(grow, __builtin_amdgcn_rcpf(rec[3]), rec[3]);
        unsigned long long M = wave_ballot(cull_test(rec, cp) && (base + lane < n));
        ev_cull += (base + lane < n) ? 1u : 0u;
        while (M != 0ull) {
            const uint32_t b = (uint32_t)__builtin_ctzll(M);
            M &= M - 1ull;
            const uint32_t i = base + b;
            const SphereRec s = load_sphere_uniform(P.spheres, i);
            float dist = 0.0f;
            const bool hit = sphere_exact(s, o, d, dist);
            ev_sphere += active ? 1u : 0u;
            if constexpr (SHADOW) {
                const bool pass = hit && active && !done && (load_sphere_meta_uniform(P.sphere_meta, i).id != impact_id);
                best_idx = pass ? i : best_idx;
                best_dist = pass ? dist : best_dist;
                done = done || pass;
                if (!wave_any(active && !done)) return;
            } else {
                const bool win = hit && active && (dist < best_dist);
                best_dist = win ? dist : best_dist;
                best_idx = win ? i : best_idx;
            }
        }
    }
}

// ---------------------------------------------------------------- nearest hit (render.rs:87-101)
// `pass` is wave-uniform: PASS_SPHERES walks the sphere list, PASS_CUBES the cube list.
template <bool COUNT, int LARGE, int SPLIT = 0>
DEV uint32_t nearest_query(const Params& P, uint32_t pass, bool active, V3 o, V3 d, bool& div_ok, Counters& cnt, uint32_t key PROF_ARG)
{
    uint32_t best_idx = kNoHit;
    div_ok = false;
    if (pass == PASS_CUBES) {
        QState st;
        st.b1 = __builtin_inff(); st.b2 = __builtin_inff(); st.i1 = kNoHit;
        st.best = __builtin_inff(); st.best_i = kNoHit; st.done = false; st.blocker = kNoHit; st.tol = 0.0f; st.div_ok = false;
        st.ev_cull = 0u; st.ev_chunk = 0u; st.ev_cand = 0u; st.ev_exact = 0u; st.ev_bvh = 0u;
        cube_query<Q_NEAREST, LARGE, SPLIT>(P, active, o, d, 0u, st, key PROF_PASS);
        best_idx = active ? st.i1 : kNoHit;
        div_ok = st.div_ok;
        // contested: a second candidate within 2E of the best (exact duplicates, shared edges).  NaN -> contested.
        const bool contested = active && (st.i1 != kNoHit) && !(st.b2 - st.b1 > st.tol);
        if (wave_any(contested)) {
            cube_query<Q_NEAREST_EXACT, LARGE, SPLIT>(P, contested, o, d, 0u, st, key PROF_PASS);
            PROF_MARK(PH_CONTESTED);
            best_idx = contested ? st.best_i : best_idx;
            if constexpr (COUNT) cnt.ev[EV_QUERY] += contested ? 1u : 0u;
        }
        if constexpr (COUNT) {
            cnt.cube_tests += active ? P.n_cubes : 0u;
            cnt.ev[EV_CULL] += st.ev_cull; cnt.ev[EV_CHUNK] += st.ev_chunk; cnt.ev[EV_CAND] += st.ev_cand;
            cnt.ev[EV_EXACT] += st.ev_exact; cnt.ev[EV_BVH] += st.ev_bvh;
        }
    } else {
        float best_dist = __builtin_inff();
        uint32_t ev_cull = 0u, ev_sphere = 0u;
        sphere_query<false>(P, active, o, d, 0u, best_idx, best_dist, ev_cull, ev_sphere);
        if constexpr (COUNT) {
            cnt.sphere_tests += active ? P.n_spheres : 0u;
            cnt.ev[EV_CULL] += ev_cull; cnt.ev[EV_SPHERE] += ev_sphere;
        }
    }
    if constexpr (COUNT) {
        cnt.nearest_rays += active ? 1u : 0u;
        cnt.ev[EV_QUERY] += active ? 1u : 0u;
    }
    return best_idx;
}

// ---------------------------------------------------------------- shadow (render.rs:44-70)
// First object in array order, among those whose id differs from the impact object's
// (render.rs:113), whose ray_intersect returns Some.  Not the nearest one.
template <bool COUNT, int LARGE, int SPLIT = 0>
DEV float shadow_query(const Params& P, uint32_t pass, bool active, V3 origin, V3 light_pos, V3 dir, uint32_t impact_id, uint32_t& blocker, Counters& cnt, uint32_t key PROF_ARG)
{
    // dir = normalize(light_pos - origin) (render.rs:46), the caller's light_dir: the same expression
    blocker = kNoHit;
    float sdist = 0.0f;
    if constexpr (COUNT) {
        cnt.shadow_rays += active ? 1u : 0u;
        cnt.ev[EV_QUERY] += active ? 1u : 0u;
    }
    if (pass == PASS_CUBES) {
        QState st;
        st.b1 = 0.0f; st.b2 = 0.0f; st.i1 = kNoHit; st.best = 0.0f; st.best_i = kNoHit;
        st.done = false; st.blocker = kNoHit; st.tol = 0.0f; st.div_ok = false;
        st.ev_cull = 0u; st.ev_chunk = 0u; st.ev_cand = 0u; st.ev_exact = 0u; st.ev_bvh = 0u;
        cube_query<Q_SHADOW, LARGE, SPLIT>(P, active, origin, dir, impact_id, st, key PROF_PASS);
        blocker = (active && st.done) ? st.blocker : kNoHit;
        if constexpr (COUNT) {
            cnt.ev[EV_CULL] += st.ev_cull; cnt.ev[EV_CHUNK] += st.ev_chunk; cnt.ev[EV_CAND] += st.ev_cand;
            cnt.ev[EV_EXACT] += st.ev_exact; cnt.ev[EV_BVH] += st.ev_bvh;
            cnt.ev[EV_BLOCKER] += (blocker != kNoHit) ? 1u : 0u;
        }
        if (wave_any(blocker != kNoHit)) {
            // the blocker's exact distance (per-lane gather; the loop only proved that it passes)
            const CubeRec c = P.cubes[blocker != kNoHit ? blocker : 0u];
            sdist = cube_distance(c, origin, dir, st.div_ok);
        }
        if constexpr (COUNT) {
            // executed tests = cubes up to and including the blocker whose id differs from the impact id
            const uint32_t n = P.n_cubes;
            const uint32_t limit = (blocker != kNoHit) ? blocker + 1u : n;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t id = load_cube_uniform(P.cubes, i).id;
                cnt.cube_tests += (active && i < limit && id != impact_id) ? 1u : 0u;
            }
        }
    } else {
        uint32_t ev_cull = 0u, ev_sphere = 0u;
        sphere_query<true>(P, active, origin, dir, impact_id, blocker, sdist, ev_cull, ev_sphere);
        if constexpr (COUNT) {
            cnt.ev[EV_CULL] += ev_cull; cnt.ev[EV_SPHERE] += ev_sphere;
            cnt.ev[EV_BLOCKER] += (blocker != kNoHit) ? 1u : 0u;
            // executed tests = spheres up to and including the blocker whose id differs from the impact id
            const uint32_t n = P.n_spheres;
            const uint32_t limit = (blocker != kNoHit) ? blocker + 1u : n;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t id = load_sphere_meta_uniform(P.sphere_meta, i).id;
                cnt.sphere_tests += (active && i < limit && id != impact_id) ? 1u : 0u;
            }
        }
    }
    float shadow = 0.0f;
    if (wave_any(blocker != kNoHit)) {
        const bool has = (blocker != kNoHit) && !(sdist < 0.0f);  // render.rs:56-58: negative distance -> 0
        const uint32_t bi = (blocker != kNoHit) ? blocker : 0u;
        const V3 sp = vadd(origin, vscale(dir, sdist));
        const V3 dv = vsub(sp, light_pos);  // distance2(a,b) = norm_squared(b-a)
        const float d2 = vdot(dv, dv);
        const uint32_t bm = (pass == PASS_CUBES) ? P.cube_mat[bi] : P.sphere_meta[bi].material;
        const float sh = sdist / d2 * (1.0f - P.mats[bm].transparency);
        shadow = has ? sh : 0.0f;
    }
    PROF_MARK(PH_SHADOW_POST);
    return shadow;
}

// ---------------------------------------------------------------- the ray tree (cast_ray, render.rs:73-229)
// The reference recurses: a hit spawns a reflection and a refraction ray per light iteration, to depth
// 3.  Two restatements that leave every output bit unchanged:
//  * children are traced once per hit and reused by every light iteration (the reference re-traces them
//    with identical results).  Because every operand of the light fold is a non-negative u8 and `+`
//    saturates, ((acc + X) + Y) + Z == min(255, acc+X+Y+Z), so the fold equals
//    min(255, sum_l DS_l + L*(refl*k_r) + L*(refr*k_t)) per channel and is evaluated after the children
//    return (node_finish);
//  * the recursion is an explicit depth-first walk with ONE copy of the node code.  All 64 lanes of a
//    wave are at the same depth at the same time (`active` says which lanes have a ray), so the walk's
//    control state is wave-uniform and the per-lane frames live in LDS, indexed by depth.  (Unrolling
//    the recursion by template depth, as an earlier version did, gave 110 KB of code - more than the
//    instruction cache - and several hundred spilled registers.)
// Frame of a node waiting for its children, per lane:
enum { F_RO = 0, F_RD = 3, F_SUMF = 6, F_C0 = 7, F_KR = 8, F_KT = 9, F_H = 10, F_H0 = 11 };
constexpr uint32_t kEntryNoCubes = 1u << 15;    // queue entry flag: the tile's cube pass is answered without tracing
constexpr uint32_t kEntryNoSpheres = 1u << 14;  // ... and its sphere pass (no sphere's grown box projects into the tile)
constexpr uint32_t kEntryColMask = 0x3FFFu;     // tile column (frames are at most 65 536 wide: 8 192 columns)
constexpr uint32_t kPaintGroup = 16;  // untraced tiles painted per surplus workgroup
constexpr int kStackDepth = 3;  // nodes at depth 0..2 can wait for traced children (max_depth <= 3)
template <bool DBG, int SPLIT = 0>
constexpr int frame_words() { return DBG ? 12 : (SPLIT ? 11 : 10); }
constexpr int F_KEY = 10;  // SPLIT (plain variants only): the bundle key of the node's refraction child
constexpr uint32_t kHitFlag = 1u << 24;  // in F_SUMF, above the packed 0x00RRGGBB light sum
constexpr uint32_t RT_MAX_DEPTH_C = 3u;  // deepest level that is traced (rtx950.h RT_MAX_DEPTH; check_frame refuses more)

// min(255, sum + L*trunc(c_refl*k_r) + L*trunc(c_refr*k_t)) for lanes with a hit, sky otherwise
DEV uint32_t node_finish(const Params& P, uint32_t sumf, float k_r, float k_t, uint32_t c_refl, uint32_t c_refr)
{
    const Col sum = col_from_hex(sumf);
    const Col rl = col_mul(col_from_hex(c_refl), k_r);
    const Col rr = col_mul(col_from_hex(c_refr), k_t);
    const float fl = P.n_lights_f;
    Col lit;
    lit.r = __builtin_fminf(sum.r + fl * rl.r + fl * rr.r, 255.0f);
    lit.g = __builtin_fminf(sum.g + fl * rl.g + fl * rr.g, 255.0f);
    lit.b = __builtin_fminf(sum.b + fl * rl.b + fl * rr.b, 255.0f);
    return (sumf & kHitFlag) ? col_to_hex(lit) : P.sky;
}

// ---------------------------------------------------------------- heavy tiles: one tile on three waves
// A tile is one wave, and a wave alone on its SIMD issues an instruction every 4-5 cycles: the tiles across
// cube edges (incoherent secondary bundles, chains of four nodes with three list traversals each) take as
// long as the whole rest of the frame.  render_heavy_kernel gives such a tile a workgroup of three waves: the
// TRACER (wave 0) walks the ray tree - nearest hit, hit resolution, child rays - and posts each node's
// shading input to a mailbox in LDS instead of running the light fold itself; the SHADERS (waves 1, 2) take
// the posted nodes, each runs the fold's iterations for its share of the lights (shadow query + Phong terms,
// light_term below) and returns the packed partial sum.  The tracer is already on the next node of the chain
// while the shaders work, and collects a node's sum when it finishes the node on the way back up.  The
// fold's operands are non-negative u8 and `+` saturates, so partial sums combine in any order:
// min(255, min(255, A) + min(255, B)) = min(255, A + B).  Every pixel is computed by the same functions in
// the same arithmetic as in the one-wave walk; only the wave that executes them differs.
//
// The mailbox has one slot per tree depth: the walk holds at most one unfinished node per depth (a node's
// second child is traced after its first has returned), so a slot is free again when the tracer has
// collected its node.  posted[k] / done[k][w] are sequence numbers that only grow (release/acquire at
// workgroup scope); all waves of a workgroup are resident together, so the polling loops cannot starve.
constexpr uint32_t kShaders = kWgWaves - 1u;
constexpr uint32_t kMailDepth = 4u;   // tree depths 0..3 (max_depth <= 3)
constexpr uint32_t kRecWords = 13u;   // hit point, normal, view direction (3 each), packed texel*albedo.0, material | lit << 31, impact id, bundle key
struct MailCtl {                      // wave-uniform control words, apart from the record overlay
    uint32_t posted[kMailDepth];
    uint32_t done[kMailDepth][kShaders];
    uint32_t pass_ambient[kMailDepth][2];  // the node's pass and ambient term (uniform)
    uint32_t items_done;              // heavy items the tracer has finished
#ifdef RTX_PROFILE
    uint32_t prof_wait, prof_busy[kShaders], prof_nodes[kShaders], prof_posts;
#endif
    uint32_t error;                   // a polling loop gave up (never expected; rt_debug_status reports it)
};
struct Mailbox {
    uint32_t* rec;      // [kMailDepth][kRecWords][64]
    uint32_t* res;      // [kMailDepth][kShaders][64] packed 0x00RRGGBB partial sums
    MailCtl* ctl;
};
constexpr uint32_t kMailWords = kMailDepth * kRecWords * 64u + kMailDepth * kShaders * 64u;
constexpr uint32_t kSpinLimit = 1u << 22;  // polls before a waiting wave gives up (seconds; a tile takes microseconds)

DEV uint32_t lds_load_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void lds_store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// tracer: hand the node at `depth` to the shaders
DEV void mail_post(const Mailbox& mb, uint32_t depth, uint32_t pass, float ambient, V3 hit_p, V3 hit_n, V3 view_dir, Col tx_a, uint32_t material, bool lit,
                   uint32_t hit_id, uint32_t key)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* r = mb.rec + depth * (kRecWords * 64u) + lane;
    r[0 * 64] = __builtin_bit_cast(uint32_t, hit_p.x); r[1 * 64] = __builtin_bit_cast(uint32_t, hit_p.y); r[2 * 64] = __builtin_bit_cast(uint32_t, hit_p.z);
    r[3 * 64] = __builtin_bit_cast(uint32_t, hit_n.x); r[4 * 64] = __builtin_bit_cast(uint32_t, hit_n.y); r[5 * 64] = __builtin_bit_cast(uint32_t, hit_n.z);
    r[6 * 64] = __builtin_bit_cast(uint32_t, view_dir.x); r[7 * 64] = __builtin_bit_cast(uint32_t, view_dir.y); r[8 * 64] = __builtin_bit_cast(uint32_t, view_dir.z);
    r[9 * 64] = col_to_hex(tx_a);  // exact u8 values per channel
    r[10 * 64] = (material & 0x7FFFFFFFu) | (lit ? 0x80000000u : 0u);
    r[11 * 64] = hit_id;
    r[12 * 64] = key;
    if (lane == 0u) {
        mb.ctl->pass_ambient[depth][0] = pass;
        mb.ctl->pass_ambient[depth][1] = __builtin_bit_cast(uint32_t, ambient);
    }
    const uint32_t seq = mb.ctl->posted[depth] + 1u;  // only this wave writes it
    lds_store_release(&mb.ctl->posted[depth], seq);   // the whole wave stores the same word: the records above are visible first
}

// tracer: the node's light sum, min(255, sum over the shaders' partial sums), packed
DEV uint32_t mail_collect(const Mailbox& mb, uint32_t depth)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t want = mb.ctl->posted[depth];
#ifdef RTX_PROFILE
    const uint32_t prof_c0 = (uint32_t)__builtin_readcyclecounter();
#endif
    for (uint32_t w = 0; w < kShaders; ++w) {
        uint32_t spins = 0u;
        while (lds_load_acquire(&mb.ctl->done[depth][w]) != want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) { mb.ctl->error = 1u; break; }
        }
    }
#ifdef RTX_PROFILE
    if (lane == 0u) mb.ctl->prof_wait += (uint32_t)__builtin_readcyclecounter() - prof_c0;
#endif
    Col sum = Col{0.0f, 0.0f, 0.0f};
    for (uint32_t w = 0; w < kShaders; ++w) sum = col_add(sum, col_from_hex(mb.res[(depth * kShaders + w) * 64u + lane]));
    return col_to_hex(sum);
}

// One iteration of the light fold of cast_ray (render.rs:104-152, :221-224 without the children): the term
// (diffuse + specular) * (1 - reflectivity - transparency) of light `li` for the wave's hit points.  `lit`:
// lanes that cast the shadow ray (a hit whose direct light is not exactly black, see trace_tree).
template <bool DBG, bool COUNT, int LARGE, int SPLIT = 0>
DEV Col light_term(const Params& P, uint32_t pass, uint32_t li, bool lit, bool hit_any, V3 hit_p, V3 hit_n, uint32_t hit_id, V3 view_dir, Col tx_a,
                   float m_specular, float m_albedo1, float k_ds, float ambient, uint32_t& h, Counters& cnt, uint32_t key PROF_ARG)
{
    const LightRec L = P.lights[li];
    const V3 lp = v3(L.px, L.py, L.pz);
    const V3 light_dir = vnormalize(vsub(lp, hit_p));
    uint32_t blocker;
    PROF_MARK(PH_SHADE);
    const float shadow = shadow_query<COUNT, LARGE, SPLIT>(P, pass, lit, hit_p, lp, light_dir, hit_id, blocker, cnt, key PROF_PASS);
    if constexpr (DBG) h = hit_any ? hmix(h, blocker) : h;
    if constexpr (COUNT) cnt.ev[EV_LIGHT] += hit_any ? 1u : 0u;
    const float light_intensity = L.intensity * (1.0f - shadow);
    const float diffuse_intensity = rs_clamp(vdot(hit_n, light_dir), 0.0f, 1.0f) + ambient;
    const Col diffuse = col_mul(col_mul(tx_a, diffuse_intensity), light_intensity);
    // The specular term is trunc(trunc(trunc(c * albedo.1) * spec) * intensity) per channel with
    // c <= c_max <= 255 (the light's brightest channel): exactly black when c_max * albedo.1 * spec < 1, the
    // usual case away from highlights.
    // That can be seen before the reflection direction is normalised and the power taken:
    // reflect() of two unit vectors has length 1 (checked: within 1 %), so the cosine is at most
    // cb = the raw dot product + 0.6 % + 1e-5 (50x the rounding of either route), powf is monotone
    // in its base, and exp2(y log2 cb) is within 1e-3 of cb^y (see powf_like_libm).  NaNs fail
    // every compare and take the full route.
    Col specular = Col{0.0f, 0.0f, 0.0f};
    const uint32_t lc = L.color_0rgb;  // wave-uniform: the light's brightest channel bounds c
    const uint32_t lc_rg = ((lc >> 16) & 0xFFu) > ((lc >> 8) & 0xFFu) ? ((lc >> 16) & 0xFFu) : ((lc >> 8) & 0xFFu);
    const float c_max = (float)(lc_rg > (lc & 0xFFu) ? lc_rg : (lc & 0xFFu));
    const V3 refl_raw = reflect(vneg(light_dir), hit_n);
    const float rr = vdot(refl_raw, refl_raw);
    const float c_raw = vdot(view_dir, refl_raw);
    const float cb = c_raw + __builtin_fabsf(c_raw) * 0x1.8p-8f + 1e-5f;
    const float e_b = m_specular * __builtin_amdgcn_logf(cb);  // cb <= 0: NaN or -inf times y > 0
    const bool dark = (rr > 0.99f) && (rr < 1.01f) && (m_specular > 0.0f) &&
                      ((cb <= 0.0f) || ((cb < 1.0f) && (__builtin_amdgcn_exp2f(e_b) * (c_max * m_albedo1) < 0.98f)));
    // (adding a black specular term and scaling by k_ds = 1 - 0 - 0 leave the whole numbers up to 255 of `diffuse` as
    // they are: both are skipped where that holds for the whole wave - every matte surface away from a highlight)
    Col ds = diffuse;
    if (wave_any(lit && !dark)) {
        const V3 reflect_dir = vnormalize(refl_raw);
        const float spec_i = powf_like_libm(rs_clamp(vdot(view_dir, reflect_dir), 0.0f, 1.0f), m_specular);
        if (wave_any(!(spec_i < 0x1p-8f))) {
            specular = col_mul(col_mul(col_mul(col_from_hex(L.color_0rgb), m_albedo1), spec_i), light_intensity);
            ds = col_add(diffuse, specular);
        }
    }
    return wave_any(k_ds != 1.0f) ? col_mul(ds, k_ds) : ds;
}

// One pass of render.rs:298-312 (spheres with ambient 1, or cubes with ambient_light) for the wave's
// 64 primary rays: returns the packed colour, the primary hit index and the tree hash per lane.
// ASYNC (render_heavy_kernel's tracer, plain variant only): the light fold of a node is posted to the
// workgroup's shader waves (`mb`) and its sum collected when the node is finished.
template <bool DBG, bool COUNT, int LARGE, bool DARK, bool ASYNC = false, int SPLIT = (ASYNC ? 1 : 0)>
DEV uint32_t trace_tree(const Params& P, uint32_t pass, bool active0, V3 o, V3 d, float ambient, uint32_t* stk, uint32_t& hash, uint32_t& hit_idx,
                        Totals& tot, const Mailbox& mb PROF_ARG)
{
    static_assert(!ASYNC || (!DBG && !COUNT), "the shader waves compute neither blocker hashes nor counters");
    static_assert(!SPLIT || !DBG, "the debug variant's frame has no room for bundle keys");
    constexpr int F = frame_words<DBG, SPLIT>();
    uint32_t postbits = 0u;  // ASYNC, bit k: the node waiting at depth k has a posted light fold (wave-uniform)
    // SPLIT: the bundle a lane's current ray belongs to (cube_query's SPLIT): the faces and child kinds its
    // tree has come through, 4 bits a level; primary rays are one bundle
    uint32_t bkey = 0u;
    const uint32_t lane = threadIdx.x & 63u;
    hit_idx = kNoHit;
    hash = kHashCut;
    if (!wave_any(active0)) return P.sky;
    uint32_t depth = 0u;     // wave-uniform
    uint32_t slotbits = 0u;  // bit k: the node waiting at depth k is tracing its second child
    bool act = active0;
    bool unwinding = false;
    uint32_t res = P.sky, res_h = kHashCut;
    for (;;) {
        if (!unwinding) {
            // ------------------------------------------------ one node: nearest hit, shading, child rays
            Counters cnt = {};
            PROF_MARK(PH_OTHER);
            PROF_KIND(depth * 2u);
            if constexpr (ASYNC) postbits &= ~(1u << depth);
            bool div_ok;
            const uint32_t idx = nearest_query<COUNT, LARGE, SPLIT>(P, pass, act, o, d, div_ok, cnt, bkey PROF_PASS);
            if (depth == 0u) hit_idx = idx;
            uint32_t h = hmix(kHashBasis, idx);
            const bool hit_any = act && (idx != kNoHit);
            Col sum = Col{0.0f, 0.0f, 0.0f};
            float k_r = 0.0f, k_t = 0.0f;
            bool has_refl = false, has_refr = false;
            V3 refl_o = o, refl_d = d, refr_o = o, refr_d = d;
            uint32_t nkey = 0u;  // SPLIT: this node's bundle key = its ray's key, then the face that was hit
            if (wave_any(hit_any)) {
                const uint32_t sidx = hit_any ? idx : 0u;  // lanes without a hit shade object 0 and discard the result
                if constexpr (COUNT) {
                    cnt.shaded_hits += hit_any ? 1u : 0u;
                    cnt.ev[EV_RESOLVE] += hit_any ? 1u : 0u;
                }
                Hit hit;
                if (pass == PASS_CUBES) hit = cube_resolve(P, sidx, o, d, div_ok);
                else hit = sphere_resolve(P, sidx, o, d);
                const MatRec m = P.mats[hit.material];
                PROF_MARK(PH_RESOLVE);
                if constexpr (SPLIT) nkey = (bkey * 16u + (uint32_t)hit.face * 2u) & 0x00FFFFFEu;

                Col tx;
                if (m.texture >= 0) tx = texel_of_face(P, m.texture, hit.face, hit.u * P.sprite, hit.v * P.sprite);
                else tx = col_from_hex(m.diffuse_0rgb);
                const Col tx_a = col_mul(tx, m.albedo0);
                const V3 view_dir = vnormalize(vsub(o, hit.p));
                const float k_ds = 1.0f - m.reflectivity - m.transparency;

                // Direct light enters the colour as trunc(c * k_ds) per channel with c <= 255: exactly black for
                // k_ds < 1/255 (water: 1 - 0.1 - 0.9), whatever the shadow rays find.  Such lanes cast none - the
                // variants that report blockers or count rays still do - and a wave of them skips the loop.
                // (DARK: compiled in only for scenes that hold such a material, rt_scene_upload decides; the
                // other scenes' kernel does not carry the test.)
                const uint32_t nl = P.n_lights;
                bool lit = hit_any;
                uint32_t nl_cast = nl;
                if constexpr (DARK && !DBG && !COUNT) {
                    lit = hit_any && !(k_ds < 0.0039f);
                    nl_cast = wave_any(lit) ? nl : 0u;
                }
                if constexpr (ASYNC) {
                    if (nl_cast != 0u) {
                        mail_post(mb, depth, pass, ambient, hit.p, hit.n, view_dir, tx_a, hit.material, lit, hit.id, nkey);
                        postbits |= 1u << depth;
                    }
                } else {
                    for (uint32_t li = 0; li < nl_cast; ++li) {
                        PROF_KIND(depth * 2u + 1u);
                        const Col term = light_term<DBG, COUNT, LARGE, SPLIT>(P, pass, li, lit, hit_any, hit.p, hit.n, hit.id, view_dir, tx_a, m.specular,
                                                                              m.albedo1, k_ds, ambient, h, cnt, nkey PROF_PASS);
                        sum = col_add(sum, term);
                    }
                }
                PROF_MARK(PH_SHADE);
                // children (render.rs:154-193)
                k_r = m.reflectivity;
                k_t = m.transparency;
                // the reference casts the secondary rays inside the light loop: without lights there are none
                has_refl = hit_any && (k_r > 0.0f) && (nl != 0u);
                has_refr = hit_any && (k_t > 0.0f) && (nl != 0u);
                // each kind only where some lane of the wave has it (a mirror-like material has no refraction ray)
                if (wave_any(has_refl)) {
                    refl_d = vnormalize(reflect(vneg(d), hit.n));
                    refl_o = vadd(hit.p, vscale(hit.n, 1e-2f));
                }
                if (wave_any(has_refr)) {
                    refr_d = refract(d, hit.n, m.ior);
                    refr_o = vadd(hit.p, vscale(hit.n, 1e-3f));
                }
                if constexpr (COUNT) cnt.ev[EV_CHILD] += (has_refl || has_refr) ? 1u : 0u;
                PROF_MARK(PH_CHILDSETUP);
            }
            if constexpr (COUNT) {
                unsigned long long w = 1ull;  // n_lights^depth: how often the reference runs this node
                for (uint32_t k = 0; k < depth; ++k) w *= P.n_lights;
                totals_add(tot, cnt, w);
            }
            uint32_t sumf = col_to_hex(sum) | (hit_any ? kHitFlag : 0u);  // ASYNC: the light sum is collected when the node is finished
            const bool any_child = has_refl || has_refr;
            if (depth < P.max_depth && wave_any(any_child)) {
                // wait for the children: the lane's first child is its reflection if it has one, else its
                // refraction; lanes with both trace the refraction in a second round (slot 1)
                uint32_t* fr = stk + (depth * F) * 64u + lane;
                fr[(F_RO + 0) * 64] = __builtin_bit_cast(uint32_t, refr_o.x);
                fr[(F_RO + 1) * 64] = __builtin_bit_cast(uint32_t, refr_o.y);
                fr[(F_RO + 2) * 64] = __builtin_bit_cast(uint32_t, refr_o.z);
                fr[(F_RD + 0) * 64] = __builtin_bit_cast(uint32_t, refr_d.x);
                fr[(F_RD + 1) * 64] = __builtin_bit_cast(uint32_t, refr_d.y);
                fr[(F_RD + 2) * 64] = __builtin_bit_cast(uint32_t, refr_d.z);
                fr[F_SUMF * 64] = sumf;
                if constexpr (SPLIT) fr[F_KEY * 64] = nkey;
                fr[F_KR * 64] = __builtin_bit_cast(uint32_t, k_r);
                fr[F_KT * 64] = __builtin_bit_cast(uint32_t, k_t);
                if constexpr (DBG) fr[F_H * 64] = h;
                o = has_refl ? refl_o : refr_o;
                d = has_refl ? refl_d : refr_d;
                if constexpr (SPLIT) bkey = has_refl ? (nkey | 1u) : nkey;
                act = any_child;
                slotbits &= ~(1u << depth);
                depth += 1u;
                continue;
            }
            // children beyond max_depth return the sky colour without tracing (render.rs:80-82)
            const uint32_t zero = 0u;
            if constexpr (ASYNC) {
                if ((postbits >> depth) & 1u) sumf |= mail_collect(mb, depth);
            }
            // (a wave without a child anywhere - every matte tile - adds 0 + 0 to light sums that are whole numbers
            // up to 255 already: node_finish would return them unchanged)
            if (wave_any(any_child)) res = node_finish(P, sumf, k_r, k_t, has_refl ? P.sky : zero, has_refr ? P.sky : zero);
            else res = hit_any ? (sumf & 0x00FFFFFFu) : P.sky;
            if constexpr (DBG) res_h = hit_any ? hmix(hmix(h, has_refl ? kHashCut : 0u), has_refr ? kHashCut : 0u) : h;
            unwinding = true;
        }
        // ---------------------------------------------------- hand (res, res_h) to the waiting parent
        if (depth == 0u) break;
        depth -= 1u;
        PROF_MARK(PH_OTHER);
        uint32_t* fr = stk + (depth * F) * 64u + lane;
        uint32_t sumf = fr[F_SUMF * 64];
        const float k_r = __builtin_bit_cast(float, fr[F_KR * 64]);
        const float k_t = __builtin_bit_cast(float, fr[F_KT * 64]);
        const bool hit_any = (sumf & kHitFlag) != 0u;
        const bool has_refl = hit_any && (k_r > 0.0f) && (P.n_lights != 0u);
        const bool has_refr = hit_any && (k_t > 0.0f) && (P.n_lights != 0u);
        uint32_t c0, c1 = 0u, h0 = 0u, h1 = 0u;
        if (((slotbits >> depth) & 1u) == 0u) {
            c0 = res;
            if constexpr (DBG) h0 = res_h;
            if (wave_any(has_refl && has_refr)) {
                fr[F_C0 * 64] = c0;
                if constexpr (DBG) fr[F_H0 * 64] = h0;
                o = v3(__builtin_bit_cast(float, fr[(F_RO + 0) * 64]), __builtin_bit_cast(float, fr[(F_RO + 1) * 64]),
                       __builtin_bit_cast(float, fr[(F_RO + 2) * 64]));
                d = v3(__builtin_bit_cast(float, fr[(F_RD + 0) * 64]), __builtin_bit_cast(float, fr[(F_RD + 1) * 64]),
                       __builtin_bit_cast(float, fr[(F_RD + 2) * 64]));
                act = has_refl && has_refr;
                if constexpr (SPLIT) bkey = fr[F_KEY * 64];  // the refraction child's bundle
                slotbits |= 1u << depth;
                depth += 1u;
                unwinding = false;
                continue;
            }
        } else {
            c0 = fr[F_C0 * 64];
            c1 = res;
            if constexpr (DBG) {
                h0 = fr[F_H0 * 64];
                h1 = res_h;
            }
        }
        const uint32_t zero = 0u;
        const uint32_t c_refl = has_refl ? c0 : zero;
        const uint32_t c_refr = has_refr ? (has_refl ? c1 : c0) : zero;
        if constexpr (ASYNC) {
            if ((postbits >> depth) & 1u) sumf |= mail_collect(mb, depth);  // the shaders have had the children's time
        }
        res = node_finish(P, sumf, k_r, k_t, c_refl, c_refr);
        if constexpr (DBG) {
            const uint32_t h = fr[F_H * 64];
            const uint32_t h_refl = has_refl ? h0 : 0u;
            const uint32_t h_refr = has_refr ? (has_refl ? h1 : h0) : 0u;
            res_h = hit_any ? hmix(hmix(h, h_refl), h_refr) : h;
        }
    }
    hash = res_h;
    return res;
}

// ---------------------------------------------------------------- render (render.rs:234-324)
// Geometry of a launch: the band [y0,y1) is cut into 16-row blocks, part p of n_parts owns blocks p,
// p+n_parts, ... and stores them compactly; each block is two rows of 8x8 pixel tiles.  Tile (bx, by) of
// the launch's grid covers frame rows tile_y0(by) .. +7 and output rows by*8 .. +7.
DEV uint32_t tile_y0(const Params& P, uint32_t by) { return P.y0 + ((by >> 1) * P.n_parts + P.part) * kRowBlock + (by & 1u) * 8u; }

// No cube's projection touches the tile (classify_tiles)?  The render tile is aligned to the band
// start y0, so it can straddle two rows of the frame-aligned classification tiles.  Wave-uniform.
// `plane`: the fine words of the cubes' or the spheres' stamps, the coarse level behind them.
DEV bool tile_plane_clear(const Params& P, const uint32_t* plane, uint32_t bx, uint32_t ty0)
{
    const uint32_t last = P.tiles_y - 1u;
    const uint32_t r0 = (ty0 >> 3) < last ? (ty0 >> 3) : last;
    const uint32_t r1 = ((ty0 + 7u) >> 3) < last ? ((ty0 + 7u) >> 3) : last;
    const uint32_t s_all = P.tile_stamps[0], s0 = plane[r0 * P.tiles_x + bx], s1 = plane[r1 * P.tiles_x + bx];
    // coarse level: blocks of 8x8 tiles, stamped instead of the tiles by objects with a large projection
    const uint32_t* cs = plane + P.tiles_x * P.tiles_y;
    const uint32_t ctx = (P.tiles_x + 7u) >> 3;
    const uint32_t c0 = cs[(r0 >> 3) * ctx + (bx >> 3)], c1 = cs[(r1 >> 3) * ctx + (bx >> 3)];
    return (s_all != P.frame_id) && (s0 != P.frame_id) && (s1 != P.frame_id) && (c0 != P.frame_id) && (c1 != P.frame_id);
}
DEV bool tile_has_no_cubes(const Params& P, uint32_t bx, uint32_t ty0)
{
    if (!P.tile_stamps) return false;
    return tile_plane_clear(P, P.tile_stamps + kStampBase, bx, ty0);
}
// a scene without spheres has an empty sphere pass everywhere; one with spheres only where no sphere projects
DEV bool tile_has_no_spheres(const Params& P, uint32_t bx, uint32_t ty0)
{
    if (P.n_spheres == 0u) return true;
    if (!P.tile_stamps || !P.sphere_stamps) return false;
    return tile_plane_clear(P, P.sphere_stamps, bx, ty0);
}

// build_queue: one thread per tile of the launch appends the tile to one of the queue's lists.  A
// tile whose primary rays can hit nothing (no spheres in the scene and no cube projecting into it) goes
// to the last list (class kCostClasses): the render kernel paints it without tracing, after all traced
// tiles have been handed out.  Every other tile goes to the list of its cost class.
// HEAVY: the launch uses or writes a heavy list (Params::heavy); launches that do neither run the instantiation
// without that code (with it compiled in, unused, the 4K frame with two frames in flight took 3 % longer).
template <bool HEAVY>
__global__ __launch_bounds__(1024) void build_queue(const Params P)
{
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t tiles = P.grid_tx * P.grid_ty;
    const bool valid = t < tiles;
    const uint32_t bx = valid ? t % P.grid_tx : 0u, by = valid ? t / P.grid_tx : 0u;
    const uint32_t ty0 = tile_y0(P, by);
    const bool in_band = valid && (ty0 < P.y1);
    const bool no_cubes = tile_has_no_cubes(P, bx, ty0);
    const bool no_spheres = tile_has_no_spheres(P, bx, ty0);
    const bool traced = in_band && !(no_spheres && no_cubes);
    // cost class from the tile's cycle count in the previous launch: < 8K cycles -> 0, then one class per
    // doubling, >= 512K -> 7
    const uint32_t cost = valid ? P.tile_cost[t] : 0u;
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(cost | 1u);
    const uint32_t cls = !traced ? kCostClasses : (bits <= 13u ? 0u : (bits - 13u > kCostClasses - 1u ? kCostClasses - 1u : bits - 13u));
    // Heavy tiles (Params::heavy): a tile on this launch's heavy list is render_heavy_kernel's, which is
    // tracing it on the second stream already: it is neither listed nor painted here.  (Its cost word may be
    // rewritten by that kernel while this thread reads it: either value will do, costs only order things.)
    bool heavy_now = false;
    if constexpr (HEAVY) heavy_now = valid && (P.heavy_now != 0u) && (P.heavy_mark[t] == P.heavy_now);
    // ... and the next launch's list: the traced tiles whose cost class is at least heavy_class (one class
    // less for a tile that is on the list already: its three-wave cost is scaled back to one-wave terms
    // only roughly, and a tile should not change sides every frame)
    bool heavy_next = false;
    if (HEAVY && P.heavy_next != 0u) {  // wave-uniform
        const uint32_t fit_class = P.queue[kHeavyClassWord];  // classify_tiles: the classes from here up fit the side kernel
        const uint32_t min_class = P.heavy_class > fit_class ? P.heavy_class : fit_class;
        heavy_next = traced && (cls + (heavy_now ? 1u : 0u) >= min_class);
    }
    // block-aggregated append to the class's list: one atomicAdd per block and class (same-address
    // atomics are the slow part: ~3 ns each, device-wide); slot kCostClasses + 1 = the next heavy list
    constexpr uint32_t kSlots = kCostClasses + (HEAVY ? 2u : 1u);
    __shared__ uint32_t n_cls[kSlots], base_cls[kSlots];
    if (threadIdx.x < kSlots) n_cls[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t rank = 0u, hrank = 0u;
    // a prefilled frame (Params::prefilled) needs no painting: tiles that are not traced are not listed
    const bool listed = in_band && !heavy_now && (traced || P.prefilled == 0u);
    if (listed) rank = atomicAdd(&n_cls[cls], 1u);
    if constexpr (HEAVY) {
        if (heavy_next) hrank = atomicAdd(&n_cls[kCostClasses + 1u], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= kCostClasses) base_cls[threadIdx.x] = n_cls[threadIdx.x] ? atomicAdd(&P.queue[P.qsel * 16u + threadIdx.x], n_cls[threadIdx.x]) : 0u;
    if constexpr (HEAVY) {
        if (threadIdx.x == kCostClasses + 1u)
            base_cls[threadIdx.x] = n_cls[threadIdx.x] ? atomicAdd(&P.queue[(P.qsel ^ 1u) * 16u + kHeavyCountWord], n_cls[threadIdx.x]) : 0u;
    }
    __syncthreads();
    if (listed) P.queue[kQueueBase + cls * tiles + base_cls[cls] + rank] = bx | (no_cubes ? kEntryNoCubes : 0u) | (no_spheres ? kEntryNoSpheres : 0u) | (by << 16);
    if (HEAVY && heavy_next) {
        const uint32_t slot = base_cls[kCostClasses + 1u] + hrank;
        if (slot < kHeavyCap) {  // a longer list is cut: the tiles beyond it stay with render_kernel (no mark)
            P.heavy[(P.qsel ^ 1u) * kHeavyCap + slot] = bx | (by << 16);
            P.heavy_mark[t] = P.heavy_next;
        }
    }
    // ... except for the reference's never-painted pixel i == width (render.rs:316-322; row 1, column 0): it is 0
    // whether its tile is traced (the render kernel writes 0 there too) or sky
    if (P.prefilled == 1u && t == 0u && P.part == 0u && P.y0 <= 1u && 1u < P.y1 && P.out) P.out[(size_t)(1u - P.y0) * P.width] = 0u;
    // prefilled == 2: this kernel is the fill.  The block paints the tiles it has just found to need no
    // tracing: thread j takes half a tile row (4 pixels, one 16-byte store), so a wave writes 1 KB of
    // consecutive addresses per pixel row wherever the tiles are consecutive; disjoint from every pixel
    // the render kernel writes.  (The host chooses this only for a 16-byte aligned buffer and width % 4 == 0.)
    if (P.prefilled == 2u) {
        __shared__ uint8_t paint[1024];
        paint[threadIdx.x] = (in_band && !traced && !heavy_now) ? 1u : 0u;
        __syncthreads();
        const uint32_t t0 = blockIdx.x * 1024u;
        for (uint32_t j = threadIdx.x; j < 2048u; j += 1024u) {
            const uint32_t lt = j >> 1;
            if (paint[lt] == 0u) continue;
            const uint32_t tt = t0 + lt;
            const uint32_t pbx = tt % P.grid_tx, pby = tt / P.grid_tx;
            const uint32_t x = pbx * 8u + (j & 1u) * 4u;
            if (x >= P.width) continue;
            const uint32_t fy0 = tile_y0(P, pby);
            uint32_t* row = P.out + (size_t)(pby * 8u) * P.width + x;
            for (uint32_t r = 0; r < 8u && fy0 + r < P.y1; ++r) {
                u32x4 v = {P.sky, P.sky, P.sky, P.sky};
                if (fy0 + r == 1u && x == 0u) v[0] = 0u;  // the never-painted pixel i == width
                *(u32x4*)(row + (size_t)r * P.width) = v;
            }
        }
    }
}

// render_kernel: single-wave workgroups, one lane per pixel of an 8x8 tile (coherent rays give tight
// wave-level bounds in the cull; a wave retires on its own instead of holding a 4-wave workgroup's
// slots).  One workgroup per POSSIBLE tile; workgroup i takes the i-th entry of the queue in
// heaviest-class-first order and leaves at once if there is none.  The hardware dispatcher is the
// dynamic scheduler: the frame time is bounded below by "start of the slowest tile + its duration", so
// the slow tiles (as measured in the previous launch) go first.  (A persistent variant taking tickets
// with atomicAdd lost 75 us per 4K frame to same-address atomic traffic.)
#ifndef RTX_WAVES_PER_EU
#define RTX_WAVES_PER_EU 4
#endif
#ifndef RTX_WAVES_PER_EU_LARGE
#define RTX_WAVES_PER_EU_LARGE 4  // long lists: 128 registers hold the per-lane BVH walk since pow_f64_rounded is out of line (three waves and 168 before)
#endif
template <bool DBG, bool COUNT, int LARGE, bool DARK = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LARGE == 1 ? RTX_WAVES_PER_EU_LARGE : RTX_WAVES_PER_EU, LARGE == 1 ? RTX_WAVES_PER_EU_LARGE : RTX_WAVES_PER_EU))) void render_kernel(const Params P)
{
    constexpr int KEYS = (!DBG && !COUNT && LARGE == 3) ? 2 : 0;  // bundle keys: cube_query's SPLIT == 2
    __shared__ uint32_t stk[kStackDepth * frame_words<DBG, KEYS>() * 64];
    const uint32_t lane = threadIdx.x & 63u;
    Totals cnt;
#pragma unroll
    for (int k = 0; k < 10 + EV_COUNT; ++k) cnt.v[k] = 0ull;
    // The queue is final (build_queue has completed).  Item i = the i-th traced tile in heaviest-first
    // order, then the paint groups; a workgroup takes items blockIdx.x, + gridDim.x, ...: with a grid of a
    // few times the resident workgroups the dispatcher still balances the heavy first items, the later
    // (light) ones ride along, and there are no surplus workgroups to launch (one per possible tile cost
    // the 4K frame 30 us of dispatching at its end).
    for (uint32_t item = blockIdx.x;; item += gridDim.x) {
    const unsigned long long t_start = __builtin_readcyclecounter();
    uint32_t idx = item, entry = 0xFFFFFFFFu;
    bool painted = false;
    {
        const uint32_t tiles = P.grid_tx * P.grid_ty;
        KCONST uint32_t* q = (KCONST uint32_t*)P.queue;
        bool found = false;
        for (uint32_t k = kCostClasses; k-- > 0u;) {
            const uint32_t n_k = q[P.qsel * 16u + k];
            if (!found && idx < n_k) {
                entry = q[kQueueBase + k * tiles + idx];
                found = true;
            }
            idx -= found ? 0u : n_k;
        }
        if (!found) {
            // The traced tiles are all handed out: this workgroup paints kPaintGroup of the others (sky
            // colour, "no hit" planes) and counts the rays the reference would still have cast for them.
            // plain frames are prefilled with the sky colour (rt_render_device_part) and list no paint-only tiles: no
            // traced tile left means done; the variants with debug planes or counters paint and count here
            if constexpr (!DBG && !COUNT) break;
            KCONST uint32_t* sky_list = q + kQueueBase + kCostClasses * tiles;
            const uint32_t n_sky = q[P.qsel * 16u + kCostClasses];
            if (idx * kPaintGroup >= n_sky) break;  // no items left (every wave gets here: items only grow)
            const uint32_t no_hash = hmix(kHashBasis, kNoHit);
            const uint32_t px_hash = hmix(hmix(kHashBasis, no_hash), no_hash);
            for (uint32_t j = idx * kPaintGroup; j < n_sky && j < (idx + 1u) * kPaintGroup; ++j) {
                const uint32_t e = sky_list[j];
                const uint32_t bx = e & kEntryColMask, by = e >> 16;
                const uint32_t x = bx * 8u + (lane & 7u), y = tile_y0(P, by) + (lane >> 3);
                if (x < P.width && y < P.y1) {
                    const size_t i_band = (size_t)(by * 8u + (lane >> 3)) * P.width + x;
                    if constexpr (!COUNT) {
                        // render.rs:316-322: pixel i == width is never painted
                        if (P.out) P.out[i_band] = ((size_t)y * P.width + x == (size_t)P.width) ? 0u : P.sky;
                    }
                    if constexpr (DBG) {
                        if (P.hit_cube) P.hit_cube[i_band] = kNoHit;
                        if (P.hit_sphere) P.hit_sphere[i_band] = kNoHit;
                        if (P.tree_hash) P.tree_hash[i_band] = px_hash;
                    }
                    if constexpr (COUNT) {  // one nearest-hit walk of the sphere list, one of the cube list
                        cnt.v[0] += 2ull; cnt.v[5] += 2ull;
                        cnt.v[2] += P.n_cubes; cnt.v[7] += P.n_cubes;
                        cnt.v[3] += P.n_spheres; cnt.v[8] += P.n_spheres;
                    }
                }
            }
            painted = true;
        }
    }
    if (!painted) {
        const float width = P.width_f, height = P.height_f;
        const float aspect = P.aspect;
        const V3 right = v3(P.right[0], P.right[1], P.right[2]);
        const V3 up = v3(P.up[0], P.up[1], P.up[2]);
        const V3 fwd = v3(P.fwd[0], P.fwd[1], P.fwd[2]);
        const V3 eye = v3(P.eye[0], P.eye[1], P.eye[2]);
        const uint32_t bx = entry & kEntryColMask, by = entry >> 16;
        const uint32_t ty0 = tile_y0(P, by);
        const uint32_t x = bx * 8u + (lane & 7u);
        const uint32_t y = ty0 + (lane >> 3);
        const uint32_t out_row = by * 8u + (lane >> 3);
        PROF_DECL
        const bool active = (x < P.width) && (y < P.y1);  // lanes outside the frame stay in the wave, inactive

        // render.rs:298-312: a spheres-only pass with ambient 1, a cubes-only pass with ambient_light; the
        // sphere colour wins unless it equals the sky colour.  Passes that cannot hit anything are not
        // traced (wave-uniform): no spheres at all, or no cube projecting into the tile.
        const bool skip_s = (P.n_spheres == 0u) || ((entry & kEntryNoSpheres) != 0u);  // no spheres, or none projects into the tile
        const bool skip_c = (entry & kEntryNoCubes) != 0u;  // no cube projects into the tile (classification)
        V3 dir = v3(0.0f, 0.0f, 0.0f);
        if (!(skip_s && skip_c)) {
            float sx = (2.0f * (float)x) / width - 1.0f;
            const float sy = -(2.0f * (float)y) / height + 1.0f;
            sx = sx * aspect;
            const V3 v = vnormalize(v3(sx, sy, -1.0f));
            // camera.rs:38-40: v.x*right + v.y*up - v.z*forward, normalised
            dir = vnormalize(vsub(vadd(vscale(right, v.x), vscale(up, v.y)), vscale(fwd, v.z)));
        }
        PROF_MARK(PH_RAYGEN);
        if constexpr (COUNT) cnt.v[10 + EV_PRIMARY] += (active && !(skip_s && skip_c)) ? 1ull : 0ull;
        uint32_t px = 0u, hs = 0u, hc = 0u, hit_s = kNoHit, hit_c = kNoHit;
        for (uint32_t pass = PASS_SPHERES; pass <= PASS_CUBES; ++pass) {
            uint32_t hh, hi, c;
            const bool skip = (pass == PASS_SPHERES) ? skip_s : skip_c;
            if (skip) {
                c = P.sky;
                hi = kNoHit;
                hh = hmix(kHashBasis, kNoHit);
                if constexpr (COUNT) {  // the reference still walks the list for this ray
                    const unsigned long long one = active ? 1ull : 0ull;
                    const unsigned long long tests = active ? (pass == PASS_CUBES ? P.n_cubes : P.n_spheres) : 0ull;
                    const int slot = (pass == PASS_CUBES) ? 2 : 3;  // cube_tests / sphere_tests
                    cnt.v[0] += one; cnt.v[5] += one;
                    cnt.v[slot] += tests; cnt.v[5 + slot] += tests;
                }
            } else {
                const Mailbox none{nullptr, nullptr, nullptr};
                c = trace_tree<DBG, COUNT, LARGE, DARK, false, KEYS>(P, pass, active, eye, dir, pass == PASS_SPHERES ? 1.0f : P.ambient, stk, hh, hi, cnt, none PROF_PASS);
            }
            if (pass == PASS_SPHERES) { px = c; hs = hh; hit_s = hi; }
            else { px = (px == P.sky) ? c : px; hc = hh; hit_c = hi; }
        }
        PROF_MARK(PH_OTHER);
        const size_t i_frame = (size_t)y * P.width + x;
        if (i_frame == (size_t)P.width) px = 0u;  // render.rs:316-322: pixel i == width is never painted
        const size_t i_band = (size_t)out_row * P.width + x;
        if (active) {
            if constexpr (!COUNT) {
                if (P.out) P.out[i_band] = px;
            }
            if constexpr (DBG) {
                if (P.hit_cube) P.hit_cube[i_band] = hit_c;
                if (P.hit_sphere) P.hit_sphere[i_band] = hit_s;
                if (P.tree_hash) P.tree_hash[i_band] = hmix(hmix(kHashBasis, hs), hc);
            }
        }
        if (lane == 0u) P.tile_cost[(size_t)by * P.grid_tx + bx] = (uint32_t)(__builtin_readcyclecounter() - t_start) | 1u;
#ifdef RTX_PROFILE
        PROF_MARK(PH_OTHER);
        if (lane == 0u && P.counters) {
            // per-tile slots (no atomics: contended atomics would dominate the run); the host reads them
            uint32_t* slot = (uint32_t*)P.counters + 64 + ((size_t)by * P.grid_tx + bx) * 40u;
            for (int k = 0; k < 10; ++k) slot[k] = prof.t[k];
            slot[10] = prof_t0;
            slot[11] = prof.last;
            slot[12] = prof_r0;
            slot[13] = (uint32_t)__builtin_amdgcn_s_memrealtime();  // 100 MHz, common to the chip
            slot[14] = blockIdx.x;
            for (int k = 0; k < 8; ++k) { slot[16 + k] = prof.c[k]; slot[24 + k] = prof.qk[k]; slot[32 + k] = prof.ck[k]; }
        }
#endif
    }
    }  // items
    if constexpr (COUNT) {
        // wave reduction, then one atomic per wave and counter
#pragma unroll
        for (int k = 0; k < 10 + EV_COUNT; ++k) {
            unsigned long long v = cnt.v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0u && v != 0ull) atomicAdd(&P.counters[k], v);
        }
    }
}

// ---------------------------------------------------------------- render_heavy_kernel (heavy tiles of a plain frame)
// Launched on the context's second stream, concurrently with the frame's classification, listing and
// render_kernel: workgroups of three waves striding over the launch's heavy list (Params::heavy), the three
// waves as tracer and shaders (see "heavy tiles" above).  Every pixel is computed by trace_tree, as in
// render_kernel.  The list was written by the previous launch's build_queue, for whatever camera that launch
// had: an entry only says "trace this tile", so the cube pass is never skipped here.
template <int LARGE, bool DARK, bool ASYNC>
DEV void plain_tile(const Params& P, uint32_t entry, uint32_t* stk, const Mailbox& mb, uint32_t cost_scale)
{
    const unsigned long long t_start = __builtin_readcyclecounter();
    const uint32_t lane = threadIdx.x & 63u;
    const float width = P.width_f, height = P.height_f;
    const float aspect = P.aspect;
    const V3 right = v3(P.right[0], P.right[1], P.right[2]);
    const V3 up = v3(P.up[0], P.up[1], P.up[2]);
    const V3 fwd = v3(P.fwd[0], P.fwd[1], P.fwd[2]);
    const V3 eye = v3(P.eye[0], P.eye[1], P.eye[2]);
    const uint32_t bx = entry & kEntryColMask, by = entry >> 16;
    const uint32_t ty0 = tile_y0(P, by);
    const uint32_t x = bx * 8u + (lane & 7u);
    const uint32_t y = ty0 + (lane >> 3);
    const uint32_t out_row = by * 8u + (lane >> 3);
    const bool active = (x < P.width) && (y < P.y1);  // lanes outside the frame stay in the wave, inactive
    // render.rs:298-312 (see render_kernel)
    const bool skip_s = (P.n_spheres == 0u);
    const bool skip_c = (entry & kEntryNoCubes) != 0u;
    V3 dir = v3(0.0f, 0.0f, 0.0f);
    if (!(skip_s && skip_c)) {
        float sx = (2.0f * (float)x) / width - 1.0f;
        const float sy = -(2.0f * (float)y) / height + 1.0f;
        sx = sx * aspect;
        const V3 v = vnormalize(v3(sx, sy, -1.0f));
        dir = vnormalize(vsub(vadd(vscale(right, v.x), vscale(up, v.y)), vscale(fwd, v.z)));  // camera.rs:38-40
    }
    uint32_t px = 0u;
    Totals unused;
    PROF_DECL
    for (uint32_t pass = PASS_SPHERES; pass <= PASS_CUBES; ++pass) {
        uint32_t hh, hi, c;
        const bool skip = (pass == PASS_SPHERES) ? skip_s : skip_c;
        if (skip) c = P.sky;
        else c = trace_tree<false, false, LARGE, DARK, ASYNC>(P, pass, active, eye, dir, pass == PASS_SPHERES ? 1.0f : P.ambient, stk, hh, hi, unused, mb PROF_PASS);
        if (pass == PASS_SPHERES) px = c;
        else px = (px == P.sky) ? c : px;
    }
    const size_t i_frame = (size_t)y * P.width + x;
    if (i_frame == (size_t)P.width) px = 0u;  // render.rs:316-322: pixel i == width is never painted
    // The side kernel's stores are device-coherent ones (relaxed atomic stores at agent scope: they go through
    // to where every XCD reads them), so that a finished tile can be counted behind a wait for these stores
    // alone: a release fence at agent scope writes the XCD's whole L2 back, per tile, under render_kernel's
    // own stores (1080p frame +25 %).
    if (active && P.out) __hip_atomic_store(&P.out[(size_t)out_row * P.width + x], px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // cycles the tile took, in one-wave terms (a heavy tile's three waves shorten it by about cost_scale)
    if (lane == 0u) {
        const unsigned long long cyc = (__builtin_readcyclecounter() - t_start) * cost_scale;
        __hip_atomic_store(&P.tile_cost[(size_t)by * P.grid_tx + bx], (uint32_t)(cyc > 0xFFFFFFFFull ? 0xFFFFFFFFull : cyc) | 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A shader wave's part of one heavy item: serve the nodes the tracer posts until it reports the item done.
template <int LARGE>
DEV void shader_serve(const Params& P, const Mailbox& mb, uint32_t w, uint32_t items_target)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t spins = 0u;
    for (;;) {
        uint32_t k = kMailDepth;
        for (uint32_t j = kMailDepth; j-- > 0u;) {
            const uint32_t pj = lds_load_acquire(&mb.ctl->posted[j]);
            if (pj != mb.ctl->done[j][w]) k = j;  // done[j][w] is this wave's own word
        }
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        if (k == kMailDepth) {
            // nothing posted: the item is over once the tracer says so (it collects every node it posted first)
            if (lds_load_acquire(&mb.ctl->items_done) >= items_target) return;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) { mb.ctl->error = 2u; return; }
            continue;
        }
        spins = 0u;
#ifdef RTX_PROFILE
        const uint32_t prof_c0 = (uint32_t)__builtin_readcyclecounter();
#endif
        const uint32_t seq = mb.ctl->posted[k];
        const uint32_t* r = mb.rec + k * (kRecWords * 64u) + lane;
        const V3 hit_p = v3(__builtin_bit_cast(float, r[0 * 64]), __builtin_bit_cast(float, r[1 * 64]), __builtin_bit_cast(float, r[2 * 64]));
        const V3 hit_n = v3(__builtin_bit_cast(float, r[3 * 64]), __builtin_bit_cast(float, r[4 * 64]), __builtin_bit_cast(float, r[5 * 64]));
        const V3 view_dir = v3(__builtin_bit_cast(float, r[6 * 64]), __builtin_bit_cast(float, r[7 * 64]), __builtin_bit_cast(float, r[8 * 64]));
        const Col tx_a = col_from_hex(r[9 * 64]);
        const uint32_t mw = r[10 * 64];
        const uint32_t hit_id = r[11 * 64];
        const uint32_t key = r[12 * 64];
        const bool lit = (mw & 0x80000000u) != 0u;
        const MatRec m = P.mats[mw & 0x7FFFFFFFu];
        const float k_ds = 1.0f - m.reflectivity - m.transparency;  // the expression of trace_tree
        const uint32_t pass = mb.ctl->pass_ambient[k][0];
        const float ambient = __builtin_bit_cast(float, mb.ctl->pass_ambient[k][1]);
        Col sum = Col{0.0f, 0.0f, 0.0f};
        Counters cnt = {};
        uint32_t h = 0u;
        PROF_DECL
        for (uint32_t li = w; li < P.n_lights; li += kShaders) {
            const Col term = light_term<false, false, LARGE, true>(P, pass, li, lit, lit, hit_p, hit_n, hit_id, view_dir, tx_a, m.specular, m.albedo1, k_ds, ambient, h, cnt, key PROF_PASS);
            sum = col_add(sum, term);
        }
        mb.res[(k * kShaders + w) * 64u + lane] = col_to_hex(sum);
#ifdef RTX_PROFILE
        if (lane == 0u) { mb.ctl->prof_busy[w] += (uint32_t)__builtin_readcyclecounter() - prof_c0; mb.ctl->prof_nodes[w] += 1u; }
#endif
        lds_store_release(&mb.ctl->done[k][w], seq);
    }
}

constexpr uint32_t kHeavyCostScale = 3u;
#ifndef RTX_HEAVY_WAVES_PER_EU
#define RTX_HEAVY_WAVES_PER_EU 2  // 256 registers, nothing spilled on the critical path (1080p frame 0.090 -> 0.080 ms against 4)
#endif
template <int LARGE, bool DARK>
__global__ __launch_bounds__(64 * kWgWaves) __attribute__((amdgpu_waves_per_eu(RTX_HEAVY_WAVES_PER_EU, RTX_HEAVY_WAVES_PER_EU))) void render_heavy_kernel(const Params P)
{
    constexpr uint32_t kFrame = kStackDepth * frame_words<false, true>() * 64;
    __shared__ uint32_t stk[kFrame];      // the tracer's frame stack
    __shared__ uint32_t mail[kMailWords]; // records and partial sums, see Mailbox
    __shared__ MailCtl ctl;
    const uint32_t wave = threadIdx.x >> 6;
    if (threadIdx.x < sizeof(MailCtl) / 4u) ((uint32_t*)&ctl)[threadIdx.x] = 0u;
    __syncthreads();  // the only workgroup barrier: from here on the waves meet through the mailbox
    const Mailbox mb{mail, mail + kMailDepth * kRecWords * 64u, &ctl};
    const uint32_t* q = P.queue;
    const uint32_t listed = q[P.qsel * 16u + kHeavyCountWord];
    const uint32_t n_heavy = listed < kHeavyCap ? listed : kHeavyCap;
    const uint32_t* list = P.heavy + P.qsel * kHeavyCap;
    uint32_t items = 0u;
    for (uint32_t item = blockIdx.x; item < n_heavy; item += gridDim.x) {
        // an entry outside the launch's tile grid cannot come from this launch's list (a gate that gave up lets
        // the kernel start before the list is whole): skipped by every wave alike, and counted as finished
        const uint32_t raw = list[item];
        if ((raw & kEntryColMask) >= P.grid_tx || (raw >> 16) >= P.grid_ty) {
            if (threadIdx.x == 0u) __hip_atomic_fetch_add(&P.queue[P.qsel * 16u + kHeavyDoneWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        ++items;
        if (wave == 0u) {
            const uint32_t entry = raw & ~(kEntryNoCubes | kEntryNoSpheres);
#ifdef RTX_PROFILE
            const uint32_t prof_c0 = (uint32_t)__builtin_readcyclecounter();
            if (threadIdx.x == 0u) { ctl.prof_wait = 0u; for (uint32_t w = 0; w < kShaders; ++w) { ctl.prof_busy[w] = 0u; ctl.prof_nodes[w] = 0u; } }
#endif
            plain_tile<LARGE, DARK, true>(P, entry, stk, mb, kHeavyCostScale);
#ifdef RTX_PROFILE
            if (threadIdx.x == 0u && P.counters) {
                uint32_t* slot = (uint32_t*)P.counters + 64 + ((size_t)(entry >> 16) * P.grid_tx + (entry & kEntryColMask)) * 40u;
                slot[0] = (uint32_t)__builtin_readcyclecounter() - prof_c0;
                slot[1] = ctl.prof_wait;
                for (uint32_t w = 0; w < kShaders; ++w) { slot[2 + w] = ctl.prof_busy[w]; slot[4 + w] = ctl.prof_nodes[w]; }
            }
#endif
            lds_store_release(&ctl.items_done, items);
            // the tile's pixels and cost word are visible device-wide before it counts as finished (heavy_join):
            // see plain_tile's stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt: the wave's stores above have completed
            if (threadIdx.x == 0u) __hip_atomic_fetch_add(&P.queue[P.qsel * 16u + kHeavyDoneWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            shader_serve<LARGE>(P, mb, wave - 1u, items);
        }
    }
    if (threadIdx.x == 0u && ctl.error != 0u) P.queue[kQueueBase - 1u] = ctl.error;  // header word 63: rt_debug_status
    // This workgroup is done with the launch's header bank and list (it read the count when it started): heavy_join
    // waits for every workgroup's mark, not only for the tiles, because the launch after next reuses the bank - a
    // workgroup that started late would otherwise read that launch's count and trace its tiles with this launch's
    // camera (an empty list ends the join at once, and the side kernel shares hardware queues with other streams).
    __syncthreads();
    if (threadIdx.x == 0u) __hip_atomic_fetch_add(&P.queue[P.qsel * 16u + kHeavyExitWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// heavy_gate: the side stream waits here, one wave, until classify_tiles of the same launch says go
// (kHeavyGoWord; sequence numbers only grow).  The side stream is not fenced against the context's stream by
// events - an event record costs that stream 10 us per frame - so render_heavy_kernel, which follows this
// kernel on the side stream, could otherwise start while the previous frame is still being rendered: before
// the list it reads is complete and before the caller is done with the output buffer.  (Waiting inside
// render_heavy_kernel itself was tried: its workgroups become resident early and hold half of the register
// files while they wait; two contexts' worth of them leave no room for the kernel they wait for.)  Every wait
// between the two streams points at a kernel that was enqueued earlier, so the waits cannot form a cycle.
__global__ __launch_bounds__(64) void heavy_gate(const Params P)
{
    uint32_t spins = 0u;
    // ">=", wrap-safe: a launch with an empty list is not waited for by heavy_join, so the next launch's
    // classify_tiles may have stored its own, later number before this wave looks
    while ((int32_t)(__hip_atomic_load(&P.queue[kHeavyGoWord], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - P.heavy_now) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 24)) {
            if (threadIdx.x == 0u) P.queue[kQueueBase - 1u] = 4u;
            break;
        }
    }
}

// heavy_join: the context's stream meets the side stream here.  One wave polls the launch's count of
// finished heavy tiles until it reaches the list's length; the kernel's end then orders everything after it
// on the stream behind those tiles' stores.  The side kernel needs nothing from this one, so the wait cannot
// deadlock; a wave that gives up (seconds) reports through rt_debug_status.
__global__ __launch_bounds__(64) void heavy_join(const Params P)
{
    const uint32_t listed = __hip_atomic_load(&P.queue[P.qsel * 16u + kHeavyCountWord], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t want = listed < kHeavyCap ? listed : kHeavyCap;
    const uint32_t limit = P.heavy_spin_limit ? P.heavy_spin_limit : (1u << 24);
    // heavy_spin_limit == 1 (RT_TUNE_HEAVY_SPIN_LIMIT, the recovery path's test): no patience at all - a launch
    // with a side kernel is reported as given up without a look at the counts
    bool gave_up = (P.heavy_spin_limit == 1u);
    uint32_t spins = 0u;
    // every listed tile finished AND every workgroup of render_heavy_kernel gone (see the kernel's last lines)
    while (!gave_up && (__hip_atomic_load(&P.queue[P.qsel * 16u + kHeavyDoneWord], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want ||
                        __hip_atomic_load(&P.queue[P.qsel * 16u + kHeavyExitWord], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < kHeavyGrid)) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > limit) gave_up = true;
    }
    if (gave_up && threadIdx.x == 0u) P.queue[kQueueBase - 1u] = 3u;
}

// ---------------------------------------------------------------- the wavefront path (WfParams, rtx950_device.h)
// cast_ray's recursion (render.rs:154-193) as one launch per ray-tree level.  The tile-resident walk above keeps a
// tile on one wave from its primary rays to the last grandchild: a chain of up to four nodes with three list
// traversals each, and after the first bounce the 64 lanes of a tile across a cube edge go several ways, so the
// wave-level cull keeps every cube.  Here a node is traced by the wave of its ITEM: the rays of one level that
// left one parent wave through the same kind of child off the same cube face.  Such rays are coherent whatever the
// tile looked like, a launch's critical path is one node, and the deeper levels of all tiles fill the GPU together.
// Every pixel is computed by the same functions in the same arithmetic as in trace_tree (nearest_query,
// cube_resolve, light_term, reflect/refract, node_finish); only the wave that executes them differs.
struct WfNode {
    uint32_t sumf;   // packed light sum | kHitFlag
    uint32_t mat;    // material index of the hit (0 without one)
    uint32_t face;   // cube face of the hit: the bundle key of the node's children
    bool has_refl, has_refr;
    V3 refl_o, refl_d, refr_o, refr_d;
};

// One node of cast_ray for the wave's rays: render.rs:87-152 and the child rays of :154-193 (trace_tree's node
// code without the tree walk around it).
template <int LARGE, bool DARK>
DEV WfNode wf_node(const Params& P, uint32_t pass, bool act, V3 o, V3 d, float ambient PROF_ARG)
{
    WfNode r;
    Counters cnt = {};
    bool div_ok;
    const uint32_t idx = nearest_query<false, LARGE>(P, pass, act, o, d, div_ok, cnt, 0u PROF_PASS);
    const bool hit_any = act && (idx != kNoHit);
    Col sum = Col{0.0f, 0.0f, 0.0f};
    r.mat = 0u;
    r.face = 0u;
    r.has_refl = false;
    r.has_refr = false;
    r.refl_o = o; r.refl_d = d; r.refr_o = o; r.refr_d = d;
    if (wave_any(hit_any)) {
        const uint32_t sidx = hit_any ? idx : 0u;  // lanes without a hit shade object 0 and discard the result
        Hit hit;
        if (pass == PASS_CUBES) hit = cube_resolve(P, sidx, o, d, div_ok);
        else hit = sphere_resolve(P, sidx, o, d);
        const MatRec m = P.mats[hit.material];
        Col tx;
        if (m.texture >= 0) tx = texel_of_face(P, m.texture, hit.face, hit.u * P.sprite, hit.v * P.sprite);
        else tx = col_from_hex(m.diffuse_0rgb);
        const Col tx_a = col_mul(tx, m.albedo0);
        const V3 view_dir = vnormalize(vsub(o, hit.p));
        const float k_ds = 1.0f - m.reflectivity - m.transparency;
        const uint32_t nl = P.n_lights;
        bool lit = hit_any;
        uint32_t nl_cast = nl;
        if constexpr (DARK) {  // direct light that is exactly black casts no shadow rays (see trace_tree)
            lit = hit_any && !(k_ds < 0.0039f);
            nl_cast = wave_any(lit) ? nl : 0u;
        }
        uint32_t h_unused = 0u;
        for (uint32_t li = 0; li < nl_cast; ++li) {
            const Col term = light_term<false, false, LARGE>(P, pass, li, lit, hit_any, hit.p, hit.n, hit.id, view_dir, tx_a, m.specular, m.albedo1, k_ds,
                                                             ambient, h_unused, cnt, 0u PROF_PASS);
            sum = col_add(sum, term);
        }
        // children (render.rs:154-193); the reference casts them inside the light loop: without lights there are none
        r.has_refl = hit_any && (m.reflectivity > 0.0f) && (nl != 0u);
        r.has_refr = hit_any && (m.transparency > 0.0f) && (nl != 0u);
        if (wave_any(r.has_refl || r.has_refr)) {
            r.refl_d = vnormalize(reflect(vneg(d), hit.n));
            r.refl_o = vadd(hit.p, vscale(hit.n, 1e-2f));
            r.refr_d = refract(d, hit.n, m.ior);
            r.refr_o = vadd(hit.p, vscale(hit.n, 1e-3f));
        }
        r.mat = hit_any ? hit.material : 0u;
        r.face = (uint32_t)hit.face;
    }
    r.sumf = col_to_hex(sum) | (hit_any ? kHitFlag : 0u);
    return r;
}

DEV WfItem wf_load_item(const WfItem* base, uint32_t i)
{
    const u32x8 v = ((KCONST u32x8*)base)[i];
    const uint32_t e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3], e4 = v[4], e5 = v[5], e6 = v[6], e7 = v[7];
    return WfItem{e0, e1, e2, e3, e4, e5, e6, e7};
}

// The tile behind `entry` cannot be finished by the wavefront launches (its items did not fit): render_kernel
// traces it whole after them.  Once per tile and launch.
DEV void wf_mark_redo(const Params& P, const WfParams& W, uint32_t entry)
{
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t t = (entry >> 16) * P.grid_tx + (entry & kEntryColMask);
        if (atomicExch(&W.redo_mark[t], W.seq) != W.seq) {
            const uint32_t slot = atomicAdd(&W.header[kWfRedoBase], 1u);
            W.header[kWfRedoBase + kQueueBase + slot] = entry;  // at most one entry per tile of the launch
        }
    }
}

// The bundles of the lanes in `m`: lanes whose node hit the same face go together, the kWfBundles-th takes the rest.
// f(sel, j) is called per bundle j with its lane mask; returns the number of bundles.  Wave-uniform.
template <typename F>
DEV uint32_t wf_split(unsigned long long m, uint32_t face, F f)
{
    uint32_t nb = 0u;
    while (m != 0ull) {
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)face, (int)__builtin_ctzll(m));
        const unsigned long long sel = (nb + 1u == kWfBundles) ? m : (m & wave_ballot(face == f0));
        f(sel, nb);
        m &= ~sel;
        ++nb;
    }
    return nb;
}

// The children of the node `n` of item `item` at `level`: one item per kind and bundle at level + 1, their rays
// stored per lane.  One atomic add per emitting wave.
DEV void wf_emit(const Params& P, const WfParams& W, uint32_t level, uint32_t item, uint32_t pass, uint32_t entry, const WfNode& n, uint32_t& first_child,
                 uint32_t& n_children)
{
    first_child = 0u;
    n_children = 0u;
    if (!(level < P.max_depth)) return;  // children beyond max_depth return the sky colour without tracing (render.rs:80-82)
    const unsigned long long m_refl = wave_ballot(n.has_refl), m_refr = wave_ballot(n.has_refr);
    if ((m_refl | m_refr) == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t want = wf_split(m_refl, n.face, [](unsigned long long, uint32_t) {}) + wf_split(m_refr, n.face, [](unsigned long long, uint32_t) {});
    // allocation in the sub-queue of this item's index (see kWfQueues)
    const uint32_t sq = item & (kWfQueues - 1u);
    const uint32_t cap = W.cap[level + 1u] / kWfQueues;
    uint32_t base = 0u;
    if (lane == 0u) base = atomicAdd(&W.header[(level * kWfQueues + sq) * 16u], want);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= cap || want > cap - base) {
        wf_mark_redo(P, W, entry);
        return;
    }
    base += sq * cap;
    WfItem* items = W.items[level + 1u];
    float* rays = W.rays[level + 1u];
    uint32_t ci = base;
    for (uint32_t kind = 0u; kind < 2u; ++kind) {
        const V3 co = kind ? n.refr_o : n.refl_o, cd = kind ? n.refr_d : n.refl_d;
        wf_split(kind ? m_refr : m_refl, n.face, [&](unsigned long long sel, uint32_t) {
            if ((sel >> lane) & 1ull) {
                float* r = rays + (size_t)ci * kWfRayWords + lane;
                r[0 * 64] = co.x; r[1 * 64] = co.y; r[2 * 64] = co.z;
                r[3 * 64] = cd.x; r[4 * 64] = cd.y; r[5 * 64] = cd.z;
            }
            if (lane == 0u) {
                u32x8 rec = {item, kind | (pass << 1), (uint32_t)sel, (uint32_t)(sel >> 32), entry, 0u, 0u, W.seq};
                *(u32x8*)(items + ci) = rec;
            }
            ++ci;
        });
    }
    first_child = base;
    n_children = want;
}

template <bool PRIMARY, int LARGE, bool DARK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LARGE == 1 ? RTX_WAVES_PER_EU_LARGE : RTX_WAVES_PER_EU, LARGE == 1 ? RTX_WAVES_PER_EU_LARGE : RTX_WAVES_PER_EU))) void wf_trace_kernel(const Params P, const WfParams W)
{
    const uint32_t lane = threadIdx.x & 63u;
    PROF_DECL
    if constexpr (PRIMARY) {
        // level 0: the traced tiles of the queue, heaviest first, as in render_kernel.  A work unit is one PASS of a
        // tile (render.rs:248-306: spheres with ambient 1, cubes with ambient_light): one unit per tile when the scene
        // has no spheres, two otherwise, so that the node code below has no loop around it (a pass loop around it
        // doubled the kernel's spills).
        const uint32_t two = P.n_spheres != 0u ? 1u : 0u;
        for (uint32_t unit = blockIdx.x;; unit += gridDim.x) {
            const unsigned long long t_start = __builtin_readcyclecounter();
            const uint32_t item = unit >> two;
            const uint32_t pass = two ? (unit & 1u) : (uint32_t)PASS_CUBES;
            uint32_t idx = item, entry = 0xFFFFFFFFu;
            const uint32_t tiles = P.grid_tx * P.grid_ty;
            KCONST uint32_t* q = (KCONST uint32_t*)P.queue;
            bool found = false;
            for (uint32_t k = kCostClasses; k-- > 0u;) {
                const uint32_t n_k = q[P.qsel * 16u + k];
                if (!found && idx < n_k) {
                    entry = q[kQueueBase + k * tiles + idx];
                    found = true;
                }
                idx -= found ? 0u : n_k;
            }
            if (!found) break;  // plain frames list no paint-only tiles (build_queue paints them)
            const uint32_t bx = entry & kEntryColMask, by = entry >> 16;
            const uint32_t x = bx * 8u + (lane & 7u);
            const uint32_t y = tile_y0(P, by) + (lane >> 3);
            const bool active = (x < P.width) && (y < P.y1);
            // a pass that cannot hit anything is not traced: its root stays "no hit" (the sky colour)
            const bool skip = (pass == PASS_SPHERES) ? ((P.n_spheres == 0u) || ((entry & kEntryNoSpheres) != 0u)) : ((entry & kEntryNoCubes) != 0u);
            WfNode n;
            n.sumf = 0u; n.mat = 0u; n.face = 0u; n.has_refl = false; n.has_refr = false;
            uint32_t fc = 0u, nc = 0u;
            if (!skip) {
                // render.rs:253-263, camera.rs:38-40
                const float width = P.width_f, height = P.height_f;
                const float aspect = P.aspect;
                float sx = (2.0f * (float)x) / width - 1.0f;
                const float sy = -(2.0f * (float)y) / height + 1.0f;
                sx = sx * aspect;
                const V3 v = vnormalize(v3(sx, sy, -1.0f));
                const V3 right = v3(P.right[0], P.right[1], P.right[2]), up = v3(P.up[0], P.up[1], P.up[2]), fwd = v3(P.fwd[0], P.fwd[1], P.fwd[2]);
                const V3 dir = vnormalize(vsub(vadd(vscale(right, v.x), vscale(up, v.y)), vscale(fwd, v.z)));
                n = wf_node<LARGE, DARK>(P, pass, active, v3(P.eye[0], P.eye[1], P.eye[2]), dir, pass == PASS_SPHERES ? 1.0f : P.ambient PROF_PASS);
                wf_emit(P, W, 0u, 2u * item + pass, pass, entry, n, fc, nc);
            }
            if (!two && nc == 0u) {
                // one pass and no traced children: the pixel is final (the sphere pass is the sky colour, so the merge
                // of render.rs:308-312 takes this pass; children beyond max_depth are the sky colour)
                const MatRec* mp = P.mats + n.mat;
                const float k_r = mp->reflectivity, k_t = mp->transparency;
                const bool hit_any = (n.sumf & kHitFlag) != 0u;
                const bool has_refl = hit_any && (k_r > 0.0f) && (P.n_lights != 0u);
                const bool has_refr = hit_any && (k_t > 0.0f) && (P.n_lights != 0u);
                const uint32_t zero = 0u;
                uint32_t px = node_finish(P, n.sumf, k_r, k_t, has_refl ? P.sky : zero, has_refr ? P.sky : zero);
                if ((size_t)y * P.width + x == (size_t)P.width) px = 0u;  // render.rs:316-322: pixel i == width is never painted
                if (active && P.out) P.out[(size_t)(by * 8u + (lane >> 3)) * P.width + x] = px;
            } else {
                // the root waits: wf_combine_kernel finishes the tile from both passes' roots
                const unsigned long long am = wave_ballot(active);
                const uint32_t ri = 2u * item + pass;
                uint32_t* nd = W.nodes[0] + (size_t)ri * kWfNodeWords + lane;
                nd[0] = n.sumf;
                nd[64] = n.mat;
                if (lane == 0u) {
                    u32x8 rec = {0u, pass << 1, (uint32_t)am, (uint32_t)(am >> 32), entry, fc, nc, W.seq};
                    *(u32x8*)(W.items[0] + ri) = rec;
                    if (pass == PASS_CUBES) {  // once per tile
                        const uint32_t sq = item & (kWfQueues - 1u);  // at most ceil(tiles / kWfQueues) positions map here: it cannot overflow
                        W.pending[sq * W.pending_cap + atomicAdd(&W.header[(kWfPendingCounter + sq) * 16u], 1u)] = item;
                    }
                }
            }
            if (lane == 0u && pass == PASS_CUBES) P.tile_cost[(size_t)by * P.grid_tx + bx] = (uint32_t)(__builtin_readcyclecounter() - t_start) | 1u;
        }
    } else {
        const uint32_t level = W.level;
        KCONST uint32_t* hdr = (KCONST uint32_t*)W.header;
        // workgroup b serves sub-queue b % kWfQueues (the grid is a multiple of kWfQueues)
        const uint32_t sq = blockIdx.x & (kWfQueues - 1u);
        const uint32_t cap = W.cap[level] / kWfQueues;
        uint32_t n = hdr[((level - 1u) * kWfQueues + sq) * 16u];
        n = n < cap ? n : cap;
        const WfItem* items = W.items[level];
        const float* rays = W.rays[level];
        for (uint32_t j = blockIdx.x / kWfQueues; j < n; j += gridDim.x / kWfQueues) {
            const uint32_t it = sq * cap + j;
            const unsigned long long t_item = __builtin_readcyclecounter();
            const WfItem rec = wf_load_item(items, it);
            if (rec.stamp != W.seq) continue;  // an allocation that did not fit left a hole (its tile is on the redo list)
            const unsigned long long mask = ((unsigned long long)rec.mask_hi << 32) | rec.mask_lo;
            const bool act = ((mask >> lane) & 1ull) != 0ull;
            const float* r = rays + (size_t)it * kWfRayWords + lane;
            V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 0.0f);
            if (act) {
                o = v3(r[0 * 64], r[1 * 64], r[2 * 64]);
                d = v3(r[3 * 64], r[4 * 64], r[5 * 64]);
            }
            const uint32_t pass = (rec.info >> 1) & 1u;
            const WfNode nd = wf_node<LARGE, DARK>(P, pass, act, o, d, pass == PASS_SPHERES ? 1.0f : P.ambient PROF_PASS);
            uint32_t fc, nc;
            wf_emit(P, W, level, it, pass, rec.entry, nd, fc, nc);
            uint32_t* out = W.nodes[level] + (size_t)it * kWfNodeWords + lane;
            out[0] = nd.sumf;
            out[64] = nd.mat;
            if (lane == 0u) {
                u32x2 cw = {fc, nc};
                *(u32x2*)&W.items[level][it].first_child = cw;
                W.items[level][it].parent = (uint32_t)(__builtin_readcyclecounter() - t_item);  // diagnostic (rt_debug_wavefront_items); nothing reads the parent link
            }
        }
    }
}

// The colour of item `it` of LEVEL per lane (valid for the lanes of its mask): its children's colours first
// (depth-first over the item tree, post-order), then node_finish - cast_ray's fold of render.rs:221-224 in the
// closed form described at node_finish, children in the reference's order (reflection, refraction).
template <uint32_t LEVEL>
DEV uint32_t wf_finish(const Params& P, const WfParams& W, uint32_t it, const WfItem& rec)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t* nd = W.nodes[LEVEL] + (size_t)it * kWfNodeWords + lane;
    const uint32_t sumf = nd[0], mat = nd[64];
    const MatRec* mp = P.mats + mat;
    const float k_r = mp->reflectivity, k_t = mp->transparency;
    const bool hit_any = (sumf & kHitFlag) != 0u;
    const bool has_refl = hit_any && (k_r > 0.0f) && (P.n_lights != 0u);
    const bool has_refr = hit_any && (k_t > 0.0f) && (P.n_lights != 0u);
    uint32_t c_refl = P.sky, c_refr = P.sky;  // children beyond max_depth
    if constexpr (LEVEL < RT_MAX_DEPTH_C) {
        if (LEVEL < P.max_depth) {
            c_refl = 0u;
            c_refr = 0u;
            for (uint32_t j = 0; j < rec.n_children; ++j) {
                const uint32_t ch = rec.first_child + j;
                const WfItem crec = wf_load_item(W.items[LEVEL + 1u], ch);
                const uint32_t c = wf_finish<LEVEL + 1u>(P, W, ch, crec);
                const unsigned long long cm = ((unsigned long long)crec.mask_hi << 32) | crec.mask_lo;
                const bool in = ((cm >> lane) & 1ull) != 0ull;
                if (crec.info & 1u) c_refr = in ? c : c_refr;
                else c_refl = in ? c : c_refl;
            }
        }
    }
    const uint32_t zero = 0u;
    return node_finish(P, sumf, k_r, k_t, has_refl ? c_refl : zero, has_refr ? c_refr : zero);
}

// One wave per tile whose roots wait for children: both passes' trees, the merge of render.rs:308-312, the pixel.
__global__ __launch_bounds__(64) void wf_combine_kernel(const Params P, const WfParams W)
{
    const uint32_t lane = threadIdx.x & 63u;
    KCONST uint32_t* hdr = (KCONST uint32_t*)W.header;
    const uint32_t sq = blockIdx.x & (kWfQueues - 1u);
    const uint32_t n = hdr[(kWfPendingCounter + sq) * 16u];
    for (uint32_t i = blockIdx.x / kWfQueues; i < n; i += gridDim.x / kWfQueues) {
        const uint32_t item = ((KCONST uint32_t*)W.pending)[sq * W.pending_cap + i];
        uint32_t px = 0u, entry = 0u;
        for (uint32_t pass = PASS_SPHERES; pass <= PASS_CUBES; ++pass) {
            uint32_t c = P.sky;
            if (pass == PASS_CUBES || P.n_spheres != 0u) {  // without spheres level 0 writes no sphere roots: that pass is the sky colour
                const uint32_t ri = 2u * item + pass;
                const WfItem rec = wf_load_item(W.items[0], ri);
                entry = rec.entry;
                c = wf_finish<0u>(P, W, ri, rec);
            }
            if (pass == PASS_SPHERES) px = c;
            else px = (px == P.sky) ? c : px;
        }
        const uint32_t bx = entry & kEntryColMask, by = entry >> 16;
        const uint32_t x = bx * 8u + (lane & 7u);
        const uint32_t y = tile_y0(P, by) + (lane >> 3);
        const uint32_t out_row = by * 8u + (lane >> 3);
        if ((size_t)y * P.width + x == (size_t)P.width) px = 0u;  // render.rs:316-322: pixel i == width is never painted
        if ((x < P.width) && (y < P.y1) && P.out) P.out[(size_t)out_row * P.width + x] = px;
    }
}

// ---------------------------------------------------------------- classify_tiles
// Which 8x8 pixel tiles can contain a primary ray that meets a cube?  A pixel's ray is the line
// eye + t (sx right + sy up + fwd) with sx = (2x/W - 1) W/H, sy = 1 - 2y/H (render.rs:245-262,
// camera.rs:38-40).  The reference's slab test has no t < 0 rejection (cube.rs:90-145): it is a LINE
// test, so the projection argument needs the whole (padded) cube in front of the eye; a cube with a
// corner beside or behind it sets the trace-everything flag instead.  In exact arithmetic the line
// meets the cube only if (sx, sy) lies in the convex hull of the projected corners, hence inside
// their bounding rectangle.  The reference's f32 arithmetic can move a silhouette by a few ulp of
// the coordinates involved; every bound is padded by 64 ulp of the largest coordinate (host: pad)
// and the rectangle by two pixels.  f64 throughout; one wave per cube, lanes 0..7 = the corners.
__global__ __launch_bounds__(64) void classify_tiles(const TileParams T)
{
    const uint32_t cube = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (cube == 0u) {
        // The other header bank still holds the previous launch's counts: from them, the lowest cost class
        // whose tiles and all heavier ones fit the side kernel (kHeavyTarget workgroups at once), for this
        // launch's build_queue (header word kHeavyClassWord).  The tiles that were on the previous launch's heavy
        // list are not in its class counts: they count as heavier than everything.
        // (One load for the whole bank, lane k its word k; the sums run over readlanes.  Launches that keep no
        // heavy list skip this: heavy_on == 0.)
        if (T.heavy_on != 0u) {
            const uint32_t word = lane < 16u ? T.queue[(T.qsel ^ 1u) * 16u + lane] : 0u;
            uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)kHeavyCountWord);
            acc = acc < kHeavyCap ? acc : kHeavyCap;
            uint32_t cls = kCostClasses;
#pragma unroll
            for (uint32_t k = kCostClasses; k-- > 0u;) {
                acc += (uint32_t)__builtin_amdgcn_readlane((int)word, (int)k);
                if (acc <= kHeavyTarget && cls == k + 1u) cls = k;
            }
            if (lane == 0u) T.queue[kHeavyClassWord] = cls;
        }
        // The side kernel of this launch may start: everything enqueued on the context's stream before this
        // launch has completed (the previous launch's build_queue, which wrote the list, and whatever the
        // caller did with the output buffer).  A device-coherent store: the waiting waves run on any XCD.
        if (lane == 0u && T.heavy_go != 0u) __hip_atomic_store(&T.queue[kHeavyGoWord], T.heavy_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 16u) T.queue[(T.qsel ^ 1u) * 16u + lane] = 0u;  // the queue header bank of the NEXT launch
        if (T.wf_header) {  // a wavefront launch: its item counters, pending count and redo list start at zero
            for (uint32_t c = lane; c < kWfPendingCounter + kWfQueues; c += 64u) T.wf_header[c * 16u] = 0u;  // every sub-queue's counter
            if (lane < 16u) T.wf_header[kWfRedoBase + lane] = 0u;
        }
    }
    if (cube >= T.n_cubes + T.n_spheres) return;
    // blocks behind the cubes: the spheres, as the bounding box of the sphere grown by sphere_grow / r (sphere_query:
    // how far a line can miss and still be reported by the reference's f32 discriminant); r -> 0 gives an infinite
    // box, which cannot be projected: everything is traced
    const bool is_sphere = cube >= T.n_cubes;
    const float* rec = is_sphere ? T.sphere_filt + (size_t)(cube - T.n_cubes) * 4u : T.cube_filt + (size_t)cube * 4u;
    const double h0 = __builtin_fabs((double)rec[3]);
    const double h = (is_sphere ? h0 + T.sphere_grow / h0 : h0) + T.pad;
    uint32_t* plane = is_sphere ? T.sphere_stamps : T.stamps + kStampBase;  // fine words, then the coarse level
    double q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = (double)rec[k] + (((lane >> k) & 1u) ? h : -h) - T.eye[k];
    const double* r = T.right; const double* u = T.up; const double* f = T.fwd;
    // q = a right + b up + c fwd (Cramer)
    const double a = (q[0] * (u[1] * f[2] - u[2] * f[1]) - q[1] * (u[0] * f[2] - u[2] * f[0]) + q[2] * (u[0] * f[1] - u[1] * f[0])) * T.inv_det;
    const double b = (r[0] * (q[1] * f[2] - q[2] * f[1]) - r[1] * (q[0] * f[2] - q[2] * f[0]) + r[2] * (q[0] * f[1] - q[1] * f[0])) * T.inv_det;
    const double c = (r[0] * (u[1] * q[2] - u[2] * q[1]) - r[1] * (u[0] * q[2] - u[2] * q[0]) + r[2] * (u[0] * q[1] - u[1] * q[0])) * T.inv_det;
    const double len = __builtin_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double px = ((a / c) / T.aspect + 1.0) * (double)T.width * 0.5;
    double py = (1.0 - b / c) * (double)T.height * 0.5;
    // negated compares: NaN anywhere counts as "cannot project"
    bool bad = !(c > 1e-3 * len) || !(c > 16.0 * T.pad) || !(__builtin_fabs(px) < 1e9) || !(__builtin_fabs(py) < 1e9);
    double x0 = px, x1 = px, y0 = py, y1 = py;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        x0 = __builtin_fmin(x0, __shfl_xor(x0, m, 64)); x1 = __builtin_fmax(x1, __shfl_xor(x1, m, 64));
        y0 = __builtin_fmin(y0, __shfl_xor(y0, m, 64)); y1 = __builtin_fmax(y1, __shfl_xor(y1, m, 64));
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
        if (lane == 0u) T.stamps[0] = T.frame_id;
        return;
    }
    // pixel rectangle, padded, clipped to the frame; empty -> nothing to stamp
    const double fx0 = __builtin_floor(x0) - 2.0, fx1 = __builtin_ceil(x1) + 2.0;
    const double fy0 = __builtin_floor(y0) - 2.0, fy1 = __builtin_ceil(y1) + 2.0;
    if (fx1 < 0.0 || fy1 < 0.0 || fx0 > (double)T.width - 1.0 || fy0 > (double)T.height - 1.0) return;
    const uint32_t px0 = fx0 < 0.0 ? 0u : (uint32_t)fx0, py0 = fy0 < 0.0 ? 0u : (uint32_t)fy0;
    const uint32_t px1 = fx1 > (double)T.width - 1.0 ? T.width - 1u : (uint32_t)fx1;
    const uint32_t py1 = fy1 > (double)T.height - 1.0 ? T.height - 1u : (uint32_t)fy1;
    const uint32_t tx0 = px0 >> 3, tx1 = px1 >> 3, ty0 = py0 >> 3, ty1 = py1 >> 3;  // < tiles_x, < tiles_y
    const uint32_t nx = tx1 - tx0 + 1u, n = nx * (ty1 - ty0 + 1u);
    if (n <= 1024u) {
        for (uint32_t i = lane; i < n; i += 64u) {
            const uint32_t tx = tx0 + i % nx, ty = ty0 + i / nx;
            plane[(size_t)ty * T.tiles_x + tx] = T.frame_id;
        }
    } else {
        // a large projection (a cube near the eye): stamp blocks of 8x8 tiles instead, 64x fewer stores
        uint32_t* cs = plane + (size_t)T.tiles_x * T.tiles_y;
        const uint32_t ctx = (T.tiles_x + 7u) >> 3;
        const uint32_t cx0 = tx0 >> 3, cy0 = ty0 >> 3, cnx = (tx1 >> 3) - cx0 + 1u, cn = cnx * ((ty1 >> 3) - cy0 + 1u);
        for (uint32_t i = lane; i < cn; i += 64u) cs[(size_t)(cy0 + i / cnx) * ctx + cx0 + i % cnx] = T.frame_id;
    }
}

// ---------------------------------------------------------------- rt_selftest_division
// Device check of the claim under "correctly rounded a / d": for pseudo-random operands that cover the
// whole in-range exponent span, special significands and zero numerators, div_shared == the compiled
// `/`, bit for bit.  out[0] += mismatches, out[1] += cases.
__global__ __launch_bounds__(256) void selftest_division(uint32_t seed, uint32_t iters, unsigned long long* out)
{
    uint32_t s = seed ^ ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ 0x9E3779B9u;
    unsigned long long bad = 0ull, cases = 0ull;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; w[k] = s; }
        // significands: random, or (1 in 8) one of the hard shapes: all ones, 1.0, 1.0 + ulp, all ones - 1
        auto mant = [](uint32_t r) -> uint32_t {
            const uint32_t m = r & 0x7FFFFFu;
            switch ((r >> 23) & 31u) {
            case 0: return 0x7FFFFFu;
            case 1: return 0u;
            case 2: return 1u;
            case 3: return 0x7FFFFEu;
            default: return m;
            }
        };
        const uint32_t ed = 127u - 60u + (w[2] % 121u), ea = 127u - 60u + ((w[2] >> 8) % 121u);  // exponents -60..60
        const float d = __builtin_bit_cast(float, ((w[2] >> 31) << 31) | (ed << 23) | mant(w[0]));
        float a = __builtin_bit_cast(float, ((w[3] >> 31) << 31) | (ea << 23) | mant(w[1]));
        if (((w[3] >> 8) & 63u) == 0u) a = (w[3] & 1u) ? 0.0f : -0.0f;
        if (!(div_in_range(d) && num_in_range(a))) continue;  // 2^60 with a non-zero significand is out of range
        const float fast = div_shared(a, d, rcp_refined(d));
        const float ref = a / d;
        ++cases;
        if (__builtin_bit_cast(uint32_t, fast) != __builtin_bit_cast(uint32_t, ref)) ++bad;
    }
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off, 64); cases += __shfl_down(cases, off, 64); }
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&out[0], bad); atomicAdd(&out[1], cases); }
}

}  // namespace

int launch_selftest_division(uint32_t seed, uint32_t blocks, uint32_t iters, unsigned long long* d_out, void* stream)
{
    hipLaunchKernelGGL(selftest_division, dim3(blocks, 1u, 1u), dim3(256u, 1u, 1u), 0, (hipStream_t)stream, seed, iters, d_out);
    return (int)hipGetLastError();
}

int launch_classify(const TileParams& t, void* stream)
{
    // at least one block: block 0 also zeroes the work queue header
    const uint32_t objects = t.n_cubes + t.n_spheres;
    hipLaunchKernelGGL(classify_tiles, dim3(objects ? objects : 1u, 1u, 1u), dim3(64u, 1u, 1u), 0, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

int launch_queue(const Params& p, int variant, void* stream)
{
    const uint32_t tiles = p.grid_tx * p.grid_ty;
    if (tiles == 0u) return (int)hipSuccess;
    dim3 grid((tiles + 1023u) / 1024u, 1u, 1u), block(1024u, 1u, 1u);
    hipStream_t s = (hipStream_t)stream;
    (void)variant;
    if (p.heavy_now != 0u || p.heavy_next != 0u) hipLaunchKernelGGL(build_queue<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(build_queue<false>, grid, block, 0, s, p);
    return (int)hipGetLastError();
}

int launch_render(const Params& p, int variant, void* stream, uint32_t max_workgroups)
{
    const uint32_t tiles = p.grid_tx * p.grid_ty;
    if (tiles == 0u) return (int)hipSuccess;
    // single-wave workgroups striding over the items (traced tiles, then paint groups)
    const uint32_t items = tiles + (tiles + kPaintGroup - 1u) / kPaintGroup;
    dim3 grid(items < max_workgroups ? items : max_workgroups, 1u, 1u), block(64u, 1u, 1u);
    hipStream_t s = (hipStream_t)stream;
    const bool large = p.n_chunks > 0u;  // long cube list: Morton-ordered stream, chunk boxes, BVH
    const uint32_t lds = large ? p.bvh_levels * 256u : 0u;  // the walk's per-lane stack
    switch (variant) {
    case kVariantDebug:
        if (large) hipLaunchKernelGGL((render_kernel<true, false, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((render_kernel<true, false, false>), grid, block, lds, s, p);
        break;
    case kVariantCount:
        if (large) hipLaunchKernelGGL((render_kernel<false, true, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((render_kernel<false, true, false>), grid, block, lds, s, p);
        break;
    default:
        if (p.dark_materials != 0u) {
            if (large) hipLaunchKernelGGL((render_kernel<false, false, 1, true>), grid, block, lds, s, p);
            else if (p.refine == 2u) hipLaunchKernelGGL((render_kernel<false, false, 3, true>), grid, block, lds, s, p);
            else if (p.refine != 0u) hipLaunchKernelGGL((render_kernel<false, false, 2, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((render_kernel<false, false, 0, true>), grid, block, lds, s, p);
        } else {
            if (large) hipLaunchKernelGGL((render_kernel<false, false, 1>), grid, block, lds, s, p);
            else if (p.refine == 2u) hipLaunchKernelGGL((render_kernel<false, false, 3>), grid, block, lds, s, p);
            else if (p.refine != 0u) hipLaunchKernelGGL((render_kernel<false, false, 2>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((render_kernel<false, false, 0>), grid, block, lds, s, p);
        }
        break;
    }
    return (int)hipGetLastError();
}

int launch_wf_trace(const Params& p, const WfParams& w, void* stream, uint32_t max_workgroups)
{
    const uint32_t tiles = p.grid_tx * p.grid_ty;
    if (tiles == 0u) return (int)hipSuccess;
    const bool primary = w.level == 0u;
    // deeper levels: the item count is only known on the device; a grid of a few thousand workgroups (a multiple of
    // kWfQueues) strides over the sub-queues
    const uint32_t units = p.n_spheres != 0u ? 2u * tiles : tiles;  // level 0: a work unit per tile and traced pass
    uint32_t g = primary ? (units < max_workgroups ? units : max_workgroups) : (max_workgroups < RTX_WF_GRID ? max_workgroups : RTX_WF_GRID);
    if (!primary) g = g < kWfQueues ? kWfQueues : (g / kWfQueues) * kWfQueues;
    dim3 grid(g, 1u, 1u), block(64u, 1u, 1u);
    hipStream_t s = (hipStream_t)stream;
    const bool large = p.n_chunks > 0u, dark = p.dark_materials != 0u;
    const uint32_t lds = large ? p.bvh_levels * 256u : 0u;
#define RTX_WF_LAUNCH(PRIM, LG, DK) hipLaunchKernelGGL((wf_trace_kernel<PRIM, LG, DK>), grid, block, lds, s, p, w)
    if (primary) {
        if (large) { if (dark) RTX_WF_LAUNCH(true, 1, true); else RTX_WF_LAUNCH(true, 1, false); }
        else { if (dark) RTX_WF_LAUNCH(true, 2, true); else RTX_WF_LAUNCH(true, 2, false); }  // short lists: with the cull refinement
    } else {
        if (large) { if (dark) RTX_WF_LAUNCH(false, 1, true); else RTX_WF_LAUNCH(false, 1, false); }
        else { if (dark) RTX_WF_LAUNCH(false, 2, true); else RTX_WF_LAUNCH(false, 2, false); }
    }
#undef RTX_WF_LAUNCH
    return (int)hipGetLastError();
}

int launch_wf_combine(const Params& p, const WfParams& w, void* stream, uint32_t max_workgroups)
{
    const uint32_t tiles = p.grid_tx * p.grid_ty;
    if (tiles == 0u) return (int)hipSuccess;
    uint32_t g = max_workgroups < 8192u ? max_workgroups : 8192u;
    g = g < kWfQueues ? kWfQueues : (g / kWfQueues) * kWfQueues;
    dim3 grid(g, 1u, 1u), block(64u, 1u, 1u);
    hipLaunchKernelGGL(wf_combine_kernel, grid, block, 0, (hipStream_t)stream, p, w);
    return (int)hipGetLastError();
}

int launch_heavy(const Params& p, void* stream, bool gate)
{
    if (p.grid_tx * p.grid_ty == 0u || !p.heavy) return (int)hipSuccess;
    dim3 grid(kHeavyGrid, 1u, 1u), block(64u * kWgWaves, 1u, 1u);
    hipStream_t s = (hipStream_t)stream;
    const bool large = p.n_chunks > 0u;
    const uint32_t lds = large ? kWgWaves * p.bvh_levels * 256u : 0u;
    if (gate) hipLaunchKernelGGL(heavy_gate, dim3(1u, 1u, 1u), dim3(64u, 1u, 1u), 0, s, p);
    if (p.dark_materials != 0u) {
        if (large) hipLaunchKernelGGL((render_heavy_kernel<true, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((render_heavy_kernel<false, true>), grid, block, lds, s, p);
    } else {
        if (large) hipLaunchKernelGGL((render_heavy_kernel<true, false>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((render_heavy_kernel<false, false>), grid, block, lds, s, p);
    }
    return (int)hipGetLastError();
}

int launch_heavy_join(const Params& p, void* stream)
{
    hipLaunchKernelGGL(heavy_join, dim3(1u, 1u, 1u), dim3(64u, 1u, 1u), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

}  // namespace rtx950
