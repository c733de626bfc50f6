/*
 * rt_oracle.c — CPU restatement of the reference's per-pixel ray-cast path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (raytracing_amd/csrc, include/rtx950.h) never routes through it.
 *
 * What it restates (reference = ElrohirGT/Raytracing @ 2024-11-15, paths relative to its root):
 *   src/render.rs (all), src/cube.rs:51-66,77-218, src/sphere.rs:25-56, src/lib.rs:29-35,
 *   src/texture.rs:171-196, src/color.rs:39-149, src/camera.rs:33-41,61-86,
 *   src/framebuffer.rs:44-109, src/bmp.rs:19-113.
 * Third-party arithmetic the reference calls but does not vendor: nalgebra-glm 0.19.0 ->
 * nalgebra 0.33.0 -> simba 0.9.0 (Cargo.lock).  Restated from their published algorithms:
 * 3-vector dot = (a0*b0 + a1*b1) + a2*b2, norm = sqrt(dot(v,v)), normalize = per-component
 * division by the norm, cross = the textbook formula, no FMA anywhere.
 *
 * PINNING STATUS: the reference cannot be built here (no cargo/rustc, dependencies not vendored)
 * and its own tests cover only `Color * f32` (src/color.rs:155-177).  Those two known-answer
 * tests pin col_mul below (tests/test_oracle_kat.py).  Every other function is "parity
 * unpinned": a from-spec restatement checked only by hand-computed cases and by an independent
 * numpy restatement (tests/np_restatement.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).  No -march
 * flag: baseline x86-64 has no FMA, so a*b+c is never fused, like rustc's output.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../include/rtx950.h"

/* ------------------------------------------------------------------ scalar helpers (Rust semantics) */

/* f32::clamp (core): NaN passes through. */
static inline float rs_clamp(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
/* f32::max / f32::min: if one operand is NaN the other is returned. */
static inline float rs_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline float rs_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
/* `x as u8` / `x as u32`: saturating, truncating toward zero, NaN -> 0. */
static inline uint8_t rs_as_u8(float x)
{
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
static inline uint32_t rs_as_u32(float x)
{
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

/* reference src/lib.rs:29-35 */
static inline void minmax(float a, float b, float* mn, float* mx)
{
    if (a < b) { *mn = a; *mx = b; } else { *mn = b; *mx = a; }
}

/* ------------------------------------------------------------------ Color (reference src/color.rs) */

typedef struct { uint8_t r, g, b; } color;

static inline color col_from_hex(uint32_t v) /* color.rs:49-55 */
{
    color c = { (uint8_t)((v >> 16) & 0xFF), (uint8_t)((v >> 8) & 0xFF), (uint8_t)(v & 0xFF) };
    return c;
}
static inline uint32_t col_to_hex(color c) /* color.rs:76-79 */
{
    return ((uint32_t)c.r << 16) | ((uint32_t)c.g << 8) | (uint32_t)c.b;
}
static inline uint8_t sat_add_u8(uint8_t a, uint8_t b) { unsigned s = (unsigned)a + b; return s > 255 ? 255 : (uint8_t)s; }
static inline uint8_t sat_sub_u8(uint8_t a, uint8_t b) { return a > b ? (uint8_t)(a - b) : 0; }
static inline color col_add(color a, color b) /* color.rs:99-116 */
{
    color c = { sat_add_u8(a.r, b.r), sat_add_u8(a.g, b.g), sat_add_u8(a.b, b.b) };
    return c;
}
static inline color col_sub(color a, color b) /* color.rs:132-149 */
{
    color c = { sat_sub_u8(a.r, b.r), sat_sub_u8(a.g, b.g), sat_sub_u8(a.b, b.b) };
    return c;
}
static inline color col_mul(color a, float f) /* color.rs:118-130 */
{
    color c = { rs_as_u8(rs_clamp((float)a.r * f, 0.0f, 255.0f)),
                rs_as_u8(rs_clamp((float)a.g * f, 0.0f, 255.0f)),
                rs_as_u8(rs_clamp((float)a.b * f, 0.0f, 255.0f)) };
    return c;
}
static inline int col_eq(color a, color b) { return col_to_hex(a) == col_to_hex(b); } /* color.rs:39-44 */

/* ------------------------------------------------------------------ Vec3 (nalgebra-glm semantics) */

typedef struct { float x, y, z; } vec3;

static inline vec3 v3(float x, float y, float z) { vec3 v = { x, y, z }; return v; }
static inline vec3 v_from(rt_vec3 a) { return v3(a.x, a.y, a.z); }
static inline vec3 v_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v_neg(vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline vec3 v_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float v_dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 v_cross(vec3 a, vec3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline vec3 v_normalize(vec3 a)
{
    float n = sqrtf(v_dot(a, a));
    return v3(a.x / n, a.y / n, a.z / n);
}

/* ------------------------------------------------------------------ scene */

enum { FACE_NONE = 0, FACE_TOP, FACE_BOTTOM, FACE_FORWARDS, FACE_BACKWARDS, FACE_LEFT, FACE_RIGHT }; /* texture.rs:12-20 */

typedef struct {
    vec3 bmin, bmax; /* cube.rs:51-66 */
    float size;
    uint32_t id;
    uint32_t material;
} ocube;

typedef struct {
    vec3 center;
    float radius;
    uint32_t id;
    uint32_t material;
} osphere;

typedef struct orc_scene {
    ocube* cubes;
    size_t n_cubes;
    osphere* spheres;
    size_t n_spheres;
    rt_material* mats;
    size_t n_mats;
    uint8_t* atlas; /* n_tex * tex_w * tex_h * 3, texture.rs:151-161 layout */
    size_t n_tex;
    uint32_t tex_w, tex_h, sprite;
} orc_scene;

typedef struct { /* raytracer.rs:8-16 */
    float distance;
    vec3 point;
    vec3 normal;
    uint32_t material;
    float u, v;
    int face;
} isect;

/* reference src/cube.rs:77-218 */
static int cube_intersect(const ocube* c, vec3 o, vec3 d, isect* out)
{
    float t0x = (c->bmin.x - o.x) / d.x;
    float t1x = (c->bmax.x - o.x) / d.x;
    float tmin, tmax;
    minmax(t0x, t1x, &tmin, &tmax);

    float t0y = (c->bmin.y - o.y) / d.y;
    float t1y = (c->bmax.y - o.y) / d.y;
    float tymin, tymax;
    minmax(t0y, t1y, &tymin, &tymax);

    if (tmin > tymax || tymin > tmax) return 0;
    tmin = rs_max(tmin, tymin);
    tmax = rs_min(tmax, tymax);

    float t0z = (c->bmin.z - o.z) / d.z;
    float t1z = (c->bmax.z - o.z) / d.z;
    float tzmin, tzmax;
    minmax(t0z, t1z, &tzmin, &tzmax);

    if (tmin > tzmax || tzmin > tmax) return 0;
    tmin = rs_max(tmin, tzmin);
    tmax = rs_min(tmax, tzmax);

    float distance = (tmin < 0.0f) ? tmax : tmin; /* :145 — may be negative, still Some */
    vec3 p = v_add(o, v_scale(d, distance));      /* :146 */

    vec3 n = v3(0, 0, 0);
    int face = FACE_NONE;
    float u = 0.0f, v = 0.0f;
    const float limit = 1e-3f;
    if (fabsf(p.x - c->bmin.x) < limit) {
        n = v3(-1, 0, 0); face = FACE_LEFT;
        u = (p.y - c->bmin.y) / c->size; v = (p.z - c->bmin.z) / c->size;
    } else if (fabsf(p.x - c->bmax.x) < limit) {
        n = v3(1, 0, 0); face = FACE_RIGHT;
        u = 1.0f - (p.y - c->bmin.y) / c->size; v = (p.z - c->bmin.z) / c->size;
    } else if (fabsf(p.y - c->bmin.y) < limit) {
        n = v3(0, -1, 0); face = FACE_BOTTOM;
        u = (p.x - c->bmin.x) / c->size; v = (p.z - c->bmin.z) / c->size;
    } else if (fabsf(p.y - c->bmax.y) < limit) {
        n = v3(0, 1, 0); face = FACE_TOP;
        u = (p.x - c->bmin.x) / c->size; v = (p.z - c->bmin.z) / c->size;
    } else if (fabsf(p.z - c->bmin.z) < limit) {
        n = v3(0, 0, -1); face = FACE_BACKWARDS;
        u = (p.x - c->bmin.x) / c->size; v = 1.0f - (p.y - c->bmin.y) / c->size;
    } else if (fabsf(p.z - c->bmax.z) < limit) {
        n = v3(0, 0, 1); face = FACE_FORWARDS;
        u = (p.x - c->bmin.x) / c->size; v = (p.y - c->bmin.y) / c->size;
    }
    out->distance = distance;
    out->point = p;
    out->normal = n;
    out->material = c->material;
    out->u = u;
    out->v = v;
    out->face = face;
    return 1;
}

/* reference src/sphere.rs:25-56 */
static int sphere_intersect(const osphere* s, vec3 o, vec3 d, isect* out)
{
    vec3 oc = v_sub(o, s->center);
    float a = v_dot(d, d);
    float b = 2.0f * v_dot(oc, d);
    float c = v_dot(oc, oc) - s->radius * s->radius;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float t = (-b - sqrtf(disc)) / (2.0f * a);
        if (t > 0.0f) {
            vec3 p = v_add(o, v_scale(d, t));
            out->distance = t;
            out->point = p;
            out->normal = v_normalize(v_sub(p, s->center));
            out->material = s->material;
            out->u = 0.0f;
            out->v = 0.0f;
            out->face = FACE_NONE;
            return 1;
        }
    }
    return 0;
}

/* reference src/texture.rs:171-196 */
static color texture_face_color(const orc_scene* S, int tex, int face, float x, float y)
{
    float sp = (float)S->sprite;
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
    uint32_t xi = rs_as_u32(rs_clamp(px, ox, ox + sp - 1.0f));
    uint32_t yi = rs_as_u32(rs_clamp(py, oy, oy + sp - 1.0f));
    size_t idx = (size_t)yi * S->tex_w + xi; /* :171-174 */
    const uint8_t* t = S->atlas + (size_t)tex * S->tex_w * S->tex_h * 3 + idx * 3;
    color c = { t[0], t[1], t[2] };
    return c;
}

/* reference src/render.rs:17-19 */
static inline vec3 reflect(vec3 i, vec3 n) { return v_sub(i, v_scale(n, 2.0f * v_dot(i, n))); }

/* reference src/render.rs:21-42 */
static vec3 refract(vec3 i, vec3 n, float eta_t)
{
    float cosi = -rs_clamp(v_dot(i, n), -1.0f, 1.0f);
    float n_cosi = cosi, eta = eta_t;
    vec3 nn = n;
    if (cosi < 0.0f) {
        n_cosi = -cosi;
        eta = 1.0f / eta_t;
        nn = v_neg(n);
    }
    float k = 1.0f - eta * eta * (1.0f - n_cosi * n_cosi);
    if (k < 0.0f) return reflect(i, nn);
    return v_add(v_scale(i, eta), v_scale(nn, eta * n_cosi - sqrtf(k)));
}

/* ------------------------------------------------------------------ render context */

#define HASH_BASIS 2166136261u
#define HASH_PRIME 16777619u
#define HASH_CUT 0x9E3779B9u /* child beyond max_depth: returns sky without traversal */
static inline uint32_t hmix(uint32_t h, uint32_t v) { return (h ^ v) * HASH_PRIME; }

typedef struct {
    const orc_scene* S;
    const rt_light* lights;
    uint32_t n_lights;
    color sky;
    uint32_t max_depth;
    int reuse_secondary; /* 0: literal reference (re-cast per light); 1: trace each secondary once per hit */
} rctx;

enum { PASS_SPHERES = 0, PASS_CUBES = 1 };

static inline size_t n_objects(const rctx* R, int pass) { return pass == PASS_CUBES ? R->S->n_cubes : R->S->n_spheres; }
static inline uint32_t object_id(const rctx* R, int pass, size_t i)
{
    return pass == PASS_CUBES ? R->S->cubes[i].id : R->S->spheres[i].id;
}
static inline int object_intersect(const rctx* R, int pass, size_t i, vec3 o, vec3 d, isect* out, rt_counters* C)
{
    if (pass == PASS_CUBES) { C->cube_tests++; return cube_intersect(&R->S->cubes[i], o, d, out); }
    C->sphere_tests++;
    return sphere_intersect(&R->S->spheres[i], o, d, out);
}

/* reference src/render.rs:44-70; the object list is `objects.filter(|o| o != impact_object)`
 * (render.rs:113) where equality is by id (cube.rs:69-73, sphere.rs:17-21). */
static float cast_shadow(const rctx* R, int pass, const isect* hit, const rt_light* light, uint32_t impact_id,
                         uint32_t* blocker, rt_counters* C)
{
    vec3 lp = v_from(light->position);
    vec3 light_dir = v_normalize(v_sub(lp, hit->point));
    vec3 origin = hit->point;
    float shadow = 0.0f;
    *blocker = RT_NO_HIT;
    C->shadow_rays++;
    size_t n = n_objects(R, pass);
    for (size_t i = 0; i < n; ++i) {
        if (object_id(R, pass, i) == impact_id) continue;
        isect si;
        if (object_intersect(R, pass, i, origin, light_dir, &si, C)) {
            *blocker = (uint32_t)i;
            if (si.distance < 0.0f) break;
            vec3 dv = v_sub(si.point, lp); /* distance2(a,b) = norm_squared(b - a) */
            float d2 = v_dot(dv, dv);
            shadow = si.distance / d2 * (1.0f - R->S->mats[si.material].transparency);
            break;
        }
    }
    return shadow;
}

/* reference src/render.rs:73-229.  Returns the colour; *out_hash gets the node's tree hash. */
static color cast_ray(const rctx* R, vec3 o, vec3 d, int pass, float ambient, uint32_t depth, uint32_t* out_hash,
                      uint32_t* out_hit, rt_counters* C)
{
    if (depth > R->max_depth) { /* :83-85 */
        *out_hash = HASH_CUT;
        if (out_hit) *out_hit = RT_NO_HIT;
        return R->sky;
    }
    /* :87-101 — fold with strict `<`, accumulator starts at +INF */
    C->nearest_rays++;
    size_t n = n_objects(R, pass);
    isect best;
    memset(&best, 0, sizeof(best));
    float best_d = INFINITY;
    uint32_t best_i = RT_NO_HIT;
    for (size_t i = 0; i < n; ++i) {
        isect cur;
        if (object_intersect(R, pass, i, o, d, &cur, C)) {
            if (cur.distance < best_d) {
                best_d = cur.distance;
                best = cur;
                best_i = (uint32_t)i;
            }
        }
    }
    if (out_hit) *out_hit = best_i;
    uint32_t h = hmix(HASH_BASIS, best_i);
    if (best_i == RT_NO_HIT) { /* :226-228 */
        *out_hash = h;
        return R->sky;
    }
    C->shaded_hits++;
    const rt_material* m = &R->S->mats[best.material];
    uint32_t impact_id = object_id(R, pass, best_i);

    color acc = { 0, 0, 0 };
    int have_refl = 0, have_refr = 0;
    color refl_cached = { 0, 0, 0 }, refr_cached = { 0, 0, 0 };
    uint32_t h_refl = 0, h_refr = 0;
    for (uint32_t li = 0; li < R->n_lights; ++li) { /* :104-225 */
        const rt_light* L = &R->lights[li];
        vec3 lp = v_from(L->position);
        vec3 light_dir = v_normalize(v_sub(lp, best.point));
        vec3 view_dir = v_normalize(v_sub(o, best.point));
        vec3 reflect_dir = v_normalize(reflect(v_neg(light_dir), best.normal));
        uint32_t blocker;
        float shadow = cast_shadow(R, pass, &best, L, impact_id, &blocker, C);
        h = hmix(h, blocker);
        float light_intensity = L->intensity * (1.0f - shadow);

        float diffuse_intensity = rs_clamp(v_dot(best.normal, light_dir), 0.0f, 1.0f) + ambient;
        color tx;
        if (m->texture != RT_TEXTURE_NONE) {
            float sp = (float)R->S->sprite;
            tx = texture_face_color(R->S, m->texture, best.face, best.u * sp, best.v * sp);
        } else {
            tx = col_from_hex(m->diffuse_0rgb);
        }
        color diffuse = col_mul(col_mul(col_mul(tx, m->albedo0), diffuse_intensity), light_intensity);

        float specular_intensity = powf(rs_clamp(v_dot(view_dir, reflect_dir), 0.0f, 1.0f), m->specular);
        color specular =
            col_mul(col_mul(col_mul(col_from_hex(L->color_0rgb), m->albedo1), specular_intensity), light_intensity);

        color reflect_color = { 0, 0, 0 };
        float reflectivity = m->reflectivity;
        if (reflectivity > 0.0f) {
            if (!R->reuse_secondary || !have_refl) {
                vec3 rdir = v_normalize(reflect(v_neg(d), best.normal));
                vec3 rorig = v_add(best.point, v_scale(best.normal, 1e-2f));
                refl_cached = cast_ray(R, rorig, rdir, pass, ambient, depth + 1, &h_refl, NULL, C);
                have_refl = 1;
            }
            reflect_color = refl_cached;
        }
        color refract_color = { 0, 0, 0 };
        float transparency = m->transparency;
        if (transparency > 0.0f) {
            if (!R->reuse_secondary || !have_refr) {
                vec3 rdir = refract(d, best.normal, m->refractive_index);
                vec3 rorig = v_add(best.point, v_scale(best.normal, 1e-3f));
                refr_cached = cast_ray(R, rorig, rdir, pass, ambient, depth + 1, &h_refr, NULL, C);
                have_refr = 1;
            }
            refract_color = refr_cached;
        }
        acc = col_add(col_add(col_add(acc, col_mul(col_add(diffuse, specular), 1.0f - reflectivity - transparency)),
                              col_mul(reflect_color, reflectivity)),
                      col_mul(refract_color, transparency));
    }
    h = hmix(h, h_refl);
    h = hmix(h, h_refr);
    *out_hash = h;
    return acc;
}

/* reference src/camera.rs:33-41 (recomputed per pixel there; a pure function of the camera) */
static vec3 change_basis(const rt_camera* cam, vec3 v)
{
    vec3 forward = v_normalize(v_sub(v_from(cam->center), v_from(cam->eye)));
    vec3 right = v_normalize(v_cross(forward, v_from(cam->up)));
    vec3 up = v_normalize(v_cross(right, forward));
    vec3 cb = v_sub(v_add(v_scale(right, v.x), v_scale(up, v.y)), v_scale(forward, v.z));
    return v_normalize(cb);
}

typedef struct {
    rctx R;
    const rt_frame* F;
    uint32_t row_step;
    uint32_t* out; /* band-relative, (y1-y0)*W; rows not sampled keep their previous content */
    rt_debug_planes dbg;
    volatile uint32_t* next_row;
    rt_counters counters;
} worker;

static void render_row(worker* w, uint32_t y)
{
    const rt_frame* F = w->F;
    const rctx* R = &w->R;
    float width = (float)F->width, height = (float)F->height;
    float aspect = width / height; /* render.rs:237-239 */
    vec3 eye = v_from(F->camera.eye);
    size_t row_off = (size_t)(y - F->y0) * F->width;
    for (uint32_t x = 0; x < F->width; ++x) {
        float sx = (2.0f * (float)x) / width - 1.0f;   /* :253 */
        float sy = -(2.0f * (float)y) / height + 1.0f; /* :254 */
        sx = sx * aspect;                              /* :257 */
        vec3 dir = v_normalize(v3(sx, sy, -1.0f));     /* :260 */
        dir = change_basis(&F->camera, dir);           /* :263 */
        uint32_t hs, hc, hit_s, hit_c;
        color cs = cast_ray(R, eye, dir, PASS_SPHERES, 1.0f, 0, &hs, &hit_s, &w->counters);           /* :264-273 */
        color cc = cast_ray(R, eye, dir, PASS_CUBES, F->ambient_light, 0, &hc, &hit_c, &w->counters); /* :294-303 */
        color px = col_eq(cs, R->sky) ? cc : cs;                                                       /* :308-312 */
        /* paint loop :314-323: i == width gives x = width -> XTooLarge -> pixel keeps clear colour 0 */
        size_t i = (size_t)y * F->width + x;
        uint32_t v = (i == F->width) ? 0u : col_to_hex(px);
        if (w->out) w->out[row_off + x] = v;
        if (w->dbg.hit_cube) w->dbg.hit_cube[row_off + x] = hit_c;
        if (w->dbg.hit_sphere) w->dbg.hit_sphere[row_off + x] = hit_s;
        if (w->dbg.tree_hash) w->dbg.tree_hash[row_off + x] = hmix(hmix(HASH_BASIS, hs), hc);
    }
}

static void* worker_main(void* arg)
{
    worker* w = (worker*)arg;
    const rt_frame* F = w->F;
    uint32_t nrows = (F->y1 - F->y0 + w->row_step - 1) / w->row_step;
    for (;;) {
        uint32_t k = __atomic_fetch_add(w->next_row, 1u, __ATOMIC_RELAXED);
        if (k >= nrows) break;
        render_row(w, F->y0 + k * w->row_step);
    }
    return NULL;
}

/* ------------------------------------------------------------------ exported API (ctypes) */

orc_scene* orc_scene_create(const rt_cube* cubes, size_t n_cubes, const rt_sphere* spheres, size_t n_spheres,
                            const rt_material* mats, size_t n_mats, const uint8_t* atlas_rgb8, size_t n_tex,
                            uint32_t tex_w, uint32_t tex_h, uint32_t sprite)
{
    orc_scene* S = (orc_scene*)calloc(1, sizeof(*S));
    if (!S) return NULL;
    S->n_cubes = n_cubes;
    S->cubes = (ocube*)calloc(n_cubes ? n_cubes : 1, sizeof(ocube));
    for (size_t i = 0; i < n_cubes; ++i) {
        float half = cubes[i].size / 2.0f; /* cube.rs:52 */
        vec3 c = v_from(cubes[i].center);
        S->cubes[i].bmin = v3(c.x - half, c.y - half, c.z - half);
        S->cubes[i].bmax = v3(c.x + half, c.y + half, c.z + half);
        S->cubes[i].size = cubes[i].size;
        S->cubes[i].id = cubes[i].id;
        S->cubes[i].material = cubes[i].material;
    }
    S->n_spheres = n_spheres;
    S->spheres = (osphere*)calloc(n_spheres ? n_spheres : 1, sizeof(osphere));
    for (size_t i = 0; i < n_spheres; ++i) {
        S->spheres[i].center = v_from(spheres[i].center);
        S->spheres[i].radius = spheres[i].radius;
        S->spheres[i].id = spheres[i].id;
        S->spheres[i].material = spheres[i].material;
    }
    S->n_mats = n_mats;
    S->mats = (rt_material*)calloc(n_mats ? n_mats : 1, sizeof(rt_material));
    memcpy(S->mats, mats, n_mats * sizeof(rt_material));
    S->n_tex = n_tex;
    S->tex_w = tex_w;
    S->tex_h = tex_h;
    S->sprite = sprite;
    size_t ab = n_tex * (size_t)tex_w * tex_h * 3;
    S->atlas = (uint8_t*)malloc(ab ? ab : 1);
    if (ab) memcpy(S->atlas, atlas_rgb8, ab);
    return S;
}

void orc_scene_destroy(orc_scene* S)
{
    if (!S) return;
    free(S->cubes);
    free(S->spheres);
    free(S->mats);
    free(S->atlas);
    free(S);
}

/* render(): rows y0, y0+row_step, ... < y1 of the frame.  row_step = 1 renders the whole band.
 * n_threads <= 0 -> all online cores (rayon's default pool size).  Returns threads used, <0 on error. */
int orc_render(const orc_scene* S, const rt_frame* F, uint32_t* out_band, const rt_debug_planes* dbg,
               int reuse_secondary, uint32_t row_step, int n_threads, rt_counters* counters_out)
{
    if (!S || !F || F->width == 0 || F->height == 0 || F->y1 > F->height || F->y0 >= F->y1) return -1;
    if (F->max_depth > RT_MAX_DEPTH || row_step == 0) return -1;
    for (size_t i = 0; i < S->n_cubes; ++i) if (S->cubes[i].material >= S->n_mats) return -1;
    for (size_t i = 0; i < S->n_spheres; ++i) if (S->spheres[i].material >= S->n_mats) return -1;
    if (n_threads <= 0) n_threads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    volatile uint32_t next = 0;
    worker* ws = (worker*)calloc((size_t)n_threads, sizeof(worker));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        ws[t].R.S = S;
        ws[t].R.lights = F->lights;
        ws[t].R.n_lights = F->n_lights;
        ws[t].R.sky = col_from_hex(F->sky_0rgb);
        ws[t].R.max_depth = F->max_depth;
        ws[t].R.reuse_secondary = reuse_secondary;
        ws[t].F = F;
        ws[t].row_step = row_step;
        ws[t].out = out_band;
        if (dbg) ws[t].dbg = *dbg;
        ws[t].next_row = &next;
    }
    for (int t = 1; t < n_threads; ++t) pthread_create(&th[t], NULL, worker_main, &ws[t]);
    worker_main(&ws[0]);
    for (int t = 1; t < n_threads; ++t) pthread_join(th[t], NULL);
    if (counters_out) {
        memset(counters_out, 0, sizeof(*counters_out)); /* only the first five fields are filled: in
            literal mode they are the reference's counts, in reuse mode the unique counts */
        for (int t = 0; t < n_threads; ++t) {
            counters_out->nearest_rays += ws[t].counters.nearest_rays;
            counters_out->shadow_rays += ws[t].counters.shadow_rays;
            counters_out->cube_tests += ws[t].counters.cube_tests;
            counters_out->sphere_tests += ws[t].counters.sphere_tests;
            counters_out->shaded_hits += ws[t].counters.shaded_hits;
        }
    }
    free(ws);
    free(th);
    return n_threads;
}

/* ---- known-answer-test entry points (thin wrappers so tests can poke single functions) ---- */

uint32_t orc_color_mul(uint32_t c, float f) { return col_to_hex(col_mul(col_from_hex(c), f)); }
uint32_t orc_color_add(uint32_t a, uint32_t b) { return col_to_hex(col_add(col_from_hex(a), col_from_hex(b))); }
uint32_t orc_color_sub(uint32_t a, uint32_t b) { return col_to_hex(col_sub(col_from_hex(a), col_from_hex(b))); }

/* out[0]=hit, [1]=distance, [2..4]=point, [5..7]=normal, [8]=u, [9]=v, [10]=face */
void orc_cube_intersect(const float center[3], float size, const float o[3], const float d[3], float out[11])
{
    ocube c;
    float half = size / 2.0f;
    c.bmin = v3(center[0] - half, center[1] - half, center[2] - half);
    c.bmax = v3(center[0] + half, center[1] + half, center[2] + half);
    c.size = size;
    c.id = 0;
    c.material = 0;
    isect it;
    memset(&it, 0, sizeof(it));
    int hit = cube_intersect(&c, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), &it);
    out[0] = (float)hit; out[1] = it.distance;
    out[2] = it.point.x; out[3] = it.point.y; out[4] = it.point.z;
    out[5] = it.normal.x; out[6] = it.normal.y; out[7] = it.normal.z;
    out[8] = it.u; out[9] = it.v; out[10] = (float)it.face;
}

void orc_sphere_intersect(const float center[3], float radius, const float o[3], const float d[3], float out[11])
{
    osphere s;
    s.center = v3(center[0], center[1], center[2]);
    s.radius = radius;
    s.id = 0;
    s.material = 0;
    isect it;
    memset(&it, 0, sizeof(it));
    int hit = sphere_intersect(&s, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), &it);
    out[0] = (float)hit; out[1] = it.distance;
    out[2] = it.point.x; out[3] = it.point.y; out[4] = it.point.z;
    out[5] = it.normal.x; out[6] = it.normal.y; out[7] = it.normal.z;
    out[8] = it.u; out[9] = it.v; out[10] = (float)it.face;
}

uint32_t orc_texel(const orc_scene* S, int tex, int face, float x, float y)
{
    return col_to_hex(texture_face_color(S, tex, face, x, y));
}

void orc_reflect(const float i[3], const float n[3], float out[3])
{
    vec3 r = reflect(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float i[3], const float n[3], float eta_t, float out[3])
{
    vec3 r = refract(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), eta_t);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_change_basis(const rt_camera* cam, const float v[3], float out[3])
{
    vec3 r = change_basis(cam, v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* reference src/camera.rs:61-86 */
void orc_rotate_cam(rt_camera* cam, float delta_yaw, float delta_pitch)
{
    const float PI = 3.14159265358979323846f;
    vec3 rv = v_sub(v_from(cam->eye), v_from(cam->center));
    float radius = sqrtf(v_dot(rv, rv));
    float current_yaw = atan2f(rv.z, rv.x);
    float radius_xz = sqrtf(rv.x * rv.x + rv.z * rv.z);
    float current_pitch = atan2f(-rv.y, radius_xz);
    float new_yaw = fmodf(current_yaw + delta_yaw, 2.0f * PI);
    float new_pitch = rs_clamp(current_pitch + delta_pitch, -PI / 2.0f + 0.1f, PI / 2.0f - 0.1f);
    cam->eye.x = cam->center.x + radius * cosf(new_yaw) * cosf(new_pitch);
    cam->eye.y = cam->center.y + -radius * sinf(new_pitch);
    cam->eye.z = cam->center.z + radius * sinf(new_yaw) * cosf(new_pitch);
}

/* reference src/bmp.rs:19-113.  Writes the file image into `out` (caller sizes it with
 * orc_bmp_size) and returns the number of bytes. */
static size_t bmp_padding(size_t width) /* bmp.rs:108-113 */
{
    size_t row = width * 3;
    return (size_t)floorf((4.0f * ceilf((float)row / 4.0f)) - (float)row);
}
size_t orc_bmp_size(size_t width, size_t height)
{
    size_t pad = bmp_padding(width);
    size_t n = width * height;
    size_t data = n * 3;
    for (size_t i = 0; i < n; ++i)
        if ((i + 1) % width == 0 && i != 0) data += pad; /* bmp.rs:88 */
    return 54 + data;
}
size_t orc_bmp_encode(const uint32_t* buffer, size_t width, size_t height, uint8_t* out)
{
    size_t pad = bmp_padding(width);
    size_t n = width * height;
    size_t total = orc_bmp_size(width, height);
    size_t data_len = total - 54;
    uint8_t* flat = (uint8_t*)malloc(data_len ? data_len : 1);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) { /* bmp.rs:79-94 */
        color c = col_from_hex(buffer[i]);
        flat[k++] = c.b; flat[k++] = c.g; flat[k++] = c.r;
        if ((i + 1) % width == 0 && i != 0)
            for (size_t p = 0; p < pad; ++p) flat[k++] = 0;
    }
    uint8_t* h = out; /* bmp.rs:38-72 */
    uint32_t fsz = (uint32_t)(54 + data_len), off = 54, dib = 40, w32 = (uint32_t)width, h32 = (uint32_t)height,
             bpp = 24, dl = (uint32_t)data_len;
    memset(h, 0, 54);
    h[0] = 'B'; h[1] = 'M';
    memcpy(h + 2, &fsz, 4);
    memcpy(h + 10, &off, 4);
    memcpy(h + 14, &dib, 4);
    memcpy(h + 18, &w32, 4);
    memcpy(h + 22, &h32, 4);
    h[26] = 1; h[27] = 0;
    memcpy(h + 28, &bpp, 4); /* 4 bytes of BMP_BITS_PER_PIXEL, then 2 zero bytes: bits(2)+compression(4) */
    memcpy(h + 34, &dl, 4);
    /* chunks(width*3+pad).rev().flatten()  bmp.rs:99-105 */
    size_t chunk = width * 3 + pad;
    size_t nchunks = chunk ? (data_len + chunk - 1) / chunk : 0;
    uint8_t* dst = out + 54;
    for (size_t ci = nchunks; ci-- > 0;) {
        size_t start = ci * chunk;
        size_t len = (start + chunk <= data_len) ? chunk : data_len - start;
        memcpy(dst, flat + start, len);
        dst += len;
    }
    free(flat);
    return total;
}
