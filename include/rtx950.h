/*
 * rtx950.h — C ABI of the MI355X-native ray-cast renderer (drop-in boundary).
 *
 * The reference (ElrohirGT/Raytracing, a pure-Rust crate) has no FFI of its own; its de-facto
 * operator boundary is
 *
 *     pub fn render(framebuffer: &mut Framebuffer, data: &Model)      reference src/render.rs:234
 *     pub fn init_render(framebuffer: &mut Framebuffer, data: &Model) reference src/render.rs:13
 *
 * Every entry point below is what a Rust `extern "C"` block for that path would bind (the
 * binding itself is shown in INTEGRATION.md).  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary.  All structs are plain-old-data with natural alignment and are
 * layout-compatible with `#[repr(C)]` Rust structs of the same field order.
 *
 * Conventions
 *   - every function returns an `int` status: 0 = RT_OK, negative = error (see rt_status);
 *     nothing throws or aborts across the boundary; `rt_last_error` gives the message.
 *   - colours are packed 0x00RRGGBB, exactly `From<Color> for u32` (reference src/color.rs:76-79).
 *   - a context is used by one host thread at a time; distinct contexts are independent
 *     (reference `render` needs `&mut Framebuffer`, so it is not re-entrant either).
 *   - every call makes its context's GPU current for the duration of the call and restores the calling
 *     thread's current HIP device before it returns.
 *   - test and tuning knobs are explicit calls (rt_set_tuning, rt_set_grid_limit).  The library reads ONE
 *     environment variable, HSA_TOOLS_LIB, and only to see whether a profiler that serialises kernels is attached
 *     (see RT_TUNE_HEAVY_SYNC, which overrides what it concludes: set it explicitly when profiling the side kernel).
 */
#ifndef RTX950_H
#define RTX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX950_ABI_VERSION 2u

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,   /* null pointer, zero size, bad band, bad index in a record   */
    RT_ERR_NO_DEVICE = -2,     /* HIP runtime reports no usable GPU / bad device ordinal     */
    RT_ERR_HIP = -3,           /* a HIP call failed; message carries hipGetErrorString       */
    RT_ERR_NO_SCENE = -4,      /* rt_render* called before rt_scene_upload                   */
    RT_ERR_UNSUPPORTED = -5,   /* e.g. max_depth > 3, more lights than RT_MAX_LIGHTS         */
    RT_ERR_LIMIT = -6,         /* a fixed-capacity entry point was given too large an input  */
    RT_ERR_DEVICE_WAIT = -7    /* rt_sync: a device-side wait between the context's two streams (side kernel for
                                  heavy tiles) gave up; frames rendered since the last rt_sync may be incomplete */
} rt_status;

/* reference src/render.rs:83 — `if depth > 3 { return sky }` */
#define RT_MAX_DEPTH 3u
/* capacity of the per-frame light table (reference uses 2: src/render.rs:241-243) */
#define RT_MAX_LIGHTS 16u
/* `texture: None` (reference src/material.rs:6, RUBBER at :114-122) */
#define RT_TEXTURE_NONE (-1)
/* value stored in hit-index planes where the depth-0 ray hit nothing */
#define RT_NO_HIT 0xFFFFFFFFu

typedef struct rt_vec3 { float x, y, z; } rt_vec3;

/* reference src/cube.rs:13-19 `Cube { id, center, size, material, bounds }`; bounds are derived
 * inside rt_scene_upload exactly as `Cube::compute_bounds` (src/cube.rs:51-66): center -/+ size/2.
 * `material` indexes the rt_material table passed to rt_scene_upload. */
typedef struct rt_cube {
    uint32_t id;
    rt_vec3  center;
    float    size;
    uint32_t material;
} rt_cube;

/* reference src/sphere.rs:10-15 `Sphere { id, center, radius, material }` */
typedef struct rt_sphere {
    uint32_t id;
    rt_vec3  center;
    float    radius;
    uint32_t material;
} rt_sphere;

/* reference src/material.rs:4-18 `Material`.  `texture` is an index into the atlas array given
 * to rt_scene_upload (the order of `enum Textures`, src/texture.rs:35-45) or RT_TEXTURE_NONE. */
typedef struct rt_material {
    uint32_t diffuse_0rgb;
    int32_t  texture;
    float    specular;
    float    albedo0;
    float    albedo1;
    float    reflectivity;
    float    transparency;
    float    refractive_index;
} rt_material;

/* reference src/light.rs:5-9 `Light { position, color, intensity }` */
typedef struct rt_light {
    rt_vec3  position;
    uint32_t color_0rgb;
    float    intensity;
} rt_light;

/* reference src/camera.rs:5-17 (eye, center, up); `has_changed` is host-loop state, not render input */
typedef struct rt_camera {
    rt_vec3 eye;
    rt_vec3 center;
    rt_vec3 up;
} rt_camera;

/* Everything `render` reads from `&Model` (reference src/lib.rs:42-52) that is per-frame, plus the
 * framebuffer geometry (src/framebuffer.rs:6-13) and the scanline band this call renders.
 *   lights      — `data.lights` followed by `data.top_light`, the order render.rs:241-243 builds
 *   ambient     — `data.ambient_light`, used for the cube pass (render.rs:299); the sphere pass
 *                 uses the literal 1.0 (render.rs:269) inside the library
 *   y0,y1       — rows [y0,y1) of the full width x height frame are rendered; 0,height = whole frame
 *   max_depth   — reference constant 3 (render.rs:83); 0..3 accepted
 */
typedef struct rt_frame {
    rt_camera       camera;
    uint32_t        width;
    uint32_t        height;
    uint32_t        y0;
    uint32_t        y1;
    const rt_light* lights;
    uint32_t        n_lights;
    float           ambient_light;
    uint32_t        sky_0rgb;
    uint32_t        max_depth;
} rt_frame;

/* Optional per-pixel debug planes for the bit-exact parity check (no reference analogue: the
 * reference keeps these values only transiently, src/render.rs:87-101).  Any pointer may be NULL.
 * Each plane has (y1-y0)*width u32 entries, row-major, band-relative.
 *   hit_cube   — array index of the nearest cube hit by the depth-0 ray of the cube pass, or RT_NO_HIT
 *   hit_sphere — same for the sphere pass
 *   tree_hash  — FNV-1a style hash over every object index the pixel's whole ray tree selected
 *                (nearest hit and shadow blocker per light, every depth, both passes)
 */
typedef struct rt_debug_planes {
    uint32_t* hit_cube;
    uint32_t* hit_sphere;
    uint32_t* tree_hash;
} rt_debug_planes;

/* Ray/test counters of one rt_render* call (summed over the band).  "uniq" counts what the GPU
 * executes (secondary rays traced once per hit); the reference re-casts them once per light
 * (src/render.rs:154-193 sits inside the fold at :104-106). */
typedef struct rt_counters {
    uint64_t nearest_rays;    /* object-list traversals for a nearest hit (render.rs:87-101) */
    uint64_t shadow_rays;     /* object-list traversals for a shadow query (render.rs:44-70) */
    uint64_t cube_tests;      /* executed Cube::ray_intersect calls (early exit honoured)    */
    uint64_t sphere_tests;    /* executed Sphere::ray_intersect calls                        */
    uint64_t shaded_hits;     /* nearest-hit queries that found an object                    */
    /* the same five quantities as the REFERENCE algorithm executes them: a ray-tree node at
     * depth k runs n_lights^k times there because the recursion sits inside the light fold */
    uint64_t ref_nearest_rays;
    uint64_t ref_shadow_rays;
    uint64_t ref_cube_tests;
    uint64_t ref_sphere_tests;
    uint64_t ref_shaded_hits;
} rt_counters;

/* What the kernels EXECUTE for a band (rt_count_executed), per lane, as opposed to the reference's brute-force
 * tests in rt_counters: the wave-level cull, the two-sided slab filter and the tile classification prove most
 * of those tests away.  bench.py prices each event in FP32 operations ("Roofline") to state the
 * kernel's executed arithmetic rate.  No reference counterpart. */
typedef struct rt_exec_counters {
    uint64_t cull_tests;        /* cube or sphere cull-stream records tested against a wave's ray bundle  */
    uint64_t chunk_tests;       /* chunk bounding boxes tested against a wave's ray bundle (long lists)   */
    uint64_t candidate_evals;   /* two-sided slab filter evaluations: (candidate cube, active ray) pairs  */
    uint64_t exact_evals;       /* Cube::ray_intersect evaluated exactly for a pair the filter left open  */
    uint64_t bvh_steps;         /* per-lane BVH walk steps (two sibling boxes each; long lists)           */
    uint64_t sphere_evals;      /* Sphere::ray_intersect evaluations: (candidate sphere, active ray)      */
    uint64_t queries;           /* per-ray query set-ups (nearest, exact re-run, shadow)                  */
    uint64_t light_iterations;  /* (shaded hit, light) pairs of the Phong fold                            */
    uint64_t resolved_hits;     /* winning hits resolved (exact distance, face, uv, material)             */
    uint64_t blockers;          /* shadow blockers whose exact distance was computed                      */
    uint64_t child_setups;      /* hits that set up reflection/refraction rays                            */
    uint64_t primary_rays;      /* primary rays generated (pixels of traced tiles)                        */
} rt_exec_counters;
#define RT_EXEC_EVENTS 12u

/* Largest frame, in 8x8 pixel tiles, that rt_render_device_noalloc accepts (4K = 129600 tiles). */
#define RT_NOALLOC_MAX_TILES 131072u

/* Keys of rt_set_tuning. */
typedef enum rt_tuning {
    RT_TUNE_COHERENT_SCALE = 1, /* factor on the long-list traversal threshold: 0 = every wave walks the BVH,
                                   a huge value = every wave takes the chunk traversal; default 1.
                                   Takes effect at the next rt_scene_upload                                */
    RT_TUNE_SHADOW_LIST = 2,    /* -1 automatic (default), 0 / 1: shadow queries of long lists walk the
                                   Morton-ordered / the list-ordered stream.  Next rt_scene_upload         */
    RT_TUNE_DARK_KERNEL = 3,    /* -1 automatic (default), 0 / 1: kernel variant without / with the
                                   "direct light is exactly black" test.  Next rt_scene_upload             */
    RT_TUNE_FILL_MEMSET = 4,    /* 0 (default) / 1: sky pixels are written by a fill ahead of the launches
                                   instead of by build_queue.  Next render call                            */
    RT_TUNE_HEAVY_CLASS = 6,    /* -1 (default, automatic) or 0..8: the tiles of a plain frame whose cost class in the
                                   context's previous launch was at least this (class c = 8 K << c cycles and up;
                                   at most about 512 tiles, heaviest classes first) are traced by a side kernel on
                                   the context's second stream - a workgroup of three waves per tile, concurrent
                                   with the frame's other launches and joined to the context's stream before the
                                   call returns - instead of one wave each: shortens a frame that is bound by
                                   its slowest tile (one frame at a time, small frames, the parts of an N-GPU
                                   split), costs about 12 us of event traffic per frame, and should stay off
                                   where several contexts keep frames in flight (more streams than hardware
                                   queues).  8 = never.  Automatic: class 5 for the synchronous rt_render of
                                   frames of up to 65536 tiles, never for the asynchronous entry points.
                                   Results do not depend on it.  Next render call                            */
    RT_TUNE_HEAVY_SYNC = 7,     /* -1 (default, automatic), 0 or 1: how the side kernel's stream and the context's stream
                                   meet.  1: device-side - classify_tiles releases a one-wave gate kernel on the side
                                   stream, a one-wave join kernel on the context's stream counts finished tiles
                                   (3 us per frame).  0: events both ways (18 us per frame).  Automatic: device-side
                                   unless a tool that runs one kernel at a time is attached to the process
                                   (a rocprofiler-sdk tool such as rocprofv3, or HSA_TOOLS_LIB set): under such a
                                   tool a polling kernel can wait for a kernel that is not allowed to start.
                                   Next render call                                                          */
    RT_TUNE_HEAVY_SPIN_LIMIT = 8, /* polls the join kernel of the side stream makes before it gives up: 0 = the default
                                   (2^24, seconds), 1 = none at all: every launch with a side kernel is REPORTED as
                                   given up (fault injection for the test of the recovery path: rt_render renders
                                   the frame again without the side kernel and disables it for the context, rt_sync
                                   returns RT_ERR_DEVICE_WAIT).  Setting 0 puts a disabled side kernel back in
                                   service.  Next render call                                                 */
    RT_TUNE_WAVEFRONT = 9,      /* -1 (default, automatic), 0 or 1: plain frames are rendered by the wavefront launches - one
                                   per ray-tree level over bundles of coherent rays, then a combine launch - instead of
                                   the tile-resident walk of the whole tree.  Results do not depend on it.  Next render call */
    RT_TUNE_WAVEFRONT_CAP = 10, /* items (ray bundles) per deeper level that the wavefront buffers hold; 0 (default) = from
                                   the frame size.  Tiles whose bundles do not fit are rendered by the tile-resident
                                   kernel instead, so tests lower it to force that path.  Next render call            */
    RT_TUNE_CULL_REFINE = 11,   /* -1 (default, automatic), 0, 1 or 2: the plain short-list kernel without / with the
                                   cull refinement (1: the box form of the wave-level cull applied where the one-radius
                                   form leaves 20 candidates or more) / with the refinement and the walk by bundles (2:
                                   a wave whose secondary rays left through different faces is walked bundle by bundle
                                   where its cull is loose).  Automatic: 2 for launches of at most 65536 tiles (bound by
                                   their slowest tiles), 1 for larger ones, 0 when the side kernel for heavy tiles is in
                                   use (it takes the slowest tiles anyway).  Results do not depend on it.  Next render
                                   call                                                                        */
    RT_TUNE_BVH_DEPTH_LIMIT = 5 /* deepest BVH the per-lane walk takes (1..28, default 28 = the walk's stack);
                                   a deeper tree sends every wave through the chunk traversal.  28 levels
                                   cannot be exceeded by a real list, so tests lower it.  Next rt_scene_upload */
} rt_tuning;

typedef struct rt_ctx rt_ctx;

/* ABI / build identification.  No reference counterpart (library housekeeping). */
uint32_t rt_abi_version(void);

/* Create / destroy one context bound to HIP device `device` (one per GPU; owns a stream, the
 * device-resident scene, the device framebuffer rt_render copies from, and the launches' work buffers).
 * Replaces: construction of `Framebuffer::new` (reference src/framebuffer.rs:44-58) + `Model`
 * (src/main.rs:249-259) as far as device state is concerned. */
int rt_ctx_create(int device, rt_ctx** out_ctx);
int rt_ctx_destroy(rt_ctx* ctx);

/* Message of the last failing call on this context (or on the calling thread if ctx is NULL).
 * Replaces the reference's `unwrap`/`expect` panics (src/texture.rs:142, src/main.rs:41,141). */
const char* rt_last_error(const rt_ctx* ctx);

/* Upload the static part of `Model`: cubes, spheres (src/lib.rs:43-44), the material table
 * (src/material.rs:34-122) and the texture atlases (src/texture.rs:22-32, loaded by
 * `Texture::new`, :141-169, as row-major RGB8: 3 bytes per texel, `idx = y*width + x`).
 * Host arrays are copied; the caller keeps ownership.  n_cubes / n_spheres may be 0; more than 2^26 cubes or
 * 2^31 - 1 spheres are refused (RT_ERR_UNSUPPORTED).
 *   atlas_rgb8 — n_textures consecutive images of tex_w*tex_h*3 bytes
 *   sprite     — `sprite_size` (16 in the reference, src/texture.rs:59-67)
 */
int rt_scene_upload(rt_ctx* ctx,
                    const rt_cube* cubes, size_t n_cubes,
                    const rt_sphere* spheres, size_t n_spheres,
                    const rt_material* materials, size_t n_materials,
                    const uint8_t* atlas_rgb8, size_t n_textures,
                    uint32_t tex_w, uint32_t tex_h, uint32_t sprite);

/* render(framebuffer, data) — reference src/render.rs:234-324.  Synchronous, like the reference.
 * Renders rows [y0,y1) and copies them to host memory `out_0rgb` ((y1-y0)*width u32, 0x00RRGGBB,
 * `framebuffer.buffer` layout, src/framebuffer.rs:15,105).  Reproduces the reference's unpainted
 * pixel i == width (src/render.rs:316-322 -> XTooLarge, src/framebuffer.rs:101-102), which keeps
 * the clear colour 0x000000 (src/framebuffer.rs:45,63-65).
 * `dbg_host` (nullable) receives the debug planes in host memory.
 * The single read-back is a DMA straight into `out_0rgb`: give it memory from rt_host_alloc (pinned) and the
 * 4K frame's 33 MB arrive at PCIe rate; pageable memory works too, staged by the HIP runtime at a third of it. */
int rt_render(rt_ctx* ctx, const rt_frame* frame, uint32_t* out_0rgb,
              const rt_debug_planes* dbg_host);

/* Same computation, but the output (and debug planes, if any) are DEVICE pointers on the
 * context's GPU and nothing is copied to the host; the call returns after the kernel has been
 * enqueued on the context's stream (rt_sync waits).  This is what a multi-GPU host uses to render
 * its band straight into a buffer it then gathers over xGMI, and what bench.py times.
 * Replaces the same reference function (src/render.rs:234). */
int rt_render_device(rt_ctx* ctx, const rt_frame* frame, uint32_t* d_out_0rgb,
                     const rt_debug_planes* dbg_device);

/* Scanline-interleaved partition of rows [y0,y1) for multi-GPU load balance (sky rows are cheap,
 * diorama rows are not): the band is cut into 16-row blocks and part `part` of `n_parts` renders
 * blocks part, part+n_parts, part+2*n_parts, ...  The output is COMPACT: the part's blocks are
 * stored one after another (rt_part_rows gives the row count), so each GPU's buffer is one
 * contiguous message for the gather to GPU0.  n_parts = 1 is rt_render_device.  Debug planes use
 * the same compact layout.  No reference counterpart (the reference has one address space; its
 * pixel loop is reference src/render.rs:248-306). */
int rt_render_device_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part,
                          uint32_t* d_out_0rgb, const rt_debug_planes* dbg_device);

/* Number of rows part `part` of `n_parts` holds for a band of `rows` rows (16-row blocks). */
uint32_t rt_part_rows(uint32_t rows, uint32_t n_parts, uint32_t part);

/* Rows of the band that can hold anything but sky: on return every pixel of rows [y0, *out_y_first)
 * and [*out_y_end, y1) of the frame is the sky colour in the rendered image (except the reference's
 * unpainted pixel i == width, see rt_render), so a multi-GPU host need not move those rows: GPU0
 * fills them itself.  Computed on the host from the uploaded scene (no device work, no
 * synchronisation) with the projection the tile classification uses: conservative, never exact; the
 * whole band when the scene has spheres or the camera is irregular; y_first == y_end == y0 when
 * nothing projects into the band.  No reference counterpart (the reference has one address space;
 * its pixel loop is reference src/render.rs:248-306). */
int rt_frame_row_extent(rt_ctx* ctx, const rt_frame* frame, uint32_t* out_y_first, uint32_t* out_y_end);

/* Page-locked host memory for framebuffers that rt_render writes (`Framebuffer::new`, reference
 * src/framebuffer.rs:44-58, allocates its `Vec<u32>` from the heap; the host mirror's Framebuffer uses
 * this instead).  Usable with every context of the process. */
int rt_host_alloc(size_t bytes, void** out_ptr);
int rt_host_free(void* ptr);

/* Allocation-free variant of rt_render_device, for callers that cannot call hipMalloc: code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  NO COUNTERPART in the reference.  Same
 * kernels and same output as rt_render_device, but the launches' work buffers (tile stamps, tile queue, tile
 * costs) live in one fixed array per GPU inside the library (6.4 MB, held until the process ends) instead of
 * buffers the context allocates and grows; the call performs no hipMalloc/hipFree, creates no events and takes
 * no timing samples.  Frames of more than RT_NOALLOC_MAX_TILES 8x8 tiles are refused with RT_ERR_LIMIT.  Every
 * call on a GPU, from any context, shares that GPU's array: the array keeps the last call's contents (the
 * call clears what must start at zero), and calls that overlap in time on one GPU race - wait for one
 * (rt_sync) before issuing the next from another context. */
int rt_render_device_noalloc(rt_ctx* ctx, const rt_frame* frame, uint32_t* d_out_0rgb);

/* One frame on n GPUs, driven by one host thread (native counterpart of the RCCL gather in
 * raytracing_amd/multigpu.py; pixel independence: reference src/render.rs:248-251,278-281).  ctxs[g] is a
 * context on GPU g with the scene uploaded (several contexts may share a GPU); the rows of the band that can
 * hold geometry (rt_frame_row_extent) are cut into 16-row blocks dealt round-robin to the contexts, each
 * context renders its blocks into a buffer it owns, and the blocks are copied into frame order at
 * `d_frame_on_ctx0` (device memory of ctxs[0]'s GPU, (y1-y0)*width u32) by peer copies over xGMI, one strided
 * 2-D copy per context on that context's stream; ctxs[0] fills the sky rows itself.  Returns once everything
 * is enqueued; rt_sync(ctxs[0]) waits for the whole frame.  `out_bytes_moved` (nullable): bytes that crossed
 * between contexts.  The caller must not reuse `d_frame_on_ctx0` before the previous frame into it has been
 * consumed (swap buffers or rt_sync). */
int rt_render_multi(rt_ctx* const* ctxs, uint32_t n_ctx, const rt_frame* frame, uint32_t* d_frame_on_ctx0,
                    uint64_t* out_bytes_moved);

/* The gather step alone (SURVEY 8(b) `rt_gather_bands`): d_parts[g] is the compact output of
 * rt_render_device_part(ctxs[g], .., n_ctx, g, ..) for a band of `rows` rows (device memory of ctxs[g]'s GPU,
 * written on ctxs[g]'s stream); the parts are scattered into band order at `d_band_on_ctx0`.  Each copy is
 * enqueued on its part's stream, ctxs[0]'s stream waits for all of them.  No reference counterpart. */
int rt_gather_bands(rt_ctx* const* ctxs, uint32_t n_ctx, const uint32_t* const* d_parts, uint32_t width, uint32_t rows,
                    uint32_t* d_band_on_ctx0, uint64_t* out_bytes_moved);

/* Wait for all work enqueued on the context's stream.  No reference counterpart (the reference
 * is synchronous).  Returns RT_ERR_DEVICE_WAIT if a launch since the last rt_sync used the side kernel for heavy
 * tiles (RT_TUNE_HEAVY_CLASS) and one of its device-side waits gave up: those frames may be incomplete and must be
 * rendered again; the side kernel is then out of service for the context (RT_TUNE_HEAVY_SPIN_LIMIT 0 restores it).
 * The synchronous rt_render handles that case itself: it renders the frame again without the side kernel. */
int rt_sync(rt_ctx* ctx);

/* Run the counting variant of the render kernel on the band (no image output) and return the
 * ray / test counters.  No reference counterpart; used for the Mrays/s metric and checked
 * against the oracle's counters in tests. */
int rt_count_rays(rt_ctx* ctx, const rt_frame* frame, rt_counters* out);
/* Same, restricted to interleaved part `part` of `n_parts` (see rt_render_device_part). */
int rt_count_rays_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, rt_counters* out);
/* Same launch, also returning what the kernels executed (either pointer may be NULL). */
int rt_count_executed(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, rt_counters* out,
                      rt_exec_counters* exec);

/* Average device time, in milliseconds, of the render kernel launches issued since the last call
 * (measured with HIP events on the context's stream), and their number.  No reference counterpart
 * (the reference shows fps in its window title, src/main.rs:122-131). */
int rt_kernel_time_ms(rt_ctx* ctx, double* out_avg_ms, uint32_t* out_launches);

/* HIP-event timing of rt_render_device launches (off by default): enable = 0 off, 1 every launch,
 * k > 1 every k-th launch (a pair of event records costs the stream about 6 us per launch, 4 % of the
 * 4K frame: a sample keeps a live measurement from slowing what it measures). */
int rt_enable_timing(rt_ctx* ctx, int enable);

/* Tuning, no counterpart in the reference: upper limit of the render kernel's grid (single-wave workgroups
 * striding over the tile queue); 0 restores the default of 32768, the best for one frame at a time.  A
 * host that keeps two frames in flight on two contexts gains from a smaller grid (12288: the 4K diorama
 * 3 % faster, the other benchmark frames within 2 % either way), because the two
 * frames' kernels then share the GPU from the start instead of queueing behind each other. */
int rt_set_grid_limit(rt_ctx* ctx, uint32_t max_workgroups);

/* Test and tuning knobs, no counterpart in the reference (keys: rt_tuning).  The parity tests use them to
 * force each traversal and kernel variant on any scene; results never depend on them. */
int rt_set_tuning(rt_ctx* ctx, int key, double value);

/* Diagnostic, no counterpart in the reference: runs about n_cases pseudo-random f32 divisions on the
 * device, once through the kernels' shared-reciprocal sequence and once through the compiler's IEEE
 * division, and reports how many results differ in any bit (must be 0: the kernels' claim of
 * reference-identical rounding of normalize() and of the slab quotients rests on it). */
int rt_selftest_division(rt_ctx* ctx, uint32_t seed, uint64_t n_cases, uint64_t* out_mismatches, uint64_t* out_cases);

/* Name of the render kernel instantiation that rt_render / rt_render_device launch for the uploaded scene,
 *  e.g. "render_kernel<false,false,0,false>" (template arguments: debug planes, counters, cube list kind - 0 short,
 * 1 long, 2 short with the cull refinement, 3 short with the refinement and the walk by bundles, as chosen by the
 * context's last plain launch (RT_TUNE_CULL_REFINE) -, dark-material test), as it
 * appears in a rocprofv3 kernel trace (there with spaces after the commas); with RT_TUNE_HEAVY_CLASS in effect
 * render_heavy_kernel<long list, dark-material test> runs beside it.  No reference counterpart. */
const char* rt_kernel_name(const rt_ctx* ctx);

/* Diagnostic, no counterpart in the reference: 0 unless a device-side wait of the side kernel for heavy tiles
 * (RT_TUNE_HEAVY_CLASS) gave up in one of the context's launches and neither rt_render nor rt_sync has dealt
 * with it yet (never expected; see rt_sync).  Waits for the stream. */
int rt_debug_status(rt_ctx* ctx, uint32_t* out_word);

/* Diagnostic, no counterpart in the reference: cycles each 8x8 tile of the context's last launch took
 * (0 = not traced), in the launch's tile grid (ceil(width/8) columns, two tile rows per 16-row block of the
 * part); the library uses them only to order the next launch's queue.  Copies min(cap, tiles) words after
 * waiting for the stream; *out_tiles = tiles of the last launch's grid buffer. */
int rt_debug_tile_costs(rt_ctx* ctx, uint32_t* out, size_t cap, size_t* out_tiles);

/* Diagnostic, no counterpart in the reference: the item records (ray bundles, 8 words each: cycles the item took,
 * kind | pass << 1, lane mask low / high, root tile entry, first child, children, launch stamp) that the context's last
 * wavefront frame (RT_TUNE_WAVEFRONT) allocated at ray-tree level `level` (1..3; 0: the root slots, two per tile
 * position; level 4: nothing is copied and *out_items = the tiles on the frame's redo list).  Copies min(cap_items, items)
 * records after waiting for the stream; *out_items = items allocated. */
int rt_debug_wavefront_items(rt_ctx* ctx, uint32_t level, uint32_t* out_words, size_t cap_items, size_t* out_items);

/* Raw HIP stream of the context as an opaque pointer (hipStream_t), for hosts that order their
 * own work (e.g. an RCCL gather) after the render.  No reference counterpart. */
int rt_get_stream(rt_ctx* ctx, void** out_hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RTX950_H */
