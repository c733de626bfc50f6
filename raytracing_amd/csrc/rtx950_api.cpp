// rtx950_api.cpp — the extern "C" boundary declared in include/rtx950.h.
//
// Host-side work per call is what the reference does outside its pixel loops: validate, derive
// cube bounds (reference src/cube.rs:51-66), hoist the camera basis (src/camera.rs:34-36, a pure
// function of the camera that the reference recomputes per pixel), assemble kernel parameters,
// launch.  There is no CPU rendering path here: if HIP or the GPU is missing every call fails
// with RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtx950.h"
#include "rtx950_device.h"

using namespace rtx950;

// Heavy tiles, automatic mode: tiles of >= 128 K cycles (one-wave terms) go to render_heavy_kernel on the second
// stream when the call is the synchronous rt_render (one frame at a time by construction: the frame's slowest
// tile is its duration), the launch is small enough to be bound by that tile rather than by throughput, and
// the cube list is a short one (the slow tiles of a long list are per-lane BVH walks, which three waves do
// not shorten: 10k-cube world 1.55 -> 1.76 ms with the side kernel).
// The side stream costs about 12 us per frame in event traffic, and a process that keeps several contexts'
// frames in flight has more streams than hardware queues (a waiting side kernel then blocks another context's
// launches): asynchronous callers opt in with RT_TUNE_HEAVY_CLASS.
constexpr uint32_t kDefaultHeavyClass = 5;
constexpr size_t kAutoHeavyMaxTiles = 65536;  // 1080p: 32 400 tiles, 4K: 129 600
constexpr uint32_t kDefaultGrid = 32768;  // swept at the end of round 1, one frame at a time: 24576 +2 %, 28672..49152 within 0.5 %, 65536 +1.5 %

// Work buffers of a launch (device memory beyond inputs and outputs): the tile stamps classify_tiles
// writes, the tile queue build_queue fills, the tile costs render_kernel records.  A context owns a set
// that grows with the frame size; rt_render_device_noalloc uses the fixed-size set in `scratch` below.
struct TileScratch {
    uint32_t* stamps = nullptr;
    size_t stamps_cap = 0;  // in words
    uint32_t* queue = nullptr;
    size_t queue_cap = 0;   // in words
    uint32_t* cost = nullptr;
    size_t cost_cap = 0;    // in words
    uint32_t frame_id = 0;    // stamps equal to the launch's frame id are this launch's (ids only grow)
    uint32_t launch_seq = 0;  // selects the queue header bank
    bool growable = true;     // false: fixed capacity, a frame that does not fit is refused (RT_ERR_LIMIT)
    // heavy tiles (growable sets only): two banks of kHeavyCap list entries, a mark word per tile, and which
    // launch (sequence number, tile grid) the list that the last launch's build_queue wrote is for
    uint32_t* heavy = nullptr;
    size_t heavy_cap = 0;     // in words
    uint32_t* heavy_mark = nullptr;
    size_t mark_cap = 0;      // in words
    bool heavy_list_ready = false;
    uint32_t heavy_list_seq = 0;
    uint32_t heavy_geom[6] = {0, 0, 0, 0, 0, 0};  // width, height, y0, y1, n_parts, part
};

// Work buffers of the wavefront path (WfParams): one allocation per context, sized by the launch's tile grid.
struct WfScratch {
    char* base = nullptr;
    size_t bytes = 0;
    size_t tiles = 0;            // tile grid the buffers were sized for
    uint32_t cap_tune = 0;       // RT_TUNE_WAVEFRONT_CAP the buffers were sized with
    uint32_t seq = 0;            // stamp of the last launch
    WfParams w{};                // pointers and capacities (seq, level set per launch)
    size_t zero_bytes = 0;       // leading part that must start at zero: header, pending, redo marks, item records
};

// ---- the allocation-free entry point's work buffers (rt_render_device_noalloc) -------------------------
// One array per GPU, held until the process ends: queue (64 + 9 x tiles words), stamps (64 + 2 x (tiles + coarse)
// words: a plane for the cubes, one for the spheres) and tile costs (tiles + 1 words) for frames of up to RT_NOALLOC_MAX_TILES 8x8 tiles (6.4 MB).
constexpr size_t kNaTiles = RT_NOALLOC_MAX_TILES;
constexpr size_t kNaQueueWords = kQueueBase + (kCostClasses + 1) * kNaTiles;
constexpr size_t kNaStampWords = kStampBase + 2 * (kNaTiles + kNaTiles / 16 + 64);  // two planes (cubes, spheres); coarse level: <= tiles/64 + (tiles_x + tiles_y)/8 + 1
constexpr size_t kNaCostWords = kNaTiles + 64;
constexpr size_t kNaWords = kNaQueueWords + kNaStampWords + kNaCostWords;
static __device__ __align__(256) uint32_t scratch[kNaWords];

constexpr int kMaxDevices = 64;
struct DeviceScratch {  // host-side view of one GPU's copy of `scratch`
    std::mutex mu;
    bool ready = false;
    TileScratch ts;
};
static DeviceScratch g_dev_scratch[kMaxDevices];


struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // render_heavy_kernel runs here.  It starts when classify_tiles of the same launch, on `stream`, says so, and
    // `stream` meets it again in the heavy_join kernel behind its own launches (words in the queue header,
    // no events): callers see one stream, as before
    hipStream_t stream_heavy = nullptr;
    // ... unless a tool that serialises kernels is attached to the process (rocprofv3 --pmc runs one kernel at a
    // time and does not hand the device out in submission order: a polling kernel then waits for one that is
    // not allowed to start, until it gives up): the two streams are then ordered by events instead, which such
    // tools follow.  Decided when the side stream is created; RT_TUNE_HEAVY_SYNC overrides it.
    bool heavy_events = false;
    hipEvent_t heavy_fork = nullptr, heavy_join = nullptr;
    int tune_heavy_sync = -1;  // -1: automatic, 0: events, 1: device-side
    uint32_t tune_heavy_spin = 0;  // heavy_join's patience in polls (0: the default, 1: none - the recovery path's test)
    // A device-side wait of the side kernel's machinery that gave up leaves a status word in the queue header and,
    // possibly, an incomplete frame.  Every launch with a side kernel sets heavy_unchecked; the next host wait on
    // the context (rt_render's, rt_sync's) reads the word back and, if it is set, takes the side kernel out of
    // service for the context (heavy_disabled) - rt_render then renders the frame again without it.
    bool heavy_unchecked = false;
    bool heavy_disabled = false;
    uint32_t* h_status = nullptr;  // one pinned word: the read-back target
    std::string err;
    // scene
    bool has_scene = false;
    CubeRec* d_cubes = nullptr;
    float* d_cube_filt = nullptr;
    std::vector<float> h_cube_box;  // centre xyz + |half size| per cube, list order (rt_frame_row_extent)
    // rt_frame_row_extent's last answer (a host loop over all cubes; hosts ask once per frame, cameras often stand still)
    bool extent_valid = false;
    rt_camera extent_camera{};
    uint32_t extent_dims[4] = {0, 0, 0, 0}, extent_rows[2] = {0, 0};
    size_t h_n_spheres = 0;
    uint32_t* d_cube_id = nullptr;
    uint32_t* d_cube_perm = nullptr;
    BvhPair* d_bvh = nullptr;
    float* d_sphere_filt = nullptr;
    float* d_chunk_box = nullptr;
    float* d_cube_filt_list = nullptr;   // list-ordered stream for shadow queries of long lists (when compact)
    uint32_t* d_cube_id_list = nullptr;
    float* d_chunk_box_list = nullptr;
    uint32_t* d_cube_mat = nullptr;
    SphereRec* d_spheres = nullptr;
    SphereMeta* d_sphere_meta = nullptr;
    MatRec* d_mats = nullptr;
    uint32_t* d_atlas = nullptr;
    Params base{};
    // framebuffer + debug planes owned by the context (used by rt_render)
    uint32_t* d_fb = nullptr;
    size_t fb_capacity = 0;  // in pixels
    uint32_t* d_dbg[3] = {nullptr, nullptr, nullptr};
    size_t dbg_capacity[3] = {0, 0, 0};
    unsigned long long* d_counters = nullptr;
    TileScratch own;  // tile stamps, work queue, tile costs of this context's launches
    WfScratch wf;     // item, node and ray buffers of the wavefront path
    int tune_wavefront = -1;      // -1: automatic, 0: never, 1: every plain frame
    int tune_cull_refine = -1;    // -1: automatic (refine_choice), 0 / 1 / 2: plain short-list kernel without / with the cull refinement / with refinement and bundles
    uint32_t last_refine = 0;     // what the last plain launch used (rt_kernel_name)
    uint32_t tune_wf_cap = 0;     // items per deeper level (0: from the tile grid)
    bool scene_has_children = false;  // some material reflects or refracts: the ray tree can be deeper than its root
    // rt_render_multi: this context's compact part of the frame, and the event that says it has reached GPU0
    uint32_t* d_part = nullptr;
    size_t part_capacity = 0;  // in pixels
    hipEvent_t part_done = nullptr;
    // rt_gather_bands: what this context knows about reaching GPU `peer_target` from its own GPU (asked once per
    // target): -1 not asked yet, 1 peer access enabled (a device-to-device copy goes over xGMI directly), 0 no peer
    // access between the two GPUs (the copy is a hipMemcpyPeerAsync, which the runtime stages)
    int peer_target = -1, peer_state = -1;
    // tuning / test knobs (rt_set_tuning); the library reads no environment variables
    float tune_coherent_scale = 1.0f;
    int tune_shadow_list = -1, tune_dark_kernel = -1;
    bool tune_fill_memset = false;
    uint32_t tune_bvh_depth_limit = kBvhStackMax;
    int tune_heavy_class = -1;  // -1: automatic (see render_part), 0..7: threshold class, 8: never
    uint32_t max_workgroups = kDefaultGrid;  // grid of render_kernel: 8x the resident single-wave workgroups of an MI355X
    // timing
    bool timing = false;
    uint32_t timing_every = 1, timing_phase = 0;  // every k-th launch is bracketed by events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // recorded pairs since last query
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
};

namespace {

#ifdef RTX_PROFILE
constexpr size_t kProfWaves = 1u << 18;  // per-wave phase slots of the diagnostic build
constexpr size_t kCounterBytes = 256 + kProfWaves * 40 * sizeof(uint32_t);
#else
constexpr size_t kCounterBytes = 32 * sizeof(unsigned long long);
#endif

constexpr size_t kSmallListMax = 256;  // cube lists up to this length are streamed in list order without chunk boxes

thread_local std::string g_tls_err;

int fail(rt_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    g_tls_err = msg;
    return code;
}

int fail_hip(rt_ctx* ctx, hipError_t e, const char* what)
{
    return fail(ctx, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// Makes `device` current for the calling thread and restores the caller's device on scope exit.
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != device) {
            status = hipSetDevice(device);
            if (status == hipSuccess) prev = cur;
        }
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ENTER_DEVICE(ctx)                                                     \
    DeviceGuard guard__((ctx)->device);                                       \
    if (guard__.status != hipSuccess) return fail_hip(ctx, guard__.status, "hipSetDevice")

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return fail_hip(ctx, e__, #expr); \
    } while (0)

struct H3 {
    float x, y, z;
};
// nalgebra-glm semantics (see oracle/rt_oracle.c): dot = (a0*b0 + a1*b1) + a2*b2, normalize divides
H3 h_sub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
float h_dot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
H3 h_cross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
H3 h_normalize(H3 a)
{
    float n = std::sqrt(h_dot(a, a));
    return H3{a.x / n, a.y / n, a.z / n};
}

void free_scene(rt_ctx* c)
{
    (void)hipFree(c->d_cubes);
    (void)hipFree(c->d_cube_filt);
    c->d_cube_filt = nullptr;
    (void)hipFree(c->d_cube_id);
    c->d_cube_id = nullptr;
    (void)hipFree(c->d_cube_perm);
    c->d_cube_perm = nullptr;
    (void)hipFree(c->d_bvh);
    c->d_bvh = nullptr;
    (void)hipFree(c->d_sphere_filt);
    c->d_sphere_filt = nullptr;
    (void)hipFree(c->d_chunk_box);
    c->d_chunk_box = nullptr;
    (void)hipFree(c->d_cube_filt_list);
    c->d_cube_filt_list = nullptr;
    (void)hipFree(c->d_cube_id_list);
    c->d_cube_id_list = nullptr;
    (void)hipFree(c->d_chunk_box_list);
    c->d_chunk_box_list = nullptr;
    (void)hipFree(c->d_cube_mat);
    (void)hipFree(c->d_spheres);
    (void)hipFree(c->d_sphere_meta);
    (void)hipFree(c->d_mats);
    (void)hipFree(c->d_atlas);
    c->d_cubes = nullptr;
    c->d_cube_mat = nullptr;
    c->d_spheres = nullptr;
    c->d_sphere_meta = nullptr;
    c->d_mats = nullptr;
    c->d_atlas = nullptr;
    c->has_scene = false;
}

template <typename T>
int upload(rt_ctx* ctx, T** dptr, const std::vector<T>& host)
{
    // one spare element so zero-length arrays still have a valid address
    size_t bytes = (host.size() + 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc((void**)dptr, bytes));
    HIP_TRY(ctx, hipMemset(*dptr, 0, bytes));
    if (!host.empty()) HIP_TRY(ctx, hipMemcpy(*dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

int check_frame(rt_ctx* ctx, const rt_frame* f)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    if (!f) return fail(ctx, RT_ERR_INVALID_ARG, "frame is null");
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called on this context");
    if (f->width == 0 || f->height == 0) return fail(ctx, RT_ERR_INVALID_ARG, "frame width/height must be non-zero");
    if (f->width > 65536u || f->height > 65536u) return fail(ctx, RT_ERR_UNSUPPORTED, "frame larger than 65536 in a dimension");
    if (f->y0 >= f->y1 || f->y1 > f->height) return fail(ctx, RT_ERR_INVALID_ARG, "band must satisfy y0 < y1 <= height");
    if (f->max_depth > RT_MAX_DEPTH) return fail(ctx, RT_ERR_UNSUPPORTED, "max_depth > 3 is not supported (reference constant is 3)");
    if (f->n_lights > RT_MAX_LIGHTS) return fail(ctx, RT_ERR_UNSUPPORTED, "more than RT_MAX_LIGHTS lights");
    if (f->n_lights > 0 && !f->lights) return fail(ctx, RT_ERR_INVALID_ARG, "lights is null but n_lights > 0");
    return RT_OK;
}

void fill_frame(const rt_ctx* ctx, const rt_frame* f, Params& p)
{
    p = ctx->base;
    const H3 eye{f->camera.eye.x, f->camera.eye.y, f->camera.eye.z};
    const H3 center{f->camera.center.x, f->camera.center.y, f->camera.center.z};
    const H3 upv{f->camera.up.x, f->camera.up.y, f->camera.up.z};
    const H3 forward = h_normalize(h_sub(center, eye));     // camera.rs:34
    const H3 right = h_normalize(h_cross(forward, upv));    // camera.rs:35
    const H3 up = h_normalize(h_cross(right, forward));     // camera.rs:36
    p.eye[0] = eye.x; p.eye[1] = eye.y; p.eye[2] = eye.z;
    p.right[0] = right.x; p.right[1] = right.y; p.right[2] = right.z;
    p.up[0] = up.x; p.up[1] = up.y; p.up[2] = up.z;
    p.fwd[0] = forward.x; p.fwd[1] = forward.y; p.fwd[2] = forward.z;
    p.width = f->width;
    p.height = f->height;
    p.y0 = f->y0;
    p.y1 = f->y1;
    p.n_parts = 1u;
    p.part = 0u;
    p.heavy = nullptr;
    p.heavy_mark = nullptr;
    p.heavy_now = 0u;
    p.heavy_next = 0u;
    p.heavy_class = kCostClasses;
    {
        float ob = std::max(p.bound_max, p.sphere_bound_max);
        for (int k = 0; k < 3; ++k) ob = std::max(ob, std::fabs(p.eye[k]));
        ob += 1.0f;
        p.origin_bound = (ob <= 0x1p20f) ? ob : 0x1p20f;  // NaN -> 2^20
    }
    // what every wave would compute for itself (Params: "wave-uniform float values"), in the kernels' f32 operations
    p.width_f = (float)f->width;
    p.height_f = (float)f->height;
    p.aspect = p.width_f / p.height_f;
    p.n_lights_f = (float)f->n_lights;
    p.cull_slack = 0x1p-17f * (p.bound_max + p.origin_bound + 1.0f);
    p.sphere_cull_slack = 0x1p-17f * (p.sphere_bound_max + p.origin_bound + 1.0f);
    {
        const float t = p.origin_bound + p.sphere_bound_max + 1.0f;
        p.sphere_grow = 0x1p-17f * t * t;
    }
    p.ambient = f->ambient_light;
    p.sky = f->sky_0rgb & 0x00FFFFFFu;
    p.max_depth = f->max_depth;
    p.n_lights = f->n_lights;
    for (uint32_t i = 0; i < f->n_lights; ++i) {
        p.lights[i].px = f->lights[i].position.x;
        p.lights[i].py = f->lights[i].position.y;
        p.lights[i].pz = f->lights[i].position.z;
        p.lights[i].color_0rgb = f->lights[i].color_0rgb & 0x00FFFFFFu;
        p.lights[i].intensity = f->lights[i].intensity;
    }
    p.out = nullptr;
    p.hit_cube = nullptr;
    p.hit_sphere = nullptr;
    p.tree_hash = nullptr;
    p.counters = nullptr;
}

int ensure(rt_ctx* ctx, uint32_t** buf, size_t* cap, size_t need)
{
    if (*cap >= need) return RT_OK;
    if (*buf) HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)buf, need * sizeof(uint32_t)));
    *cap = need;
    return RT_OK;
}

// Camera part of the cube projection (classify_tiles on the device, rt_frame_row_extent on the host):
// the f32 basis of the frame in f64, its determinant, the padding.  False on anything irregular
// (non-finite camera or bounds, skewed basis): then nothing may be concluded from projections.
bool projection_setup(const Params& p, TileParams& t)
{
    std::memset(&t, 0, sizeof(t));
    bool regular = std::isfinite(p.bound_max);
    double mag = 1.0;
    for (int k = 0; k < 3; ++k) {
        t.eye[k] = p.eye[k]; t.right[k] = p.right[k]; t.up[k] = p.up[k]; t.fwd[k] = p.fwd[k];
        if (!std::isfinite(t.eye[k]) || !std::isfinite(t.right[k]) || !std::isfinite(t.up[k]) || !std::isfinite(t.fwd[k])) regular = false;
        mag = std::max(mag, std::fabs(t.eye[k]));
    }
    mag = std::max(mag, (double)p.bound_max);
    if (p.n_spheres != 0u && std::isfinite(p.sphere_bound_max)) mag = std::max(mag, (double)p.sphere_bound_max);
    const double* r = t.right; const double* u = t.up; const double* f = t.fwd;
    const double det = r[0] * (u[1] * f[2] - u[2] * f[1]) - r[1] * (u[0] * f[2] - u[2] * f[0]) + r[2] * (u[0] * f[1] - u[1] * f[0]);
    if (!(std::fabs(det) > 0.5 && std::fabs(det) < 2.0)) regular = false;  // an orthonormal basis has |det| = 1
    if (!regular) return false;
    t.inv_det = 1.0 / det;
    t.pad = mag * 0x1p-17;  // >= 64 ulp of the largest coordinate involved
    t.aspect = (double)((float)p.width / (float)p.height);
    t.width = p.width; t.height = p.height;
    return true;
}

// Per-launch preparation on the context's stream: the work queue (sized for the launch's tile grid, its
// header zeroed) and the tile stamps (classify_tiles).  On anything irregular (non-finite camera, skewed
// basis, non-finite bounds) p.tile_stamps stays null and every tile is traced.
int grow(rt_ctx* ctx, TileScratch& ts, uint32_t** buf, size_t* cap, size_t need, const char* what)
{
    if (*cap >= need) return RT_OK;
    if (!ts.growable)
        return fail(ctx, RT_ERR_LIMIT, std::string("rt_render_device_noalloc: the frame needs more ") + what +
                                           " than the fixed work buffer holds (RT_NOALLOC_MAX_TILES 8x8 tiles per frame)");
    if (*buf) HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)buf, need * sizeof(uint32_t)));
    *cap = need;
    HIP_TRY(ctx, hipMemsetAsync(*buf, 0, need * sizeof(uint32_t), ctx->stream));
    return RT_OK;
}

int prepare(rt_ctx* ctx, Params& p, TileScratch& ts, bool heavy_on = false, uint32_t heavy_go = 0u, bool* out_classified = nullptr,
            uint32_t* wf_header = nullptr)
{
    if (out_classified) *out_classified = false;
    p.tile_stamps = nullptr;
    p.sphere_stamps = nullptr;
    p.tiles_x = (p.width + 7u) / 8u;
    p.tiles_y = (p.height + 7u) / 8u;
    p.frame_id = 0u;
    // tile grid of this launch: the part's 16-row blocks, two tile rows each
    const uint32_t rows = p.y1 - p.y0;
    const uint32_t blocks = (rows + kRowBlock - 1u) / kRowBlock;
    const uint32_t my_blocks = (p.part < blocks) ? (blocks - p.part + p.n_parts - 1u) / p.n_parts : 0u;
    p.grid_tx = p.tiles_x;
    p.grid_ty = my_blocks * 2u;
    const size_t n_tiles = (size_t)p.grid_tx * p.grid_ty;
    const size_t qwords = kQueueBase + (kCostClasses + 1) * n_tiles;
    // fine stamps per 8x8 tile of the FRAME, then coarse stamps per block of 8x8 tiles
    // ... for the cubes, and the same two levels once more for the spheres
    const size_t plane_words = (size_t)p.tiles_x * p.tiles_y + (size_t)((p.tiles_x + 7u) / 8u) * ((p.tiles_y + 7u) / 8u);
    const size_t swords = kStampBase + 2 * plane_words;
    int rc;
    // a fresh buffer is all zero: both queue header banks, "unknown" tile costs, no stamps
    if ((rc = grow(ctx, ts, &ts.queue, &ts.queue_cap, qwords, "queue entries")) != RT_OK) return rc;
    if ((rc = grow(ctx, ts, &ts.cost, &ts.cost_cap, n_tiles + 1, "tile cost words")) != RT_OK) return rc;
    const size_t stamps_before = ts.stamps_cap;
    if ((rc = grow(ctx, ts, &ts.stamps, &ts.stamps_cap, swords, "tile stamps")) != RT_OK) return rc;
    if (ts.stamps_cap != stamps_before) ts.frame_id = 0u;
    if (ts.growable) {
        const size_t heavy_before = ts.heavy_cap, mark_before = ts.mark_cap;
        if ((rc = grow(ctx, ts, &ts.heavy, &ts.heavy_cap, 2 * (size_t)kHeavyCap, "heavy list entries")) != RT_OK) return rc;
        if ((rc = grow(ctx, ts, &ts.heavy_mark, &ts.mark_cap, n_tiles + 1, "heavy marks")) != RT_OK) return rc;
        if (ts.heavy_cap != heavy_before || ts.mark_cap != mark_before) ts.heavy_list_ready = false;
    }
    p.queue = ts.queue;
    // Header banks alternate: this launch counts into bank qsel, which the previous launch (or the fresh
    // buffer) left zeroed, and zeroes the other one for the next launch.
    p.qsel = (ts.launch_seq++) & 1u;
    p.tile_cost = ts.cost;
    // The fixed work buffer is shared by every context of its GPU and keeps the last call's contents: its
    // header bank is cleared here whatever the last call left (a call that failed between its launches, or
    // another context's launch that is still in flight: such calls race, see rtx950.h).
    if (!ts.growable) HIP_TRY(ctx, hipMemsetAsync(ts.queue + p.qsel * 16u, 0, 16 * sizeof(uint32_t), ctx->stream));

    TileParams t;
    if (!projection_setup(p, t)) {
        HIP_TRY(ctx, hipMemsetAsync(ts.queue + (p.qsel ^ 1u) * 16u, 0, 16 * sizeof(uint32_t), ctx->stream));
        if (wf_header) HIP_TRY(ctx, hipMemsetAsync(wf_header, 0, (kWfRedoBase + 16u) * sizeof(uint32_t), ctx->stream));  // what classify_tiles would have zeroed
        return RT_OK;
    }
    t.cube_filt = p.cube_filt;
    t.n_cubes = p.n_cubes;
    // spheres are classified when the cull can bound them (Params::sphere_ok); otherwise their pass is traced everywhere
    // (origin_bound is capped at 2^20: an eye beyond that is outside the bound sphere_grow was derived for - the kernels then
    // keep every sphere, ray_wild - so nothing is concluded here either)
    const bool classify_spheres = p.n_spheres != 0u && p.sphere_ok != 0u && std::isfinite(p.sphere_bound_max) && std::isfinite(p.sphere_grow) &&
                                  p.origin_bound < 0x1p20f;
    t.sphere_filt = p.sphere_filt;
    t.n_spheres = classify_spheres ? p.n_spheres : 0u;
    t.sphere_grow = (double)p.sphere_grow;
    t.sphere_stamps = ts.stamps + kStampBase + plane_words;
    t.tiles_x = p.tiles_x; t.tiles_y = p.tiles_y;
    t.queue = ts.queue;
    t.qsel = p.qsel;
    t.heavy_on = heavy_on ? 1u : 0u;
    t.heavy_go = heavy_go;
    t.wf_header = wf_header;
    if (++ts.frame_id == 0u) {  // wrapped: stale stamps could alias, start over
        HIP_TRY(ctx, hipMemsetAsync(ts.stamps, 0, ts.stamps_cap * sizeof(uint32_t), ctx->stream));
        ts.frame_id = 1u;
    }
    t.stamps = ts.stamps;
    t.frame_id = ts.frame_id;
    hipError_t e = (hipError_t)launch_classify(t, (void*)ctx->stream);  // also zeroes the next header bank
    if (e != hipSuccess) return fail_hip(ctx, e, "classify kernel launch");
    p.tile_stamps = ts.stamps;
    p.sphere_stamps = classify_spheres ? t.sphere_stamps : nullptr;
    p.frame_id = ts.frame_id;
    if (out_classified) *out_classified = true;
    return RT_OK;
}

// The launch that fills the work queue from the stamps that classify_tiles wrote.  (One fused launch
// that projects the cubes per block and tests tiles against the rectangles took 27 us for the 97-cube
// scene against 5.6 + 5.1 us for the two kernels, launch gap included: dropped.)
int listing(rt_ctx* ctx, const Params& p, int variant)
{
    hipError_t e = (hipError_t)launch_queue(p, variant, (void*)ctx->stream);
    if (e != hipSuccess) return fail_hip(ctx, e, "queue kernel launch");
    return RT_OK;
}

int launch_timed(rt_ctx* ctx, const Params& p, int variant)
{
    int rc = listing(ctx, p, variant);
    if (rc != RT_OK) return rc;
    hipError_t e = hipSuccess;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = ctx->timing && (ctx->timing_phase++ % ctx->timing_every == 0u);
    if (timed) {
        if (!ctx->event_pool.empty()) {
            ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(ctx, hipEventCreate(&ev.first));
            HIP_TRY(ctx, hipEventCreate(&ev.second));
        }
        HIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream));
    }
    e = (hipError_t)launch_render(p, variant, (void*)ctx->stream, ctx->max_workgroups);
    if (e != hipSuccess) return fail_hip(ctx, e, "render kernel launch");
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream));
        ctx->events.push_back(ev);
    }
    return RT_OK;
}


// Buffers of the wavefront path for a launch of n_tiles tiles.  Level 0 has two items (passes) per tile; a deeper
// level holds up to twice the tile count (every tile's rays in two bundles), within [4096, 2^19] - a launch whose
// items do not fit hands the tiles concerned to render_kernel (redo list), so the capacity bounds memory, not
// correctness.  4K: 1.75 GB, 8K: 3.9 GB per context, next to 288 GB.
int ensure_wavefront(rt_ctx* ctx, size_t n_tiles)
{
    WfScratch& f = ctx->wf;
    if (f.base && f.tiles >= n_tiles && f.cap_tune == ctx->tune_wf_cap) return RT_OK;
    if (f.base) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(f.base));
    }
    f = WfScratch{};
    const size_t tiles = std::max<size_t>(n_tiles, 64);
    uint32_t cap[kWfLevels];
    cap[0] = (uint32_t)(2 * tiles);
    uint32_t deep = ctx->tune_wf_cap ? ctx->tune_wf_cap : (uint32_t)std::min<size_t>(std::max<size_t>(2 * tiles, 4096), (size_t)1 << 19);
    deep = ((deep + kWfQueues - 1u) / kWfQueues) * kWfQueues;  // kWfQueues sub-queues of equal size
    for (uint32_t k = 1; k < kWfLevels; ++k) cap[k] = deep;
    const size_t pending_cap = (tiles + kWfQueues - 1) / kWfQueues + 1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t o_header = off; off = up(off + (kWfRedoBase + kQueueBase + tiles) * 4);
    const size_t o_pending = off; off = up(off + pending_cap * kWfQueues * 4);
    const size_t o_mark = off; off = up(off + tiles * 4);
    size_t o_items[kWfLevels], o_nodes[kWfLevels], o_rays[kWfLevels];
    for (uint32_t k = 0; k < kWfLevels; ++k) { o_items[k] = off; off = up(off + (size_t)cap[k] * sizeof(WfItem)); }
    const size_t zero_bytes = off;
    for (uint32_t k = 0; k < kWfLevels; ++k) { o_nodes[k] = off; off = up(off + (size_t)cap[k] * kWfNodeWords * 4); }
    for (uint32_t k = 1; k < kWfLevels; ++k) { o_rays[k] = off; off = up(off + (size_t)cap[k] * kWfRayWords * 4); }
    o_rays[0] = o_rays[1];
    HIP_TRY(ctx, hipMalloc((void**)&f.base, off));
    f.bytes = off;
    f.tiles = tiles;
    f.cap_tune = ctx->tune_wf_cap;
    f.zero_bytes = zero_bytes;
    HIP_TRY(ctx, hipMemsetAsync(f.base, 0, zero_bytes, ctx->stream));
    f.w.header = (uint32_t*)(f.base + o_header);
    f.w.pending = (uint32_t*)(f.base + o_pending);
    f.w.redo_mark = (uint32_t*)(f.base + o_mark);
    f.w.pending_cap = (uint32_t)pending_cap;
    for (uint32_t k = 0; k < kWfLevels; ++k) {
        f.w.items[k] = (WfItem*)(f.base + o_items[k]);
        f.w.nodes[k] = (uint32_t*)(f.base + o_nodes[k]);
        f.w.rays[k] = (float*)(f.base + o_rays[k]);
        f.w.cap[k] = cap[k];
    }
    return RT_OK;
}

// The launches of a wavefront frame behind classification and listing: level 0, the deeper levels the frame can
// have, the combine, and render_kernel over the redo list.  Bracketed by one event pair when the launch is timed.
int launch_wavefront(rt_ctx* ctx, const Params& p)
{
    int rc = listing(ctx, p, kVariantRender);
    if (rc != RT_OK) return rc;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = ctx->timing && (ctx->timing_phase++ % ctx->timing_every == 0u);
    if (timed) {
        if (!ctx->event_pool.empty()) {
            ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(ctx, hipEventCreate(&ev.first));
            HIP_TRY(ctx, hipEventCreate(&ev.second));
        }
        HIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream));
    }
    WfParams w = ctx->wf.w;
    w.seq = ctx->wf.seq;
    w.level = 0u;
    hipError_t e = (hipError_t)launch_wf_trace(p, w, (void*)ctx->stream, ctx->max_workgroups);
    if (e != hipSuccess) return fail_hip(ctx, e, "wavefront level-0 launch");
    // children exist only if some material reflects or refracts, a light is on and the depth limit allows them
    const bool deeper = ctx->scene_has_children && p.n_lights != 0u && p.max_depth != 0u;
    if (deeper) {
        for (uint32_t level = 1; level <= p.max_depth; ++level) {
            w.level = level;
            e = (hipError_t)launch_wf_trace(p, w, (void*)ctx->stream, ctx->max_workgroups);
            if (e != hipSuccess) return fail_hip(ctx, e, "wavefront level launch");
        }
    }
    // level 0 paints a tile itself only if the tile has one pass and no children: with spheres in the scene every
    // root waits for the merge of its two passes (render.rs:308-312), children or not
    if (deeper || p.n_spheres != 0u) {
        e = (hipError_t)launch_wf_combine(p, w, (void*)ctx->stream, ctx->max_workgroups);
        if (e != hipSuccess) return fail_hip(ctx, e, "wavefront combine launch");
    }
    if (deeper) {
        Params redo = p;  // the tiles whose items did not fit: render_kernel traces them whole (usually none)
        redo.queue = w.header + kWfRedoBase;
        redo.qsel = 0u;
        e = (hipError_t)launch_render(redo, kVariantRender, (void*)ctx->stream, 256u);
        if (e != hipSuccess) return fail_hip(ctx, e, "wavefront redo launch");
    }
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream));
        ctx->events.push_back(ev);
    }
    return RT_OK;
}

// Params::refine of a plain launch of `launch_tiles` tiles: which short-list kernel renders it.
//  1  the cull refinement (render_kernel<.., 2, ..>): where the one-radius cull leaves many candidates the box form of
//     the cull is applied as well.  It shortens the slowest tiles (1080p one frame at a time -15 %, 4K with the water
//     platform -14 %, 4K diorama -4 %, 8K -4 %); its code used to cost the common path 4 % in register allocation and was
//     kept to launches bound by their slowest tiles, until the kernels got 35 registers back (pow_f64_rounded).
//  2  the refinement and the walk by bundles (render_kernel<.., 3, ..>; cube_query's SPLIT == 2): a wave whose secondary
//     rays left in several bundles (a tile across the edge between two faces) and whose cull is loose is walked bundle
//     by bundle.  The slowest tiles' candidates drop several times (1080p 0.0570 -> 0.0479 ms with two frames in flight,
//     0.1106 -> 0.0874 one at a time; 4K with water one at a time 0.272 -> 0.215), and the keys' bookkeeping costs every
//     other tile 7 % (4K diorama 0.1333 -> 0.1429, 8K +9 %): for launches bound by their slowest tiles, i.e. at most
//     65 536 tiles (1080p; the parts of an N-GPU split of a 4K frame).
//  0  neither: with the side kernel, which takes the slowest tiles anyway (1080p one frame at a time: 0.0707 ms with
//     the plain kernel beside it, 0.0716 with the refinement, 0.0728 with bundles; before the register changes the
//     refinement cost more there, 0.099: it changes the tiles' one-wave costs against their three-wave costs, and
//     tiles then change sides from frame to frame); long lists.
uint32_t refine_choice(const rt_ctx* ctx, bool heavy_on, size_t launch_tiles)
{
    if (ctx->base.n_chunks != 0u) return 0u;
    if (ctx->tune_cull_refine >= 0) return (uint32_t)ctx->tune_cull_refine;
    if (heavy_on) return 0u;
    return launch_tiles <= kAutoHeavyMaxTiles ? 2u : 1u;
}

// The host has just waited for the context's stream.  If a launch since the last check used the side kernel, read
// the status word of the queue header back; non-zero = one of the device-side waits between the two streams (or
// inside a heavy tile's workgroup) gave up, and the frames rendered since the last check may be incomplete.  The
// side kernel is then taken out of service for this context, its stream drained, the word cleared.  Returns the
// word (0: all is well) through *out_word.
int check_heavy_status(rt_ctx* ctx, uint32_t* out_word)
{
    *out_word = 0u;
    if (!ctx->heavy_unchecked) return RT_OK;
    ctx->heavy_unchecked = false;
    if (!ctx->own.queue) return RT_OK;
    *ctx->h_status = 0u;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, ctx->own.queue + kQueueBase - 1u, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t word = *ctx->h_status;
    if (word == 0u) return RT_OK;
    ctx->heavy_disabled = true;
    ctx->own.heavy_list_ready = false;
    if (ctx->stream_heavy) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_heavy));  // whatever of the side kernel is still running
    HIP_TRY(ctx, hipMemsetAsync(ctx->own.queue + kQueueBase - 1u, 0, sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_word = word;
    return RT_OK;
}

const char* heavy_wait_name(uint32_t word)
{
    switch (word) {
    case 1u: return "a tracer wave's wait for its shader waves";
    case 2u: return "a shader wave's wait for its tracer";
    case 3u: return "the join of the side kernel with the context's stream";
    case 4u: return "the side kernel's start gate";
    default: return "a device-side wait";
    }
}

}  // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RTX950_ABI_VERSION; }

int rt_ctx_create(int device, rt_ctx** out_ctx)
{
    if (!out_ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(nullptr, RT_ERR_NO_DEVICE, "device ordinal out of range");
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return fail_hip(nullptr, guard.status, "hipSetDevice");
    rt_ctx* c = new rt_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail_hip(nullptr, e, "hipStreamCreateWithFlags");
    }
    e = hipMalloc((void**)&c->d_counters, kCounterBytes);
    if (e == hipSuccess) {
        e = hipHostMalloc((void**)&c->h_status, 64, hipHostMallocDefault);
        if (e != hipSuccess) (void)hipFree(c->d_counters);
    }
    if (e != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail_hip(nullptr, e, "hipMalloc(counters)");
    }
    // The fixed work buffer of rt_render_device_noalloc: this device's copy of `scratch`, looked up once per
    // device with the device current (in host code the array's name is not a device address).
    if (device < kMaxDevices) {
        DeviceScratch& ds = g_dev_scratch[device];
        std::lock_guard<std::mutex> lock(ds.mu);
        if (!ds.ready) {
            uint32_t* base = nullptr;
            e = hipGetSymbolAddress((void**)&base, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !base) {
                (void)hipFree(c->d_counters);
                (void)hipHostFree(c->h_status);
                (void)hipStreamDestroy(c->stream);
                delete c;
                return fail_hip(nullptr, e, "hipGetSymbolAddress(scratch)");
            }
            ds.ts.queue = base;
            ds.ts.queue_cap = kNaQueueWords;
            ds.ts.stamps = base + kNaQueueWords;
            ds.ts.stamps_cap = kNaStampWords;
            ds.ts.cost = base + kNaQueueWords + kNaStampWords;
            ds.ts.cost_cap = kNaCostWords;
            ds.ts.growable = false;
            ds.ready = true;
        }
    }
    *out_ctx = c;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx)
{
    if (!ctx) return RT_OK;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_heavy) (void)hipStreamSynchronize(ctx->stream_heavy);
    free_scene(ctx);
    (void)hipFree(ctx->d_fb);
    for (auto& b : ctx->d_dbg) (void)hipFree(b);
    (void)hipFree(ctx->d_counters);
    (void)hipHostFree(ctx->h_status);
    (void)hipFree(ctx->own.stamps);
    (void)hipFree(ctx->own.queue);
    (void)hipFree(ctx->own.cost);
    (void)hipFree(ctx->own.heavy);
    (void)hipFree(ctx->own.heavy_mark);
    (void)hipFree(ctx->wf.base);
    (void)hipFree(ctx->d_part);
    if (ctx->part_done) (void)hipEventDestroy(ctx->part_done);
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : ctx->event_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->heavy_fork) (void)hipEventDestroy(ctx->heavy_fork);
    if (ctx->heavy_join) (void)hipEventDestroy(ctx->heavy_join);
    if (ctx->stream_heavy) (void)hipStreamDestroy(ctx->stream_heavy);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_tls_err.c_str(); }

int rt_scene_upload(rt_ctx* ctx, const rt_cube* cubes, size_t n_cubes, const rt_sphere* spheres, size_t n_spheres,
                    const rt_material* materials, size_t n_materials, const uint8_t* atlas_rgb8, size_t n_textures,
                    uint32_t tex_w, uint32_t tex_h, uint32_t sprite)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    if ((n_cubes && !cubes) || (n_spheres && !spheres)) return fail(ctx, RT_ERR_INVALID_ARG, "object array is null");
    if (!materials || n_materials == 0) return fail(ctx, RT_ERR_INVALID_ARG, "material table is empty");
    // 2^26 cubes: the BVH walk addresses a pair of sibling boxes (64 bytes) by a 32-bit byte offset (cube_query_bvh)
    if (n_cubes > 0x04000000u || n_spheres > 0x7FFFFFFFu) return fail(ctx, RT_ERR_UNSUPPORTED, "too many objects (at most 2^26 cubes, 2^31 - 1 spheres)");
    if (n_textures && (!atlas_rgb8 || tex_w == 0 || tex_h == 0 || sprite == 0)) return fail(ctx, RT_ERR_INVALID_ARG, "bad atlas description");
    if (n_textures && ((uint64_t)n_textures * tex_w * tex_h > 0x3FFFFFFFull)) return fail(ctx, RT_ERR_UNSUPPORTED, "atlas too large");
    for (size_t i = 0; i < n_materials; ++i) {
        if (materials[i].texture != RT_TEXTURE_NONE && (materials[i].texture < 0 || (size_t)materials[i].texture >= n_textures))
            return fail(ctx, RT_ERR_INVALID_ARG, "material references a texture outside the atlas");
    }
    for (size_t i = 0; i < n_cubes; ++i)
        if (cubes[i].material >= n_materials) return fail(ctx, RT_ERR_INVALID_ARG, "cube references a material outside the table");
    for (size_t i = 0; i < n_spheres; ++i)
        if (spheres[i].material >= n_materials) return fail(ctx, RT_ERR_INVALID_ARG, "sphere references a material outside the table");

    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);

    std::vector<CubeRec> hc(n_cubes);
    std::vector<uint32_t> hcm(n_cubes);
    float bmax = 0.0f;
    bool fast_ok = true;
    auto check_bound = [&](float b) {
        float a = std::fabs(b);
        if (!(a == 0.0f || (a >= 0x1p-30f && a <= 0x1p20f))) fast_ok = false;  // also catches NaN/inf
        if (a > bmax) bmax = a;
    };
    for (size_t i = 0; i < n_cubes; ++i) {
        const float half = cubes[i].size / 2.0f;  // cube.rs:52
        CubeRec& r = hc[i];
        r.minx = cubes[i].center.x - half; r.maxx = cubes[i].center.x + half;
        r.miny = cubes[i].center.y - half; r.maxy = cubes[i].center.y + half;
        r.minz = cubes[i].center.z - half; r.maxz = cubes[i].center.z + half;
        r.id = cubes[i].id;
        r.size = cubes[i].size;
        hcm[i] = cubes[i].material;
        check_bound(r.minx); check_bound(r.maxx); check_bound(r.miny);
        check_bound(r.maxy); check_bound(r.minz); check_bound(r.maxz);
    }
    // Cull stream, padded to a multiple of 64 records (one per lane) with far-away zero-size boxes.  A
    // short list is streamed in list order.  A long one is streamed in Morton order of the cube centres,
    // with the list index of every record (cube_perm) and a bounding box per chunk of 64 (chunk_box).
    const size_t n_filt = ((n_cubes + 63) / 64) * 64 + 64;
    const bool large = n_cubes > kSmallListMax;
    std::vector<uint32_t> perm(n_filt, 0u);
    for (size_t i = 0; i < n_cubes; ++i) perm[i] = (uint32_t)i;
    if (large) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = 0; i < n_cubes; ++i) {
            const float c[3] = {cubes[i].center.x, cubes[i].center.y, cubes[i].center.z};
            for (int k = 0; k < 3; ++k)
                if (std::isfinite(c[k])) { lo[k] = std::min(lo[k], (double)c[k]); hi[k] = std::max(hi[k], (double)c[k]); }
        }
        auto spread = [](uint32_t v) {  // 10 bits -> every third bit
            v &= 0x3FFu;
            v = (v | (v << 16)) & 0x030000FFu;
            v = (v | (v << 8)) & 0x0300F00Fu;
            v = (v | (v << 4)) & 0x030C30C3u;
            v = (v | (v << 2)) & 0x09249249u;
            return v;
        };
        std::vector<uint32_t> code(n_cubes, 0u);
        for (size_t i = 0; i < n_cubes; ++i) {
            const float c[3] = {cubes[i].center.x, cubes[i].center.y, cubes[i].center.z};
            uint32_t q[3] = {0u, 0u, 0u};
            for (int k = 0; k < 3; ++k) {
                if (!std::isfinite(c[k]) || !(hi[k] > lo[k])) continue;
                q[k] = (uint32_t)std::min(1023.0, std::max(0.0, ((double)c[k] - lo[k]) / (hi[k] - lo[k]) * 1023.0));
            }
            code[i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        }
        std::stable_sort(perm.begin(), perm.begin() + n_cubes, [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    }
    std::vector<uint32_t> hid(n_filt, 0u);
    std::vector<float> hf(n_filt * 4);
    for (size_t j = 0; j < n_filt; ++j) {
        hf[4 * j + 0] = 1e18f; hf[4 * j + 1] = -1e18f; hf[4 * j + 2] = 1e18f; hf[4 * j + 3] = 0.0f;
    }
    for (size_t j = 0; j < n_cubes; ++j) {
        const rt_cube& c = cubes[perm[j]];
        hf[4 * j + 0] = c.center.x; hf[4 * j + 1] = c.center.y; hf[4 * j + 2] = c.center.z;
        hf[4 * j + 3] = std::fabs(c.size / 2.0f);  // min > max (negative size) is the same box to the slab test (lib.rs:29-35 sorts the two t)
        hid[j] = c.id;
    }
    ctx->h_cube_box.resize(n_cubes * 4);
    for (size_t i = 0; i < n_cubes; ++i) {
        ctx->h_cube_box[4 * i + 0] = cubes[i].center.x; ctx->h_cube_box[4 * i + 1] = cubes[i].center.y;
        ctx->h_cube_box[4 * i + 2] = cubes[i].center.z; ctx->h_cube_box[4 * i + 3] = std::fabs(cubes[i].size / 2.0f);
    }
    ctx->h_n_spheres = n_spheres;
    ctx->extent_valid = false;
    const size_t n_chunks = large ? (n_cubes + 63) / 64 : 0;
    const size_t n_box = ((n_chunks + 63) / 64) * 64 + 64;
    // bounding box per chunk of 64 consecutive records of a stream given by its list indices; returns the
    // sum of the boxes' half-extent sums (a measure of how compact the stream's chunks are)
    auto chunk_boxes = [&](const uint32_t* order, std::vector<float>& out) {
        double total = 0.0;
        out.assign(n_box * 8, 0.0f);
        for (size_t c = 0; c < n_chunks; ++c) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            for (size_t j = c * 64; j < std::min(n_cubes, c * 64 + 64); ++j) {
                const CubeRec& r = hc[order[j]];
                const float b[3][2] = {{r.minx, r.maxx}, {r.miny, r.maxy}, {r.minz, r.maxz}};
                for (int k = 0; k < 3; ++k)
                    for (int e = 0; e < 2; ++e) {
                        if (!std::isfinite(b[k][e])) finite = false;
                        lo[k] = std::min(lo[k], (double)b[k][e]); hi[k] = std::max(hi[k], (double)b[k][e]);
                    }
            }
            for (int k = 0; k < 3; ++k) {
                // centre rounded to f32, half extent rounded up so that centre +- half still covers [lo, hi]
                const float ctr = finite ? (float)((lo[k] + hi[k]) * 0.5) : 0.0f;
                float half = finite ? (float)std::max(hi[k] - (double)ctr, (double)ctr - lo[k]) : INFINITY;
                if (finite) half = std::nextafter(half, INFINITY);
                out[8 * c + k] = ctr;
                out[8 * c + 4 + k] = half;
                total += half;
            }
            // lowest list index in the chunk (shadow queries want the lowest index that passes)
            uint32_t first = 0xFFFFFFFFu;
            for (size_t j = c * 64; j < std::min(n_cubes, c * 64 + 64); ++j) first = std::min(first, order[j]);
            std::memcpy(&out[8 * c + 3], &first, sizeof(first));
        }
        return total;
    };
    std::vector<float> hbox;
    const double morton_extent = chunk_boxes(perm.data(), hbox);
    // Shadow queries want the LOWEST list index that passes (render.rs:44-70 stops at the first), so a
    // stream in list order lets them stop at the first chunk that answers every ray - if consecutive list
    // entries are neighbours in space (a generated world is; a shuffled list is not), i.e. if the chunks
    // of the list-ordered stream are not much larger than those of the Morton-ordered one.
    std::vector<float> hf_list, hbox_list;
    std::vector<uint32_t> hid_list;
    bool shadow_list = false;
    if (large) {
        std::vector<uint32_t> ident(n_filt, 0u);
        for (size_t i = 0; i < n_cubes; ++i) ident[i] = (uint32_t)i;
        const double list_extent = chunk_boxes(ident.data(), hbox_list);
        shadow_list = std::isfinite(list_extent) && std::isfinite(morton_extent) && list_extent <= 3.0 * morton_extent;
        if (ctx->tune_shadow_list >= 0) shadow_list = ctx->tune_shadow_list != 0;  // rt_set_tuning: tests force either stream
        if (shadow_list) {
            hf_list.resize(n_filt * 4);
            hid_list.assign(n_filt, 0u);
            for (size_t j = 0; j < n_filt; ++j) {
                hf_list[4 * j + 0] = 1e18f; hf_list[4 * j + 1] = -1e18f; hf_list[4 * j + 2] = 1e18f; hf_list[4 * j + 3] = 0.0f;
            }
            for (size_t j = 0; j < n_cubes; ++j) {
                hf_list[4 * j + 0] = cubes[j].center.x; hf_list[4 * j + 1] = cubes[j].center.y; hf_list[4 * j + 2] = cubes[j].center.z;
                hf_list[4 * j + 3] = std::fabs(cubes[j].size / 2.0f);
                hid_list[j] = cubes[j].id;
            }
        } else {
            hbox_list.clear();
        }
    }
    // BVH for the incoherent bundles of a long list, over the stream records (already in Morton order)
    std::vector<BvhNode> nodes;
    uint32_t level = 0u, bvh_depth = 0u;  // pairs on the longest root-to-leaf path = deferred siblings a walk can hold
    if (large) {
        auto fill = [&](size_t lo_i, size_t hi_i, BvhNode& nd) {  // box, and the leaf or inner word; link is set by the caller
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            uint32_t min_idx = 0xFFFFFFFFu;  // lowest list index below this node (shadow pruning and ordering)
            for (size_t j = lo_i; j < hi_i; ++j) {
                const CubeRec& r = hc[perm[j]];
                min_idx = std::min(min_idx, perm[j]);
                const float b[3][2] = {{r.minx, r.maxx}, {r.miny, r.maxy}, {r.minz, r.maxz}};
                for (int k = 0; k < 3; ++k)
                    for (int e = 0; e < 2; ++e) {
                        if (!std::isfinite(b[k][e])) finite = false;
                        lo[k] = std::min(lo[k], (double)b[k][e]); hi[k] = std::max(hi[k], (double)b[k][e]);
                    }
            }
            float c[3], h[3];
            for (int k = 0; k < 3; ++k) {
                c[k] = finite ? (float)((lo[k] + hi[k]) * 0.5) : 0.0f;
                h[k] = finite ? std::nextafter((float)std::max(hi[k] - (double)c[k], (double)c[k] - lo[k]), INFINITY) : INFINITY;
            }
            nd.cx = c[0]; nd.cy = c[1]; nd.cz = c[2]; nd.hx = h[0]; nd.hy = h[1]; nd.hz = h[2];
            nd.link = 0u;
            nd.first_count = min_idx << 3;
            if (hi_i - lo_i == 1) {  // a leaf is its cube: the cull stream's record (centre, |half size|), list index, id
                nd.cx = hf[4 * lo_i + 0]; nd.cy = hf[4 * lo_i + 1]; nd.cz = hf[4 * lo_i + 2];
                nd.hx = nd.hy = nd.hz = hf[4 * lo_i + 3];
                nd.first_count = (perm[lo_i] << 3) | 1u;
                nd.link = hid[lo_i];
            }
        };
        // pair p holds the two halves of a range; depth is about log2(n), one deferred sibling per level at most
        std::function<uint32_t(size_t, size_t)> build_pair = [&](size_t lo_i, size_t hi_i) -> uint32_t {
            ++level;
            bvh_depth = std::max(bvh_depth, level);
            const size_t p = nodes.size() / 2;
            nodes.push_back(BvhNode{});
            nodes.push_back(BvhNode{});
            const size_t mid = (lo_i + hi_i) / 2;
            const size_t range[2][2] = {{lo_i, mid}, {mid, hi_i}};
            for (int side = 0; side < 2; ++side) {
                BvhNode nd;
                fill(range[side][0], range[side][1], nd);
                if (range[side][1] - range[side][0] > 1) nd.link = build_pair(range[side][0], range[side][1]);
                nodes[2 * p + side] = nd;  // after the recursion: the vector may have moved
            }
            --level;
            return (uint32_t)p;
        };
        build_pair(0, n_cubes);  // n_cubes > kSmallListMax >= 2: both halves are non-empty
    }
    double half_sum = 0.0;
    size_t half_n = 0;
    for (size_t i = 0; i < n_cubes; ++i) {
        const float h = std::fabs(cubes[i].size / 2.0f);
        if (std::isfinite(h)) { half_sum += h; ++half_n; }
    }
    std::vector<SphereRec> hs(n_spheres);
    std::vector<SphereMeta> hsm(n_spheres);
    const size_t n_sfilt = ((n_spheres + 63) / 64) * 64 + 64;
    std::vector<float> hsf(n_sfilt * 4);
    for (size_t i = 0; i < n_sfilt; ++i) {
        hsf[4 * i + 0] = 1e18f; hsf[4 * i + 1] = -1e18f; hsf[4 * i + 2] = 1e18f; hsf[4 * i + 3] = 1.0f;
    }
    float sbmax = 0.0f;
    bool sphere_ok = true;
    for (size_t i = 0; i < n_spheres; ++i) {
        hs[i] = SphereRec{spheres[i].center.x, spheres[i].center.y, spheres[i].center.z, spheres[i].radius};
        hsm[i] = SphereMeta{spheres[i].id, spheres[i].material};
        const float v[4] = {spheres[i].center.x, spheres[i].center.y, spheres[i].center.z, std::fabs(spheres[i].radius)};
        for (int k = 0; k < 4; ++k) {
            hsf[4 * i + k] = v[k];
            if (!(std::fabs(v[k]) <= 0x1p20f)) sphere_ok = false;  // also NaN
        }
        for (int k = 0; k < 3; ++k) sbmax = std::max(sbmax, std::fabs(v[k]) + v[3]);
    }
    if (!std::isfinite(sbmax)) { sbmax = 0.0f; sphere_ok = false; }
    std::vector<MatRec> hm(n_materials);
    for (size_t i = 0; i < n_materials; ++i) {
        hm[i] = MatRec{materials[i].specular, materials[i].albedo0, materials[i].albedo1, materials[i].reflectivity,
                       materials[i].transparency, materials[i].refractive_index, materials[i].diffuse_0rgb & 0x00FFFFFFu,
                       materials[i].texture};
    }
    const size_t texels = n_textures * (size_t)tex_w * tex_h;
    std::vector<uint32_t> ha(texels);
    for (size_t t = 0; t < texels; ++t)
        ha[t] = ((uint32_t)atlas_rgb8[3 * t] << 16) | ((uint32_t)atlas_rgb8[3 * t + 1] << 8) | (uint32_t)atlas_rgb8[3 * t + 2];

    int rc;
    if ((rc = upload(ctx, &ctx->d_cubes, hc)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_mat, hcm)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_filt, hf)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_id, hid)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_perm, perm)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_chunk_box, hbox)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_filt_list, hf_list)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_cube_id_list, hid_list)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_chunk_box_list, hbox_list)) != RT_OK) return rc;
    std::vector<BvhPair> pairs(nodes.size() / 2);
    for (size_t p = 0; p < pairs.size(); ++p)
        for (int side = 0; side < 2; ++side) {
            const BvhNode& nd = nodes[2 * p + side];
            BvhPair& pr = pairs[p];
            float* xy = side ? pr.b_xy : pr.a_xy;
            float* hxy = side ? pr.b_hxy : pr.a_hxy;
            xy[0] = nd.cx; xy[1] = nd.cy; hxy[0] = nd.hx; hxy[1] = nd.hy;
            pr.cz[side] = nd.cz; pr.hz[side] = nd.hz;
            pr.first_count[side] = nd.first_count; pr.link[side] = nd.link;
        }
    if ((rc = upload(ctx, &ctx->d_bvh, pairs)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_spheres, hs)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_sphere_meta, hsm)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_sphere_filt, hsf)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_mats, hm)) != RT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_atlas, ha)) != RT_OK) return rc;

    Params& b = ctx->base;
    std::memset(&b, 0, sizeof(b));
    b.cubes = ctx->d_cubes;
    b.cube_mat = ctx->d_cube_mat;
    b.cube_filt = ctx->d_cube_filt;
    b.cube_id = ctx->d_cube_id;
    b.cube_perm = ctx->d_cube_perm;
    b.chunk_box = ctx->d_chunk_box;
    b.cube_filt_list = ctx->d_cube_filt_list;
    b.cube_id_list = ctx->d_cube_id_list;
    b.chunk_box_list = ctx->d_chunk_box_list;
    b.shadow_list = shadow_list ? 1u : 0u;
    b.n_chunks = (uint32_t)n_chunks;
    b.bvh = ctx->d_bvh;
    b.n_bvh_nodes = (uint32_t)nodes.size();
    b.coherent_thr = half_n ? (float)(8.0 * half_sum / (double)half_n) : 0.0f;  // 8 average half sizes (swept with the ordered walk: 4..16 within 10 %)
    b.split_thr = half_n ? (float)(0.5 * half_sum / (double)half_n) : 0.0f;  // heavy tiles: half an average half size of inflation per cube
    b.bvh_levels = std::min(std::max(bvh_depth, 1u), kBvhStackMax);
    if (bvh_depth > ctx->tune_bvh_depth_limit) b.coherent_thr = INFINITY;  // a list too long for the walk's per-lane stack: every wave takes the chunk traversal
    if (std::isfinite(b.coherent_thr)) b.coherent_thr *= ctx->tune_coherent_scale;  // rt_set_tuning: 1 unless a test forces one traversal
    b.spheres = ctx->d_spheres;
    b.sphere_meta = ctx->d_sphere_meta;
    b.sphere_filt = ctx->d_sphere_filt;
    b.sphere_bound_max = sbmax;
    b.sphere_ok = sphere_ok ? 1u : 0u;
    b.mats = ctx->d_mats;
    // a material whose direct light is exactly black (render kernel, trace_tree): anything not provably above
    // the kernel's threshold selects the kernel variant that tests for it per lane
    b.dark_materials = 0u;
    for (size_t i = 0; i < n_materials; ++i)
        if (!(1.0f - materials[i].reflectivity - materials[i].transparency >= 0.01f)) b.dark_materials = 1u;
    if (ctx->tune_dark_kernel >= 0) b.dark_materials = ctx->tune_dark_kernel ? 1u : 0u;  // rt_set_tuning: either kernel for any scene
    ctx->scene_has_children = false;  // cast_ray recurses where reflectivity > 0 or transparency > 0 (render.rs:154,172)
    for (size_t i = 0; i < n_materials; ++i)
        if (materials[i].reflectivity > 0.0f || materials[i].transparency > 0.0f) ctx->scene_has_children = true;
    b.atlas = ctx->d_atlas;
    b.n_cubes = (uint32_t)n_cubes;
    b.n_spheres = (uint32_t)n_spheres;
    b.tex_w = n_textures ? tex_w : 1u;
    b.tex_h = n_textures ? tex_h : 1u;
    b.sprite = (float)sprite;  // `sprite_size as f32`, texture.rs:179
    b.sprite_x0 = b.sprite * 0.0f;
    b.sprite_x2 = b.sprite * 2.0f;
    b.sprite_x3 = b.sprite * 3.0f;
    b.bound_max = bmax;
    b.fast_ok = fast_ok ? 1u : 0u;
    ctx->has_scene = true;
    return RT_OK;
}

uint32_t rt_part_rows(uint32_t rows, uint32_t n_parts, uint32_t part)
{
    if (n_parts == 0u || part >= n_parts) return 0u;
    const uint32_t blocks = (rows + kRowBlock - 1u) / kRowBlock;
    uint32_t total = 0u;
    for (uint32_t b = part; b < blocks; b += n_parts) {
        const uint32_t first = b * kRowBlock;
        total += (rows - first < kRowBlock) ? rows - first : kRowBlock;
    }
    return total;
}

static int render_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, uint32_t* d_out_0rgb,
                       const rt_debug_planes* dbg_device, bool synchronous);

int rt_render_device(rt_ctx* ctx, const rt_frame* frame, uint32_t* d_out_0rgb, const rt_debug_planes* dbg_device)
{
    return render_part(ctx, frame, 1u, 0u, d_out_0rgb, dbg_device, false);
}

int rt_render_device_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, uint32_t* d_out_0rgb,
                          const rt_debug_planes* dbg_device)
{
    return render_part(ctx, frame, n_parts, part, d_out_0rgb, dbg_device, false);
}

static int render_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, uint32_t* d_out_0rgb,
                       const rt_debug_planes* dbg_device, bool synchronous)
{
    int rc = check_frame(ctx, frame);
    if (rc != RT_OK) return rc;
    if (!d_out_0rgb) return fail(ctx, RT_ERR_INVALID_ARG, "output pointer is null");
    if (n_parts == 0u || part >= n_parts) return fail(ctx, RT_ERR_INVALID_ARG, "need 0 <= part < n_parts");
    ENTER_DEVICE(ctx);
    Params p;
    fill_frame(ctx, frame, p);
    p.n_parts = n_parts;
    p.part = part;
    p.out = d_out_0rgb;
#ifdef RTX_PROFILE
    p.counters = ctx->d_counters;
#endif
    int variant = kVariantRender;
    if (dbg_device && (dbg_device->hit_cube || dbg_device->hit_sphere || dbg_device->tree_hash)) {
        p.hit_cube = dbg_device->hit_cube;
        p.hit_sphere = dbg_device->hit_sphere;
        p.tree_hash = dbg_device->tree_hash;
        variant = kVariantDebug;
    }
    // Heavy tiles: a plain frame whose untraced tiles build_queue paints (no fill that could overwrite the
    // side kernel's pixels) hands the tiles on the list that the previous launch wrote - same tile grid, next
    // sequence number - to render_heavy_kernel on the second stream, and writes the list for the next launch.
    TileScratch& ts = ctx->own;
    const bool paints = ((uintptr_t)d_out_0rgb & 15u) == 0u && (frame->width & 3u) == 0u && !ctx->tune_fill_memset;
    const size_t launch_tiles = (size_t)((frame->width + 7u) / 8u) * ((rt_part_rows(frame->y1 - frame->y0, n_parts, part) + 7u) / 8u);
    const uint32_t heavy_class = ctx->tune_heavy_class >= 0 ? (uint32_t)ctx->tune_heavy_class
                                 : ((synchronous && launch_tiles <= kAutoHeavyMaxTiles && ctx->base.n_chunks == 0u) ? kDefaultHeavyClass : kCostClasses);
    // The wavefront path (plain frames): a launch per ray-tree level over bundles of coherent rays instead of the
    // tile-resident walk; explicit for now (RT_TUNE_WAVEFRONT).
    const bool wavefront = variant == kVariantRender && ctx->tune_wavefront == 1;
    if (wavefront) {
        // the launch's tile grid (prepare): two tile rows per 16-row block of the part, a ragged last block included
        const size_t wf_tiles = (size_t)((frame->width + 7u) / 8u) * 2u * ((rt_part_rows(frame->y1 - frame->y0, n_parts, part) + 15u) / 16u + 1u);
        if ((rc = ensure_wavefront(ctx, wf_tiles)) != RT_OK) return rc;
        if (++ctx->wf.seq == 0u) {  // wrapped: stale stamps could alias
            HIP_TRY(ctx, hipMemsetAsync(ctx->wf.base, 0, ctx->wf.zero_bytes, ctx->stream));
            ctx->wf.seq = 1u;
        }
    }
    const bool heavy_on = variant == kVariantRender && paints && heavy_class < kCostClasses && !ctx->heavy_disabled && !wavefront;
    const uint32_t seq = ts.launch_seq;  // this launch's; prepare() increments it
    const uint32_t geom[6] = {frame->width, frame->height, frame->y0, frame->y1, n_parts, part};
    if (heavy_on && !ctx->stream_heavy) {
        // created on first use: a context that never takes this path keeps to one stream (a process has few
        // hardware queues, and streams beyond them share).  Default priority on purpose: on a high-priority
        // stream the one-wave gate, polling for a store that a kernel on a normal-priority stream has yet to
        // make, kept that kernel from being dispatched once three contexts had side streams (priority
        // inversion: every frame waited until the gate gave up), and it gained nothing with one or two.
        // (the events first: the stream's existence is what says that all three are there)
        if (!ctx->heavy_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->heavy_fork, hipEventDisableTiming));
        if (!ctx->heavy_join) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->heavy_join, hipEventDisableTiming));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream_heavy, hipStreamNonBlocking));
        // a rocprofiler-sdk tool in the process (rocprofv3, rocprof-sys) exports rocprofiler_configure; the older
        // profilers are loaded by the HSA runtime through HSA_TOOLS_LIB
        const char* tools = std::getenv("HSA_TOOLS_LIB");
        ctx->heavy_events = dlsym(RTLD_DEFAULT, "rocprofiler_configure") != nullptr || (tools && *tools);
    }
    const bool by_events = ctx->tune_heavy_sync >= 0 ? ctx->tune_heavy_sync == 0 : ctx->heavy_events;
    bool heavy_now = heavy_on && ts.heavy_list_ready && ts.heavy_list_seq == seq && std::memcmp(ts.heavy_geom, geom, sizeof(geom)) == 0 &&
                     seq + 1u != 0u;
    // classify_tiles of this launch tells the side kernel when to start (kHeavyGoWord): no event between the
    // streams - unless events it must be (see rt_ctx::heavy_events): then everything enqueued on the context's
    // stream so far comes before the side kernel, this launch's own kernels do not
    if (heavy_now && by_events) {
        HIP_TRY(ctx, hipEventRecord(ctx->heavy_fork, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_heavy, ctx->heavy_fork, 0));
    }
    bool classified = false;
    if ((rc = prepare(ctx, p, ts, heavy_on, heavy_now ? seq + 1u : 0u, &classified, wavefront ? ctx->wf.w.header : nullptr)) != RT_OK) return rc;
    // a buffer that had to grow took the list with it (no mark can match then: the fresh marks are zero), and
    // without classify_tiles (irregular camera) nobody would say go
    heavy_now = heavy_now && ts.heavy_list_ready && (classified || by_events);
    ts.heavy_list_ready = false;
    if (heavy_on) {
        p.heavy = ts.heavy;
        p.heavy_mark = ts.heavy_mark;
        p.heavy_class = heavy_class;
        p.split_thr = ctx->base.split_thr;
        p.heavy_spin_limit = ctx->tune_heavy_spin;
        p.heavy_now = heavy_now ? seq + 1u : 0u;
        p.heavy_next = (seq + 2u != 0u) ? seq + 2u : 0u;
        if (p.heavy_next != 0u) {
            ts.heavy_list_ready = true;
            ts.heavy_list_seq = seq + 1u;
            std::memcpy(ts.heavy_geom, geom, sizeof(geom));
        }
    }
    if (variant == kVariantRender) {
        // Three quarters of a typical frame are sky.  Those tiles are painted by build_queue, which has just
        // found that they need no tracing (16-byte stores: 25 MB of the 4K frame in 3 us), instead of 16-tile
        // paint groups at the end of the render kernel (whose plain variant has no painting code: the variants
        // with debug planes or counters still paint and count).  A buffer or width that 16-byte stores do not
        // fit is filled by the runtime ahead of the launches instead (33 MB at 4K in 20 us).
        const size_t npx = (size_t)rt_part_rows(frame->y1 - frame->y0, n_parts, part) * frame->width;
        if (((uintptr_t)d_out_0rgb & 15u) == 0u && (frame->width & 3u) == 0u && !ctx->tune_fill_memset) {
            p.prefilled = 2u;  // build_queue paints the tiles it does not list (16-byte stores)
        } else {
            if (npx) HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_out_0rgb, (int)p.sky, npx, ctx->stream));
            p.prefilled = 1u;
        }
    }
    p.refine = refine_choice(ctx, heavy_on, launch_tiles);
    if (variant == kVariantRender) ctx->last_refine = p.refine;
    if (wavefront) return launch_wavefront(ctx, p);
    if ((rc = launch_timed(ctx, p, variant)) != RT_OK) return rc;
    if (heavy_now) {
        ctx->heavy_unchecked = true;
        // after the stream's own launches, so that the host's time for these calls is not ahead of them
        hipError_t e = (hipError_t)launch_heavy(p, (void*)ctx->stream_heavy, !by_events);
        if (e != hipSuccess) return fail_hip(ctx, e, "heavy kernel launch");
        // the frame is whole when the stream gets past this
        if (by_events) {
            HIP_TRY(ctx, hipEventRecord(ctx->heavy_join, ctx->stream_heavy));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->heavy_join, 0));
        } else {
            e = (hipError_t)launch_heavy_join(p, (void*)ctx->stream);
            if (e != hipSuccess) return fail_hip(ctx, e, "heavy join launch");
        }
    }
    return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_frame* frame, uint32_t* out_0rgb, const rt_debug_planes* dbg_host)
{
    int rc = check_frame(ctx, frame);
    if (rc != RT_OK) return rc;
    if (!out_0rgb) return fail(ctx, RT_ERR_INVALID_ARG, "output pointer is null");
    ENTER_DEVICE(ctx);
    const size_t npx = (size_t)(frame->y1 - frame->y0) * frame->width;
    if ((rc = ensure(ctx, &ctx->d_fb, &ctx->fb_capacity, npx)) != RT_OK) return rc;
    rt_debug_planes dev{nullptr, nullptr, nullptr};
    uint32_t* host_planes[3] = {nullptr, nullptr, nullptr};
    if (dbg_host) {
        host_planes[0] = dbg_host->hit_cube;
        host_planes[1] = dbg_host->hit_sphere;
        host_planes[2] = dbg_host->tree_hash;
        for (int k = 0; k < 3; ++k) {
            if (!host_planes[k]) continue;
            if ((rc = ensure(ctx, &ctx->d_dbg[k], &ctx->dbg_capacity[k], npx)) != RT_OK) return rc;
        }
        dev.hit_cube = host_planes[0] ? ctx->d_dbg[0] : nullptr;
        dev.hit_sphere = host_planes[1] ? ctx->d_dbg[1] : nullptr;
        dev.tree_hash = host_planes[2] ? ctx->d_dbg[2] : nullptr;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = render_part(ctx, frame, 1u, 0u, ctx->d_fb, &dev, true)) != RT_OK) return rc;
        // the single read-back
        HIP_TRY(ctx, hipMemcpyAsync(out_0rgb, ctx->d_fb, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        for (int k = 0; k < 3; ++k)
            if (host_planes[k])
                HIP_TRY(ctx, hipMemcpyAsync(host_planes[k], ctx->d_dbg[k], npx * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // A frame that a device-side wait of the side kernel gave up on is never handed out: the side kernel is out
        // of service for this context from here on, and the frame is rendered again without it (once: the second
        // pass has no side kernel to wait for).  rt_last_error keeps the reason although the call succeeds.
        uint32_t word = 0u;
        if ((rc = check_heavy_status(ctx, &word)) != RT_OK) return rc;
        if (word == 0u) break;
        (void)fail(ctx, RT_OK, std::string("rt_render: ") + heavy_wait_name(word) + " gave up (status " + std::to_string(word) +
                                   "); the side kernel for heavy tiles is disabled for this context and the frame was rendered again without it");
    }
    return RT_OK;
}

int rt_sync(rt_ctx* ctx)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t word = 0u;
    int rc = check_heavy_status(ctx, &word);
    if (rc != RT_OK) return rc;
    if (word != 0u)
        return fail(ctx, RT_ERR_DEVICE_WAIT, std::string("rt_sync: ") + heavy_wait_name(word) + " gave up (status " + std::to_string(word) +
                                                 "): frames rendered with the side kernel since the last rt_sync may be incomplete - render them again; "
                                                 "the side kernel for heavy tiles is disabled for this context from here on");
    return RT_OK;
}

int rt_count_rays(rt_ctx* ctx, const rt_frame* frame, rt_counters* out) { return rt_count_rays_part(ctx, frame, 1u, 0u, out); }

int rt_count_rays_part(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, rt_counters* out)
{
    return rt_count_executed(ctx, frame, n_parts, part, out, nullptr);
}

int rt_count_executed(rt_ctx* ctx, const rt_frame* frame, uint32_t n_parts, uint32_t part, rt_counters* out, rt_exec_counters* exec)
{
    int rc = check_frame(ctx, frame);
    if (rc != RT_OK) return rc;
    if (!out && !exec) return fail(ctx, RT_ERR_INVALID_ARG, "out is null");
    if (n_parts == 0u || part >= n_parts) return fail(ctx, RT_ERR_INVALID_ARG, "need 0 <= part < n_parts");
    ENTER_DEVICE(ctx);
    Params p;
    fill_frame(ctx, frame, p);
    p.n_parts = n_parts;
    p.part = part;
    p.counters = ctx->d_counters;
    if ((rc = prepare(ctx, p, ctx->own)) != RT_OK) return rc;
    constexpr size_t kWords = 10 + RT_EXEC_EVENTS;
    static_assert(kWords * sizeof(unsigned long long) <= kCounterBytes, "counter buffer too small");
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, kWords * sizeof(unsigned long long), ctx->stream));
    if ((rc = listing(ctx, p, kVariantCount)) != RT_OK) return rc;
    hipError_t e = (hipError_t)launch_render(p, kVariantCount, (void*)ctx->stream, ctx->max_workgroups);
    if (e != hipSuccess) return fail_hip(ctx, e, "count kernel launch");
    unsigned long long h[kWords];
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out) {
        out->nearest_rays = h[0];
        out->shadow_rays = h[1];
        out->cube_tests = h[2];
        out->sphere_tests = h[3];
        out->shaded_hits = h[4];
        out->ref_nearest_rays = h[5];
        out->ref_shadow_rays = h[6];
        out->ref_cube_tests = h[7];
        out->ref_sphere_tests = h[8];
        out->ref_shaded_hits = h[9];
    }
    if (exec) {
        exec->cull_tests = h[10];
        exec->chunk_tests = h[11];
        exec->candidate_evals = h[12];
        exec->exact_evals = h[13];
        exec->bvh_steps = h[14];
        exec->sphere_evals = h[15];
        exec->queries = h[16];
        exec->light_iterations = h[17];
        exec->resolved_hits = h[18];
        exec->blockers = h[19];
        exec->child_setups = h[20];
        exec->primary_rays = h[21];
    }
    return RT_OK;
}

int rt_selftest_division(rt_ctx* ctx, uint32_t seed, uint64_t n_cases, uint64_t* out_mismatches, uint64_t* out_cases)
{
    if (!ctx || !out_mismatches || !out_cases) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    ENTER_DEVICE(ctx);
    const uint32_t iters = 4096u;
    uint64_t blocks64 = (n_cases + 256ull * iters - 1ull) / (256ull * iters);
    if (blocks64 == 0) blocks64 = 1;
    if (blocks64 > 65536ull) return fail(ctx, RT_ERR_LIMIT, "rt_selftest_division: at most 2^36 cases per call");
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipError_t e = (hipError_t)launch_selftest_division(seed, (uint32_t)blocks64, iters, ctx->d_counters, (void*)ctx->stream);
    if (e != hipSuccess) return fail_hip(ctx, e, "selftest kernel launch");
    unsigned long long h[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_mismatches = h[0];
    *out_cases = h[1];
    return RT_OK;
}

int rt_enable_timing(rt_ctx* ctx, int enable)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    if (enable < 0) return fail(ctx, RT_ERR_INVALID_ARG, "enable must be 0 (off), 1 (every launch) or k > 1 (every k-th launch)");
    ctx->timing = enable != 0;
    ctx->timing_every = enable > 1 ? (uint32_t)enable : 1u;
    ctx->timing_phase = 0u;
    return RT_OK;
}

int rt_set_grid_limit(rt_ctx* ctx, uint32_t max_workgroups)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    if (max_workgroups >= (1u << 30)) return fail(ctx, RT_ERR_INVALID_ARG, "grid limit out of range");
    ctx->max_workgroups = max_workgroups ? max_workgroups : kDefaultGrid;
    return RT_OK;
}

int rt_kernel_time_ms(rt_ctx* ctx, double* out_avg_ms, uint32_t* out_launches)
{
    if (!ctx || !out_avg_ms || !out_launches) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    for (auto& ev : ctx->events) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
    }
#ifdef RTX_PROFILE
    {
        std::vector<uint32_t> h(kProfWaves * 40);
        HIP_TRY(ctx, hipMemcpy(h.data(), (char*)ctx->d_counters + 256, h.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemset((char*)ctx->d_counters + 256, 0, h.size() * 4));
        const char* names[10] = {"raygen", "pre(ray_pre+cull_pre)", "cull(+chunk load wait)", "candidates", "contested", "resolve", "shade", "shadow_post", "childsetup", "other"};
        double sum[10] = {0}; double tot = 0; size_t waves = 0; double mx = 0;
        for (size_t w = 0; w < kProfWaves; ++w) {
            double t = 0;
            for (int k = 0; k < 10; ++k) { sum[k] += h[w * 40 + k]; t += h[w * 40 + k]; }
            if (t > 0) { ++waves; tot += t; if (t > mx) mx = t; }
        }
        fprintf(stderr, "[rtx950 profile, last launch] waves=%zu avg wave-cycles=%.0f max=%.0f\n", waves, tot / (waves ? waves : 1), mx);
        {   // slowest waves: tile position and phase split
            std::vector<std::pair<double, size_t>> top;
            for (size_t w = 0; w < kProfWaves; ++w) { double t = 0; for (int k = 0; k < 10; ++k) t += h[w * 40 + k]; if (t > 0) top.push_back({t, w}); }
            std::sort(top.begin(), top.end()); std::reverse(top.begin(), top.end());
            const size_t gx = 480;  // 3840/8: tile grid of the 4K bench frame (diagnostic only)
            for (size_t i = 0; i < top.size() && i < 12; ++i) {
                size_t w = top[i].second;
                fprintf(stderr, "  slow wave %zu: tile x=%zu y=%zu cyc=%.0f :", i, (w % gx) * 8, (w / gx) * 8, top[i].first);
                for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", h[w * 40 + k]);
                fprintf(stderr, "\n");
            }
            if (const char* dump = getenv("RTX950_PROF_DUMP")) {
                if (FILE* f = fopen(dump, "wb")) { fwrite(h.data(), 4, (size_t)129600 * 40, f); fclose(f); }
            }
            size_t hist[8] = {0}; for (auto& t : top) { int b = 0; double v = t.first; while (v > 8192 && b < 7) { v /= 2; ++b; } hist[b]++; }
            fprintf(stderr, "  histogram of wave cycles (<=8K,16K,32K,64K,128K,256K,512K,more):"); for (int b = 0; b < 8; ++b) fprintf(stderr, " %zu", hist[b]); fprintf(stderr, "\n");
        }
        for (int k = 0; k < 10; ++k) fprintf(stderr, "  %-24s %6.2f %%  %8.0f cyc/wave\n", names[k], 100.0 * sum[k] / (tot > 0 ? tot : 1), sum[k] / (double)(waves ? waves : 1));
    }
#endif
    *out_launches = (uint32_t)ctx->events.size();
    *out_avg_ms = ctx->events.empty() ? 0.0 : total / (double)ctx->events.size();
    for (auto& ev : ctx->events) ctx->event_pool.push_back(ev);
    ctx->events.clear();
    return RT_OK;
}

// Host-side twin of classify_tiles, rows only: the same padded f64 projection of every cube's corners.
int rt_frame_row_extent(rt_ctx* ctx, const rt_frame* frame, uint32_t* out_y_first, uint32_t* out_y_end)
{
    int rc = check_frame(ctx, frame);
    if (rc != RT_OK) return rc;
    if (!out_y_first || !out_y_end) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    *out_y_first = frame->y0;
    *out_y_end = frame->y1;
    if (ctx->h_n_spheres != 0) return RT_OK;  // the sphere pass is traced everywhere
    const uint32_t dims[4] = {frame->width, frame->height, frame->y0, frame->y1};
    if (ctx->extent_valid && std::memcmp(&ctx->extent_camera, &frame->camera, sizeof(rt_camera)) == 0 &&
        std::memcmp(ctx->extent_dims, dims, sizeof(dims)) == 0) {
        *out_y_first = ctx->extent_rows[0];
        *out_y_end = ctx->extent_rows[1];
        return RT_OK;
    }
    auto remember = [&]() {
        ctx->extent_camera = frame->camera;
        std::memcpy(ctx->extent_dims, dims, sizeof(dims));
        ctx->extent_rows[0] = *out_y_first;
        ctx->extent_rows[1] = *out_y_end;
        ctx->extent_valid = true;
    };
    Params p;
    fill_frame(ctx, frame, p);
    TileParams t;
    if (!projection_setup(p, t)) { remember(); return RT_OK; }
    const double* r = t.right; const double* u = t.up; const double* f = t.fwd;
    double lo = 1e300, hi = -1e300;
    const size_t n = ctx->h_cube_box.size() / 4;
    for (size_t i = 0; i < n; ++i) {
        const float* rec = &ctx->h_cube_box[4 * i];
        const double h = std::fabs((double)rec[3]) + t.pad;
        for (unsigned corner = 0; corner < 8u; ++corner) {
            double q[3];
            for (int k = 0; k < 3; ++k) q[k] = (double)rec[k] + (((corner >> k) & 1u) ? h : -h) - t.eye[k];
            // q = a right + b up + c fwd (Cramer); only b and c matter for the row
            const double b = (r[0] * (q[1] * f[2] - q[2] * f[1]) - r[1] * (q[0] * f[2] - q[2] * f[0]) + r[2] * (q[0] * f[1] - q[1] * f[0])) * t.inv_det;
            const double c = (r[0] * (u[1] * q[2] - u[2] * q[1]) - r[1] * (u[0] * q[2] - u[2] * q[0]) + r[2] * (u[0] * q[1] - u[1] * q[0])) * t.inv_det;
            const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            const double py = (1.0 - b / c) * (double)t.height * 0.5;
            // a corner beside or behind the eye (the slab test is a line test), or NaN: no conclusion
            if (!(c > 1e-3 * len) || !(c > 16.0 * t.pad) || !(std::fabs(py) < 1e9)) { remember(); return RT_OK; }
            lo = std::min(lo, py);
            hi = std::max(hi, py);
        }
    }
    // same two-pixel padding as the tile rectangles; the last row that can hold geometry is ceil(hi) + 2
    const double first = std::floor(lo) - 2.0, last = std::ceil(hi) + 2.0;
    uint32_t y_first = frame->y0, y_end = frame->y0;  // nothing projects into the band: empty extent
    if (n != 0 && last >= (double)frame->y0 && first <= (double)frame->y1 - 1.0) {
        y_first = first <= (double)frame->y0 ? frame->y0 : (uint32_t)first;
        y_end = last >= (double)frame->y1 - 1.0 ? frame->y1 : (uint32_t)last + 1u;
    }
    *out_y_first = y_first;
    *out_y_end = y_end;
    remember();
    return RT_OK;
}

int rt_set_tuning(rt_ctx* ctx, int key, double value)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID_ARG, "ctx is null");
    const bool tri = (value == -1.0 || value == 0.0 || value == 1.0);
    switch (key) {
    case RT_TUNE_COHERENT_SCALE:
        if (!(value >= 0.0)) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_COHERENT_SCALE must be >= 0");
        ctx->tune_coherent_scale = (float)value;
        return RT_OK;
    case RT_TUNE_SHADOW_LIST:
        if (!tri) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_SHADOW_LIST must be -1 (automatic), 0 or 1");
        ctx->tune_shadow_list = (int)value;
        return RT_OK;
    case RT_TUNE_DARK_KERNEL:
        if (!tri) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_DARK_KERNEL must be -1 (automatic), 0 or 1");
        ctx->tune_dark_kernel = (int)value;
        return RT_OK;
    case RT_TUNE_FILL_MEMSET:
        if (!(value == 0.0 || value == 1.0)) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_FILL_MEMSET must be 0 or 1");
        ctx->tune_fill_memset = value != 0.0;
        return RT_OK;
    case RT_TUNE_HEAVY_CLASS:
        if (!(value >= -1.0 && value <= (double)kCostClasses && value == (double)(int)value))
            return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_HEAVY_CLASS must be -1 (automatic) or an integer in 0..8");
        ctx->tune_heavy_class = (int)value;
        return RT_OK;
    case RT_TUNE_HEAVY_SYNC:
        if (!tri) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_HEAVY_SYNC must be -1 (automatic), 0 (events) or 1 (device-side)");
        ctx->tune_heavy_sync = (int)value;
        return RT_OK;
    case RT_TUNE_HEAVY_SPIN_LIMIT:
        if (!(value >= 0.0 && value <= 4294967295.0 && value == (double)(uint32_t)value))
            return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_HEAVY_SPIN_LIMIT must be an integer in 0..2^32-1 (0: default)");
        ctx->tune_heavy_spin = (uint32_t)value;
        if (value == 0.0) ctx->heavy_disabled = false;  // the default restores a side kernel that a give-up took out of service
        return RT_OK;
    case RT_TUNE_WAVEFRONT:
        if (!tri) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_WAVEFRONT must be -1 (automatic), 0 or 1");
        ctx->tune_wavefront = (int)value;
        return RT_OK;
    case RT_TUNE_CULL_REFINE:
        if (!(tri || value == 2.0)) return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_CULL_REFINE must be -1 (automatic), 0, 1 or 2");
        ctx->tune_cull_refine = (int)value;
        return RT_OK;
    case RT_TUNE_WAVEFRONT_CAP:
        if (!(value >= 0.0 && value <= (double)(1u << 22) && value == (double)(uint32_t)value))
            return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_WAVEFRONT_CAP must be an integer in 0..2^22 (0: from the frame size)");
        ctx->tune_wf_cap = (uint32_t)value;
        return RT_OK;
    case RT_TUNE_BVH_DEPTH_LIMIT:
        if (!(value >= 1.0 && value <= (double)kBvhStackMax && value == (double)(uint32_t)value))
            return fail(ctx, RT_ERR_INVALID_ARG, "RT_TUNE_BVH_DEPTH_LIMIT must be an integer in 1..28");
        ctx->tune_bvh_depth_limit = (uint32_t)value;
        return RT_OK;
    default:
        return fail(ctx, RT_ERR_INVALID_ARG, "unknown tuning key");
    }
}

int rt_host_alloc(size_t bytes, void** out_ptr)
{
    if (!out_ptr) return fail(nullptr, RT_ERR_INVALID_ARG, "out_ptr is null");
    *out_ptr = nullptr;
    if (bytes == 0) return fail(nullptr, RT_ERR_INVALID_ARG, "zero bytes");
    HIP_TRY(nullptr, hipHostMalloc(out_ptr, bytes, hipHostMallocPortable));
    return RT_OK;
}

int rt_host_free(void* ptr)
{
    if (!ptr) return RT_OK;
    HIP_TRY(nullptr, hipHostFree(ptr));
    return RT_OK;
}

// The allocation-free variant of rt_render_device: the same kernels, with the tile stamps, the tile queue
// and the tile costs in this GPU's copy of the file-scope array `scratch` instead of buffers the context
// allocates and grows.  Calls no hipMalloc/hipFree, creates no events.
int rt_render_device_noalloc(rt_ctx* ctx, const rt_frame* frame, uint32_t* d_out_0rgb)
{
    int rc = check_frame(ctx, frame);
    if (rc != RT_OK) return rc;
    if (!d_out_0rgb) return fail(ctx, RT_ERR_INVALID_ARG, "output pointer is null");
    if (ctx->device >= kMaxDevices || !g_dev_scratch[ctx->device].ready)
        return fail(ctx, RT_ERR_UNSUPPORTED, "no fixed work buffer for this device");
    const uint64_t frame_tiles = (uint64_t)((frame->width + 7u) / 8u) * ((frame->height + 7u) / 8u);
    if (frame_tiles > RT_NOALLOC_MAX_TILES)
        return fail(ctx, RT_ERR_LIMIT, "rt_render_device_noalloc: frame has " + std::to_string(frame_tiles) +
                                           " 8x8 tiles, the fixed work buffer holds RT_NOALLOC_MAX_TILES = " + std::to_string(RT_NOALLOC_MAX_TILES));
    ENTER_DEVICE(ctx);
    DeviceScratch& ds = g_dev_scratch[ctx->device];
    std::lock_guard<std::mutex> lock(ds.mu);  // host-side bookkeeping only: launches of overlapping calls still race on the device
    Params p;
    fill_frame(ctx, frame, p);
    p.out = d_out_0rgb;
    if ((rc = prepare(ctx, p, ds.ts)) != RT_OK) return rc;
    if (((uintptr_t)d_out_0rgb & 15u) == 0u && (frame->width & 3u) == 0u && !ctx->tune_fill_memset) {
        p.prefilled = 2u;
    } else {
        const size_t npx = (size_t)(frame->y1 - frame->y0) * frame->width;
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_out_0rgb, (int)p.sky, npx, ctx->stream));
        p.prefilled = 1u;
    }
    // the same choice of kernel variant as rt_render_device (no side kernel here)
    p.refine = refine_choice(ctx, false, (size_t)frame_tiles);
    ctx->last_refine = p.refine;
    if ((rc = listing(ctx, p, kVariantRender)) != RT_OK) return rc;
    hipError_t e = (hipError_t)launch_render(p, kVariantRender, (void*)ctx->stream, ctx->max_workgroups);
    if (e != hipSuccess) return fail_hip(ctx, e, "render kernel launch");
    return RT_OK;
}

// Scatter of the compact parts into frame order on ctxs[0]'s GPU: part g's k-th 16-row block is block
// k * n + g of the band, so a part is one strided 2-D copy (rows of 16 * width pixels, destination pitch
// n such rows) plus, for the part that owns the band's ragged last block, one short copy.  Each copy runs
// on the stream of the context that rendered the part, i.e. behind its render with no host wait; GPU0's
// stream then waits for every part's event.
int rt_gather_bands(rt_ctx* const* ctxs, uint32_t n_ctx, const uint32_t* const* d_parts, uint32_t width, uint32_t rows,
                    uint32_t* d_band_on_ctx0, uint64_t* out_bytes_moved)
{
    if (!ctxs || n_ctx == 0u || !ctxs[0]) return fail(nullptr, RT_ERR_INVALID_ARG, "ctxs is null or empty");
    rt_ctx* c0 = ctxs[0];
    if (!d_parts || !d_band_on_ctx0 || width == 0u) return fail(c0, RT_ERR_INVALID_ARG, "null argument");
    for (uint32_t g = 0; g < n_ctx; ++g)
        if (!ctxs[g] || !d_parts[g]) return fail(c0, RT_ERR_INVALID_ARG, "a context or part buffer is null");
    const uint32_t blocks = (rows + kRowBlock - 1u) / kRowBlock;
    const size_t block_bytes = (size_t)kRowBlock * width * sizeof(uint32_t);
    uint64_t moved = 0;
    for (uint32_t g = 0; g < n_ctx; ++g) {
        rt_ctx* c = ctxs[g];
        ENTER_DEVICE(c);
        bool direct = true;  // the destination is addressable from this context's GPU
        if (c->device != c0->device) {
            if (c->peer_target != c0->device || c->peer_state < 0) {  // once per (context, destination GPU), not per frame
                int can = 0;
                HIP_TRY(c0, hipDeviceCanAccessPeer(&can, c->device, c0->device));
                c->peer_target = c0->device;
                c->peer_state = 0;
                if (can) {
                    hipError_t pe = hipDeviceEnablePeerAccess(c0->device, 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // another context of this GPU did it
                    else if (pe != hipSuccess) return fail_hip(c0, pe, "hipDeviceEnablePeerAccess");
                    c->peer_state = 1;
                }
            }
            direct = c->peer_state == 1;
        }
        const uint32_t my_blocks = (g < blocks) ? (blocks - g + n_ctx - 1u) / n_ctx : 0u;
        if (my_blocks == 0u) continue;
        // the band's last block may be ragged; it belongs to part (blocks - 1) % n_ctx and is that part's last
        const bool ragged = (rows % kRowBlock != 0u) && ((blocks - 1u) % n_ctx == g);
        const uint32_t full = ragged ? my_blocks - 1u : my_blocks;
        uint32_t* dst = d_band_on_ctx0 + (size_t)g * kRowBlock * width;
        if (direct) {
            if (full)
                HIP_TRY(c0, hipMemcpy2DAsync(dst, block_bytes * n_ctx, d_parts[g], block_bytes, block_bytes, full, hipMemcpyDeviceToDevice, c->stream));
            if (ragged) {
                const size_t tail_bytes = (size_t)(rows % kRowBlock) * width * sizeof(uint32_t);
                HIP_TRY(c0, hipMemcpyAsync(dst + (size_t)full * n_ctx * kRowBlock * width, d_parts[g] + (size_t)full * kRowBlock * width, tail_bytes,
                                           hipMemcpyDeviceToDevice, c->stream));
            }
        } else {
            // no peer access between the two GPUs: one runtime-staged peer copy per 16-row block (slow, but correct)
            for (uint32_t k = 0; k < my_blocks; ++k) {
                const size_t bytes = (ragged && k + 1u == my_blocks) ? (size_t)(rows % kRowBlock) * width * sizeof(uint32_t) : block_bytes;
                HIP_TRY(c0, hipMemcpyPeerAsync(dst + (size_t)k * n_ctx * kRowBlock * width, c0->device, d_parts[g] + (size_t)k * kRowBlock * width, c->device, bytes,
                                               c->stream));
            }
        }
        if (c != c0) moved += (uint64_t)rt_part_rows(rows, n_ctx, g) * width * sizeof(uint32_t);
        if (c->stream != c0->stream) {
            if (!c->part_done) HIP_TRY(c0, hipEventCreateWithFlags(&c->part_done, hipEventDisableTiming));
            HIP_TRY(c0, hipEventRecord(c->part_done, c->stream));
        }
    }
    {
        ENTER_DEVICE(c0);
        for (uint32_t g = 1; g < n_ctx; ++g)
            if (ctxs[g]->stream != c0->stream && ctxs[g]->part_done) HIP_TRY(c0, hipStreamWaitEvent(c0->stream, ctxs[g]->part_done, 0));
    }
    if (out_bytes_moved) *out_bytes_moved = moved;
    return RT_OK;
}

// One frame on n GPUs from one host thread: the rows that can hold geometry (rt_frame_row_extent) are cut
// into 16-row blocks dealt round-robin to the contexts, each renders its blocks into a buffer of its own
// and rt_gather_bands moves them into the frame on ctxs[0]'s GPU; that GPU fills the sky rows itself.
int rt_render_multi(rt_ctx* const* ctxs, uint32_t n_ctx, const rt_frame* frame, uint32_t* d_frame_on_ctx0, uint64_t* out_bytes_moved)
{
    if (!ctxs || n_ctx == 0u || !ctxs[0]) return fail(nullptr, RT_ERR_INVALID_ARG, "ctxs is null or empty");
    rt_ctx* c0 = ctxs[0];
    int rc = check_frame(c0, frame);
    if (rc != RT_OK) return rc;
    if (!d_frame_on_ctx0) return fail(c0, RT_ERR_INVALID_ARG, "output pointer is null");
    for (uint32_t g = 1; g < n_ctx; ++g) {
        if (!ctxs[g]) return fail(c0, RT_ERR_INVALID_ARG, "a context is null");
        if ((rc = check_frame(ctxs[g], frame)) != RT_OK) return fail(c0, rc, std::string("context ") + std::to_string(g) + ": " + ctxs[g]->err);
        for (uint32_t k = 0; k < g; ++k)
            if (ctxs[k] == ctxs[g]) return fail(c0, RT_ERR_INVALID_ARG, "the same context is listed twice (one context per part)");
    }
    uint32_t ya = frame->y0, yb = frame->y1;
    if ((rc = rt_frame_row_extent(c0, frame, &ya, &yb)) != RT_OK) return rc;
    // the sub-band that is rendered: from the 16-row block of the band that holds ya to yb
    const uint32_t w = frame->width;
    const uint32_t s0 = frame->y0 + ((ya - frame->y0) / kRowBlock) * kRowBlock, s1 = yb;
    {
        ENTER_DEVICE(c0);
        const uint32_t sky = frame->sky_0rgb & 0x00FFFFFFu;
        const size_t head = (size_t)((s1 > s0 ? s0 : frame->y1) - frame->y0) * w;  // nothing to render: the whole band is sky
        if (head) HIP_TRY(c0, hipMemsetD32Async((hipDeviceptr_t)d_frame_on_ctx0, (int)sky, head, c0->stream));
        if (s1 > s0 && s1 < frame->y1)
            HIP_TRY(c0, hipMemsetD32Async((hipDeviceptr_t)(d_frame_on_ctx0 + (size_t)(s1 - frame->y0) * w), (int)sky, (size_t)(frame->y1 - s1) * w, c0->stream));
        // the reference's never-painted pixel i == width (render.rs:316-322) when its row is one of the filled ones
        if (frame->y0 <= 1u && 1u < frame->y1 && !(s1 > s0 && s0 <= 1u && 1u < s1))
            HIP_TRY(c0, hipMemsetD32Async((hipDeviceptr_t)(d_frame_on_ctx0 + (size_t)(1u - frame->y0) * w), 0, 1, c0->stream));
    }
    if (!(s1 > s0)) {
        if (out_bytes_moved) *out_bytes_moved = 0;
        return RT_OK;
    }
    rt_frame sub = *frame;
    sub.y0 = s0;
    sub.y1 = s1;
    std::vector<const uint32_t*> parts(n_ctx, nullptr);
    for (uint32_t g = 0; g < n_ctx; ++g) {
        rt_ctx* c = ctxs[g];
        ENTER_DEVICE(c);
        const size_t need = (size_t)rt_part_rows(s1 - s0, n_ctx, g) * w;
        if ((rc = ensure(c, &c->d_part, &c->part_capacity, need ? need : 1)) != RT_OK) return fail(c0, rc, c->err);
        parts[g] = c->d_part;
        if (need && (rc = rt_render_device_part(c, &sub, n_ctx, g, c->d_part, nullptr)) != RT_OK) return fail(c0, rc, c->err);
    }
    // GPU0's fills above run on its stream; the parts' copies run on theirs and touch other rows
    return rt_gather_bands(ctxs, n_ctx, parts.data(), w, s1 - s0, d_frame_on_ctx0 + (size_t)(s0 - frame->y0) * w, out_bytes_moved);
}

int rt_debug_status(rt_ctx* ctx, uint32_t* out_word)
{
    if (!ctx || !out_word) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    *out_word = 0u;
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own.queue) HIP_TRY(ctx, hipMemcpy(out_word, ctx->own.queue + kQueueBase - 1u, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_tile_costs(rt_ctx* ctx, uint32_t* out, size_t cap, size_t* out_tiles)
{
    if (!ctx || !out_tiles) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t tiles = ctx->own.cost_cap ? ctx->own.cost_cap - 1 : 0;
    *out_tiles = tiles;
    const size_t n = cap < tiles ? cap : tiles;
    if (out && n) HIP_TRY(ctx, hipMemcpy(out, ctx->own.cost, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_debug_wavefront_items(rt_ctx* ctx, uint32_t level, uint32_t* out_words, size_t cap_items, size_t* out_items)
{
    if (!ctx || !out_items || level > kWfLevels) return fail(ctx, RT_ERR_INVALID_ARG, "null argument or level > 4");
    *out_items = 0;
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->wf.base) return RT_OK;
    if (level == kWfLevels) {  // the redo list's length: tiles that render_kernel traced again
        uint32_t n = 0;
        HIP_TRY(ctx, hipMemcpy(&n, ctx->wf.w.header + kWfRedoBase, sizeof(n), hipMemcpyDeviceToHost));
        *out_items = n;
        return RT_OK;
    }
    std::vector<WfItem> all;
    if (level == 0u) {
        all.resize(ctx->wf.w.cap[0]);
        HIP_TRY(ctx, hipMemcpy(all.data(), ctx->wf.w.items[0], all.size() * sizeof(WfItem), hipMemcpyDeviceToHost));
    } else {
        std::vector<uint32_t> hdr(kWfRedoBase);
        HIP_TRY(ctx, hipMemcpy(hdr.data(), ctx->wf.w.header, hdr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t cap = ctx->wf.w.cap[level] / kWfQueues;
        for (uint32_t q = 0; q < kWfQueues; ++q) {
            const uint32_t n = std::min(hdr[((level - 1u) * kWfQueues + q) * 16u], cap);
            const size_t at = all.size();
            all.resize(at + n);
            if (n) HIP_TRY(ctx, hipMemcpy(all.data() + at, ctx->wf.w.items[level] + (size_t)q * cap, n * sizeof(WfItem), hipMemcpyDeviceToHost));
        }
    }
    *out_items = all.size();
    const size_t n = std::min<size_t>(cap_items, all.size());
    if (out_words && n) std::memcpy(out_words, all.data(), n * sizeof(WfItem));
    return RT_OK;
}

const char* rt_kernel_name(const rt_ctx* ctx)
{
    if (!ctx || !ctx->has_scene) return "";
    const bool large = ctx->base.n_chunks > 0u, dark = ctx->base.dark_materials != 0u;  // launch_render's choice
    if (large) return dark ? "render_kernel<false,false,1,true>" : "render_kernel<false,false,1,false>";
    if (ctx->last_refine == 2u) return dark ? "render_kernel<false,false,3,true>" : "render_kernel<false,false,3,false>";
    if (ctx->last_refine) return dark ? "render_kernel<false,false,2,true>" : "render_kernel<false,false,2,false>";
    return dark ? "render_kernel<false,false,0,true>" : "render_kernel<false,false,0,false>";
}

int rt_get_stream(rt_ctx* ctx, void** out_hip_stream)
{
    if (!ctx || !out_hip_stream) return fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    *out_hip_stream = (void*)ctx->stream;
    return RT_OK;
}

}  // extern "C"
