// rtx950_device.h — device-side data layout shared by the kernels and the C-ABI host code.
//
// HBM layout (all read-only during a launch, all tiny next to 288 GB; the only per-pixel HBM
// traffic is the 4 B/px framebuffer store and the occasional texel gather):
//   cubes    : n_cubes  x CubeRec (32 B, one s_load_dwordx8 per ray-cube test, wave-uniform)
//   cube_filt: ceil64(n_cubes) x 16 B (centre, half size): the cull stream, one coalesced 1 KB wave
//              load per 64 cubes (lane j <- record base+j); cube_id: the same for ids
//   cube_mat : n_cubes  x u32 material index (read once per winning hit)
//   spheres  : n_spheres x SphereRec (16 B) + sphere_meta (id, material)
//   mats     : n_materials x MatRec (32 B)
//   atlas    : n_textures x tex_w*tex_h packed 0x00RRGGBB texels (row-major, idx = y*w + x,
//              the layout of reference src/texture.rs:151-161)
#pragma once
#include <stdint.h>

namespace rtx950 {

// reference src/cube.rs:13-25: bounds (min,max), size, id.  Stored as (min,max) pairs per axis so a
// wave-uniform 64-bit SGPR pair feeds one packed FMA.
struct CubeRec {
    float minx, maxx;
    float miny, maxy;
    float minz, maxz;
    uint32_t id;
    float size;
};
static_assert(sizeof(CubeRec) == 32, "CubeRec must be one dwordx8");

// Bounding-volume hierarchy over the cubes of a long list, for incoherent ray bundles (cube_query_bvh):
// binary, median splits of the Morton order of the cube centres, stored as PAIRS of sibling nodes (pair p
// = nodes 2p and 2p+1, the two children of one inner node; pair 0 = the children of the root), so that
// one 64-byte fetch brings both boxes a walk has to choose between.
struct BvhNode {
    float cx, cy, cz;      // box centre; leaf: the cube's centre
    uint32_t first_count;  // leaf (one cube): (its list index << 3) | 1; inner node: lowest list index below it << 3
    float hx, hy, hz;      // half extents (rounded up); leaf: the cube's |half size| three times
    uint32_t link;         // inner node: index of the pair holding its children; leaf: the cube's id
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be two dwordx4");
// The device reads a pair laid out for PACKED FMAs (cube_query_bvh): x and y of a box side by side, as the ray's
// per-axis coefficients are (RayPre::r_xy ...), and the z components of the two boxes side by side, so that the slab
// filter of both boxes is nine v_pk_fma_f32 with four splatted coefficients.
struct BvhPair {
    float a_xy[2], a_hxy[2];           // dwordx4 0: (a.cx, a.cy, a.hx, a.hy)
    float b_xy[2], b_hxy[2];           // dwordx4 1: the same of box b
    float cz[2], hz[2];                // dwordx4 2: (a.cz, b.cz, a.hz, b.hz)
    uint32_t first_count[2], link[2];  // dwordx4 3
};
static_assert(sizeof(BvhPair) == 64, "BvhPair must be four dwordx4");

// reference src/sphere.rs:10-15
struct SphereRec {
    float cx, cy, cz, radius;
};
struct SphereMeta {
    uint32_t id, material;
};

// reference src/material.rs:4-18
struct MatRec {
    float specular, albedo0, albedo1, reflectivity;
    float transparency, ior;
    uint32_t diffuse_0rgb;
    int32_t texture;
};
static_assert(sizeof(MatRec) == 32, "MatRec must be two dwordx4");

// reference src/light.rs:5-9
struct LightRec {
    float px, py, pz;
    uint32_t color_0rgb;
    float intensity;
};

constexpr int kMaxLights = 16;
constexpr uint32_t kBvhStackMax = 28;  // deferred siblings per lane: one per level (rt_scene_upload takes at most 2^26 cubes)
constexpr uint32_t kRowBlock = 16;  // rows per workgroup tile and per interleave block

struct Params {
    // ---- scene (device pointers)
    const CubeRec* cubes;
    const float* cube_filt;  // ceil64(n_cubes) x (centre.x, centre.y, centre.z, half size): lane j of a wave loads record base+j
    const uint32_t* cube_id; // ceil64(n_cubes) x id (shadow exclusion, reference src/render.rs:113)
    const uint32_t* cube_mat;
    // long lists only (n_chunks > 0): the cull stream (cube_filt, cube_id) is in Morton order, cube_perm[j]
    // is the list index of stream record j, chunk_box holds ceil64(n_chunks) x 32 B (centre.xyz, -, half
    // extents.xyz, -): the bounding box of the 64 cubes of a chunk
    const uint32_t* cube_perm;
    const float* chunk_box;
    // shadow_list != 0 (long lists whose list order is spatially compact): the same stream and chunk boxes in
    // LIST order, so that a shadow query can stop at the first chunk that answers every ray
    const float* cube_filt_list;
    const uint32_t* cube_id_list;
    const float* chunk_box_list;
    uint32_t shadow_list;
    uint32_t n_chunks;
    // every scene: cube_perm is valid (identity for short lists); bvh: long lists only (n_bvh_nodes = 0 otherwise)
    const BvhPair* bvh;
    uint32_t n_bvh_nodes;
    float coherent_thr;  // a wave whose cull inflation (2 B dr + wr) exceeds this walks the BVH per lane
    const SphereRec* spheres;
    const SphereMeta* sphere_meta;
    const float* sphere_filt;   // ceil64(n_spheres) x (centre.xyz, |radius|): the spheres' cull stream
    float sphere_bound_max;     // max |centre coordinate| + |radius|
    uint32_t sphere_ok;         // 1 if every sphere is finite and within 2^20
    const MatRec* mats;
    const uint32_t* atlas;
    uint32_t n_cubes, n_spheres;
    uint32_t tex_w, tex_h;
    float sprite;
    float bound_max;   // max |coordinate| over all cube bounds (filter tolerance)
    uint32_t fast_ok;  // 1 if every cube bound is 0 or has magnitude in [2^-30, 2^20]
    // ---- frame
    float eye[3], right[3], up[3], fwd[3];  // camera basis, reference src/camera.rs:34-36 (hoisted)
    uint32_t width, height, y0, y1;
    uint32_t n_parts, part;  // 16-row blocks b with b % n_parts == part are rendered, output compact
    // Tile stamps written by classify_tiles for this frame (null: every tile is traced).  Word 0 ==
    // frame_id: some cube could not be projected, trace every tile; word 64 + ty*tiles_x + tx ==
    // frame_id: some cube's padded projection touches the 8x8 pixel tile (tx, ty) of the frame.  After the
    // tiles_x * tiles_y fine words: one coarse word per block of 8x8 tiles, used by large projections.
    const uint32_t* tile_stamps;
    // the same two levels for the spheres (no header word of its own: word 0 of tile_stamps covers both); null: the
    // sphere pass of a scene with spheres is traced for every tile
    const uint32_t* sphere_stamps;
    uint32_t tiles_x, tiles_y;
    uint32_t frame_id;
    // Work queue of this launch (build_queue -> render_kernel): two header banks of 16
    // words, bank qsel holds the number of listed tiles per class (classes 0..7 = traced tiles by cost,
    // class 8 = tiles that are only painted); class c's entries start at word kQueueBase + c * grid_tx *
    // grid_ty; an entry is tile column | no-cubes flag << 15 | tile row << 16 in the launch's own grid of grid_tx x
    // grid_ty 8x8 tiles.  tile_cost[row * grid_tx + column] = cycles the tile took in the previous launch
    // on this context (0: unknown); it only orders the queue, heaviest class first.
    uint32_t* queue;
    uint32_t* tile_cost;
    uint32_t qsel;  // header bank of this launch (0/1): counts at words qsel * 16 + class; the other bank is zeroed for the next launch
    uint32_t grid_tx, grid_ty;
    // upper bound of the origin coordinates of every ray of the frame: rays start at the eye or on an
    // object's surface (+ 1e-2 along the normal); min(2^20, max(|eye|, object bounds) + 1).  A lane
    // beyond it is treated like a non-finite one (the wave keeps every object).
    float origin_bound;
    float ambient;
    uint32_t sky;
    uint32_t max_depth;
    uint32_t n_lights;
    LightRec lights[kMaxLights];
    // ---- outputs (device pointers; band-relative, (y1-y0)*width entries each)
    uint32_t* out;
    uint32_t* hit_cube;
    uint32_t* hit_sphere;
    uint32_t* tree_hash;
    unsigned long long* counters;  // 10 x u64, rt_counters order
    // 0: build_queue lists the tiles that are only painted and the render kernel paints them (variants with
    // debug planes or counters).  1: `out` holds the sky colour everywhere already (a fill ahead of the
    // launches); 2: build_queue paints those tiles itself; in both cases it lists none of them, so the
    // render kernel has no paint groups.  (Last member on purpose: the render kernel does not read it, and
    // its code stays the same to the instruction.)
    uint32_t prefilled;
    // 1: some material's direct light is exactly black (1 - reflectivity - transparency < 1/255, or not a
    // number the host could bound): launch_render picks the kernel that tests for it per lane.  Host-side
    // only, the kernels do not read it.
    uint32_t dark_materials;
    // ---- heavy tiles (render_heavy_kernel; the members below are read by it and by build_queue only, and sit
    // behind everything render_kernel reads so that its code stays the same to the instruction)
    // The tiles whose cost class in this context's earlier launches was >= heavy_class are traced by
    // render_heavy_kernel on a second stream, concurrently with this launch's classification, listing and
    // render_kernel: a workgroup of three waves per tile instead of one wave.  The list a launch uses was
    // written by the PREVIOUS launch's build_queue (the side kernel must not wait for this launch's): entries
    // heavy[heavy_bank * kHeavyCap + i], i < min(queue[qsel * 16 + kHeavyCountWord], kHeavyCap), a tile column |
    // tile row << 16 each, and heavy_mark[tile] == heavy_now for exactly those tiles, which build_queue then
    // neither lists nor paints.  build_queue also writes the next launch's list (bank heavy_bank ^ 1, marks
    // heavy_next, count in the other header bank).  0 in heavy_now / heavy_next: no list in use / none wanted.
    uint32_t* heavy;
    uint32_t* heavy_mark;
    uint32_t heavy_now, heavy_next;
    uint32_t heavy_class;
    // levels of the per-lane stack of cube_query_bvh that the launch allocates (dynamic LDS, 256 B a level and
    // wave): the depth of the scene's BVH, at most kBvhStackMax.  14 levels for 10 000 cubes: with the stack at
    // its full 28 levels the long-list kernels held 2 waves per SIMD, now 3 (10k-cube world, two frames in
    // flight: 1.23 -> 0.94 ms)
    uint32_t bvh_levels;
    float split_thr;  // heavy tiles: a wave whose cull inflation exceeds this is walked bundle by bundle (cube_query, SPLIT)
    // which plain short-list kernel: 0 render_kernel<.., 0, ..>, 1 with the cull refinement (<.., 2, ..>), 2 with the
    // refinement and the walk by bundles (<.., 3, ..>); the host's choice per launch (refine_choice).  Host-side only.
    uint32_t refine;
    uint32_t heavy_spin_limit;  // heavy_join: polls before it gives up (0: 2^24; 1: report at once, RT_TUNE_HEAVY_SPIN_LIMIT)
    // ---- wave-uniform float values that the kernels would otherwise compute once per wave and then hold in vector
    // registers from the first instruction to the last (the scalar unit has no float arithmetic): computed by the
    // host with the same f32 operations (fill_frame / rt_scene_upload), read through the scalar cache where used
    float width_f, height_f, aspect;        // (f32) width, (f32) height, width_f / height_f (render.rs:253-257)
    float n_lights_f;                       // (f32) n_lights (node_finish)
    float sprite_x0, sprite_x2, sprite_x3;  // sprite * 0.0f, * 2.0f, * 3.0f (texture.rs:178-196)
    float cull_slack, sphere_cull_slack;    // 2^-17 (bound_max + origin_bound + 1), the cube list's and the sphere list's (cull_pre)
    float sphere_grow;                      // 2^-17 (origin_bound + sphere_bound_max + 1)^2 (sphere_query)
};
constexpr uint32_t kHeavyCap = 1024;       // heavy tiles per launch; the rest stay with render_kernel
constexpr uint32_t kHeavyTarget = 512;     // classify_tiles raises the class threshold until about this many tiles qualify:
                                           // what the side kernel's three-wave workgroups can hold at once beside render_kernel
constexpr uint32_t kHeavyCountWord = 9;    // word of a queue header bank that counts the heavy list's entries
constexpr uint32_t kHeavyDoneWord = 10;    // ... and the one that counts the entries render_heavy_kernel has finished
constexpr uint32_t kHeavyExitWord = 11;    // ... and the one that counts the workgroups of render_heavy_kernel that have ended
constexpr uint32_t kHeavyGoWord = 48;      // queue header word (a cache line of its own): classify_tiles stores the launch's heavy_now
                                           // here, render_heavy_kernel starts its tiles when it reads it
constexpr uint32_t kHeavyClassWord = 62;   // queue header word: the lowest class that fits (classify_tiles -> build_queue)
constexpr uint32_t kHeavyGrid = 576;       // workgroups of render_heavy_kernel (striding over the list)

// classify_tiles: one wave per cube projects the cube's padded corners through the frame's camera and
// stamps the 8x8 pixel tiles its bounding rectangle touches.  Primary rays of unstamped tiles cannot
// meet any cube (see the kernel's comment), so render_kernel answers their cube pass without tracing.
struct TileParams {
    const float* cube_filt;  // (centre.xyz, half size) per cube
    uint32_t n_cubes;
    // spheres are classified like cubes, by the bounding box of the sphere grown as in sphere_query (the reference's
    // f32 discriminant can report a hit for a line that misses by a little): blocks n_cubes .. n_cubes + n_spheres - 1
    // stamp the sphere plane.  n_spheres == 0: no sphere plane (no spheres, or a sphere list the cull cannot bound).
    const float* sphere_filt;  // (centre.xyz, |radius|) per sphere
    uint32_t n_spheres;
    double sphere_grow;        // Params::sphere_grow
    uint32_t* sphere_stamps;   // the sphere plane: per-tile words, then the coarse level (no header of its own)
    double eye[3], right[3], up[3], fwd[3];
    double inv_det;          // 1 / det[right up fwd]
    double pad;              // absolute padding of every cube bound
    double aspect;           // (f32) width / height, as the kernel computes it
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t* stamps;
    uint32_t frame_id;
    uint32_t* queue;  // block 0 zeroes header bank qsel ^ 1 (the next launch's)
    uint32_t qsel;
    uint32_t heavy_on;  // the launch keeps a heavy list: block 0 works out which cost classes fit (kHeavyClassWord)
    uint32_t heavy_go;  // Params::heavy_now of the launch (0: no side kernel): block 0 stores it to queue[kHeavyGoWord]
    uint32_t* wf_header;  // wavefront launches: block 0 zeroes the counters of WfParams::header (null otherwise)
};
constexpr uint32_t kStampBase = 64;  // first per-tile word
constexpr uint32_t kQueueBase = 64;  // first tile entry of Params::queue
constexpr uint32_t kCostClasses = 8;

enum KernelVariant { kVariantRender = 0, kVariantDebug = 1, kVariantCount = 2 };

// ---- the wavefront path: one launch per ray-tree level instead of a tile-resident walk of the whole tree
// (cast_ray's recursion, reference src/render.rs:154-193).  A launch traces ITEMS: bundles of up to 64 rays of one
// level that left one parent wave through the same kind of child (reflection / refraction) off the same cube face,
// i.e. coherent rays, whatever the tile looked like.  Level 0 = the primary rays of a traced tile (one item per
// pass); an item that spawns children allocates their items and ray blocks with one atomic add; the last launch
// (wf_combine_kernel) folds every tile's item tree bottom-up with node_finish, in the reference's operation order,
// and writes the pixels.  A tile whose items did not fit the buffers is rendered again by render_kernel (redo list).
struct WfItem {            // 32 B, wave-uniform
    uint32_t parent;       // the parent item (index in the level above); level 0: unused
    uint32_t info;         // bit 0: child kind (0 reflection, 1 refraction), bit 1: pass (PASS_SPHERES / PASS_CUBES)
    uint32_t mask_lo, mask_hi;  // the lanes (pixels of the root tile) that have a ray in this bundle
    uint32_t entry;        // the root tile's queue entry (column | no-cubes flag << 15 | row << 16)
    uint32_t first_child, n_children;  // written by the wave that traces the item: its child items, level + 1
    uint32_t stamp;        // WfParams::seq of the launch that wrote the item (anything else: not an item of this launch)
};
static_assert(sizeof(WfItem) == 32, "WfItem must be one dwordx8");
constexpr uint32_t kWfLevels = 4;          // ray-tree depths 0..3 (max_depth <= 3)
constexpr uint32_t kWfBundles = 3;         // bundles a wave's children of one kind are split into at most (by face; the last takes the rest)
constexpr uint32_t kWfNodeWords = 2 * 64;  // per item: packed light sum | hit flag, material index; per lane
constexpr uint32_t kWfRayWords = 6 * 64;   // per item: origin, direction; per lane
// Allocation counters.  A same-address atomic add costs the device about 3 ns, and the waves of a level reach their
// allocation at about the same time: 6 000 of them queued behind one counter wait 10 us on average (measured: 50 K
// of a level-1 item's 80 K cycles).  Every level therefore has kWfQueues sub-queues with a counter each, a cache line
// apart; a wave allocates in the sub-queue of its own item index modulo kWfQueues, and the next launch's workgroup b
// serves sub-queue b % kWfQueues.  Item index = sub-queue * (cap / kWfQueues) + position.  The same for the list of
// pending roots.  WfParams::header: counter c at word 16 c - c = (level - 1) * kWfQueues + q for levels 1..3, then
// the kWfQueues pending counters -, then the redo list as a queue in render_kernel's format (its header bank 0 at
// word kWfRedoBase, entries from kWfRedoBase + kQueueBase).
constexpr uint32_t kWfQueues = 64;
constexpr uint32_t kWfPendingCounter = 3 * kWfQueues;                     // first pending counter
constexpr uint32_t kWfRedoBase = (kWfPendingCounter + kWfQueues) * 16;    // = 4096
struct WfParams {
    WfItem* items[kWfLevels];     // level 0: 2 items per traced tile (index 2 * tile's queue position + pass)
    uint32_t* nodes[kWfLevels];   // [item][2][64]
    float* rays[kWfLevels];       // [item][6][64], levels 1..3
    uint32_t cap[kWfLevels];      // items per level (levels 1..3: a multiple of kWfQueues)
    uint32_t pending_cap;         // entries per pending sub-list
    uint32_t* header;             // counters, then the redo queue (one entry per tile of the launch at most)
    uint32_t* pending;            // level-0 positions of the tiles whose items wait for children
    uint32_t* redo_mark;          // per tile of the launch: == seq when the tile is on the redo list
    uint32_t seq;                 // stamp of this launch (only grows)
    uint32_t level;               // the level a wf_trace_kernel launch traces
};

// Launches the render kernel for rows [y0,y1) on `stream` (hipStream_t passed as void*).
// Returns a hipError_t as int.
// launch_queue: lists the tiles that need tracing and paints the others (build_queue); launch_render:
// at most max_workgroups single-wave workgroups striding over the queue.  Both return a hipError_t.
int launch_queue(const Params& p, int variant, void* stream);
int launch_render(const Params& p, int variant, void* stream, uint32_t max_workgroups);
// launch_heavy: render_heavy_kernel for the launch's heavy list (plain frames only), on `stream`
// (gate: heavy_gate ahead of it, for the device-side meeting of the streams; false when events order them)
int launch_heavy(const Params& p, void* stream, bool gate);
// launch_heavy_join: one wave on the context's stream that ends when render_heavy_kernel has finished every
// tile of the launch's list (a kernel behind a kernel costs the stream 2-3 us, an event wait 9)
int launch_heavy_join(const Params& p, void* stream);
int launch_classify(const TileParams& t, void* stream);
// wavefront path: level 0 (striding over the tile queue like render_kernel), then one launch per deeper level
// (w.level), the combine launch, and render_kernel over the redo list
int launch_wf_trace(const Params& p, const WfParams& w, void* stream, uint32_t max_workgroups);
int launch_wf_combine(const Params& p, const WfParams& w, void* stream, uint32_t max_workgroups);
// selftest_division: blocks x 256 threads x iters operand pairs; d_out[0] += mismatches, d_out[1] += cases
int launch_selftest_division(uint32_t seed, uint32_t blocks, uint32_t iters, unsigned long long* d_out, void* stream);

}  // namespace rtx950
