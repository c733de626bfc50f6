// scene.hpp — scene construction: the reference's default "Minecraft diorama" and the seeded
// synthetic voxel world of BASELINE.json config 4.
//
// Mirrors reference src/main.rs:137-260 (`init`), :318-357 (`generate_platform`), :359-410
// (`generate_rectangle`), :412-462 (`generate_portal_wall`), including their id bookkeeping
// quirks (ids repeat per platform row; `object_id += cubes.len() + 1` uses the pre-append length).
// The reference draws platform materials from an unseeded thread_rng (main.rs:336-346); here the
// draw comes from PCG32 with a caller-supplied seed so a frame is reproducible.
#pragma once

#include <algorithm>
#include <tuple>

#include "raytracer.hpp"

namespace raytracer {

// PCG32 (XSH-RR 64/32, O'Neill), fixed stream.
struct Pcg32 {
    uint64_t state = 0, inc = 0;
    explicit Pcg32(uint64_t seed, uint64_t seq = 0xda3e39cb94b95bdbull)
    {
        inc = (seq << 1u) | 1u;
        next_u32();
        state += seed;
        next_u32();
    }
    uint32_t next_u32()
    {
        uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    // uniform in [0,1) with 24 bits, exactly representable
    float next_f32() { return (float)(next_u32() >> 8) * 0x1p-24f; }
};

// reference src/main.rs:318-357.  `material_of(rn)` replaces the rng draw + threshold chain when
// `fixed` is non-null (the commented-out water platform of main.rs:220-229 passes a material).
inline std::vector<Cube> generate_platform(uint32_t id_count, Vec3 center, uint16_t size, float gap, float cube_size, Pcg32& rng,
                                           const Material* fixed = nullptr)
{
    std::vector<Cube> out;
    int32_t half_size = (int32_t)(size / 2);
    for (int32_t zi = -half_size; zi < half_size; ++zi) {
        float z = (float)zi * (cube_size + gap);
        // the inner `move` closure copies object_id: ids restart at id_count on every row (main.rs:325,332-334)
        uint32_t object_id = id_count;
        for (int32_t xi = -half_size; xi < half_size; ++xi) {
            float x = (float)xi * (cube_size + gap);
            uint32_t id = object_id;
            object_id += 1;
            Material material;
            if (fixed) {
                material = *fixed;
            } else {
                float rn = rng.next_f32();
                material = rn < 0.6f ? NETHERRACK : rn < 0.8f ? STONE : rn < 0.95f ? MAGMA : GOLD;  // main.rs:338-346
            }
            out.emplace_back(id, Vec3(x, 0.0f, z) + center, cube_size, material, Vec3(0.0f, 0.0f, 1.0f).normalize());
        }
    }
    return out;
}

// reference src/main.rs:359-410
inline std::vector<Cube> generate_rectangle(uint32_t id_count, Vec2 start, Vec2 end, float z, float cube_size, const Material& material)
{
    std::vector<Cube> out;
    int32_t endx = (int32_t)(end.x - start.x);
    int32_t endy = (int32_t)(end.y - start.y);
    for (float ypos : {start.y, end.y}) {
        uint32_t object_id = id_count;  // closure copy per flat_map item (main.rs:371-377)
        for (int32_t xi = 0; xi < endx; ++xi) {
            float xpos = (float)xi * cube_size;
            uint32_t id = object_id;
            object_id += 1;
            out.emplace_back(id, Vec3(xpos + start.x, ypos, z), cube_size, material, Vec3(0.0f, 0.0f, 1.0f).normalize());
        }
    }
    for (float xpos : {start.x, end.x}) {
        uint32_t object_id = id_count;
        for (int32_t yi = 1; yi < endy - 1; ++yi) {
            float ypos = (float)yi * cube_size;
            uint32_t id = object_id;
            object_id += 1;
            out.emplace_back(id, Vec3(xpos, ypos + start.y, z), cube_size, material, Vec3(0.0f, 0.0f, 1.0f).normalize());
        }
    }
    return out;
}

// reference src/main.rs:412-462
inline std::tuple<std::vector<Cube>, std::vector<Light>, std::vector<Sphere>> generate_portal_wall(uint32_t id_count, Vec2 start, Vec2 end,
                                                                                                float z, float cube_size,
                                                                                                const Material& material)
{
    std::vector<Cube> cubes;
    int32_t endx = (int32_t)(end.x - start.x);
    int32_t endy = (int32_t)(end.y - start.y);
    for (int32_t yi = 0; yi < endy; ++yi) {
        float ypos = (float)yi * cube_size;
        uint32_t object_id = id_count;
        for (int32_t xi = 0; xi < endx; ++xi) {
            float xpos = (float)xi * cube_size;
            uint32_t id = object_id;
            object_id += 1;
            cubes.emplace_back(id, Vec3(xpos + start.x, ypos, z), cube_size, material, Vec3(0.0f, 0.0f, 1.0f).normalize());
        }
    }
    Vec3 light_position((end.x + start.x) / 2.0f, (end.y + start.y) / 2.0f, z + 1.0f);
    std::vector<Light> lights{Light{light_position, Color::from_hex(0x361B6F), 0.25f}};
    return {cubes, lights, std::vector<Sphere>{}};
}

constexpr uint64_t kDefaultMaterialSeed = 0xC0FFEEull;

// reference src/main.rs:137-260.  `with_water` adds the commented-out 2x2 WATER platform of
// main.rs:220-229 (scene variant "minecraft+water").
inline Model init(const GameTextures& textures, uint64_t material_seed = kDefaultMaterialSeed, bool with_water = false)
{
    Pcg32 rng(material_seed);
    uint32_t object_id = 0;
    const uint16_t p_width_height = 8;
    const float cube_size = 1.5f;
    const float gap = 0.0f;

    std::vector<Cube> cubes = generate_platform(object_id, Vec3(0.0f, -cube_size * 1.0f, 0.0f), p_width_height, gap, cube_size, rng);
    object_id += (uint32_t)(cubes.size() + 1);

    Vec2 delta = Vec2(-1.0f, 0.0f) * cube_size;
    std::vector<Cube> obsidian_frame = generate_rectangle(object_id, Vec2(-1.0f, 0.0f) * cube_size + delta, Vec2(2.0f, 4.0f) * cube_size + delta,
                                                          -3.0f * cube_size, cube_size, OBSIDIAN);
    object_id += (uint32_t)(cubes.size() + 1);
    cubes.insert(cubes.end(), obsidian_frame.begin(), obsidian_frame.end());

    auto [portal_cubes, portal_lights, spheres] =
        generate_portal_wall(object_id, Vec2(-1.0f, 0.0f) * cube_size, Vec2(1.0f, 3.0f) * cube_size, -3.0f * cube_size, cube_size, PORTAL);
    object_id += (uint32_t)(cubes.size() + 1);
    object_id += (uint32_t)(spheres.size() + 1);
    cubes.insert(cubes.end(), portal_cubes.begin(), portal_cubes.end());

    const Vec3 upv = Vec3(0.0f, 1.0f, 0.0f).normalize();
    cubes.emplace_back(object_id, Vec3(0.0f, 0.0f, -0.0f) * cube_size, cube_size, GOLD, upv);
    cubes.emplace_back(object_id, Vec3(0.0f, 0.0f, -1.0f) * cube_size, cube_size, MAGMA, upv);
    cubes.emplace_back(object_id, Vec3(-1.0f, 0.0f, -1.0f) * cube_size, cube_size, MAGMA, upv);
    cubes.emplace_back(object_id, Vec3(-1.0f, 0.0f, -1.0f) * cube_size, cube_size, MAGMA, upv);
    cubes.emplace_back(object_id, Vec3(1.0f, 0.0f, 1.0f) * cube_size, cube_size, STONE, upv);

    if (with_water) {  // main.rs:220-229, un-commented
        std::vector<Cube> water = generate_platform(object_id, Vec3(0.0f, cube_size * 2.0f, 0.0f), 2, gap, cube_size, rng, &WATER);
        object_id += (uint32_t)(cubes.size() + 1);
        cubes.insert(cubes.end(), water.begin(), water.end());
    }

    Model m;
    m.spheres = spheres;
    m.cubes = cubes;
    m.camera = Camera(Vec3(0.0f, 0.0f, 10.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f));
    m.top_light = Light{Vec3(0.0f, 20.0f, 0.0f), Color::white(), 1.0f};
    m.lights = portal_lights;
    m.ambient_light = 0.15f;
    m.textures = textures;
    m.daytime = TimeOfDay::Day;
    m.sky_color = DAY_COLOR;
    return m;
}

// ------------------------------------------------------------------ synthetic voxel world (no reference analogue)
// BASELINE.json config 4 / SURVEY.md section 8(d)-4.  Fully specified here:
//   * columns (ix,iz), ix,iz in [-50,50), visited in increasing (ix^2+iz^2, ix, iz);
//   * column height hcol = 1 + (H(ix>>2, iz>>2) & 3) + ((H(ix,iz) & 7) == 0), H = 32-bit integer hash;
//     cubes at iy = -4 .. -4+hcol-1, centre (ix,iy,iz)*1.5, size 1.5, id = array index;
//   * filling stops when n_cubes is reached;
//   * material draw u = rng.next_f32(): u<0.05 WATER, u<0.10 GOLD, else uniform over
//     {DIRT, STONE, OBSIDIAN, NETHERRACK, MAGMA, PORTAL, GOLD};
//   * n_spheres spheres, radius 0.5+1.5u, centre x,z = -75+150u, y = 3+12u (drawn in the order
//     radius, x, y, z), material RUBBER, id = n_cubes + index;
//   * lights, ambient, sky as the default scene; camera eye (0,25,90) -> (0,0,0), up +Y.
constexpr uint64_t kSyntheticSeed = 0x5EED5EEDull;
inline uint32_t hash2i(int32_t a, int32_t b)
{
    uint32_t h = (uint32_t)a * 0x9E3779B1u ^ ((uint32_t)b * 0x85EBCA77u + 0xC2B2AE3Du);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
inline Model synthetic_world(const GameTextures& textures, uint32_t n_cubes = 10000, uint32_t n_spheres = 256, uint64_t seed = kSyntheticSeed)
{
    Pcg32 rng(seed);
    std::vector<std::tuple<int32_t, int32_t, int32_t>> cols;
    for (int32_t ix = -50; ix < 50; ++ix)
        for (int32_t iz = -50; iz < 50; ++iz) cols.emplace_back(ix * ix + iz * iz, ix, iz);
    std::sort(cols.begin(), cols.end());
    const float cube_size = 1.5f;
    const Material* palette[7] = {&DIRT, &STONE, &OBSIDIAN, &NETHERRACK, &MAGMA, &PORTAL, &GOLD};
    Model m;
    const Vec3 upv(0.0f, 1.0f, 0.0f);
    for (auto& [r2, ix, iz] : cols) {
        (void)r2;
        if (m.cubes.size() >= n_cubes) break;
        int32_t hcol = 1 + (int32_t)(hash2i(ix >> 2, iz >> 2) & 3u) + ((hash2i(ix, iz) & 7u) == 0u ? 1 : 0);
        for (int32_t k = 0; k < hcol && m.cubes.size() < n_cubes; ++k) {
            int32_t iy = -4 + k;
            float u = rng.next_f32();
            const Material* mat;
            if (u < 0.05f) mat = &WATER;
            else if (u < 0.10f) mat = &GOLD;
            else mat = palette[rng.next_u32() % 7u];
            m.cubes.emplace_back((uint32_t)m.cubes.size(), Vec3((float)ix * cube_size, (float)iy * cube_size, (float)iz * cube_size), cube_size, *mat, upv);
        }
    }
    for (uint32_t i = 0; i < n_spheres; ++i) {
        float radius = 0.5f + 1.5f * rng.next_f32();
        float x = -75.0f + 150.0f * rng.next_f32();
        float y = 3.0f + 12.0f * rng.next_f32();
        float z = -75.0f + 150.0f * rng.next_f32();
        Sphere s;
        s.id = n_cubes + i;
        s.center = Vec3(x, y, z);
        s.radius = radius;
        s.material = RUBBER;
        m.spheres.push_back(s);
    }
    m.camera = Camera(Vec3(0.0f, 25.0f, 90.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f));
    m.top_light = Light{Vec3(0.0f, 20.0f, 0.0f), Color::white(), 1.0f};
    m.lights = {Light{Vec3(0.0f, 2.25f, -3.5f), Color::from_hex(0x361B6F), 0.25f}};
    m.ambient_light = 0.15f;
    m.textures = textures;
    m.daytime = TimeOfDay::Day;
    m.sky_color = DAY_COLOR;
    return m;
}

}  // namespace raytracer
