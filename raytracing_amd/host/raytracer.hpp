// raytracer.hpp — C++ host mirror of the reference crate's public surface for the render path.
//
// The reference host is Rust; no Rust toolchain exists in the build image, so the host side above
// the C ABI (include/rtx950.h) is C++ with the reference's names, argument meaning and error
// behaviour.  Each type cites the reference item it mirrors (paths relative to the reference root).
// Geometry/colour arithmetic done on the host (camera orbit, scene builders, BMP) follows the same
// f32 operation order as the reference so results are bit-identical; compile without -ffast-math
// and with -ffp-contract=off.
//
// The only way pixels are produced is raytracer::render() -> rt_render() -> the HIP kernel.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtx950.h"

namespace raytracer {

// ------------------------------------------------------------------ glm::Vec3 / Vec2 subset
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    Vec3 operator*(float s) const { return {x * s, y * s, z * s}; }
    Vec3& operator+=(const Vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    float dot(const Vec3& o) const { return (x * o.x + y * o.y) + z * o.z; }
    Vec3 cross(const Vec3& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    float magnitude() const { return std::sqrt(dot(*this)); }
    Vec3 normalize() const { float n = magnitude(); return {x / n, y / n, z / n}; }
    rt_vec3 c() const { return rt_vec3{x, y, z}; }
};
inline Vec3 operator*(float s, const Vec3& v) { return v * s; }
struct Vec2 {
    float x = 0, y = 0;
    Vec2() = default;
    Vec2(float x_, float y_) : x(x_), y(y_) {}
    Vec2 operator+(const Vec2& o) const { return {x + o.x, y + o.y}; }
    Vec2 operator*(float s) const { return {x * s, y * s}; }
};

// ------------------------------------------------------------------ Color (reference src/color.rs)
struct Color {
    uint8_t r = 0, g = 0, b = 0;
    constexpr Color() = default;
    constexpr Color(uint8_t r_, uint8_t g_, uint8_t b_) : r(r_), g(g_), b(b_) {}
    static constexpr Color black() { return {0, 0, 0}; }        // color.rs:14
    static constexpr Color white() { return {255, 255, 255}; }  // color.rs:17
    static constexpr Color pink() { return {0xff, 0x00, 0xff}; }  // color.rs:21
    static constexpr Color from_hex(uint32_t v)                 // color.rs:49-55
    {
        return {(uint8_t)((v >> 16) & 0xFF), (uint8_t)((v >> 8) & 0xFF), (uint8_t)(v & 0xFF)};
    }
    constexpr uint32_t to_hex() const { return ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b; }  // color.rs:76-79
    bool operator==(const Color& o) const { return to_hex() == o.to_hex(); }                             // color.rs:39-44
    Color operator+(const Color& o) const { return {sat_add(r, o.r), sat_add(g, o.g), sat_add(b, o.b)}; }  // color.rs:99-116
    Color operator-(const Color& o) const { return {sat_sub(r, o.r), sat_sub(g, o.g), sat_sub(b, o.b)}; }  // color.rs:132-149
    Color operator*(float f) const { return {mul(r, f), mul(g, f), mul(b, f)}; }                           // color.rs:118-130
private:
    static uint8_t sat_add(uint8_t a, uint8_t b) { unsigned s = (unsigned)a + b; return s > 255 ? 255 : (uint8_t)s; }
    static uint8_t sat_sub(uint8_t a, uint8_t b) { return a > b ? (uint8_t)(a - b) : 0; }
    static uint8_t mul(uint8_t c, float f)
    {
        float x = (float)c * f;
        if (x < 0.0f) x = 0.0f;      // f32::clamp, NaN passes
        if (x > 255.0f) x = 255.0f;
        if (!(x == x)) return 0;      // `NaN as u8` == 0
        return (uint8_t)x;            // truncation
    }
};

// ------------------------------------------------------------------ textures (reference src/texture.rs)
enum class Textures : int32_t { DIRT = 0, STONE, MOSS, WATER, OBSIDIAN, PORTAL, NETHERRACK, MAGMA, GOLD };  // texture.rs:35-45
constexpr int kTextureCount = 9;
inline const char* texture_file_stem(Textures t)  // texture.rs:49-57
{
    static const char* names[kTextureCount] = {"dirt", "stone", "moss", "water", "obsidian", "portal", "netherrack", "magma", "gold"};
    return names[(int)t];
}
enum class CubeFace { NONE, TOP, BOTTOM, FORWARDS, BACKWARDS, LEFT, RIGHT };  // texture.rs:12-20

struct Texture {  // texture.rs:98-103
    uint32_t width = 0, height = 0;
    size_t sprite_size = 16;
    std::vector<Color> colors;  // row-major, idx = y*width + x (texture.rs:158)
    static Texture from_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h, size_t sprite)
    {
        Texture t;
        t.width = w; t.height = h; t.sprite_size = sprite;
        t.colors.resize((size_t)w * h);
        for (size_t i = 0; i < t.colors.size(); ++i) t.colors[i] = Color(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        return t;
    }
    Color get_pixel_color(uint32_t x, uint32_t y) const { return colors.at((size_t)y * width + x); }  // texture.rs:171-174
};

// reference src/texture.rs:22-32,47-95.  The reference decodes 9 PNGs from `asset_dir`; this build
// ships the decoded texels as one blob (tools/bake_textures.py) and, when a PNG decoder is
// available (png_loader.hpp), decodes the same files itself.
struct GameTextures {
    Texture tex[kTextureCount];
    const Texture& get_texture(Textures t) const { return tex[(int)t]; }  // texture.rs:82-94
    // blob = kTextureCount images of w*h*3 bytes in `enum Textures` order
    static GameTextures from_baked(const std::string& path, uint32_t w = 48, uint32_t h = 64, size_t sprite = 16)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open baked atlas: " + path);  // reference: unwrap() panic, texture.rs:142
        std::vector<uint8_t> buf((size_t)kTextureCount * w * h * 3);
        f.read((char*)buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) throw std::runtime_error("baked atlas has wrong size: " + path);
        GameTextures g;
        for (int i = 0; i < kTextureCount; ++i) g.tex[i] = Texture::from_rgb8(buf.data() + (size_t)i * w * h * 3, w, h, sprite);
        return g;
    }
    // The C ABI takes ONE width/height/sprite for all atlases, while the reference keeps them per Texture
    // (texture.rs:98-103,141-169) and indexes each with its own width (texture.rs:171-174).  Atlases of
    // different sizes (a user's asset directory) are padded, top-left aligned, to the largest width and
    // height: texel (x, y) keeps its coordinates, so every face lookup (x < 3*sprite, y < 4*sprite) reads
    // what the reference reads - provided each atlas holds the whole cross layout, which is checked.
    struct Layout { uint32_t width = 0, height = 0, sprite = 0; };
    Layout common_layout() const
    {
        Layout l;
        l.sprite = (uint32_t)tex[0].sprite_size;
        for (const Texture& t : tex) {
            if (t.sprite_size != tex[0].sprite_size) throw std::runtime_error("GameTextures: atlases differ in sprite size");
            if (t.colors.size() != (size_t)t.width * t.height) throw std::runtime_error("GameTextures: an atlas holds fewer texels than width*height");
            if (t.width < 3u * t.sprite_size || t.height < 4u * t.sprite_size)
                throw std::runtime_error("GameTextures: an atlas is smaller than the 3x4-sprite cross layout (texture.rs:178-196 would index outside it)");
            l.width = std::max(l.width, t.width);
            l.height = std::max(l.height, t.height);
        }
        return l;
    }
    // kTextureCount images of layout.width x layout.height x 3 bytes (smaller atlases padded with black)
    std::vector<uint8_t> to_rgb8() const
    {
        const Layout l = common_layout();
        std::vector<uint8_t> out((size_t)kTextureCount * l.width * l.height * 3, 0);
        for (int i = 0; i < kTextureCount; ++i) {
            const Texture& t = tex[i];
            uint8_t* img = out.data() + (size_t)i * l.width * l.height * 3;
            for (uint32_t y = 0; y < t.height; ++y)
                for (uint32_t x = 0; x < t.width; ++x) {
                    const Color& c = t.colors[(size_t)y * t.width + x];
                    uint8_t* px = img + ((size_t)y * l.width + x) * 3;
                    px[0] = c.r; px[1] = c.g; px[2] = c.b;
                }
        }
        return out;
    }
};

// ------------------------------------------------------------------ Material (reference src/material.rs)
struct Material {
    Color diffuse;
    int32_t texture = RT_TEXTURE_NONE;  // Option<Textures>
    float specular = 0;
    float albedo0 = 0, albedo1 = 0;
    float reflectivity = 0, transparency = 0, refractive_index = 0;
    bool operator==(const Material& o) const
    {
        return diffuse == o.diffuse && texture == o.texture && specular == o.specular && albedo0 == o.albedo0 &&
               albedo1 == o.albedo1 && reflectivity == o.reflectivity && transparency == o.transparency &&
               refractive_index == o.refractive_index;
    }
    rt_material c() const
    {
        return rt_material{diffuse.to_hex(), texture, specular, albedo0, albedo1, reflectivity, transparency, refractive_index};
    }
};
inline Material make_material(Textures t, float spec, float a0, float a1, float refl, float transp, float ior)
{
    Material m;
    m.diffuse = Color::pink(); m.texture = (int32_t)t; m.specular = spec; m.albedo0 = a0; m.albedo1 = a1;
    m.reflectivity = refl; m.transparency = transp; m.refractive_index = ior;
    return m;
}
// material.rs:34-122
inline const Material DIRT = make_material(Textures::DIRT, 1.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.42f);
inline const Material STONE = make_material(Textures::STONE, 1.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.42f);
inline const Material WATER = make_material(Textures::WATER, 100.0f, 0.8f, 0.1f, 0.1f, 0.9f, 1.25f);
inline const Material OBSIDIAN = make_material(Textures::OBSIDIAN, 100.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.25f);
inline const Material PORTAL = make_material(Textures::PORTAL, 150.0f, 0.95f, 0.05f, 0.0f, 0.4f, 1.2f);
inline const Material NETHERRACK = make_material(Textures::NETHERRACK, 100.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.2f);
inline const Material MAGMA = make_material(Textures::MAGMA, 1.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.2f);
inline const Material GOLD = make_material(Textures::GOLD, 2.0f, 0.5f, 0.5f, 0.1f, 0.0f, 1.2f);
inline const Material RUBBER = [] {
    Material m = make_material(Textures::DIRT, 1.0f, 0.9f, 0.1f, 0.0f, 0.0f, 1.51f);
    m.texture = RT_TEXTURE_NONE;
    return m;
}();

// ------------------------------------------------------------------ Light (reference src/light.rs)
struct Light {
    Vec3 position;
    Color color;
    float intensity = 0;
    rt_light c() const { return rt_light{position.c(), color.to_hex(), intensity}; }
};
using AmbientLightIntensity = float;

// ------------------------------------------------------------------ Camera (reference src/camera.rs)
struct Camera {
    Vec3 eye, center, up;
    Camera() = default;
    Camera(Vec3 e, Vec3 c, Vec3 u) : eye(e), center(c), up(u), has_changed_(true) {}  // camera.rs:25-32
    Vec3 change_basis(const Vec3& v) const  // camera.rs:33-41
    {
        Vec3 forward = (center - eye).normalize();
        Vec3 right = forward.cross(up).normalize();
        Vec3 upv = right.cross(forward).normalize();
        Vec3 cb = v.x * right + v.y * upv - v.z * forward;
        return cb.normalize();
    }
    void zoom_cam(float delta) { has_changed_ = true; Vec3 f = (center - eye).normalize(); eye += f * delta; }  // camera.rs:44-48
    void zoom(float delta) { zoom_cam(delta); }                                                                // camera.rs:51-55
    void rotate_cam(float delta_yaw, float delta_pitch)  // camera.rs:61-86
    {
        const float PI = 3.14159265358979323846f;
        has_changed_ = true;
        Vec3 rv = eye - center;
        float radius = rv.magnitude();
        float current_yaw = std::atan2(rv.z, rv.x);
        float radius_xz = std::sqrt(rv.x * rv.x + rv.z * rv.z);
        float current_pitch = std::atan2(-rv.y, radius_xz);
        float new_yaw = std::fmod(current_yaw + delta_yaw, 2.0f * PI);
        float new_pitch = current_pitch + delta_pitch;
        float lo = -PI / 2.0f + 0.1f, hi = PI / 2.0f - 0.1f;
        if (new_pitch < lo) new_pitch = lo;
        if (new_pitch > hi) new_pitch = hi;
        eye = center + Vec3(radius * std::cos(new_yaw) * std::cos(new_pitch), -radius * std::sin(new_pitch),
                            radius * std::sin(new_yaw) * std::cos(new_pitch));
    }
    bool has_changed() const { return has_changed_; }      // camera.rs:88-90
    void reset_change() { has_changed_ = false; }          // camera.rs:92-94
    void move_focus(Vec3 d) { has_changed_ = true; center += d; }  // camera.rs:96-99
    Vec3 direction() const { return (center - eye).normalize(); }  // camera.rs:101-103
    rt_camera c() const { return rt_camera{eye.c(), center.c(), up.c()}; }
private:
    bool has_changed_ = true;
};

// ------------------------------------------------------------------ Cube / Sphere (reference src/cube.rs, src/sphere.rs)
struct BoxBounds { Vec3 min, max; };
struct Cube {
    uint32_t id = 0;
    Vec3 center;
    float size = 0;
    Material material;
    BoxBounds bounds;
    Cube() = default;
    Cube(uint32_t id_, Vec3 center_, float size_, Material material_, Vec3 up)  // cube.rs:35-45
        : id(id_), center(center_), size(size_), material(material_), bounds(compute_bounds(center_, up, size_)) {}
    static BoxBounds compute_bounds(const Vec3& c, const Vec3& /*up: ignored, as in the reference*/, float size)  // cube.rs:51-66
    {
        float half = size / 2.0f;
        return BoxBounds{Vec3(c.x - half, c.y - half, c.z - half), Vec3(c.x + half, c.y + half, c.z + half)};
    }
    bool operator==(const Cube& o) const { return id == o.id; }  // cube.rs:69-73
};
struct Sphere {
    uint32_t id = 0;
    Vec3 center;
    float radius = 0;
    Material material;
    bool operator==(const Sphere& o) const { return id == o.id; }  // sphere.rs:17-21
};

// ------------------------------------------------------------------ BMP (reference src/bmp.rs)
namespace bmp {
inline size_t compute_padding_bytes_per_row(size_t width)  // bmp.rs:108-113
{
    size_t row = width * 3;
    return (size_t)std::floor((4.0f * std::ceil((float)row / 4.0f)) - (float)row);
}
inline std::vector<uint8_t> encode(const uint32_t* buffer, size_t width, size_t height)
{
    const size_t pad = compute_padding_bytes_per_row(width);
    std::vector<uint8_t> flat;  // pad_buffer, bmp.rs:75-94
    flat.reserve(width * height * 3 + height * pad);
    for (size_t i = 0; i < width * height; ++i) {
        Color c = Color::from_hex(buffer[i]);
        flat.push_back(c.b); flat.push_back(c.g); flat.push_back(c.r);
        if ((i + 1) % width == 0 && i != 0) flat.insert(flat.end(), pad, (uint8_t)0);
    }
    std::vector<uint8_t> out(54, 0);  // generate_header, bmp.rs:38-72
    auto put32 = [&](size_t off, uint32_t v) { std::memcpy(out.data() + off, &v, 4); };
    out[0] = 'B'; out[1] = 'M';
    put32(2, (uint32_t)(54 + flat.size()));
    put32(10, 54);
    put32(14, 40);
    put32(18, (uint32_t)width);
    put32(22, (uint32_t)height);
    out[26] = 1;
    put32(28, 24);
    put32(34, (uint32_t)flat.size());
    const size_t chunk = width * 3 + pad;  // rows reversed, bmp.rs:99-105
    const size_t nchunks = chunk ? (flat.size() + chunk - 1) / chunk : 0;
    for (size_t ci = nchunks; ci-- > 0;) {
        size_t start = ci * chunk, len = std::min(chunk, flat.size() - start);
        out.insert(out.end(), flat.begin() + (long)start, flat.begin() + (long)(start + len));
    }
    return out;
}
inline void write_bmp_file(const std::string& path, const uint32_t* buffer, size_t width, size_t height)  // bmp.rs:19-35
{
    std::vector<uint8_t> img = encode(buffer, width, height);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot create " + path);
    f.write((const char*)img.data(), (std::streamsize)img.size());
}
}  // namespace bmp

// ------------------------------------------------------------------ Framebuffer (reference src/framebuffer.rs)
enum class PaintPointErrors { Ok, XTooLarge, XTooSmall, YTooLarge, YTooSmall };  // framebuffer.rs:24-29
struct Framebuffer {
    size_t width, height;
    std::vector<uint32_t> buffer;
    Framebuffer(size_t w, size_t h)  // framebuffer.rs:44-58
        : width(w), height(h), buffer(w * h, 0u), background_color_(Color::black()), current_color_(Color::white()), empty_buffer_(w * h, 0u) {}
    void clear() { buffer = empty_buffer_; }                    // framebuffer.rs:63-65
    void set_current_color(Color c) { current_color_ = c; }     // framebuffer.rs:207-209
    void set_background_color(Color c)                          // framebuffer.rs:191-202
    {
        background_color_ = c;
        empty_buffer_.assign(width * height, c.to_hex());
    }
    PaintPointErrors paint_point(Vec2 p)  // framebuffer.rs:79-109
    {
        if (p.x < 0.0f) return PaintPointErrors::XTooSmall;
        if (p.y < 0.0f) return PaintPointErrors::YTooSmall;
        size_t x = (size_t)std::round(p.x), y = (size_t)std::round(p.y);
        if (!(x < width)) return PaintPointErrors::XTooLarge;
        if (!(y < height)) return PaintPointErrors::YTooLarge;
        buffer[y * width + x] = current_color_.to_hex();
        return PaintPointErrors::Ok;
    }
    void save(const std::string& path) const { bmp::write_bmp_file(path, buffer.data(), width, height); }  // framebuffer.rs:212-221
    Color background_color() const { return background_color_; }
private:
    Color background_color_, current_color_;
    std::vector<uint32_t> empty_buffer_;
};

// ------------------------------------------------------------------ Model / Message (reference src/lib.rs:37-60)
enum class TimeOfDay { Day, Night };
constexpr Color DAY_COLOR = Color(0x87, 0xCE, 0xEB);    // render.rs:231
constexpr Color NIGHT_COLOR = Color(0x03, 0x03, 0x05);  // render.rs:232

struct Model {
    std::vector<Sphere> spheres;
    std::vector<Cube> cubes;
    std::vector<Light> lights;
    AmbientLightIntensity ambient_light = 0;
    Camera camera;
    GameTextures textures;
    TimeOfDay daytime = TimeOfDay::Day;
    Color sky_color = DAY_COLOR;
    Light top_light;
    // bumped by the host whenever spheres/cubes/textures change, so render() re-uploads the scene
    uint64_t scene_version = 1;
};

struct Message {
    enum Kind { RotateCamera, ZoomCamera, MoveFocus, TimeToDay, TimeToNight } kind;
    float a = 0, b = 0;
    Vec3 v;
};

// reference src/main.rs:262-316
inline void update(Model& data, const Message& msg)
{
    switch (msg.kind) {
    case Message::RotateCamera: data.camera.rotate_cam(msg.a, msg.b); break;
    case Message::ZoomCamera: data.camera.zoom_cam(msg.a); break;
    case Message::MoveFocus: data.camera.move_focus(msg.v); break;
    case Message::TimeToDay:
        data.daytime = TimeOfDay::Day;
        data.sky_color = DAY_COLOR;
        data.top_light = Light{Vec3(0.0f, 20.0f, 0.0f), Color::white(), 1.0f};
        break;
    case Message::TimeToNight:
        data.daytime = TimeOfDay::Night;
        data.sky_color = NIGHT_COLOR;
        data.top_light = Light{Vec3(0.0f, 20.0f, 0.0f), Color::black(), 0.5f};
        break;
    }
}

// ------------------------------------------------------------------ lowering Model -> C ABI arrays
struct LoweredScene {
    std::vector<rt_cube> cubes;
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    std::vector<uint8_t> atlas;
    uint32_t tex_w = 0, tex_h = 0, sprite = 0;
};
inline uint32_t intern_material(std::vector<Material>& table, const Material& m)
{
    for (size_t i = 0; i < table.size(); ++i)
        if (table[i] == m) return (uint32_t)i;
    table.push_back(m);
    return (uint32_t)(table.size() - 1);
}
inline LoweredScene lower_scene(const Model& data)
{
    LoweredScene s;
    std::vector<Material> table;
    for (const Cube& c : data.cubes) s.cubes.push_back(rt_cube{c.id, c.center.c(), c.size, intern_material(table, c.material)});
    for (const Sphere& sp : data.spheres) s.spheres.push_back(rt_sphere{sp.id, sp.center.c(), sp.radius, intern_material(table, sp.material)});
    if (table.empty()) table.push_back(RUBBER);
    for (const Material& m : table) s.materials.push_back(m.c());
    const GameTextures::Layout layout = data.textures.common_layout();  // throws on atlases that cannot share one
    s.atlas = data.textures.to_rgb8();
    s.tex_w = layout.width;
    s.tex_h = layout.height;
    s.sprite = layout.sprite;
    return s;
}
// `lights` ++ [`top_light`]  (render.rs:241-243)
inline std::vector<rt_light> lower_lights(const Model& data)
{
    std::vector<rt_light> l;
    for (const Light& x : data.lights) l.push_back(x.c());
    l.push_back(data.top_light.c());
    return l;
}

// ------------------------------------------------------------------ render (reference src/render.rs:234, :13)
class Renderer {
public:
    explicit Renderer(int device = 0)
    {
        if (rt_ctx_create(device, &ctx_) != RT_OK) throw std::runtime_error(std::string("rt_ctx_create: ") + rt_last_error(nullptr));
    }
    ~Renderer() { rt_ctx_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    rt_ctx* ctx() { return ctx_; }
    void render(Framebuffer& fb, const Model& data, const rt_debug_planes* dbg = nullptr)
    {
        if (uploaded_version_ != data.scene_version || uploaded_model_ != &data) {
            LoweredScene s = lower_scene(data);
            check(rt_scene_upload(ctx_, s.cubes.data(), s.cubes.size(), s.spheres.data(), s.spheres.size(), s.materials.data(),
                                  s.materials.size(), s.atlas.data(), kTextureCount, s.tex_w, s.tex_h, s.sprite),
                  "rt_scene_upload");
            uploaded_version_ = data.scene_version;
            uploaded_model_ = &data;
        }
        std::vector<rt_light> lights = lower_lights(data);
        rt_frame f{};
        f.camera = data.camera.c();
        f.width = (uint32_t)fb.width;
        f.height = (uint32_t)fb.height;
        f.y0 = 0;
        f.y1 = (uint32_t)fb.height;
        f.lights = lights.data();
        f.n_lights = (uint32_t)lights.size();
        f.ambient_light = data.ambient_light;
        f.sky_0rgb = data.sky_color.to_hex();
        f.max_depth = RT_MAX_DEPTH;
        // render.rs:235 clears, :314-323 repaints every pixel but i == width; the kernel writes the
        // clear colour there, so one full overwrite reproduces clear + paint.
        check(rt_render(ctx_, &f, fb.buffer.data(), dbg), "rt_render");
    }
private:
    void check(int rc, const char* what)
    {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx_));
    }
    rt_ctx* ctx_ = nullptr;
    uint64_t uploaded_version_ = 0;
    const Model* uploaded_model_ = nullptr;
};

inline Renderer& default_renderer()
{
    static Renderer r(0);
    return r;
}
inline void render(Framebuffer& framebuffer, const Model& data) { default_renderer().render(framebuffer, data); }  // render.rs:234
inline void init_render(Framebuffer& framebuffer, const Model& data) { render(framebuffer, data); }              // render.rs:13

}  // namespace raytracer
