// host_c_api.cpp — flat C entry points over the C++ host mirror (raytracer.hpp, scene.hpp) so the
// Python test/bench harness exercises the same host code a C++ application would use.  Not part of
// the drop-in boundary (that is include/rtx950.h); these `rth_*` names are harness plumbing.
#include <memory>
#include <string>

#include "png_loader.hpp"
#include "raytracer.hpp"
#include "scene.hpp"

using namespace raytracer;

namespace {
thread_local std::string g_err;
struct HostModel {
    Model model;
    std::unique_ptr<Renderer> renderer;
};
template <typename F>
int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
GameTextures load_textures(const char* path)
{
    std::string p(path ? path : "");
    if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".bin") == 0) return GameTextures::from_baked(p);
    return load_game_textures_png(p);  // reference behaviour: asset dir prefix, texture.rs:48-57
}
}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

void* rth_model_default(const char* atlas_path, uint64_t material_seed, int with_water)
{
    HostModel* h = nullptr;
    if (guarded([&] {
            h = new HostModel();
            h->model = init(load_textures(atlas_path), material_seed, with_water != 0);
        }) != 0) {
        delete h;
        return nullptr;
    }
    return h;
}

void* rth_model_synthetic(const char* atlas_path, uint32_t n_cubes, uint32_t n_spheres, uint64_t seed)
{
    HostModel* h = nullptr;
    if (guarded([&] {
            h = new HostModel();
            h->model = synthetic_world(load_textures(atlas_path), n_cubes, n_spheres, seed);
        }) != 0) {
        delete h;
        return nullptr;
    }
    return h;
}

void rth_model_free(void* hm) { delete (HostModel*)hm; }

int rth_model_counts(void* hm, uint32_t* n_cubes, uint32_t* n_spheres, uint32_t* n_materials, uint32_t* n_lights)
{
    return guarded([&] {
        HostModel* h = (HostModel*)hm;
        LoweredScene s = lower_scene(h->model);
        *n_cubes = (uint32_t)s.cubes.size();
        *n_spheres = (uint32_t)s.spheres.size();
        *n_materials = (uint32_t)s.materials.size();
        *n_lights = (uint32_t)h->model.lights.size() + 1u;
    });
}

int rth_model_lower(void* hm, rt_cube* cubes, rt_sphere* spheres, rt_material* mats, rt_light* lights, rt_camera* cam, float* ambient,
                    uint32_t* sky)
{
    return guarded([&] {
        HostModel* h = (HostModel*)hm;
        LoweredScene s = lower_scene(h->model);
        std::copy(s.cubes.begin(), s.cubes.end(), cubes);
        std::copy(s.spheres.begin(), s.spheres.end(), spheres);
        std::copy(s.materials.begin(), s.materials.end(), mats);
        std::vector<rt_light> l = lower_lights(h->model);
        std::copy(l.begin(), l.end(), lights);
        *cam = h->model.camera.c();
        *ambient = h->model.ambient_light;
        *sky = h->model.sky_color.to_hex();
    });
}

// n_textures images of tex_w * tex_h * 3 bytes (the common layout of the model's atlases); `out` may be
// null to query the geometry, otherwise `cap` is its capacity in bytes and is checked
int rth_model_atlas(void* hm, uint8_t* out, size_t cap, uint32_t* n_textures, uint32_t* tex_w, uint32_t* tex_h, uint32_t* sprite)
{
    return guarded([&] {
        HostModel* h = (HostModel*)hm;
        const GameTextures::Layout l = h->model.textures.common_layout();
        *n_textures = (uint32_t)kTextureCount;
        *tex_w = l.width;
        *tex_h = l.height;
        *sprite = l.sprite;
        if (out) {
            std::vector<uint8_t> a = h->model.textures.to_rgb8();
            if (cap < a.size()) throw std::runtime_error("rth_model_atlas: buffer too small");
            std::copy(a.begin(), a.end(), out);
        }
    });
}

// kind: 0 RotateCamera(a,b)  1 ZoomCamera(a)  2 MoveFocus(v)  3 TimeToDay  4 TimeToNight   (lib.rs:54-60)
int rth_model_update(void* hm, int kind, float a, float b, float vx, float vy, float vz)
{
    return guarded([&] {
        HostModel* h = (HostModel*)hm;
        Message m;
        m.kind = (Message::Kind)kind;
        m.a = a;
        m.b = b;
        m.v = Vec3(vx, vy, vz);
        update(h->model, m);
    });
}

// render(framebuffer, data) through the C++ host class -> rt_render -> GPU.  out: W*H u32.
int rth_render(void* hm, uint32_t width, uint32_t height, uint32_t* out, uint32_t* hit_cube, uint32_t* hit_sphere, uint32_t* tree_hash)
{
    return guarded([&] {
        HostModel* h = (HostModel*)hm;
        if (!h->renderer) h->renderer = std::make_unique<Renderer>(0);
        Framebuffer fb(width, height);
        rt_debug_planes dbg{hit_cube, hit_sphere, tree_hash};
        h->renderer->render(fb, h->model, (hit_cube || hit_sphere || tree_hash) ? &dbg : nullptr);
        std::copy(fb.buffer.begin(), fb.buffer.end(), out);
    });
}

size_t rth_bmp_encode(const uint32_t* buffer, size_t width, size_t height, uint8_t* out, size_t cap)
{
    std::vector<uint8_t> img = bmp::encode(buffer, width, height);
    if (out && cap >= img.size()) std::copy(img.begin(), img.end(), out);
    return img.size();
}

uint32_t rth_color_mul(uint32_t c, float f) { return (Color::from_hex(c) * f).to_hex(); }
uint32_t rth_color_add(uint32_t a, uint32_t b) { return (Color::from_hex(a) + Color::from_hex(b)).to_hex(); }
uint32_t rth_color_sub(uint32_t a, uint32_t b) { return (Color::from_hex(a) - Color::from_hex(b)).to_hex(); }

void rth_camera_rotate(rt_camera* cam, float dyaw, float dpitch)
{
    Camera c(Vec3(cam->eye.x, cam->eye.y, cam->eye.z), Vec3(cam->center.x, cam->center.y, cam->center.z), Vec3(cam->up.x, cam->up.y, cam->up.z));
    c.rotate_cam(dyaw, dpitch);
    *cam = c.c();
}

// paint_point on a fresh w*h framebuffer; returns the PaintPointErrors code (0 = Ok)
int rth_paint_point(uint32_t w, uint32_t h, float x, float y)
{
    Framebuffer fb(w, h);
    return (int)fb.paint_point(Vec2(x, y));
}

// Decode one PNG into RGB8 (w*h*3 bytes).  Returns 0 on success.
int rth_png_decode(const char* path, uint8_t* out, size_t cap, uint32_t* w, uint32_t* h)
{
    return guarded([&] {
        PngImage img = png_decode_file(path);
        *w = img.width;
        *h = img.height;
        if (out) {
            if (cap < img.rgb.size()) throw std::runtime_error("rth_png_decode: buffer too small");
            std::copy(img.rgb.begin(), img.rgb.end(), out);
        }
    });
}

}  // extern "C"
