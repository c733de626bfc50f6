// Host time per rt_render_device call, from C (no Python in the way): how fast one thread can issue frames.
// Build: hipcc -O2 -o tools/host_issue tools/host_issue.cpp -Iinclude raytracing_amd/librtx950.so -Wl,-rpath,'$ORIGIN/../raytracing_amd'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "rtx950.h"
int main(int argc, char** argv)
{
    const uint32_t W = argc > 1 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160, parts = argc > 3 ? atoi(argv[3]) : 8;
    rt_ctx* ctx = nullptr;
    if (rt_ctx_create(0, &ctx) != RT_OK) { printf("no ctx: %s\n", rt_last_error(nullptr)); return 1; }
    std::vector<rt_cube> cubes;
    for (int k = 0; k < 64; ++k) cubes.push_back(rt_cube{(uint32_t)k, {(float)(k % 8) * 1.5f - 6.0f, -1.5f, (float)(k / 8) * 1.5f - 6.0f}, 1.5f, 0u});
    rt_material m{0xff00ffu, RT_TEXTURE_NONE, 1.0f, 0.95f, 0.05f, 0.0f, 0.0f, 1.4f};
    if (rt_scene_upload(ctx, cubes.data(), cubes.size(), nullptr, 0, &m, 1, nullptr, 0, 0, 0, 0) != RT_OK) { printf("upload: %s\n", rt_last_error(ctx)); return 1; }
    rt_light lights[2] = {{{0.f, 2.25f, -3.5f}, 0x361B6F, 0.25f}, {{0.f, 20.f, 0.f}, 0xFFFFFF, 1.0f}};
    rt_frame f{{{0.f, 0.f, 10.f}, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.f}}, W, H, 0, H, lights, 2, 0.15f, 0x87CEEB, 3};
    uint32_t* out = nullptr;
    hipMalloc((void**)&out, (size_t)W * H * 4);
    for (int mode = 0; mode < 2; ++mode) {
        if (argc > 4 && mode == 1) rt_set_tuning(ctx, atoi(argv[4]), 1.0);
        for (int rep = 0; rep < 3; ++rep) {
            const int n = 2000;
            rt_sync(ctx);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) rt_render_device_part(ctx, &f, parts, 0, out, nullptr);
            auto t1 = std::chrono::steady_clock::now();
            rt_sync(ctx);
            auto t2 = std::chrono::steady_clock::now();
            printf("mode %d: issue %.2f us/frame, with drain %.2f us/frame\n", mode, std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
        }
        if (argc <= 4) break;
    }
    rt_ctx_destroy(ctx);
    return 0;
}
