//! UNBUILT SOURCE (no Rust toolchain in this image): `extern "C"` bindings to `include/rtx950.h` and a
//! thin owner type.  Struct layouts mirror the header field for field (`#[repr(C)]`); the reference
//! types each one lowers are cited from the reference crate (ElrohirGT/Raytracing).
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RTX950_ABI_VERSION: u32 = 2;
pub const RT_EXEC_EVENTS: u32 = 12;
pub const RT_NOALLOC_MAX_TILES: u32 = 131072;
pub const RT_TUNE_COHERENT_SCALE: c_int = 1;
pub const RT_TUNE_SHADOW_LIST: c_int = 2;
pub const RT_TUNE_DARK_KERNEL: c_int = 3;
pub const RT_TUNE_FILL_MEMSET: c_int = 4;
pub const RT_TUNE_BVH_DEPTH_LIMIT: c_int = 5;
pub const RT_TUNE_HEAVY_CLASS: c_int = 6;
pub const RT_TUNE_HEAVY_SYNC: c_int = 7;
pub const RT_TUNE_HEAVY_SPIN_LIMIT: c_int = 8;
pub const RT_TUNE_WAVEFRONT: c_int = 9;
pub const RT_TUNE_WAVEFRONT_CAP: c_int = 10;
pub const RT_TUNE_CULL_REFINE: c_int = 11;
pub const RT_MAX_DEPTH: u32 = 3; // src/render.rs:83
pub const RT_MAX_LIGHTS: u32 = 16;
pub const RT_TEXTURE_NONE: i32 = -1; // `texture: None`, src/material.rs:6
pub const RT_NO_HIT: u32 = 0xFFFF_FFFF;

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID_ARG: c_int = -1;
pub const RT_ERR_NO_DEVICE: c_int = -2;
pub const RT_ERR_HIP: c_int = -3;
pub const RT_ERR_NO_SCENE: c_int = -4;
pub const RT_ERR_UNSUPPORTED: c_int = -5;
pub const RT_ERR_LIMIT: c_int = -6;
pub const RT_ERR_DEVICE_WAIT: c_int = -7;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct rt_vec3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

/// src/cube.rs:13-19 `Cube { id, center, size, material, bounds }` (bounds are derived at upload)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_cube {
    pub id: u32,
    pub center: rt_vec3,
    pub size: f32,
    pub material: u32,
}

/// src/sphere.rs:10-15 `Sphere { id, center, radius, material }`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_sphere {
    pub id: u32,
    pub center: rt_vec3,
    pub radius: f32,
    pub material: u32,
}

/// src/material.rs:4-18 `Material`; `texture` indexes the atlas array in `enum Textures` order or RT_TEXTURE_NONE
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct rt_material {
    pub diffuse_0rgb: u32,
    pub texture: i32,
    pub specular: f32,
    pub albedo0: f32,
    pub albedo1: f32,
    pub reflectivity: f32,
    pub transparency: f32,
    pub refractive_index: f32,
}

/// src/light.rs:5-9 `Light { position, color, intensity }`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_light {
    pub position: rt_vec3,
    pub color_0rgb: u32,
    pub intensity: f32,
}

/// src/camera.rs:5-17 (eye, center, up)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera {
    pub eye: rt_vec3,
    pub center: rt_vec3,
    pub up: rt_vec3,
}

/// Everything `render` reads from `&Model` per frame (src/lib.rs:42-52) plus the framebuffer geometry
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_frame {
    pub camera: rt_camera,
    pub width: u32,
    pub height: u32,
    pub y0: u32,
    pub y1: u32,
    pub lights: *const rt_light,
    pub n_lights: u32,
    pub ambient_light: f32,
    pub sky_0rgb: u32,
    pub max_depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_debug_planes {
    pub hit_cube: *mut u32,
    pub hit_sphere: *mut u32,
    pub tree_hash: *mut u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_counters {
    pub nearest_rays: u64,
    pub shadow_rays: u64,
    pub cube_tests: u64,
    pub sphere_tests: u64,
    pub shaded_hits: u64,
    pub ref_nearest_rays: u64,
    pub ref_shadow_rays: u64,
    pub ref_cube_tests: u64,
    pub ref_sphere_tests: u64,
    pub ref_shaded_hits: u64,
}

/// What the kernels execute for a band (`rt_count_executed`), per lane
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_exec_counters {
    pub cull_tests: u64,
    pub chunk_tests: u64,
    pub candidate_evals: u64,
    pub exact_evals: u64,
    pub bvh_steps: u64,
    pub sphere_evals: u64,
    pub queries: u64,
    pub light_iterations: u64,
    pub resolved_hits: u64,
    pub blockers: u64,
    pub child_setups: u64,
    pub primary_rays: u64,
}

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_ctx_create(device: c_int, out_ctx: *mut *mut rt_ctx) -> c_int;
    pub fn rt_ctx_destroy(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_scene_upload(
        ctx: *mut rt_ctx,
        cubes: *const rt_cube,
        n_cubes: usize,
        spheres: *const rt_sphere,
        n_spheres: usize,
        materials: *const rt_material,
        n_materials: usize,
        atlas_rgb8: *const u8,
        n_textures: usize,
        tex_w: u32,
        tex_h: u32,
        sprite: u32,
    ) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, frame: *const rt_frame, out_0rgb: *mut u32, dbg_host: *const rt_debug_planes) -> c_int;
    pub fn rt_render_device(ctx: *mut rt_ctx, frame: *const rt_frame, d_out_0rgb: *mut u32, dbg_device: *const rt_debug_planes) -> c_int;
    pub fn rt_render_device_part(
        ctx: *mut rt_ctx,
        frame: *const rt_frame,
        n_parts: u32,
        part: u32,
        d_out_0rgb: *mut u32,
        dbg_device: *const rt_debug_planes,
    ) -> c_int;
    pub fn rt_part_rows(rows: u32, n_parts: u32, part: u32) -> u32;
    pub fn rt_frame_row_extent(ctx: *mut rt_ctx, frame: *const rt_frame, out_y_first: *mut u32, out_y_end: *mut u32) -> c_int;
    pub fn rt_sync(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_count_rays(ctx: *mut rt_ctx, frame: *const rt_frame, out: *mut rt_counters) -> c_int;
    pub fn rt_count_rays_part(ctx: *mut rt_ctx, frame: *const rt_frame, n_parts: u32, part: u32, out: *mut rt_counters) -> c_int;
    pub fn rt_kernel_time_ms(ctx: *mut rt_ctx, out_avg_ms: *mut f64, out_launches: *mut u32) -> c_int;
    pub fn rt_enable_timing(ctx: *mut rt_ctx, enable: c_int) -> c_int;
    pub fn rt_set_grid_limit(ctx: *mut rt_ctx, max_workgroups: u32) -> c_int;
    pub fn rt_selftest_division(ctx: *mut rt_ctx, seed: u32, n_cases: u64, out_mismatches: *mut u64, out_cases: *mut u64) -> c_int;
    pub fn rt_debug_wavefront_items(ctx: *mut rt_ctx, level: u32, out_words: *mut u32, cap_items: usize, out_items: *mut usize) -> c_int;
    pub fn rt_get_stream(ctx: *mut rt_ctx, out_hip_stream: *mut *mut c_void) -> c_int;
    pub fn rt_count_executed(ctx: *mut rt_ctx, frame: *const rt_frame, n_parts: u32, part: u32, out: *mut rt_counters, exec: *mut rt_exec_counters) -> c_int;
    pub fn rt_set_tuning(ctx: *mut rt_ctx, key: c_int, value: f64) -> c_int;
    pub fn rt_kernel_name(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_debug_status(ctx: *mut rt_ctx, out_word: *mut u32) -> c_int;
    pub fn rt_debug_tile_costs(ctx: *mut rt_ctx, out: *mut u32, cap: usize, out_tiles: *mut usize) -> c_int;
    pub fn rt_host_alloc(bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn rt_host_free(ptr: *mut c_void) -> c_int;
    pub fn rt_render_device_noalloc(ctx: *mut rt_ctx, frame: *const rt_frame, d_out_0rgb: *mut u32) -> c_int;
    pub fn rt_render_multi(ctxs: *const *mut rt_ctx, n_ctx: u32, frame: *const rt_frame, d_frame_on_ctx0: *mut u32, out_bytes_moved: *mut u64) -> c_int;
    pub fn rt_gather_bands(
        ctxs: *const *mut rt_ctx,
        n_ctx: u32,
        d_parts: *const *const u32,
        width: u32,
        rows: u32,
        d_band_on_ctx0: *mut u32,
        out_bytes_moved: *mut u64,
    ) -> c_int;
}

/// Error of a failed call: the status code and `rt_last_error`'s message.
#[derive(Debug, Clone)]
pub struct RtError {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for RtError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rtx950 error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for RtError {}

/// The scene arrays `rt_scene_upload` takes, already lowered from the reference's `Model`
/// (src/lib.rs:42-52): cubes/spheres with material indices, the de-duplicated material table and the
/// atlases of `GameTextures` in `enum Textures` order as RGB8 (src/texture.rs:22-45,151-161).
pub struct SceneArrays<'a> {
    pub cubes: &'a [rt_cube],
    pub spheres: &'a [rt_sphere],
    pub materials: &'a [rt_material],
    pub atlas_rgb8: &'a [u8],
    pub n_textures: usize,
    pub tex_w: u32,
    pub tex_h: u32,
    pub sprite: u32,
}

/// One context = one GPU.  Replaces the body of `render(framebuffer, data)` (src/render.rs:234): upload
/// the static part of `Model` once, then `render` per frame into `Framebuffer::buffer`.
pub struct Renderer {
    ctx: *mut rt_ctx,
}

impl Renderer {
    pub fn new(device: i32) -> Result<Self, RtError> {
        let mut ctx: *mut rt_ctx = std::ptr::null_mut();
        let rc = unsafe { rt_ctx_create(device as c_int, &mut ctx) };
        if rc != RT_OK {
            return Err(Self::error(std::ptr::null(), rc));
        }
        Ok(Renderer { ctx })
    }

    fn error(ctx: *const rt_ctx, code: c_int) -> RtError {
        let p = unsafe { rt_last_error(ctx) };
        let message = if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() };
        RtError { code, message }
    }

    fn check(&self, rc: c_int) -> Result<(), RtError> {
        if rc == RT_OK {
            Ok(())
        } else {
            Err(Self::error(self.ctx, rc))
        }
    }

    pub fn upload(&mut self, s: &SceneArrays<'_>) -> Result<(), RtError> {
        assert_eq!(s.atlas_rgb8.len(), s.n_textures * (s.tex_w as usize) * (s.tex_h as usize) * 3);
        let rc = unsafe {
            rt_scene_upload(
                self.ctx,
                s.cubes.as_ptr(),
                s.cubes.len(),
                s.spheres.as_ptr(),
                s.spheres.len(),
                s.materials.as_ptr(),
                s.materials.len(),
                s.atlas_rgb8.as_ptr(),
                s.n_textures,
                s.tex_w,
                s.tex_h,
                s.sprite,
            )
        };
        self.check(rc)
    }

    /// `render(framebuffer, data)`: rows [y0, y1) of the frame into `out` (`Framebuffer::buffer` layout,
    /// 0x00RRGGBB, src/framebuffer.rs:15,105).  `lights` = `data.lights` followed by `data.top_light`
    /// (src/render.rs:241-243).
    pub fn render(&mut self, camera: rt_camera, width: u32, height: u32, y0: u32, y1: u32, lights: &[rt_light], ambient_light: f32,
                  sky_0rgb: u32, out: &mut [u32]) -> Result<(), RtError> {
        assert!(y0 < y1 && y1 <= height);
        assert_eq!(out.len(), ((y1 - y0) as usize) * (width as usize));
        let frame = rt_frame {
            camera,
            width,
            height,
            y0,
            y1,
            lights: lights.as_ptr(),
            n_lights: lights.len() as u32,
            ambient_light,
            sky_0rgb,
            max_depth: RT_MAX_DEPTH,
        };
        let rc = unsafe { rt_render(self.ctx, &frame, out.as_mut_ptr(), std::ptr::null()) };
        self.check(rc)
    }

    pub fn raw(&self) -> *mut rt_ctx {
        self.ctx
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { rt_ctx_destroy(self.ctx) };
    }
}
