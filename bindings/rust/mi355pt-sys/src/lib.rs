//! Raw declarations of `include/mi355pt.h`, one for one.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const MI355PT_OK: c_int = 0;
pub const MI355PT_ERR_ARG: c_int = -1;
pub const MI355PT_ERR_STATE: c_int = -2;
pub const MI355PT_ERR_DEVICE: c_int = -3;
pub const MI355PT_ERR_IO: c_int = -4;
pub const MI355PT_ERR_NOMEM: c_int = -5;

pub const MI355PT_BVH_SAH: u32 = 0;
pub const MI355PT_BVH_REFERENCE: u32 = 1;
pub const MI355PT_BVH_LBVH: u32 = 2;
pub const MI355PT_FLAG_GLASS_SHORT_CIRCUIT: u32 = 1 << 0;
pub const MI355PT_FLAG_COUNT: u32 = 1 << 1;

#[repr(C)]
pub struct mi355pt_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mi355pt_scene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355pt_desc {
    pub width: u32,
    pub height: u32,
    pub device: i32,
    pub base_seed: u32,
    pub bvh_mode: u32,
    pub flags: u32,
    pub stream: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355pt_stats {
    pub paths: u64,
    pub rays: u64,
    pub nodes: u64,
    pub tests: u64,
    pub accepted: u64,
    pub bounces: [u64; 5],
    pub phase: [u64; 12],
    pub render_ms: f64,
    pub resolve_ms: f64,
    pub launches: u32,
    pub bvh_nodes: u32,
    pub bvh_depth: u32,
    pub scene_bytes: u32,
    pub scene_in_lds: u32,
    pub node_format: u32,
    pub exchange_on: u32,
    pub exchange: [u64; 4],
    pub list_tiles: u32,
    pub list_candidates: u32,
    pub sky_tiles: u32,
    pub short_div: u32,
}

extern "C" {
    pub fn mi355pt_create(desc: *const mi355pt_desc, out: *mut *mut mi355pt_ctx) -> c_int;
    pub fn mi355pt_destroy(ctx: *mut mi355pt_ctx);
    pub fn mi355pt_last_error(ctx: *const mi355pt_ctx) -> *const c_char;
    pub fn mi355pt_set_scene(
        ctx: *mut mi355pt_ctx,
        spheres32: *const c_void, n_spheres: usize,
        lights16: *const c_void, n_lights: usize,
        lambertians16: *const c_void, n_lambertians: usize,
        metals16: *const c_void, n_metals: usize,
        glass4: *const c_void, n_glass: usize,
    ) -> c_int;
    pub fn mi355pt_set_bvh(ctx: *mut mi355pt_ctx, nodes48: *const c_void, n_nodes: usize) -> c_int;
    pub fn mi355pt_set_camera(ctx: *mut mi355pt_ctx, camera64: *const c_void) -> c_int;
    pub fn mi355pt_set_camera_config(ctx: *mut mi355pt_ctx, pos: *const f32, yaw_deg: f32, pitch_deg: f32, vfov_deg: f32) -> c_int;
    pub fn mi355pt_get_camera(ctx: *const mi355pt_ctx, camera64_out: *mut c_void) -> c_int;
    pub fn mi355pt_set_settings(ctx: *mut mi355pt_ctx, samples: i32, depth: i32) -> c_int;
    pub fn mi355pt_resize(ctx: *mut mi355pt_ctx, width: u32, height: u32) -> c_int;
    pub fn mi355pt_clear(ctx: *mut mi355pt_ctx) -> c_int;
    pub fn mi355pt_render(ctx: *mut mi355pt_ctx, n_frames: u32, seeds: *const u32) -> c_int;
    pub fn mi355pt_render_async(ctx: *mut mi355pt_ctx, n_frames: u32, seeds: *const u32) -> c_int;
    pub fn mi355pt_partition_frames(total: u32, parts: u32, part: u32, first: *mut u32, count: *mut u32);
    pub fn mi355pt_reduce(ctxs: *const *mut mi355pt_ctx, n: usize, root: usize) -> c_int;
    pub fn mi355pt_reduce_comm(ctx: *mut mi355pt_ctx, nccl_comm: *mut c_void, root_rank: c_int, my_rank: c_int, total_frames: i32) -> c_int;
    pub fn mi355pt_render_static(ctx: *mut mi355pt_ctx, n_frames: u32, seeds: *const u32) -> c_int;
    pub fn mi355pt_wait(ctx: *mut mi355pt_ctx) -> c_int;
    pub fn mi355pt_frames(ctx: *const mi355pt_ctx) -> i32;
    pub fn mi355pt_read_accum(ctx: *mut mi355pt_ctx, rgba: *mut f32, n_floats: usize) -> c_int;
    pub fn mi355pt_write_accum(ctx: *mut mi355pt_ctx, rgba: *const f32, n_floats: usize, frames: i32) -> c_int;
    pub fn mi355pt_to_rgb8(ctx: *mut mi355pt_ctx, frames: i32, rgb: *mut u8, n_bytes: usize) -> c_int;
    pub fn mi355pt_save_png(ctx: *mut mi355pt_ctx, frames: i32, path: *const c_char) -> c_int;
    pub fn mi355pt_stats_get(ctx: *mut mi355pt_ctx, out: *mut mi355pt_stats) -> c_int;
    pub fn mi355pt_accum_device_ptr(ctx: *mut mi355pt_ctx, ptr: *mut *mut c_void, stream: *mut *mut c_void) -> c_int;
    pub fn mi355pt_camera_build(pos: *const f32, yaw_deg: f32, pitch_deg: f32, vfov_deg: f32, aspect: f32, camera64_out: *mut c_void) -> c_int;
    pub fn mi355pt_bvh_build_reference(spheres32: *const c_void, n_spheres: usize, swap_mode: c_int, swap_seed: u32, nodes48_out: *mut c_void, cap_nodes: usize) -> i64;
    pub fn mi355pt_frame_seed(base_seed: u32, frame: u32) -> u32;
    pub fn mi355pt_postprocess_rgb8(accum_rgba: *const f32, width: u32, height: u32, frames: i32, rgb_out: *mut u8) -> c_int;
    pub fn mi355pt_write_png_rgb8(path: *const c_char, rgb: *const u8, width: u32, height: u32) -> c_int;
    pub fn mi355pt_scene_load_ron(path: *const c_char, out: *mut *mut mi355pt_scene, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn mi355pt_scene_parse_ron(text: *const c_char, len: usize, out: *mut *mut mi355pt_scene, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
    pub fn mi355pt_scene_default() -> *mut mi355pt_scene;
    pub fn mi355pt_scene_free(scene: *mut mi355pt_scene);
    pub fn mi355pt_scene_array(scene: *const mi355pt_scene, which: c_int, data: *mut *const c_void) -> usize;
    pub fn mi355pt_scene_camera(scene: *const mi355pt_scene, out6: *mut f32);
    pub fn mi355pt_set_scene_handle(ctx: *mut mi355pt_ctx, scene: *const mi355pt_scene) -> c_int;
    pub fn mi355pt_settings_load_ron(path: *const c_char, samples: *mut i32, depth: *mut i32, errbuf: *mut c_char, errbuf_len: usize) -> c_int;
}
