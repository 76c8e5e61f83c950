//! Declarations of `include/sxe.h`, one per C entry point, in the header's order.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const SXE_OK: c_int = 0;
pub const SXE_E_INVALID: c_int = -1;
pub const SXE_E_DEVICE: c_int = -2;
pub const SXE_E_CAPACITY: c_int = -3;
pub const SXE_E_STATE: c_int = -4;

pub const SXE_SHADE_TEXTURE: u32 = 0;
pub const SXE_SHADE_VERTEX_COLOR: u32 = 1;
pub const SXE_SHADE_COLOR_TIMES_TEXTURE: u32 = 2;
/// Flat Blinn-Phong over the texture: an extension, the reference's shaders have no lighting.
pub const SXE_SHADE_LIT: u32 = 3;

pub const SXE_FLAG_WRITE_COVERAGE: u32 = 1;
pub const SXE_FLAG_TIMING: u32 = 2;
/// The draw list as the reference binary records it, indexing flaws included (src/renderer/drawer.rs:313-347).
pub const SXE_FLAG_COMPAT_REFERENCE_INDEXING: u32 = 4;
/// Colour images in B, G, R, A byte order (the Bgra8Srgb surface format of src/renderer/presenter.rs:472-478).
pub const SXE_FLAG_BGRA: u32 = 8;
pub const SXE_FLAG_DISCARD_DEPTH: u32 = 16;

#[repr(C)]
pub struct sxe_ctx {
    _private: [u8; 0],
}

/// `Vertex` (src/primitives/vertex.rs:2-6): 32 bytes, offsets 0/12/24.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sxe_vertex {
    pub pos: [f32; 3],
    pub color: [f32; 3],
    pub uv: [f32; 2],
}

/// `Transform` (src/components/transform.rs:16-21); rotation in cgmath memory order s, x, y, z.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sxe_transform {
    pub position: [f32; 3],
    pub scale: [f32; 3],
    pub rotation: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct sxe_config {
    pub width: u32,
    pub height: u32,
    pub fov_deg: f32,
    pub z_near: f32,
    pub z_far: f32,
    pub clear: [f32; 4],
    pub device: i32,
    pub n_images: u32,
    pub scissor_y0: u32,
    pub scissor_y1: u32,
    pub shading: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct sxe_texture_desc {
    pub rgba: *const u8,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct sxe_drawable {
    pub vertices: *const sxe_vertex,
    pub n_vertices: u32,
    pub indices: *const u32,
    pub n_indices: u32,
    pub transform: sxe_transform,
    pub texture: i32,
    pub rendered: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct sxe_mesh_desc {
    pub vertices: *const sxe_vertex,
    pub n_vertices: u32,
    pub indices: *const u32,
    pub n_indices: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct sxe_instance {
    pub mesh: u32,
    pub texture: i32,
    pub rendered: u32,
    pub transform: sxe_transform,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sxe_timings {
    pub transform_ms: f32,
    pub object_ms: f32,
    pub setup_ms: f32,
    pub finish_ms: f32,
    pub raster_shade_ms: f32,
    pub frame_ms: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sxe_frame_stats {
    pub prims: u64,
    pub records: u64,
    pub bin_entries: u64,
    pub tiles: u64,
    pub instance_vertices: u64,
    pub objects: u64,
    pub tile_size: u32,
    pub redraws: u32,
}

/// Light of `SXE_SHADE_LIT` (no counterpart in the reference); see `sxe_set_light` in include/sxe.h.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sxe_light {
    pub to_light: [f32; 3],
    pub to_viewer: [f32; 3],
    pub ambient: f32,
    pub diffuse: f32,
    pub specular: f32,
    pub shininess_log2: u32,
}

extern "C" {
    pub fn sxe_default_config(cfg: *mut sxe_config, width: u32, height: u32);
    pub fn sxe_create(cfg: *const sxe_config, out: *mut *mut sxe_ctx) -> c_int;
    pub fn sxe_destroy(ctx: *mut sxe_ctx);
    pub fn sxe_last_error(ctx: *const sxe_ctx) -> *const c_char;
    pub fn sxe_version() -> *const c_char;
    pub fn sxe_set_textures(ctx: *mut sxe_ctx, textures: *const sxe_texture_desc, n: u32) -> c_int;
    pub fn sxe_update_drawables(ctx: *mut sxe_ctx, drawables: *const sxe_drawable, n: u32) -> c_int;
    pub fn sxe_update_instances(ctx: *mut sxe_ctx, meshes: *const sxe_mesh_desc, n_meshes: u32, instances: *const sxe_instance, n_instances: u32) -> c_int;
    pub fn sxe_update_uniforms(ctx: *mut sxe_ctx, models: *const f32, n: u32) -> c_int;
    pub fn sxe_update_transforms(ctx: *mut sxe_ctx, transforms: *const sxe_transform, n: u32) -> c_int;
    pub fn sxe_rotation_system(ctx: *mut sxe_ctx, seed: u64, frame: u32, delta_ms: f32) -> c_int;
    pub fn sxe_rotation_system_span(ctx: *mut sxe_ctx, seed: u64, first_frame: u32, n_frames: u32, delta_ms: f32) -> c_int;
    pub fn sxe_update_camera(ctx: *mut sxe_ctx, camera: *const sxe_transform) -> c_int;
    pub fn sxe_update_camera_matrices(ctx: *mut sxe_ctx, view: *const f32, proj: *const f32) -> c_int;
    pub fn sxe_draw(ctx: *mut sxe_ctx, image_index: u32) -> c_int;
    pub fn sxe_sync(ctx: *mut sxe_ctx) -> c_int;
    pub fn sxe_draw_views(ctx: *mut sxe_ctx, cameras: *const sxe_transform, n: u32, first_image: u32) -> c_int;
    /// draw + sync with every work buffer in the library's static device array (allocates nothing).
    pub fn sxe_draw_noalloc(ctx: *mut sxe_ctx, image_index: u32) -> c_int;
    pub fn sxe_read_color(ctx: *mut sxe_ctx, image_index: u32, rgba: *mut u8, size_bytes: usize) -> c_int;
    pub fn sxe_map_color(ctx: *mut sxe_ctx, image_index: u32, rgba: *mut *const u8) -> c_int;
    pub fn sxe_read_depth(ctx: *mut sxe_ctx, image_index: u32, depth: *mut f32, size_bytes: usize) -> c_int;
    pub fn sxe_read_coverage(ctx: *mut sxe_ctx, image_index: u32, prim_ids: *mut u32, size_bytes: usize) -> c_int;
    pub fn sxe_read_clip_positions(ctx: *mut sxe_ctx, clip: *mut f32, size_bytes: usize) -> c_int;
    pub fn sxe_read_models(ctx: *mut sxe_ctx, models: *mut f32, size_bytes: usize) -> c_int;
    pub fn sxe_read_transforms(ctx: *mut sxe_ctx, transforms: *mut sxe_transform, size_bytes: usize) -> c_int;
    pub fn sxe_color_device_ptr(ctx: *mut sxe_ctx, image_index: u32, ptr: *mut *mut c_void) -> c_int;
    pub fn sxe_stream(ctx: *mut sxe_ctx, stream: *mut *mut c_void) -> c_int;
    pub fn sxe_get_timings(ctx: *mut sxe_ctx, out: *mut sxe_timings) -> c_int;
    pub fn sxe_get_timing_totals(ctx: *mut sxe_ctx, sums: *mut sxe_timings, frames: *mut u64) -> c_int;
    pub fn sxe_reset_timing_totals(ctx: *mut sxe_ctx) -> c_int;
    pub fn sxe_set_timing_stride(ctx: *mut sxe_ctx, every: u32) -> c_int;
    pub fn sxe_read_tile_counts(ctx: *mut sxe_ctx, counts: *mut u32, size_bytes: usize) -> c_int;
    pub fn sxe_read_tile_parts(ctx: *mut sxe_ctx, parts: *mut u32, size_bytes: usize) -> c_int;
    pub fn sxe_read_tile_times(ctx: *mut sxe_ctx, clocks: *mut u64, size_bytes: usize) -> c_int;
    pub fn sxe_get_stats(ctx: *mut sxe_ctx, out: *mut sxe_frame_stats) -> c_int;
    pub fn sxe_transform_identity(t: *mut sxe_transform);
    pub fn sxe_transform_translate(t: *mut sxe_transform, x: f32, y: f32, z: f32);
    pub fn sxe_transform_rotate(t: *mut sxe_transform, x: f32, y: f32, z: f32);
    pub fn sxe_transform_to_model(t: *const sxe_transform, model: *mut f32);
    pub fn sxe_default_light(light: *mut sxe_light);
    pub fn sxe_set_light(ctx: *mut sxe_ctx, light: *const sxe_light) -> c_int;
    pub fn sxe_light_vectors(light: *const sxe_light, to_light: *mut f32, to_viewer: *mut f32, half: *mut f32) -> c_int;
    pub fn sxe_camera_matrices(camera: *const sxe_transform, fov_deg: f32, aspect: f32, z_near: f32, z_far: f32, view: *mut f32, proj: *mut f32);
    pub fn sxe_srgb_encode_host(x: *const f32, codes: *mut u8, n: usize) -> c_int;
}
