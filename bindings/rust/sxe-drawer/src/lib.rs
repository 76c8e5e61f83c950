//! `HipDrawer`: the four methods of the reference's `Drawer` trait (src/renderer/drawer.rs:26-31)
//! over the C ABI.  Inside the reference crate the `impl` block below becomes
//! `impl<B: hal::Backend> Drawer<B> for HipDrawer` and the plain structs are the crate's own
//! `Drawable`, `Transform` and `ObjectUniformBufferObject`; they are spelled out here so that this
//! crate stands alone.
use std::ffi::CStr;
use sxe_sys as sys;

/// src/components/mesh.rs:4-9
pub struct Mesh {
    pub key: String,
    pub vertices: Vec<sys::sxe_vertex>,
    pub indices: Vec<u32>,
    pub rendered: bool,
}

/// src/primitives/drawable.rs:7-12 with the texture resolved to an index into `set_textures`.
pub struct Drawable {
    pub mesh: Mesh,
    pub transform: sys::sxe_transform,
    pub texture: Option<u32>,
}

/// src/primitives/uniform_buffer_object.rs:42-53: one column-major mat4.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ObjectUniformBufferObject {
    pub model: [f32; 16],
}

/// Decoded RGBA8 texels (what `image::open(path)?.to_rgba()` yields, src/renderer/types.rs:164-170).
pub struct Texels {
    pub width: u32,
    pub height: u32,
    pub rgba: Vec<u8>,
}

pub struct HipDrawer {
    ctx: *mut sys::sxe_ctx,
}

// A context is externally synchronised; the engine has one render thread (src/main.rs:143).
unsafe impl Send for HipDrawer {}
unsafe impl Sync for HipDrawer {}

fn last_error(ctx: *const sys::sxe_ctx) -> String {
    unsafe { CStr::from_ptr(sys::sxe_last_error(ctx)).to_string_lossy().into_owned() }
}

fn check(ctx: *const sys::sxe_ctx, rc: i32) -> Result<(), String> {
    if rc == sys::SXE_OK {
        Ok(())
    } else {
        Err(format!("sxe error {}: {}", rc, last_error(ctx)))
    }
}

impl HipDrawer {
    /// `GfxDrawer::new` (src/renderer/drawer.rs:55): viewport extent and number of swapchain images.
    pub fn new(width: u32, height: u32, device: i32, n_images: u32) -> Result<Self, String> {
        let mut cfg = std::mem::MaybeUninit::<sys::sxe_config>::uninit();
        // fov 45, near 0.1, far 1000 (drawer.rs:248-253), clear (0.7, 0.2, 0, 1) (drawer.rs:307)
        unsafe { sys::sxe_default_config(cfg.as_mut_ptr(), width, height) };
        let mut cfg = unsafe { cfg.assume_init() };
        cfg.device = device;
        cfg.n_images = n_images;
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { sys::sxe_create(&cfg, &mut ctx) })?;
        Ok(Self { ctx })
    }

    /// `generate_images` (src/renderer/drawer.rs:216-236) with the files already decoded.
    pub fn set_textures(&mut self, textures: &[Texels]) -> Result<(), String> {
        let descs: Vec<sys::sxe_texture_desc> = textures
            .iter()
            .map(|t| sys::sxe_texture_desc { rgba: t.rgba.as_ptr(), width: t.width, height: t.height })
            .collect();
        check(self.ctx, unsafe { sys::sxe_set_textures(self.ctx, descs.as_ptr(), descs.len() as u32) })
    }

    /// The light of `SXE_SHADE_LIT`: an extension without a counterpart in the reference, whose
    /// shaders have no lighting.
    pub fn set_light(&mut self, light: &sys::sxe_light) -> Result<(), String> {
        check(self.ctx, unsafe { sys::sxe_set_light(self.ctx, light) })
    }

    /// src/renderer/drawer.rs:458-494
    pub fn update_drawables(&mut self, drawables: Vec<Drawable>) -> Result<(), String> {
        let flat: Vec<sys::sxe_drawable> = drawables
            .iter()
            .map(|d| sys::sxe_drawable {
                vertices: d.mesh.vertices.as_ptr(),
                n_vertices: d.mesh.vertices.len() as u32,
                indices: d.mesh.indices.as_ptr(),
                n_indices: d.mesh.indices.len() as u32,
                transform: d.transform,
                texture: d.texture.map(|t| t as i32).unwrap_or(-1),
                rendered: d.mesh.rendered as u32,
            })
            .collect();
        check(self.ctx, unsafe { sys::sxe_update_drawables(self.ctx, flat.as_ptr(), flat.len() as u32) })
    }

    /// src/renderer/drawer.rs:496-505; array order is `update_drawables`' order (src/main.rs:237-263).
    pub fn update_uniforms(&mut self, uniforms: Vec<ObjectUniformBufferObject>) -> Result<(), String> {
        check(self.ctx, unsafe { sys::sxe_update_uniforms(self.ctx, uniforms.as_ptr() as *const f32, uniforms.len() as u32) })
    }

    /// src/renderer/drawer.rs:507-518
    pub fn update_camera(&mut self, transform: sys::sxe_transform) -> Result<(), String> {
        check(self.ctx, unsafe { sys::sxe_update_camera(self.ctx, &transform) })
    }

    /// src/renderer/drawer.rs:412-456.  The semaphores of the Vulkan queue have no counterpart: the
    /// HIP stream orders the frames.  Panics where the reference panics (drawer.rs:453).
    pub fn draw(&mut self, image_index: usize) {
        let rc = unsafe { sys::sxe_draw(self.ctx, image_index as u32) };
        if rc != sys::SXE_OK {
            panic!("sxe_draw: {}", last_error(self.ctx));
        }
    }

    /// `Presenter::present` stand-in (src/renderer/presenter.rs:415-430): wait for the frame, read it back.
    pub fn present(&mut self, image_index: usize, width: u32, height: u32) -> Result<Vec<u8>, String> {
        let mut rgba = vec![0u8; width as usize * height as usize * 4];
        check(self.ctx, unsafe { sys::sxe_sync(self.ctx) })?;
        check(self.ctx, unsafe { sys::sxe_read_color(self.ctx, image_index as u32, rgba.as_mut_ptr(), rgba.len()) })?;
        Ok(rgba)
    }
}

impl Drop for HipDrawer {
    fn drop(&mut self) {
        unsafe { sys::sxe_destroy(self.ctx) } // src/renderer/drawer.rs:360-388
    }
}
