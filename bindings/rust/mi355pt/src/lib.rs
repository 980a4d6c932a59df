//! `Renderer` with the shape of the reference's (`src/renderer/mod.rs:105-320`), over the C ABI.
//! Scene data is passed as the byte blobs of the reference's `Bytes` trait, so
//! `scene.spheres.bytes()` etc. plug in directly.
use mi355pt_sys as sys;
use std::ffi::{CStr, CString};
use std::os::raw::c_int;

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "mi355pt error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}
pub type Result<T> = std::result::Result<T, Error>;

/// The five scene arrays in wire format (`SceneBuffers::new`, mod.rs:44-88).
pub struct SceneBytes<'a> {
    pub spheres: &'a [u8],     // 32 B each
    pub lights: &'a [u8],      // 16 B each
    pub lambertians: &'a [u8], // 16 B each
    pub metals: &'a [u8],      // 16 B each
    pub glass: &'a [u8],       // 4 B each
}

/// `CameraSettings` of the reference (camera.rs:3-12), angles in degrees.
#[derive(Clone, Copy)]
pub struct CameraSettings {
    pub pos: [f32; 3],
    pub yaw: f32,
    pub pitch: f32,
    pub vfov: f32,
}

pub struct Renderer {
    ctx: *mut sys::mi355pt_ctx,
    width: u32,
    height: u32,
}

// The context is thread-compatible: one owner at a time.
unsafe impl Send for Renderer {}

fn last_error(ctx: *const sys::mi355pt_ctx, code: c_int) -> Error {
    let message = unsafe { CStr::from_ptr(sys::mi355pt_last_error(ctx)) }.to_string_lossy().into_owned();
    Error { code, message }
}

impl Renderer {
    /// `Renderer::new(device, scene, settings, width, height)` (mod.rs:106-146).
    pub fn new(scene: &SceneBytes, camera: CameraSettings, samples: i32, depth: i32, width: u32, height: u32,
               device: i32, base_seed: u32) -> Result<Self> {
        let desc = sys::mi355pt_desc {
            width, height, device, base_seed,
            bvh_mode: sys::MI355PT_BVH_SAH, flags: 0, stream: std::ptr::null_mut(),
        };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::mi355pt_create(&desc, &mut ctx) };
        if rc != sys::MI355PT_OK {
            return Err(last_error(std::ptr::null(), rc));
        }
        let mut r = Renderer { ctx, width, height };
        r.reload_settings(samples, depth)?;
        r.reload_scene(scene, camera)?;
        Ok(r)
    }
    fn check(&self, rc: c_int) -> Result<()> {
        if rc == sys::MI355PT_OK { Ok(()) } else { Err(last_error(self.ctx, rc)) }
    }
    /// mod.rs:148-155: new buffers, new tree, camera from the scene, accumulator dirty.
    pub fn reload_scene(&mut self, s: &SceneBytes, camera: CameraSettings) -> Result<()> {
        self.check(unsafe {
            sys::mi355pt_set_scene(self.ctx,
                s.spheres.as_ptr().cast(), s.spheres.len() / 32,
                s.lights.as_ptr().cast(), s.lights.len() / 16,
                s.lambertians.as_ptr().cast(), s.lambertians.len() / 16,
                s.metals.as_ptr().cast(), s.metals.len() / 16,
                s.glass.as_ptr().cast(), s.glass.len() / 4)
        })?;
        self.set_camera_config(camera)
    }
    /// Keep the reference's own tree: `flatten(BVHTree::new(spheres)).bytes()` (mod.rs:73-78).
    pub fn set_bvh(&mut self, nodes48: &[u8]) -> Result<()> {
        self.check(unsafe { sys::mi355pt_set_bvh(self.ctx, nodes48.as_ptr().cast(), nodes48.len() / 48) })
    }
    /// `camera_config_mut` (mod.rs:254-257): marks the accumulator dirty.
    pub fn set_camera_config(&mut self, c: CameraSettings) -> Result<()> {
        self.check(unsafe { sys::mi355pt_set_camera_config(self.ctx, c.pos.as_ptr(), c.yaw, c.pitch, c.vfov) })
    }
    /// mod.rs:157-160: does not reset the accumulator.
    pub fn reload_settings(&mut self, samples: i32, depth: i32) -> Result<()> {
        self.check(unsafe { sys::mi355pt_set_settings(self.ctx, samples, depth) })
    }
    /// mod.rs:162-166.
    pub fn resize(&mut self, width: u32, height: u32) -> Result<()> {
        self.check(unsafe { sys::mi355pt_resize(self.ctx, width, height) })?;
        self.width = width;
        self.height = height;
        Ok(())
    }
    /// `frames` iterations of mod.rs:168-244 (+ submit + poll of offline.rs:44-52) in one call.
    pub fn render(&mut self, frames: u32) -> Result<()> {
        self.check(unsafe { sys::mi355pt_render(self.ctx, frames, std::ptr::null()) })
    }
    /// The same with caller-chosen per-frame seeds (the reference draws `rand::random()`, mod.rs:242).
    pub fn render_with_seeds(&mut self, seeds: &[u32]) -> Result<()> {
        self.check(unsafe { sys::mi355pt_render(self.ctx, seeds.len() as u32, seeds.as_ptr()) })
    }
    /// Queue frames without waiting (consecutive calls run back to back on the device); `wait` or any
    /// reading call joins.
    pub fn render_async(&mut self, frames: u32) -> Result<()> {
        self.check(unsafe { sys::mi355pt_render_async(self.ctx, frames, std::ptr::null()) })
    }
    pub fn wait(&mut self) -> Result<()> {
        self.check(unsafe { sys::mi355pt_wait(self.ctx) })
    }
    /// mod.rs:246-248.
    pub fn samples(&self) -> i32 {
        unsafe { sys::mi355pt_frames(self.ctx) }
    }
    pub fn clear(&mut self) -> Result<()> {
        self.check(unsafe { sys::mi355pt_clear(self.ctx) })
    }
    /// `start_save()` + `SaveInfo::finish(device, frames)` (mod.rs:263-315, 396-435) to `path`.
    pub fn save(&mut self, frames: i32, path: &str) -> Result<()> {
        let p = CString::new(path).map_err(|_| Error { code: sys::MI355PT_ERR_ARG, message: "path contains NUL".into() })?;
        self.check(unsafe { sys::mi355pt_save_png(self.ctx, frames, p.as_ptr()) })
    }
    /// The Rgba32Float texels, tight rows.
    pub fn read_accum(&mut self) -> Result<Vec<f32>> {
        let mut v = vec![0f32; 4 * self.width as usize * self.height as usize];
        self.check(unsafe { sys::mi355pt_read_accum(self.ctx, v.as_mut_ptr(), v.len()) })?;
        Ok(v)
    }
    pub fn stats(&mut self) -> Result<sys::mi355pt_stats> {
        let mut s = sys::mi355pt_stats::default();
        self.check(unsafe { sys::mi355pt_stats_get(self.ctx, &mut s) })?;
        Ok(s)
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { sys::mi355pt_destroy(self.ctx) }
    }
}
