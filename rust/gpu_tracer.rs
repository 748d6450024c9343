//! Reference-side binding for the MI355X render path (source only: this image has no Rust toolchain, so this
//! file has not been compiled here; tests/test_rust_binding.py checks every `#[repr(C)]` struct and every
//! `extern "C"` signature below against include/rpt.h mechanically — field order, types, offsets, sizes,
//! argument lists — and `GpuTracer::new` asserts `size_of::<RptSceneDesc>()` against the loaded library).
//!
//! Drop this file into `rust-pathtracer/src/`, add `pub mod gpu_tracer;` to `lib.rs`, link with
//! `-L <repo>/rust-pathtracer_amd -l rpt_hip`, and replace `Tracer::new(scene)` by
//! `GpuTracer::new(scene)` in `renderer/src/main.rs:42`.  `pt.render(&mut buffer)` (main.rs:118) and
//! `buffer.convert_to_u8(frame)` (main.rs:122) stay as they are.
use crate::prelude::*;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RptMaterial {
    pub mask: u32, pub proc_kind: u32,
    pub rgb: [f32; 3], pub emission: [f32; 3],
    pub anisotropic: f32, pub metallic: f32, pub roughness: f32, pub subsurface: f32, pub specular_tint: f32,
    pub sheen: f32, pub sheen_tint: f32, pub clearcoat: f32, pub clearcoat_gloss: f32, pub spec_trans: f32, pub ior: f32,
    pub proc_params: [f32; 4],
    /// `Material.medium` (material.rs:16-21, 75): read only by scenes with `RPT_SCENE_MEDIA` (project-defined, include/rpt.h).
    pub medium_type: u32, pub medium_density: f32, pub medium_color: [f32; 3], pub medium_anisotropy: f32,
}
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RptSphere { pub center: [f32; 3], pub radius: f32, pub material: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RptPlane { pub normal: [f32; 3], pub point: [f32; 3], pub min_denom: f32, pub material: u32, pub max_t: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RptLight { pub light_type: u32, pub position: [f32; 3], pub emission: [f32; 3], pub u: [f32; 3], pub v: [f32; 3], pub radius: f32, pub area: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RptCamera { pub origin: [f32; 3], pub center: [f32; 3], pub fov_deg: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RptBackground { pub kind: u32, pub colour_a: [f32; 3], pub colour_b: [f32; 3], pub gamma: f32, pub scale: f32 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RptSdfPrim { pub kind: u32, pub center: [f32; 3], pub params: [f32; 2] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct RptSdf {
    pub n_prims: u32, pub max_steps: u32, pub material: u32,
    pub smooth_k: f32, pub hit_eps: f32, pub max_t: f32, pub normal_eps: f32,
    pub prims: *const RptSdfPrim,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct RptSceneDesc {
    pub abi_version: u32, pub flags: u32,
    pub camera: RptCamera, pub background: RptBackground,
    pub eps: f32, pub max_depth: u32,
    pub n_spheres: u32, pub spheres: *const RptSphere,
    pub n_planes: u32, pub planes: *const RptPlane,
    pub n_lights: u32, pub lights: *const RptLight,
    pub n_materials: u32, pub materials: *const RptMaterial,
    pub sdf: RptSdf,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct RptUniqueId { pub bytes: [c_char; 128] }

pub const RPT_ABI_VERSION: u32 = 4;
// Constants of include/rpt.h (tests/test_rust_binding.py compares every one of them with the header).
pub const RPT_OK: i32 = 0;
pub const RPT_ERR_INVALID_ARG: i32 = -1;
pub const RPT_ERR_NO_DEVICE: i32 = -2;
pub const RPT_ERR_HIP: i32 = -3;
pub const RPT_ERR_NO_SCENE: i32 = -4;
pub const RPT_ERR_UNSUPPORTED: i32 = -5;
pub const RPT_ERR_RCCL: i32 = -6;
pub const RPT_MAT_ALL: u32 = 0x1FFF;
pub const RPT_MAT_MEDIUM: u32 = 0x2000;
pub const RPT_MEDIUM_NONE: u32 = 0;
pub const RPT_MEDIUM_ABSORB: u32 = 1;
pub const RPT_MEDIUM_SCATTER: u32 = 2;
pub const RPT_MEDIUM_EMISSIVE: u32 = 3;
pub const RPT_LIGHT_RECTANGULAR: u32 = 0;
pub const RPT_LIGHT_SPHERICAL: u32 = 1;
pub const RPT_LIGHT_DISTANT: u32 = 2;
pub const RPT_BG_CONSTANT: u32 = 0;
pub const RPT_BG_GRADIENT_Y: u32 = 1;
pub const RPT_SCENE_ANYHIT_USES_MAX_DIST: u32 = 1;
pub const RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES: u32 = 2;
pub const RPT_SCENE_MEDIA: u32 = 4;
pub const RPT_RENDER_DEFAULT: u32 = 0;
pub const RPT_RENDER_NESTED_LOOPS: u32 = 1;
pub const RPT_RENDER_FAST_MATH: u32 = 2;
pub const RPT_RENDER_RUSSIAN_ROULETTE: u32 = 0x20;
pub const RPT_RENDER_SMALL_COMPACT: u32 = 0x100;
pub const RPT_RENDER_ALL_FLAGS: u32 = 0x123;

impl RptSceneDesc {
    /// All zeros (no primitives, no SDF object): the starting point of every `describe()`.
    pub fn zeroed() -> Self { unsafe { std::mem::zeroed() } }
}

#[repr(C)] pub struct RptCtx { _private: [u8; 0] }

extern "C" {
    fn rpt_abi_version() -> u32;
    fn rpt_sizeof_scene_desc() -> u32;
    fn rpt_create(out: *mut *mut RptCtx, device_id: c_int) -> c_int;
    fn rpt_create_multi(out: *mut *mut RptCtx, device_ids: *const c_int, n_devices: c_int) -> c_int;
    #[allow(dead_code)]
    fn rpt_comm_unique_id(out: *mut RptUniqueId) -> c_int;
    #[allow(dead_code)]
    fn rpt_create_rank(out: *mut *mut RptCtx, device_id: c_int, rank: c_int, world: c_int, id: *const RptUniqueId) -> c_int;
    fn rpt_set_tile_rows(ctx: *mut RptCtx, tile_rows: u32) -> c_int;
    fn rpt_destroy(ctx: *mut RptCtx);
    fn rpt_last_error(ctx: *const RptCtx) -> *const c_char;
    fn rpt_upload_scene(ctx: *mut RptCtx, scene: *const RptSceneDesc) -> c_int;
    fn rpt_scene_analytical(out: *mut RptSceneDesc) -> c_int;
    fn rpt_render(ctx: *mut RptCtx, pixels: *mut f32, width: u32, height: u32,
                  frames_done: u64, spp: u32, seed: u64, flags: u32) -> c_int;
    fn rpt_resident_render(ctx: *mut RptCtx, width: u32, height: u32, spp: u32, seed: u64, flags: u32) -> c_int;
    fn rpt_resident_upload(ctx: *mut RptCtx, pixels: *const f32, width: u32, height: u32, frames: u64) -> c_int;
    fn rpt_resident_download(ctx: *mut RptCtx, pixels: *mut f32) -> c_int;
    fn rpt_resident_download_u8(ctx: *mut RptCtx, frame: *mut u8) -> c_int;
    fn rpt_resident_frames(ctx: *const RptCtx, frames: *mut u64) -> c_int;
    fn rpt_resident_reset(ctx: *mut RptCtx) -> c_int;
    #[allow(dead_code)]
    fn rpt_render_device(ctx: *mut RptCtx, pixels_dev: *mut f32, width: u32, height: u32, frames_done: u64, spp: u32,
                         seed: u64, flags: u32, tile_rows: u32, rank: u32, world: u32, stream: *mut c_void) -> c_int;
    fn rpt_set_dispatch(ctx: *mut RptCtx, cost_order: u32, unit_rounds: u32, unit_min_spp: u32, unit_slots: u32) -> c_int;
    fn rpt_host_pin(buffer: *mut c_void, bytes: usize) -> c_int;
    fn rpt_host_unpin(buffer: *mut c_void) -> c_int;
    fn rpt_convert_to_u8(ctx: *mut RptCtx, pixels: *const f32, frame: *mut u8, width: u32, height: u32) -> c_int;
    fn rpt_denoise(ctx: *mut RptCtx, pixels: *const f32, out: *mut f32, width: u32, height: u32, iterations: u32, edge_k: f32) -> c_int;
}

/// A scene that can describe itself as data.  `trait Scene` (scene.rs:5-90) is callbacks and cannot
/// run on the device; scenes that implement only `Scene` keep using the CPU `Tracer`.
pub trait GpuScene: Scene {
    /// The backing arrays must outlive the returned descriptor (keep them in `self`).  Start from
    /// `RptSceneDesc::zeroed()` and set `abi_version = RPT_ABI_VERSION`.
    fn describe(&self) -> RptSceneDesc;
}

/// Same surface as `Tracer` (tracer.rs:5-19, :22, :629).
pub struct GpuTracer {
    ctx: *mut RptCtx,
    scene: Box<dyn GpuScene>,
    dirty: bool,
    pub seed: u64,
    pub flags: u32,
}

// The reference's `Tracer` is `Send` (it owns a `Box<dyn Scene>` and `Scene: Sync + Send`, scene.rs:5).  `GpuTracer` owns a raw
// pointer to its library context, which the compiler will not send on its own.  The library's contract (include/rpt.h,
// "threading") is the one `&mut self` already enforces: a context is used by ONE thread at a time, any thread; different contexts
// may run concurrently.  Nothing in the context is tied to the thread that created it (every entry point sets the device it needs).
unsafe impl Send for GpuTracer {}

impl GpuTracer {
    /// `Tracer::new` (tracer.rs:13-19) on GPU 0.
    pub fn new(scene: Box<dyn GpuScene>) -> Self { Self::with_devices(scene, &[0]) }

    /// The same over several GPUs of the node: `render` fans the image rows out over them inside the call,
    /// exactly where the reference fans out over rayon's threads (tracer.rs:29-32).  A device may be listed twice
    /// (`&[0, 0]`): two streams on that GPU, whose launches fill each other's tails in a redraw loop on the resident frame.
    pub fn with_devices(scene: Box<dyn GpuScene>, devices: &[i32]) -> Self {
        assert!(unsafe { rpt_abi_version() } == RPT_ABI_VERSION, "librpt_hip ABI version mismatch");
        assert!(unsafe { rpt_sizeof_scene_desc() } as usize == std::mem::size_of::<RptSceneDesc>(),
                "RptSceneDesc does not match the library's rpt_scene_desc");
        let mut ctx: *mut RptCtx = std::ptr::null_mut();
        let rc = if devices.len() == 1 { unsafe { rpt_create(&mut ctx, devices[0] as c_int) } }
                 else { unsafe { rpt_create_multi(&mut ctx, devices.as_ptr() as *const c_int, devices.len() as c_int) } };
        assert!(rc == 0, "rpt_create failed: {}", Self::err(std::ptr::null()));
        let mut t = Self { ctx, scene, dirty: true, seed: 1, flags: 0 };
        t.upload();
        t
    }

    /// Rows per cyclic block of the multi-GPU row tiling (default 2).
    pub fn set_tile_rows(&mut self, tile_rows: u32) { unsafe { rpt_set_tile_rows(self.ctx, tile_rows); } }

    fn err(ctx: *const RptCtx) -> String {
        unsafe { std::ffi::CStr::from_ptr(rpt_last_error(ctx)).to_string_lossy().into_owned() }
    }

    fn upload(&mut self) {
        let desc = self.scene.describe();
        let rc = unsafe { rpt_upload_scene(self.ctx, &desc) };
        assert!(rc == 0, "rpt_upload_scene failed: {}", Self::err(self.ctx));
        self.dirty = false;
    }

    /// Render one frame and accumulate into the pixels buffer — the contract of tracer.rs:21-123:
    /// `buffer.pixels` is updated in place, `buffer.frames` is incremented.
    pub fn render(&mut self, buffer: &mut ColorBuffer) {
        self.render_n(buffer, 1);
    }

    /// `spp` consecutive frames in one launch; bit-identical to calling `render` `spp` times.
    pub fn render_n(&mut self, buffer: &mut ColorBuffer, spp: u32) {
        if self.dirty { self.upload(); }
        let rc = unsafe {
            rpt_render(self.ctx, buffer.pixels.as_mut_ptr(), buffer.width as u32, buffer.height as u32,
                       buffer.frames as u64, spp, self.seed, self.flags)
        };
        assert!(rc == 0, "rpt_render failed: {}", Self::err(self.ctx));
        buffer.frames += spp as usize;                       // tracer.rs:121
    }

    /// The redraw handler of renderer/src/main.rs:113-124 with the ColorBuffer kept in HBM: render one more
    /// frame into the context's resident buffer and fetch the gamma-encoded u8 frame (4 bytes per pixel
    /// cross PCIe instead of 32).  `frame.len() == width * height * 4`.
    pub fn render_resident_to_u8(&mut self, width: usize, height: usize, frame: &mut [u8]) {
        assert!(frame.len() == width * height * 4);
        if self.dirty { self.upload(); }
        let rc = unsafe { rpt_resident_render(self.ctx, width as u32, height as u32, 1, self.seed, self.flags) };
        assert!(rc == 0, "rpt_resident_render failed: {}", Self::err(self.ctx));
        let rc = unsafe { rpt_resident_download_u8(self.ctx, frame.as_mut_ptr()) };
        assert!(rc == 0, "rpt_resident_download_u8 failed: {}", Self::err(self.ctx));
    }

    /// Page-lock the window's frame once (it is handed to `render_resident_to_u8` on every redraw): the 4 bytes per pixel then
    /// cross PCIe as one DMA (1080p: 0.47 instead of 1.5 ms per redraw).  The slice must stay where it is until `unpin_frame`.
    pub fn pin_frame(frame: &mut [u8]) { unsafe { rpt_host_pin(frame.as_mut_ptr() as *mut c_void, frame.len()); } }
    pub fn unpin_frame(frame: &mut [u8]) { unsafe { rpt_host_unpin(frame.as_mut_ptr() as *mut c_void); } }

    /// `ColorBuffer::convert_to_u8` (buffer.rs:55-64) on the device, for a host buffer.
    pub fn convert_to_u8(&mut self, buffer: &ColorBuffer, frame: &mut [u8]) {
        assert!(frame.len() == buffer.width * buffer.height * 4);
        let rc = unsafe { rpt_convert_to_u8(self.ctx, buffer.pixels.as_ptr(), frame.as_mut_ptr(), buffer.width as u32, buffer.height as u32) };
        assert!(rc == 0, "rpt_convert_to_u8 failed: {}", Self::err(self.ctx));
    }

    /// The project's edge-avoiding a-trous denoiser (include/rpt.h "denoiser"; the reference lists one as a Todo, Readme.md:14):
    /// a denoised copy of `buffer`'s pixels in `out` (same size).
    pub fn denoise(&mut self, buffer: &ColorBuffer, out: &mut ColorBuffer, iterations: u32, edge_k: f32) {
        assert!(out.pixels.len() == buffer.pixels.len());
        let rc = unsafe { rpt_denoise(self.ctx, buffer.pixels.as_ptr(), out.pixels.as_mut_ptr(), buffer.width as u32, buffer.height as u32, iterations, edge_k) };
        assert!(rc == 0, "rpt_denoise failed: {}", Self::err(self.ctx));
        out.frames = buffer.frames;
    }

    /// How the launches are scheduled (include/rpt.h, rpt_set_dispatch); never changes a pixel.
    pub fn set_dispatch(&mut self, cost_order: u32, unit_rounds: u32, unit_min_spp: u32, unit_slots: u32) {
        unsafe { rpt_set_dispatch(self.ctx, cost_order, unit_rounds, unit_min_spp, unit_slots); }
    }

    /// Continue a host ColorBuffer (pixels + frames) in the resident buffer.
    pub fn resident_from(&mut self, buffer: &ColorBuffer) {
        let rc = unsafe { rpt_resident_upload(self.ctx, buffer.pixels.as_ptr(), buffer.width as u32, buffer.height as u32, buffer.frames as u64) };
        assert!(rc == 0, "rpt_resident_upload failed: {}", Self::err(self.ctx));
    }

    /// Copy the resident buffer back into a host ColorBuffer (pixels and frames).
    pub fn resident_to(&mut self, buffer: &mut ColorBuffer) {
        let rc = unsafe { rpt_resident_download(self.ctx, buffer.pixels.as_mut_ptr()) };
        assert!(rc == 0, "rpt_resident_download failed: {}", Self::err(self.ctx));
        let mut f: u64 = 0;
        unsafe { rpt_resident_frames(self.ctx, &mut f) };
        buffer.frames = f as usize;
    }

    pub fn resident_reset(&mut self) { unsafe { rpt_resident_reset(self.ctx); } }

    /// Return a mutable reference to the scene (tracer.rs:629).  The caller may mutate it through `as_any`
    /// exactly as with `Tracer`; the next `render` re-describes and re-uploads it.
    pub fn scene(&mut self) -> &mut Box<dyn GpuScene> { self.dirty = true; &mut self.scene }
}

impl Drop for GpuTracer {
    fn drop(&mut self) { unsafe { rpt_destroy(self.ctx) } }
}

/// `describe()` for renderer/src/analytical.rs: the library already knows this scene.
pub fn analytical_scene_desc() -> RptSceneDesc {
    let mut d = RptSceneDesc::zeroed();
    let rc = unsafe { rpt_scene_analytical(&mut d) };
    assert!(rc == 0);
    d
}
