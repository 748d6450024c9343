//! Reference-side binding for the MI355X render path (source only: this image has no Rust toolchain, so this
//! file has not been compiled here; tests/test_rust_binding.py checks every `#[repr(C)]` struct and every
//! `extern "C"` signature below against include/rpt.h mechanically — field order, types, offsets, sizes,
//! argument lists —, that the items INTEGRATION.md promises exist with the signatures it quotes, and — through the
//! Python mirror of `SceneDescBuilder` — that the descriptor `rust/analytical_gpu.rs` builds for the stock scene is
//! byte-identical to the library's `rpt_scene_analytical`; `GpuTracer::try_new` asserts `size_of::<RptSceneDesc>()`
//! against the loaded library).
//!
//! Drop this file into `rust-pathtracer/src/`, add `pub mod gpu_tracer;` to `lib.rs`, link with
//! `-L <repo>/rust-pathtracer_amd -l rpt_hip`; drop `rust/analytical_gpu.rs` into `renderer/src/` and replace
//! `Tracer::new(scene)` by `AutoTracer::new(scene, describer_of::<AnalyticalScene>)` in `renderer/src/main.rs:42`.
//! `pt.render(&mut buffer)` (main.rs:118) and `buffer.convert_to_u8(frame)` (main.rs:122) stay as they are; on a host
//! without a usable GPU `AutoTracer` IS the reference's `Tracer` (the CPU fallback lives here, on the Rust side: the
//! library has none).
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
pub const RPT_MAT_RGB: u32 = 0x1;
pub const RPT_MAT_EMISSION: u32 = 0x2;
pub const RPT_MAT_ANISOTROPIC: u32 = 0x4;
pub const RPT_MAT_METALLIC: u32 = 0x8;
pub const RPT_MAT_ROUGHNESS: u32 = 0x10;
pub const RPT_MAT_SUBSURFACE: u32 = 0x20;
pub const RPT_MAT_SPECULAR_TINT: u32 = 0x40;
pub const RPT_MAT_SHEEN: u32 = 0x80;
pub const RPT_MAT_SHEEN_TINT: u32 = 0x100;
pub const RPT_MAT_CLEARCOAT: u32 = 0x200;
pub const RPT_MAT_CLEARCOAT_GLOSS: u32 = 0x400;
pub const RPT_MAT_SPEC_TRANS: u32 = 0x800;
pub const RPT_MAT_IOR: u32 = 0x1000;
pub const RPT_MAT_ALL: u32 = 0x1FFF;
pub const RPT_MAT_MEDIUM: u32 = 0x2000;
pub const RPT_MEDIUM_NONE: u32 = 0;
pub const RPT_MEDIUM_ABSORB: u32 = 1;
pub const RPT_MEDIUM_SCATTER: u32 = 2;
pub const RPT_MEDIUM_EMISSIVE: u32 = 3;
pub const RPT_PROC_NONE: u32 = 0;
pub const RPT_PROC_CHECKER_DIR: u32 = 1;
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

/// What the library reports instead of unwinding across the boundary (include/rpt.h, rpt_status + rpt_last_error).
#[derive(Clone, Debug, PartialEq)]
pub struct RptError { pub status: i32, pub message: String }
impl std::fmt::Display for RptError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "rpt status {}: {}", self.status, self.message) }
}
impl std::error::Error for RptError {}
impl RptError {
    /// No usable gfx950 device (the library has no CPU fallback): what `AutoTracer` answers with the reference's `Tracer`.
    pub fn no_device(&self) -> bool { self.status == RPT_ERR_NO_DEVICE }
}

// ---- Rust-side values -> the descriptor's records ------------------------------------------------------------------

/// `Light` (globals.rs:76-84) as the library's record, field for field.
impl From<&Light> for RptLight {
    fn from(l: &Light) -> Self {
        RptLight {
            light_type: match l.light_type { LightType::Rectangular => RPT_LIGHT_RECTANGULAR, LightType::Spherical => RPT_LIGHT_SPHERICAL, LightType::Distant => RPT_LIGHT_DISTANT },
            position: [l.position.x, l.position.y, l.position.z],
            emission: [l.emission.x, l.emission.y, l.emission.z],
            u: [l.u.x, l.u.y, l.u.z], v: [l.v.x, l.v.y, l.v.z],
            radius: l.radius, area: l.area,                  // area = 4 pi r^2 was computed by AnalyticalLight::spherical (light.rs:22)
        }
    }
}
/// `AnalyticalLight` (light.rs:5-8) is a `Light` in a box.
impl From<&AnalyticalLight> for RptLight {
    fn from(l: &AnalyticalLight) -> Self { RptLight::from(&l.light) }
}

impl RptMaterial {
    /// A PATCH: the fields of `m` named by `mask` (RPT_MAT_* bits), written over whatever the material is when the primitive is
    /// accepted — what `closest_hit` does with `state.material.<field> = ...` (analytical.rs:56-58, 82-85, 115-116).  Fields outside
    /// the mask are carried along but never read.
    pub fn patch(m: &Material, mask: u32) -> Self {
        RptMaterial {
            mask, proc_kind: RPT_PROC_NONE,
            rgb: [m.rgb.x, m.rgb.y, m.rgb.z], emission: [m.emission.x, m.emission.y, m.emission.z],
            anisotropic: m.anisotropic, metallic: m.metallic, roughness: m.roughness, subsurface: m.subsurface,
            specular_tint: m.specular_tint, sheen: m.sheen, sheen_tint: m.sheen_tint, clearcoat: m.clearcoat,
            clearcoat_gloss: m.clearcoat_gloss, spec_trans: m.spec_trans, ior: m.ior,
            proc_params: [0.0; 4],
            medium_type: match m.medium.medium_type { MediumType::None => RPT_MEDIUM_NONE, MediumType::Absorb => RPT_MEDIUM_ABSORB,
                                                      MediumType::Scatter => RPT_MEDIUM_SCATTER, MediumType::Emissive => RPT_MEDIUM_EMISSIVE },
            medium_density: m.medium.density, medium_color: [m.medium.color.x, m.medium.color.y, m.medium.color.z],
            medium_anisotropy: m.medium.anisotropy,
        }
    }
    /// The same patch with every field its mask does not name set to zero (such fields are never read: canonical bytes, e.g. for
    /// comparing two descriptors).  The procedural parameters and — under RPT_MAT_MEDIUM — the medium stay.
    pub fn zero_unmasked(self) -> Self {
        let on = |bit: u32, v: f32| if self.mask & bit != 0 { v } else { 0.0 };
        let on3 = |bit: u32, v: [f32; 3]| if self.mask & bit != 0 { v } else { [0.0; 3] };
        let medium = self.mask & RPT_MAT_MEDIUM != 0;
        RptMaterial {
            mask: self.mask, proc_kind: self.proc_kind,
            rgb: on3(RPT_MAT_RGB, self.rgb), emission: on3(RPT_MAT_EMISSION, self.emission),
            anisotropic: on(RPT_MAT_ANISOTROPIC, self.anisotropic), metallic: on(RPT_MAT_METALLIC, self.metallic),
            roughness: on(RPT_MAT_ROUGHNESS, self.roughness), subsurface: on(RPT_MAT_SUBSURFACE, self.subsurface),
            specular_tint: on(RPT_MAT_SPECULAR_TINT, self.specular_tint), sheen: on(RPT_MAT_SHEEN, self.sheen),
            sheen_tint: on(RPT_MAT_SHEEN_TINT, self.sheen_tint), clearcoat: on(RPT_MAT_CLEARCOAT, self.clearcoat),
            clearcoat_gloss: on(RPT_MAT_CLEARCOAT_GLOSS, self.clearcoat_gloss), spec_trans: on(RPT_MAT_SPEC_TRANS, self.spec_trans),
            ior: on(RPT_MAT_IOR, self.ior), proc_params: self.proc_params,
            medium_type: if medium { self.medium_type } else { RPT_MEDIUM_NONE }, medium_density: if medium { self.medium_density } else { 0.0 },
            medium_color: if medium { self.medium_color } else { [0.0; 3] }, medium_anisotropy: if medium { self.medium_anisotropy } else { 0.0 },
        }
    }
    /// Every BSDF field of `m` (what scenes beyond the kernarg tables need for their spheres: include/rpt.h).
    pub fn full(m: &Material) -> Self { Self::patch(m, RPT_MAT_ALL) }
    /// ... and its `Medium` too (read only by scenes with RPT_SCENE_MEDIA).
    pub fn with_medium(mut self) -> Self { self.mask |= RPT_MAT_MEDIUM; self }
    /// The reference floor's colour (analytical.rs:107-115): `rgb = checker(dir.x / dir.y * scale + offset, dir.z / dir.y * scale + offset) ? a : b`.
    pub fn with_checker_dir(mut self, scale: f32, offset: f32, a: f32, b: f32) -> Self {
        self.proc_kind = RPT_PROC_CHECKER_DIR;
        self.proc_params = [scale, offset, a, b];
        self
    }
}

/// A scene description that OWNS its tables.  `RptSceneDesc` is plain pointers and counts (include/rpt.h); the arrays it points to
/// have to live somewhere for as long as the descriptor is used — here.  `with_desc` lends a descriptor that borrows from `self`.
#[derive(Clone, Default)]
pub struct SceneDescBuilder {
    pub flags: u32,
    pub camera: RptCamera,
    pub background: RptBackground,
    pub eps: f32,
    pub max_depth: u32,
    pub spheres: Vec<RptSphere>,
    pub planes: Vec<RptPlane>,
    pub lights: Vec<RptLight>,
    pub materials: Vec<RptMaterial>,
    pub sdf_prims: Vec<RptSdfPrim>,
    /// material, smooth_k, max_steps, hit_eps, max_t, normal_eps of the SDF object (used when `sdf_prims` is not empty)
    pub sdf: (u32, f32, u32, f32, f32, f32),
}

impl SceneDescBuilder {
    /// The reference's constants: `Tracer.eps = 0.005` (tracer.rs:16), `recursion_depth() = 4` (scene.rs:28-30), `Pinhole::new()`
    /// (pinhole.rs:16-23), a black constant background.
    pub fn new() -> Self {
        SceneDescBuilder { eps: 0.005, max_depth: 4, camera: RptCamera { origin: [0.0, 0.0, 3.0], center: [0.0, 0.0, 0.0], fov_deg: 80.0 },
                           background: RptBackground { kind: RPT_BG_CONSTANT, colour_a: [0.0; 3], colour_b: [0.0; 3], gamma: 1.0, scale: 1.0 },
                           ..Default::default() }
    }
    /// `Camera3D` has no getters (camera/mod.rs:7-18): a scene passes the numbers it configured its `Pinhole` with.
    pub fn camera(&mut self, origin: F3, center: F3, fov_deg: F) -> &mut Self {
        self.camera = RptCamera { origin: [origin.x, origin.y, origin.z], center: [center.x, center.y, center.z], fov_deg };
        self
    }
    /// `to_linear((1 - t) * a + t * b) * scale`, `t = 0.5 * (dir.y + 1)`, `to_linear = powf(gamma)` (analytical.rs:28-32, scene.rs:32-34).
    pub fn background_gradient_y(&mut self, a: F3, b: F3, gamma: F, scale: F) -> &mut Self {
        self.background = RptBackground { kind: RPT_BG_GRADIENT_Y, colour_a: [a.x, a.y, a.z], colour_b: [b.x, b.y, b.z], gamma, scale };
        self
    }
    pub fn background_constant(&mut self, c: F3, scale: F) -> &mut Self {
        self.background = RptBackground { kind: RPT_BG_CONSTANT, colour_a: [c.x, c.y, c.z], colour_b: [0.0; 3], gamma: 1.0, scale };
        self
    }
    pub fn max_depth(&mut self, depth: u16) -> &mut Self { self.max_depth = depth as u32; self }
    pub fn flags(&mut self, flags: u32) -> &mut Self { self.flags = flags; self }
    /// Adds a material, returns its index.
    pub fn material(&mut self, m: RptMaterial) -> u32 { self.materials.push(m); (self.materials.len() - 1) as u32 }
    /// Primitives are tested in the order they are added: spheres, then planes (analytical.rs:41-120).
    pub fn sphere(&mut self, center: F3, radius: F, material: u32) -> &mut Self {
        self.spheres.push(RptSphere { center: [center.x, center.y, center.z], radius, material });
        self
    }
    /// `dot(point - o, n) / dot(n, d)`, rejected when `|dot(n, d)| <= min_denom` (analytical.rs:193-204: 1e-4); `max_t` 0 = infinite.
    pub fn plane(&mut self, normal: F3, point: F3, min_denom: F, material: u32, max_t: F) -> &mut Self {
        self.planes.push(RptPlane { normal: [normal.x, normal.y, normal.z], point: [point.x, point.y, point.z], min_denom, material, max_t });
        self
    }
    pub fn light(&mut self, l: &AnalyticalLight) -> &mut Self { self.lights.push(RptLight::from(l)); self }
    /// Every light of a `Scene`, through the trait's own accessors (scene.rs:22-25).
    pub fn lights_of(&mut self, scene: &dyn Scene) -> &mut Self {
        for i in 0..scene.number_of_lights() { self.lights.push(RptLight::from(scene.light_at(i))); }
        self
    }

    /// Lends the descriptor: valid inside `f` only (it points into `self`).
    pub fn with_desc<R>(&self, f: impl FnOnce(&RptSceneDesc) -> R) -> R {
        let mut d = RptSceneDesc::zeroed();
        d.abi_version = RPT_ABI_VERSION;
        d.flags = self.flags;
        d.camera = self.camera;
        d.background = self.background;
        d.eps = self.eps;
        d.max_depth = self.max_depth;
        d.n_spheres = self.spheres.len() as u32;     d.spheres = if self.spheres.is_empty() { std::ptr::null() } else { self.spheres.as_ptr() };
        d.n_planes = self.planes.len() as u32;       d.planes = if self.planes.is_empty() { std::ptr::null() } else { self.planes.as_ptr() };
        d.n_lights = self.lights.len() as u32;       d.lights = if self.lights.is_empty() { std::ptr::null() } else { self.lights.as_ptr() };
        d.n_materials = self.materials.len() as u32; d.materials = if self.materials.is_empty() { std::ptr::null() } else { self.materials.as_ptr() };
        if !self.sdf_prims.is_empty() {
            d.sdf = RptSdf { n_prims: self.sdf_prims.len() as u32, max_steps: self.sdf.2, material: self.sdf.0, smooth_k: self.sdf.1,
                             hit_eps: self.sdf.3, max_t: self.sdf.4, normal_eps: self.sdf.5, prims: self.sdf_prims.as_ptr() };
        }
        f(&d)
    }
}

/// A scene that can describe itself as data.  `trait Scene` (scene.rs:5-90) is five callbacks that cannot run on the device;
/// what they compute for a closed set of primitives can be said as tables (include/rpt.h, "scene as data").
pub trait GpuScene: Scene {
    fn describe(&self) -> SceneDescBuilder;
}

/// How a `GpuTracer` asks the `Box<dyn Scene>` it owns for its description: through `Scene::as_any` (scene.rs:88) and a downcast to
/// the concrete scene type.  `None`: not a scene this describer knows — the caller stays on the CPU `Tracer`.
pub type Describer = fn(&mut dyn Scene) -> Option<SceneDescBuilder>;
/// The describer of one concrete `GpuScene` type: `describer_of::<AnalyticalScene>`.
pub fn describer_of<T: GpuScene + 'static>(scene: &mut dyn Scene) -> Option<SceneDescBuilder> {
    scene.as_any().downcast_ref::<T>().map(|s| s.describe())
}

/// Same surface as `Tracer` (tracer.rs:5-19, :22, :629).
pub struct GpuTracer {
    ctx: *mut RptCtx,
    scene: Box<dyn Scene>,
    describe: Describer,
    dirty: bool,
    pub seed: u64,
    pub flags: u32,
    /// What the last `render` call that could not run reported (the reference's `render` returns `()`; see `render`).
    pub last_error: Option<RptError>,
}

// The reference's `Tracer` is `Send` (it owns a `Box<dyn Scene>` and `Scene: Sync + Send`, scene.rs:5).  `GpuTracer` owns a raw
// pointer to its library context, which the compiler will not send on its own.  The library's contract (include/rpt.h,
// "threading") is the one `&mut self` already enforces: a context is used by ONE thread at a time, any thread; different contexts
// may run concurrently.  Nothing in the context is tied to the thread that created it (every entry point sets the device it needs).
unsafe impl Send for GpuTracer {}

impl GpuTracer {
    /// `Tracer::new` (tracer.rs:13-19) on GPU 0 — or the reason there is none, WITH the scene, so that the caller can hand it to
    /// the CPU `Tracer` (`AutoTracer` does).
    pub fn try_new(scene: Box<dyn Scene>, describe: Describer) -> Result<Self, (RptError, Box<dyn Scene>)> { Self::try_with_devices(scene, describe, &[0]) }

    /// The same over several GPUs of the node: `render` fans the image rows out over them inside the call,
    /// exactly where the reference fans out over rayon's threads (tracer.rs:29-32).  A device may be listed twice
    /// (`&[0, 0]`): two streams on that GPU, whose launches fill each other's tails in a redraw loop on the resident frame.
    pub fn try_with_devices(mut scene: Box<dyn Scene>, describe: Describer, devices: &[i32]) -> Result<Self, (RptError, Box<dyn Scene>)> {
        if unsafe { rpt_abi_version() } != RPT_ABI_VERSION || unsafe { rpt_sizeof_scene_desc() } as usize != std::mem::size_of::<RptSceneDesc>() {
            return Err((RptError { status: RPT_ERR_UNSUPPORTED, message: "librpt_hip: ABI version or rpt_scene_desc layout mismatch".into() }, scene));
        }
        if describe(scene.as_mut()).is_none() {
            return Err((RptError { status: RPT_ERR_UNSUPPORTED, message: "the scene cannot describe itself as data (not a GpuScene the describer knows)".into() }, scene));
        }
        let mut ctx: *mut RptCtx = std::ptr::null_mut();
        let rc = if devices.len() == 1 { unsafe { rpt_create(&mut ctx, devices[0] as c_int) } }
                 else { unsafe { rpt_create_multi(&mut ctx, devices.as_ptr() as *const c_int, devices.len() as c_int) } };
        if rc != RPT_OK { return Err((Self::error(rc, std::ptr::null()), scene)); }
        let mut t = Self { ctx, scene, describe, dirty: true, seed: 1, flags: 0, last_error: None };
        match t.upload() {
            Ok(()) => Ok(t),
            Err(e) => Err((e, t.into_scene())),
        }
    }

    /// `Tracer::new`'s signature (it cannot fail: tracer.rs:13): panics where `try_new` reports.  Prefer `AutoTracer::new`.
    pub fn new(scene: Box<dyn Scene>, describe: Describer) -> Self {
        match Self::try_new(scene, describe) { Ok(t) => t, Err((e, _)) => panic!("GpuTracer::new: {}", e) }
    }

    /// Gives the scene back (for the CPU `Tracer`) and destroys the context.
    pub fn into_scene(self) -> Box<dyn Scene> {
        let mut me = std::mem::ManuallyDrop::new(self);
        me.last_error = None;                                  // (the one other field that owns memory)
        unsafe { rpt_destroy(me.ctx); std::ptr::read(&me.scene) }
    }

    /// Rows per cyclic block of the multi-GPU row tiling (default 2).
    pub fn set_tile_rows(&mut self, tile_rows: u32) { unsafe { rpt_set_tile_rows(self.ctx, tile_rows); } }

    fn error(status: c_int, ctx: *const RptCtx) -> RptError {
        RptError { status: status as i32, message: unsafe { std::ffi::CStr::from_ptr(rpt_last_error(ctx)).to_string_lossy().into_owned() } }
    }
    fn check(&self, status: c_int) -> Result<(), RptError> { if status == RPT_OK { Ok(()) } else { Err(Self::error(status, self.ctx)) } }

    fn upload(&mut self) -> Result<(), RptError> {
        let built = (self.describe)(self.scene.as_mut())
            .ok_or_else(|| RptError { status: RPT_ERR_UNSUPPORTED, message: "the scene no longer describes itself as data".into() })?;
        let rc = built.with_desc(|d| unsafe { rpt_upload_scene(self.ctx, d) });     // the tables live in `built` for the duration of the call; the library copies them
        self.check(rc)?;
        self.dirty = false;
        Ok(())
    }

    /// Render one frame and accumulate into the pixels buffer — the contract of tracer.rs:21-123: `buffer.pixels` is updated in
    /// place, `buffer.frames` is incremented.  The reference's `render` returns `()` and cannot fail; a library error (a device lost
    /// mid-session) leaves `buffer` untouched and is kept in `last_error` — `try_render` is the same call with a `Result`, and
    /// `AutoTracer::render` continues on the CPU `Tracer` when it happens.
    pub fn render(&mut self, buffer: &mut ColorBuffer) {
        if let Err(e) = self.try_render_n(buffer, 1) { eprintln!("GpuTracer::render: {}", e); self.last_error = Some(e); }
    }
    pub fn try_render(&mut self, buffer: &mut ColorBuffer) -> Result<(), RptError> { self.try_render_n(buffer, 1) }

    /// `spp` consecutive frames in one launch; bit-identical to calling `render` `spp` times.
    pub fn try_render_n(&mut self, buffer: &mut ColorBuffer, spp: u32) -> Result<(), RptError> {
        if self.dirty { self.upload()?; }
        let rc = unsafe {
            rpt_render(self.ctx, buffer.pixels.as_mut_ptr(), buffer.width as u32, buffer.height as u32,
                       buffer.frames as u64, spp, self.seed, self.flags)
        };
        self.check(rc)?;
        buffer.frames += spp as usize;                       // tracer.rs:121
        Ok(())
    }

    /// The redraw handler of renderer/src/main.rs:113-124 with the ColorBuffer kept in HBM: render one more
    /// frame into the context's resident buffer and fetch the gamma-encoded u8 frame (4 bytes per pixel
    /// cross PCIe instead of 32).  `frame.len() == width * height * 4`.
    pub fn render_resident_to_u8(&mut self, width: usize, height: usize, frame: &mut [u8]) -> Result<(), RptError> {
        assert!(frame.len() == width * height * 4);
        if self.dirty { self.upload()?; }
        self.check(unsafe { rpt_resident_render(self.ctx, width as u32, height as u32, 1, self.seed, self.flags) })?;
        self.check(unsafe { rpt_resident_download_u8(self.ctx, frame.as_mut_ptr()) })
    }

    /// Page-lock the window's frame once (it is handed to `render_resident_to_u8` on every redraw): the 4 bytes per pixel then
    /// cross PCIe as one DMA (1080p: 0.47 instead of 1.5 ms per redraw).  The slice must stay where it is until `unpin_frame`.
    pub fn pin_frame(frame: &mut [u8]) { unsafe { rpt_host_pin(frame.as_mut_ptr() as *mut c_void, frame.len()); } }
    pub fn unpin_frame(frame: &mut [u8]) { unsafe { rpt_host_unpin(frame.as_mut_ptr() as *mut c_void); } }

    /// `ColorBuffer::convert_to_u8` (buffer.rs:55-64) on the device, for a host buffer.
    pub fn convert_to_u8(&mut self, buffer: &ColorBuffer, frame: &mut [u8]) -> Result<(), RptError> {
        assert!(frame.len() == buffer.width * buffer.height * 4);
        self.check(unsafe { rpt_convert_to_u8(self.ctx, buffer.pixels.as_ptr(), frame.as_mut_ptr(), buffer.width as u32, buffer.height as u32) })
    }

    /// The project's edge-avoiding a-trous denoiser (include/rpt.h "denoiser"; the reference lists one as a Todo, Readme.md:14):
    /// a denoised copy of `buffer`'s pixels in `out` (same size).
    pub fn denoise(&mut self, buffer: &ColorBuffer, out: &mut ColorBuffer, iterations: u32, edge_k: f32) -> Result<(), RptError> {
        assert!(out.pixels.len() == buffer.pixels.len());
        self.check(unsafe { rpt_denoise(self.ctx, buffer.pixels.as_ptr(), out.pixels.as_mut_ptr(), buffer.width as u32, buffer.height as u32, iterations, edge_k) })?;
        out.frames = buffer.frames;
        Ok(())
    }

    /// How the launches are scheduled (include/rpt.h, rpt_set_dispatch); never changes a pixel.
    pub fn set_dispatch(&mut self, cost_order: u32, unit_rounds: u32, unit_min_spp: u32, unit_slots: u32) {
        unsafe { rpt_set_dispatch(self.ctx, cost_order, unit_rounds, unit_min_spp, unit_slots); }
    }

    /// Continue a host ColorBuffer (pixels + frames) in the resident buffer.
    pub fn resident_from(&mut self, buffer: &ColorBuffer) -> Result<(), RptError> {
        self.check(unsafe { rpt_resident_upload(self.ctx, buffer.pixels.as_ptr(), buffer.width as u32, buffer.height as u32, buffer.frames as u64) })
    }

    /// Copy the resident buffer back into a host ColorBuffer (pixels and frames).
    pub fn resident_to(&mut self, buffer: &mut ColorBuffer) -> Result<(), RptError> {
        self.check(unsafe { rpt_resident_download(self.ctx, buffer.pixels.as_mut_ptr()) })?;
        let mut f: u64 = 0;
        self.check(unsafe { rpt_resident_frames(self.ctx, &mut f) })?;
        buffer.frames = f as usize;
        Ok(())
    }

    pub fn resident_reset(&mut self) { unsafe { rpt_resident_reset(self.ctx); } }

    /// Return a mutable reference to the scene — `Tracer::scene`'s own type (tracer.rs:629).  The caller may mutate it through
    /// `as_any` exactly as with `Tracer`; the next `render` describes and uploads it again.
    pub fn scene(&mut self) -> &mut Box<dyn Scene> { self.dirty = true; &mut self.scene }
}

impl Drop for GpuTracer {
    fn drop(&mut self) { unsafe { rpt_destroy(self.ctx) } }
}

/// `Tracer::new / render / scene` on the GPU when there is one, on the reference's own CPU `Tracer` when there is not: no usable
/// gfx950 device (RPT_ERR_NO_DEVICE), a scene the describer does not know, a library that does not match — or a GPU that fails
/// later, in which case the session goes on where it was (the ColorBuffer IS the state: buffer.rs:6-14).  The fallback lives here,
/// in Rust, because the library has none by design.
pub enum AutoTracer {
    Gpu(GpuTracer),
    Cpu(Tracer),
    /// (only while `render` moves the scene from one to the other)
    Moving,
}

impl AutoTracer {
    pub fn new(scene: Box<dyn Scene>, describe: Describer) -> Self {
        match GpuTracer::try_new(scene, describe) {
            Ok(t) => AutoTracer::Gpu(t),
            Err((e, scene)) => { eprintln!("AutoTracer: rendering on the CPU ({})", e); AutoTracer::Cpu(Tracer::new(scene)) }
        }
    }
    pub fn backend(&self) -> &'static str { match self { AutoTracer::Gpu(_) => "gpu", _ => "cpu" } }

    /// tracer.rs:22.
    pub fn render(&mut self, buffer: &mut ColorBuffer) {
        let failed = match self {
            AutoTracer::Gpu(t) => t.try_render(buffer).err(),
            AutoTracer::Cpu(t) => { t.render(buffer); None }
            AutoTracer::Moving => unreachable!(),
        };
        if let Some(e) = failed {
            eprintln!("AutoTracer: the GPU path failed ({}); continuing on the CPU", e);
            if let AutoTracer::Gpu(t) = std::mem::replace(self, AutoTracer::Moving) {
                let mut cpu = Tracer::new(t.into_scene());
                cpu.render(buffer);                              // the frame the GPU did not render
                *self = AutoTracer::Cpu(cpu);
            }
        }
    }

    /// tracer.rs:629.
    pub fn scene(&mut self) -> &mut Box<dyn Scene> {
        match self { AutoTracer::Gpu(t) => t.scene(), AutoTracer::Cpu(t) => t.scene(), AutoTracer::Moving => unreachable!() }
    }
}

/// The library's built-in copy of renderer/src/analytical.rs as a descriptor (its arrays are static): what
/// `rust/analytical_gpu.rs` builds from the Rust-side values must equal byte for byte (tests/test_rust_binding.py).
pub fn analytical_scene_desc() -> RptSceneDesc {
    let mut d = RptSceneDesc::zeroed();
    let rc = unsafe { rpt_scene_analytical(&mut d) };
    assert!(rc == 0);
    d
}
