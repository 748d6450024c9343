//! `impl GpuScene for AnalyticalScene`: the reference's only scene (renderer/src/analytical.rs) said as data, FROM THE RUST SIDE'S
//! OWN VALUES — the lights through `Scene::light_at` (they are `self.lights`, analytical.rs:15-16, 149-155), the depth through
//! `recursion_depth()`, and what `closest_hit` / `background` hold as literals restated next to the lines they come from — not from
//! the library's built-in `rpt_scene_analytical` (which exists for C callers; tests/test_rust_binding.py proves the two byte-identical
//! through the Python mirror of `SceneDescBuilder`).  Source only: this image has no Rust toolchain.
//!
//! Drop into `renderer/src/`, add `mod analytical_gpu;` to main.rs, and at main.rs:41-42 write
//!     let scene = Box::new(AnalyticalScene::new());
//!     let mut pt = AutoTracer::new(scene, describer_of::<AnalyticalScene>);
//! `pt.render(&mut buffer)` (main.rs:118) and `buffer.convert_to_u8(frame)` (main.rs:122) stay as they are.
use crate::analytical::AnalyticalScene;
use rust_pathtracer::gpu_tracer::*;
use rust_pathtracer::prelude::*;

impl GpuScene for AnalyticalScene {
    fn describe(&self) -> SceneDescBuilder {
        let mut b = SceneDescBuilder::new();                       // eps 0.005 (tracer.rs:16); Pinhole::new() (pinhole.rs:16-23): the scene's camera
                                                                   // is `Box::new(Pinhole::new())` (analytical.rs:20) and Camera3D has no getters
        b.max_depth(self.recursion_depth());                       // scene.rs:28-30 (not overridden by AnalyticalScene)
        // analytical.rs:28-32: t = 0.5 * (dir.y + 1); to_linear((1 - t) * (1, 1, 1) + t * (0.5, 0.7, 1.0)) * 0.5; to_linear = powf(2.2) (scene.rs:32-34)
        b.background_gradient_y(F3::new(1.0, 1.0, 1.0), F3::new(0.5, 0.7, 1.0), 2.2, 0.5);
        // Materials are what closest_hit WRITES when it accepts a primitive, over whatever state.material holds then
        // (Material::new() at the start of a bounce, tracer.rs:63; an earlier primitive's writes after that): patches.
        let mut left = Material::new();                            // analytical.rs:56-58
        left.rgb = F3::new_x(1.0);
        left.roughness = 0.05;
        left.metallic = 1.0;
        let left = b.material(RptMaterial::patch(&left, RPT_MAT_RGB | RPT_MAT_ROUGHNESS | RPT_MAT_METALLIC));
        let mut right = Material::new();                           // analytical.rs:82-85
        right.rgb = F3::new(1.0, 0.186, 0.0);
        right.clearcoat = 1.0;
        right.clearcoat_gloss = 1.0;
        right.roughness = 0.1;
        let right = b.material(RptMaterial::patch(&right, RPT_MAT_RGB | RPT_MAT_CLEARCOAT | RPT_MAT_CLEARCOAT_GLOSS | RPT_MAT_ROUGHNESS));
        let mut floor = Material::new();                           // analytical.rs:107-116: rgb from the checker, roughness 1
        floor.roughness = 1.0;
        let floor = b.material(RptMaterial::patch(&floor, RPT_MAT_ROUGHNESS).with_checker_dir(0.5, 100.0, 0.25, 0.1));
        // Fields a patch does not name are never read; the library's own descriptor leaves them zero, and so does this one
        // (`Material::new()` defaults such as rgb 1.5 / ior 1.45 would be dead bytes that differ).
        for m in b.materials.iter_mut() { *m = m.zero_unmasked(); }
        b.sphere(F3::new(-1.1, 0.0, 0.0), 1.0, left);              // analytical.rs:41
        b.sphere(F3::new(1.1, 0.0, 0.0), 1.0, right);              // analytical.rs:70
        b.plane(F3::new(0.0, 1.0, 0.0), F3::new(0.0, -1.0, 0.0), 0.0001, floor, 0.0);   // analytical.rs:193-204
        b.lights_of(self);                                         // analytical.rs:15-16 through scene.rs:22-25
        b
    }
}
