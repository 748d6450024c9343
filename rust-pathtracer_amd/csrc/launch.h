// launch.h — the seam between the host-side C ABI (capi.hip) and the kernels: one translation unit per kernel class
// (k_small.hip, k_compact.hip, k_sdf.hip, k_large.hip: each holds its kernels and the function that launches them), the utility
// kernels (k_util.hip), the denoiser (denoise.hip) and, in the test build only, the probes (k_probes.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_scene.h"
#include "dev_scene_large.h"

#include "tile_plan.h"

// The material table of a scene of FIVE TO TWELVE primitives (round 6; k_small.hip, render_small_regen_maptable_kernel).  2^n rows per
// checker colour and side do not fit the LDS beyond four primitives — but a row is a MATERIAL, and the material of a hit depends on the
// accepted set only through which primitive wrote each field last (analytical.rs:56-58, 82-85, 115-116 are field writes): accepted sets
// with the same last writers are one class, and a scene whose primitives carry whole materials has n + 1 of them.  The host sorts the
// 2^n sets into classes (capi.hip, material_class_map); up to 16 classes x 2 colours x 2 sides = 64 rows fit.
constexpr uint32_t kMatClasses = 16u;
struct MatClassMap {
    uint32_t n_classes;
    uint16_t class_set[kMatClasses];    // one accepted set of each class: spheres in bits 0-7, planes in bits 8-11 (GeomHit.code's layout)
    const uint8_t* cls;                 // [4096] in device memory: accepted spheres | accepted planes << n_spheres (8 + 4 bits at most) -> class
};

// Which instantiation of a class's kernel a launch takes (capi.hip decides from the scene and the knobs, knobs.h).
struct KernelChoice {
    bool sized = false;             // small scenes: the reference scene's table sizes are known at compile time (kernel_common.h, sized_scene)
    uint32_t sized_sdf = 0;         // SDF scenes: 1-4 primitives over one plane under one light: that many, known at compile time; 0: data
    bool material_table = false;    // a hit's material from the workgroup's table (dev_integrator.h, MaterialTable): at most 3 primitives
    bool material_table_wide = false;   // ... at most 4: the megakernel of small scenes without an SDF object only (render_small_regen_table_kernel)
    bool material_table_mapped = false; // ... five to twelve, by class (MatClassMap): the same megakernel
    MatClassMap class_map = {};
    uint32_t extra_lds = 0;         // development: pad the headline kernel's LDS (occupancy experiments)
};

namespace rptlaunch {

uint32_t max_spp_per_launch(bool sdf_object);      // samples a chunk of the state-machine kernels can hold in its LDS tables (scenes with an SDF object: fewer)
bool material_table_fits_small(const rptdev::SceneSmallSdf& scs, bool has_sdf, uint32_t max_bits = 3u);   // (kernel_common.h, material_table_fits)

// One launch on `nblocks` workgroups (16x16 tiles x chunks of samples).  `media`: the scene has participating media — the same forms
// instantiated for WithMedia<Scene> (dev_scene.h).  `nested`: the nested-loop baseline (small scenes without media only).
hipError_t render_small(const rptdev::SceneSmall& sc, bool media, bool nested, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_compact(const rptdev::SceneSmall& sc, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_sdf(const rptdev::SceneSmallSdf& scs, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_large(const rptdev::SceneLarge& scl, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st);

// Cost-ordered dispatch (kernel_common.h, block_tile): `cost` holds 4 dwords per tile, `order` one; init = bottom rows first and no
// costs; order = the tiles sorted by the costs the last launch left, most expensive first.
hipError_t sched_init(uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st);
hipError_t sched_order(const uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st);
hipError_t untile(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t tile_rows, uint32_t world,
                  uint32_t rows_padded, hipStream_t st);
hipError_t convert_to_u8(const float* pixels, uint8_t* out, uint64_t n_pixels, hipStream_t st);
hipError_t convert_to_u8_at(const float* pixels, uint32_t bw, uint32_t bh, uint8_t* frame, uint32_t at0, uint32_t at1, uint32_t width,
                            uint32_t height, hipStream_t st);
// the denoiser (denoise.hip): `iterations` a-trous passes in -> out through `scratch` (each width * height * 4 f32)
hipError_t denoise(const float* in, float* out, float* scratch, uint32_t width, uint32_t height, uint32_t iterations, float edge_k,
                   hipStream_t st);
// test build only (k_probes.hip, include/rpt_test.h)
hipError_t probe_math(uint32_t fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t st);
hipError_t probe_fn(uint32_t fn, const rptdev::DevCamera& cam, const float* in, float* out, uint64_t n, hipStream_t st);
hipError_t probe_rays(const rptdev::SceneLarge& sc, const float* rays, uint32_t* out, uint64_t n, hipStream_t st);
// development only (-DRPT_PROFILE_BLOCKS, dev_prof.h): copy out and clear the block counters of one class's object
hipError_t prof_read_small(unsigned long long* out);
hipError_t prof_read_sdf(unsigned long long* out);
hipError_t prof_read_large(unsigned long long* out);

}  // namespace rptlaunch

// the relaxed-arithmetic build of the same four translation units (-DRPT_RELAXED_BUILD): what RPT_RENDER_FAST_MATH selects
namespace rptlaunch_fast {
hipError_t render_small(const rptdev::SceneSmall& sc, bool media, bool nested, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_compact(const rptdev::SceneSmall& sc, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_sdf(const rptdev::SceneSmallSdf& scs, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc);
hipError_t render_large(const rptdev::SceneLarge& scl, bool media, const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st);
}  // namespace rptlaunch_fast
