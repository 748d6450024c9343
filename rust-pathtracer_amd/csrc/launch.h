// launch.h — the seam between the host-side C ABI (capi.hip) and the kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_scene.h"
#include "dev_scene_large.h"
#include "dev_wavefront.h"

#include "tile_plan.h"

namespace rptlaunch {

uint32_t max_spp_per_launch(bool sdf_object);      // samples a chunk of the state-machine kernels can hold in its LDS tables (scenes with an SDF object: fewer)

// One launch of the megakernel on `nblocks` 16x16 tiles: picks the instantiation (small / SDF / large,
// regenerating or nested) from the scene.  `small_scene_dev`: the same small scene in device memory (only the compacting SDF
// kernel, rp.sdf_resumable_march == 3, reads it; without it that mode falls back to the march kernel).
// `media`: the scene has participating media — the same forms instantiated for WithMedia<Scene> (dev_scene.h).
hipError_t render(const rptdev::SceneSmallSdf& small_scene, const rptdev::SceneLarge& large_scene, bool large, bool nested,
                  const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const rptdev::SceneSmallSdf* small_scene_dev = nullptr,
                  bool media = false);
// Large scenes with a grid, wavefront form (dev_wavefront.h): `spp` samples of every pixel of the tile; the buffers hold
// rp.rows_local * rp.width slots.  Needs at most 1 + 2 * (spp * max_depth + 1) launches; launches after the last useful
// iteration return at once, and the host never has more than 256 iterations enqueued without having looked at the device's
// "anything left?" flag (it waits for the stream there: a bound of up to 256 iterations is enqueued blind).
hipError_t render_wavefront(const rptdev::SceneLarge& sc, const rptdev::RenderParams& rp, const rptdev::WfBuffers& wb, hipStream_t st,
                            bool media = false);
// Cost-ordered dispatch (kernels.hip, block_tile): `cost` holds 4 dwords per tile, `order` one; init = bottom rows first and no
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
hipError_t probe_math(uint32_t fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t st);
hipError_t probe_fn(uint32_t fn, const rptdev::DevCamera& cam, const float* in, float* out, uint64_t n, hipStream_t st);
hipError_t probe_rays(const rptdev::SceneLarge& sc, const float* rays, uint32_t* out, uint64_t n, hipStream_t st);

}  // namespace rptlaunch

// the relaxed-arithmetic build of the same kernels (kernels.hip under -DRPT_RELAXED_BUILD)
namespace rptlaunch_fast {
hipError_t render_wavefront(const rptdev::SceneLarge& sc, const rptdev::RenderParams& rp, const rptdev::WfBuffers& wb, hipStream_t st,
                            bool media = false);
hipError_t render(const rptdev::SceneSmallSdf& small_scene, const rptdev::SceneLarge& large_scene, bool large, bool nested,
                  const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const rptdev::SceneSmallSdf* small_scene_dev = nullptr,
                  bool media = false);
}  // namespace rptlaunch_fast

// large scenes' and SDF scenes' kernels of the shipped library (kernels.hip under -DRPT_PEROP_BUILD); rptlaunch::render forwards to it
namespace rptlaunch_perop {
hipError_t render(const rptdev::SceneSmallSdf& small_scene, const rptdev::SceneLarge& large_scene, bool large, bool nested,
                  const rptdev::RenderParams& rp, uint32_t nblocks, hipStream_t st, const rptdev::SceneSmallSdf* small_scene_dev = nullptr,
                  bool media = false);
}  // namespace rptlaunch_perop
