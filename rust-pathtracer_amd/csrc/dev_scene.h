// dev_scene.h — device-side scene tables for "small" analytical scenes.
//
// A small scene (<= 8 spheres, 4 planes, 4 lights, 12 material patches: the
// reference's AnalyticalScene is 2/1/1/3) is passed to the kernel BY VALUE in the
// kernarg segment.  Every table index in the integrator is wave-uniform, so hipcc
// turns the reads into s_load / SGPR operands: the tables cost no VGPRs, no LDS
// traffic and no vector-memory instructions ("SGPR-resident
// scene").  Larger scenes use the LDS-tiled path (dev_scene_large.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpt.h"

// the constant address space: a load through such a pointer with a wave-uniform address is a scalar load
#ifndef RPT_CONST_AS
#define RPT_CONST_AS __attribute__((address_space(4)))
#endif

// The TYPES kernels and host share live in `rptscene`, a namespace without functions over them: the device functions (dev_math.h ...
// dev_scene_large.h) are compiled twice into one kernel — `rptdev` with the short guarded divide / square root and `rptplain` with
// hipcc's own (dev_math.h, "two passes") — and with no function here argument-dependent lookup cannot mix the two.  `rptdev` sees these names
// through a using-directive, so `rptdev::SceneSmall` stays what it was.
namespace rptscene {

constexpr int kMaxSpheres = 8;
constexpr int kMaxPlanes = 4;
constexpr int kMaxLights = 4;
constexpr int kMaxMaterials = 12;
constexpr int kMaxSdfPrims = 8;

struct DevSphere {
    float cx, cy, cz, radius;
    uint32_t material;
};

struct DevPlane {
    float nx, ny, nz;
    float px, py, pz;
    float min_denom;
    uint32_t material;
    float max_t;               // > 0: finite reach (project extension); 0: the reference's infinite plane
};

struct DevLight {
    uint32_t type;
    float px, py, pz;
    float ex, ey, ez;
    float radius, area;
    float ux, uy, uz;          // Light.u / Light.v (globals.rs:80-81): the edges of a rectangular light
    float vx, vy, vz;
};

struct DevMaterial {
    uint32_t mask, proc_kind;
    float rgb[3], emission[3];
    float anisotropic, metallic, roughness, subsurface, specular_tint, sheen, sheen_tint;
    float clearcoat, clearcoat_gloss, spec_trans, ior;
    float proc_params[4];
    uint32_t medium_type;      // Material.medium (material.rs:16-21); read only by the media kernels (WithMedia, below)
    float medium_density;
    float medium_color[3];
    float medium_anisotropy;
};

// Pinhole::gen_ray (camera/pinhole.rs:38-60) split at its frame-invariant part:
// everything up to `rd = lower_left - origin` depends only on the camera and the
// image size, so the host evaluates it once per launch with the same f32
// operation order and strict tan (host_scene.h, make_camera).
struct DevCamera {
    float ox, oy, oz;          // origin
    float rdx, rdy, rdz;       // lower_left - origin
    float hx, hy, hz;          // horizontal
    float vx, vy, vz;          // vertical
    float psx, psy;            // pixel_size = (1/width, 1/height)
};

struct DevBackground {
    uint32_t kind;
    float ax, ay, az;
    float bx, by, bz;
    float gamma, scale;
};

// One primitive of the SDF object, in the order a march step needs it: centre and first parameter (all a sphere takes) in one
// 16-byte scalar load, kind and second parameter in the next.  (With `kind` first the compiler loaded a record in three pieces and
// waited for each: two exposed scalar-cache round trips per primitive per march step.)
struct DevSdfPrim {
    float cx, cy, cz, p0;
    float p1;
    uint32_t kind;
    uint32_t pad[2];
};

// The procedural SDF object (include/rpt.h, rpt_sdf): smooth union of up to 8 primitives.
struct DevSdf {
    uint32_t n_prims, max_steps, material;
    float smooth_k, hit_eps, max_t, normal_eps;
    float inv_smooth_k;        // 1.0f / smooth_k (f32), computed by the host
    DevSdfPrim prims[kMaxSdfPrims];
};

// Scene flags above the public ones of rpt.h (rpt_scene_desc.flags): set per launch from the render flags.
constexpr uint32_t kSceneFlagRussianRoulette = 1u << 31;       // RPT_RENDER_RUSSIAN_ROULETTE

struct SceneSmall {
    static constexpr bool kMedia = false;                          // see WithMedia
    uint32_t n_spheres, n_planes, n_lights, n_materials;
    uint32_t flags, max_depth;
    float eps;
    float n_lights_f;          // number_of_lights() as F (tracer.rs:138,214)
    DevCamera cam;
    DevBackground bg;
    DevSphere spheres[kMaxSpheres];
    DevPlane planes[kMaxPlanes];
    DevLight lights[kMaxLights];
    DevMaterial materials[kMaxMaterials];
};

// A small scene that also carries the SDF object: its own type, so that the kernel for plain
// analytical scenes (the benchmark path) contains no sphere-marching code.
struct SceneSmallSdf : SceneSmall {
    DevSdf sdf;
};

// Participating media (include/rpt.h, RPT_SCENE_MEDIA; project-defined) are a COMPILE-TIME property of the scene type: the
// same tables, but the path functions (dev_integrator.h) carry the medium a path is in and act on it.  Scenes without media
// — the reference's, every benchmark scene — run kernels that contain none of that code.
template <class Base>
struct WithMedia : Base {
    static constexpr bool kMedia = true;
    WithMedia() = default;
    explicit WithMedia(const Base& b) : Base(b) {}
};

// Large scenes (dev_scene_large.h): the tables stay in global memory, the kernel argument holds pointers.
struct SceneLarge {
    static constexpr bool kMedia = false;                          // dev_scene.h, WithMedia
    uint32_t n_spheres, n_planes, n_lights, n_materials;
    uint32_t flags, max_depth;
    float eps;
    float n_lights_f;
    DevCamera cam;
    DevBackground bg;
    const float4* spheres;            // xyz = centre, w = radius
    const uint32_t* sphere_material;
    const DevLight* lights;
    const DevMaterial* materials;
    DevPlane planes[kMaxPlanes];
    // Uniform grid over the spheres (built on the host at upload, host_scene.h build_grid):
    // cell (ix,iy,iz) -> items[cell_start[c] .. cell_start[c+1]) = indices of the spheres whose
    // padded bounding box overlaps the cell, ascending.  use_accel == 0: brute-force streaming.
    uint32_t use_accel;
    uint32_t gn[3];
    float gmin[3], gmax[3], cell_size[3], inv_cell_size[3];
    float gcenter[3];
    float safe_r2;                    // rays starting farther than sqrt(safe_r2) from gcenter use the brute-force loop
    float near_r2;                    // rays starting within sqrt(near_r2) of gcenter use the second tier of cell lists (less padding:
    uint32_t near_cell_off;           // shorter), cell_start[near_cell_off + c]; near_r2 < 0: there is none
    const uint32_t* cell_start;
    const uint32_t* cell_items;
    const float4* cell_spheres;       // {centre, r * r} of spheres[cell_items[k]] stored at k: a cell's spheres are one dependent load away, not two
    uint32_t n_oversize;              // spheres kept out of the grid (far larger than the rest: host_scene.h), tested by every walk
    const uint32_t* oversize;
    // Scene::sample_lights' loop (closest_geom_finish): the spherical lights as {centre, radius} in index order, in whole groups of
    // four, with their indices into `lights`; n_light_spheres == 0xFFFFFFFF: the scene has a light of another kind that acts (a
    // rectangular one under RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES): the loop over `lights` itself runs
    const float4* light_spheres;
    const uint32_t* light_sphere_ids;
    uint32_t n_light_spheres;
};

// (seed, frame) -> the key of a frame's random streams (dev_math.h, Rng): host and device
__host__ __device__ inline uint32_t pcg_hash_hd(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// (seed, frame) -> two independently folded words: 64 bits of key per frame
struct FrameKey {
    uint32_t k0, k1;
};
__host__ __device__ inline FrameKey frame_key_hd(uint64_t seed, uint64_t frame)
{
    FrameKey fk;
    uint32_t k = pcg_hash_hd((uint32_t)(seed >> 32));
    k = pcg_hash_hd(k ^ (uint32_t)seed);
    k = pcg_hash_hd(k ^ (uint32_t)(frame >> 32));
    fk.k0 = pcg_hash_hd(k ^ (uint32_t)frame);
    uint32_t j = pcg_hash_hd((uint32_t)(seed >> 32) ^ 0x85EBCA6Bu);
    j = pcg_hash_hd(j ^ (uint32_t)seed);
    j = pcg_hash_hd(j ^ (uint32_t)(frame >> 32));
    fk.k1 = pcg_hash_hd(j ^ (uint32_t)frame);
    return fk;
}

// One launch's worth of render parameters.
struct RenderParams {
    float* pixels;             // this rank's tile buffer, rows_local * width RGBA f32
    uint32_t width, height;    // full image
    uint32_t rows_local;       // rows in `pixels`
    uint32_t tile_rows, rank, world;
    uint32_t spp;
    uint64_t frames_done;
    uint64_t seed;
    uint32_t tiles_x;          // ceil(width / 16)
    uint32_t shade_threshold;  // lanes that must be waiting before a wave runs the shading block
    uint32_t finish_threshold; // ... and before it runs the finishing block (background, blend, next camera path) instead of the fuller room
    uint32_t march_min_lanes;      // SDF scenes: a wave keeps marching while at least this many lanes are marching
    uint32_t compact;              // small scenes: the kernel that re-deals its workgroup's paths before every stage (few samples per launch)
    // Dispatch (kernel_common.h, "Dispatch: units, their order, their hand-off").  A launch of the state-machine kernels is
    // n_chunks * (tiles) workgroups; each renders chunk_spp samples (the last chunk: the rest of spp) of one tile.  n_chunks == 0:
    // a kernel without units (nested loops, the compacting kernel): workgroup b renders tile tile_order[b], all samples.
    uint32_t n_chunks, chunk_spp;
    uint32_t* sched_sync;          // ticket counter + per-tile progress (zeroed before every launch with n_chunks > 1)
    const uint32_t* tile_order;    // position -> tile, most expensive first (NULL: bottom rows first)
    uint32_t* tile_cost;           // [tile * 4 + wave] the time the wave held its slot, for the next launch's order (NULL: not recorded)
    uint32_t* tile_start;          // development (tools/dispatch_timeline.py): each wave's start stamp, like tile_cost; NULL: not recorded
};

}  // namespace rptscene

namespace rptdev { using namespace rptscene; }
namespace rptplain { using namespace rptscene; }
