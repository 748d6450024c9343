// rpt.hpp — C++ host-side mirror of the reference's render API over the C ABI (rpt.h).
//
// The reference is compiled code (Rust) and this image has no Rust toolchain, so the host side
// above the C ABI is written in C++ with the reference's names, argument meaning and behaviour:
//
//   reference                                              here
//   ColorBuffer::new(w, h)            buffer.rs:18         rpt::ColorBuffer(w, h)
//   ColorBuffer::at / convert_to_u8   buffer.rs:29, :55    .at(x, y) / tracer.convert_to_u8(buffer, frame)
//   Pinhole::new/set/set_fov          pinhole.rs:14-34     rpt::Pinhole
//   AnalyticalLight::spherical        light.rs:13          rpt::AnalyticalLight::spherical
//   AnalyticalScene::new              analytical.rs:13     rpt::AnalyticalScene
//   Tracer::new / render / scene      tracer.rs:13,22,629  rpt::Tracer
//
// Errors: the reference's render() cannot fail; here a failing C-ABI call throws rpt::Error
// (status + rpt_last_error text).  Header-only; link with -lrpt_hip.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rpt.h"

namespace rpt {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& msg) : std::runtime_error("rpt status " + std::to_string(s) + ": " + msg), status(s) {}
};

struct F3 { float x, y, z; };

// buffer.rs:6-32
struct ColorBuffer {
    size_t width, height;
    std::vector<float> pixels;      // RGBA f32, row 0 = top
    size_t frames;
    ColorBuffer(size_t w, size_t h) : width(w), height(h), pixels(w * h * 4, 0.0f), frames(0) {}
    void at(size_t x, size_t y, float out[4]) const {
        size_t i = y * width * 4 + x * 4;
        for (int c = 0; c < 4; ++c) out[c] = pixels[i + c];
    }
};

// camera/pinhole.rs:6-34
struct Pinhole {
    F3 origin{0.0f, 0.0f, 3.0f}, center{0.0f, 0.0f, 0.0f};
    float fov = 80.0f;
    void set(F3 o, F3 c) { origin = o; center = c; }
    void set_fov(float f) { fov = f; }
};

// light.rs:6-28
struct AnalyticalLight {
    rpt_light light{};
    static AnalyticalLight spherical(F3 position, float radius, F3 emission) {
        AnalyticalLight l;
        l.light.type = RPT_LIGHT_SPHERICAL;
        l.light.position[0] = position.x; l.light.position[1] = position.y; l.light.position[2] = position.z;
        l.light.emission[0] = emission.x; l.light.emission[1] = emission.y; l.light.emission[2] = emission.z;
        l.light.radius = radius;
        l.light.area = 4.0f * 3.14159265358979323846f * radius * radius;      // light.rs:22
        return l;
    }
    // LightType::Rectangular / Distant (globals.rs:69-73): declared by the reference, sampled only in scenes with
    // sample_all_light_types (project-defined, see rpt.h)
    static AnalyticalLight rectangular(F3 position, F3 u, F3 v, F3 emission) {
        AnalyticalLight l;
        l.light.type = RPT_LIGHT_RECTANGULAR;
        l.light.position[0] = position.x; l.light.position[1] = position.y; l.light.position[2] = position.z;
        l.light.emission[0] = emission.x; l.light.emission[1] = emission.y; l.light.emission[2] = emission.z;
        l.light.u[0] = u.x; l.light.u[1] = u.y; l.light.u[2] = u.z;
        l.light.v[0] = v.x; l.light.v[1] = v.y; l.light.v[2] = v.z;
        const double cx = (double)u.y * v.z - (double)u.z * v.y, cy = (double)u.z * v.x - (double)u.x * v.z, cz = (double)u.x * v.y - (double)u.y * v.x;
        l.light.area = (float)std::sqrt(cx * cx + cy * cy + cz * cz);
        return l;
    }
    static AnalyticalLight distant(F3 position, F3 emission) {
        AnalyticalLight l;
        l.light.type = RPT_LIGHT_DISTANT;
        l.light.position[0] = position.x; l.light.position[1] = position.y; l.light.position[2] = position.z;
        l.light.emission[0] = emission.x; l.light.emission[1] = emission.y; l.light.emission[2] = emission.z;
        return l;
    }
};

// Data-driven counterpart of trait Scene (scene.rs:5-90): the scene describes itself.
struct Scene {
    Pinhole camera;
    std::vector<rpt_sphere> spheres;
    std::vector<rpt_plane> planes;
    std::vector<AnalyticalLight> lights;
    std::vector<rpt_material> materials;      // patches over Material::new()
    rpt_background background{RPT_BG_CONSTANT, {0, 0, 0}, {0, 0, 0}, 2.2f, 1.0f};
    float eps = 0.005f;                        // tracer.rs:16
    uint32_t max_depth = 4;                    // scene.rs:28-30
    bool any_hit_uses_max_dist = false;
    bool sample_all_light_types = false;       // rectangular / distant lights do something (off = the reference)
    virtual ~Scene() = default;

    size_t number_of_lights() const { return lights.size(); }
    const AnalyticalLight& light_at(size_t i) const { return lights.at(i); }
    uint16_t recursion_depth() const { return (uint16_t)max_depth; }

    // keeps `lights_flat_` alive; the descriptor points into this object
    rpt_scene_desc describe() {
        lights_flat_.clear();
        for (const auto& l : lights) lights_flat_.push_back(l.light);
        rpt_scene_desc d{};
        d.abi_version = RPT_ABI_VERSION;
        d.flags = (any_hit_uses_max_dist ? RPT_SCENE_ANYHIT_USES_MAX_DIST : 0u) | (sample_all_light_types ? RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES : 0u);
        d.camera.origin[0] = camera.origin.x; d.camera.origin[1] = camera.origin.y; d.camera.origin[2] = camera.origin.z;
        d.camera.center[0] = camera.center.x; d.camera.center[1] = camera.center.y; d.camera.center[2] = camera.center.z;
        d.camera.fov_deg = camera.fov;
        d.background = background;
        d.eps = eps;
        d.max_depth = max_depth;
        d.n_spheres = (uint32_t)spheres.size(); d.spheres = spheres.data();
        d.n_planes = (uint32_t)planes.size(); d.planes = planes.data();
        d.n_lights = (uint32_t)lights_flat_.size(); d.lights = lights_flat_.data();
        d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
        return d;
    }

private:
    std::vector<rpt_light> lights_flat_;
};

// renderer/src/analytical.rs:4-205
struct AnalyticalScene : Scene {
    AnalyticalScene() {
        rpt_scene_desc d{};
        rpt_scene_analytical(&d);
        spheres.assign(d.spheres, d.spheres + d.n_spheres);
        planes.assign(d.planes, d.planes + d.n_planes);
        materials.assign(d.materials, d.materials + d.n_materials);
        for (uint32_t i = 0; i < d.n_lights; ++i) { AnalyticalLight l; l.light = d.lights[i]; lights.push_back(l); }
        background = d.background;
        eps = d.eps;
        max_depth = d.max_depth;
        camera.origin = F3{d.camera.origin[0], d.camera.origin[1], d.camera.origin[2]};
        camera.center = F3{d.camera.center[0], d.camera.center[1], d.camera.center[2]};
        camera.fov = d.camera.fov_deg;
    }
};

// tracer.rs:5-19, :22, :629
class Tracer {
public:
    explicit Tracer(Scene* scene, int device = 0, uint64_t seed = 1) : scene_(scene), seed_(seed) {
        check(rpt_create(&ctx_, device), nullptr);
        sync_scene();
    }
    /// The GPUs of a node driven by this process: render() fans the rows out over them inside the call, where the
    /// reference fans out over rayon's threads (tracer.rs:29-32).
    Tracer(Scene* scene, const std::vector<int>& devices, uint64_t seed = 1) : scene_(scene), seed_(seed) {
        check(rpt_create_multi(&ctx_, devices.data(), (int)devices.size()), nullptr);
        sync_scene();
    }
    void set_tile_rows(uint32_t tile_rows) { check(rpt_set_tile_rows(ctx_, tile_rows), ctx_); }
    // scheduling of the launches (rpt.h, rpt_set_dispatch): changes when a sample is computed, never its value
    void set_dispatch(uint32_t cost_order = 1, uint32_t unit_rounds = 12, uint32_t unit_min_spp = 64, uint32_t unit_slots = 0)
    {
        check(rpt_set_dispatch(ctx_, cost_order, unit_rounds, unit_min_spp, unit_slots), ctx_);
    }
    ~Tracer() { rpt_destroy(ctx_); }
    Tracer(const Tracer&) = delete;
    Tracer& operator=(const Tracer&) = delete;

    /// Render one frame and accumulate into the pixels buffer (tracer.rs:21-123).
    void render(ColorBuffer& buffer) { render_n(buffer, 1); }

    /// `spp` consecutive frames in one launch; bit-identical to `spp` render() calls.
    void render_n(ColorBuffer& buffer, uint32_t spp) {
        check(rpt_render(ctx_, buffer.pixels.data(), (uint32_t)buffer.width, (uint32_t)buffer.height, buffer.frames, spp, seed_, 0), ctx_);
        buffer.frames += spp;                                           // tracer.rs:121
    }

    /// ColorBuffer::convert_to_u8 (buffer.rs:55-64) on the device.
    void convert_to_u8(const ColorBuffer& buffer, uint8_t* frame) {
        check(rpt_convert_to_u8(ctx_, buffer.pixels.data(), frame, (uint32_t)buffer.width, (uint32_t)buffer.height), ctx_);
    }

    /// The interactive loop of renderer/src/main.rs:113-124 with the ColorBuffer kept in HBM: render `spp` more
    /// frames into the context's resident buffer, then fetch the gamma-encoded u8 frame (4 B per pixel).
    void render_resident(size_t width, size_t height, uint32_t spp = 1) { check(rpt_resident_render(ctx_, (uint32_t)width, (uint32_t)height, spp, seed_, 0), ctx_); }
    void resident_to_u8(uint8_t* frame) { check(rpt_resident_download_u8(ctx_, frame), ctx_); }
    void resident_to(ColorBuffer& buffer) {
        check(rpt_resident_download(ctx_, buffer.pixels.data()), ctx_);
        uint64_t f = 0; rpt_resident_frames(ctx_, &f); buffer.frames = (size_t)f;
    }
    void resident_reset() { check(rpt_resident_reset(ctx_), ctx_); }

    /// Return the scene (tracer.rs:629); call sync_scene() after mutating it.
    Scene* scene() { return scene_; }
    void sync_scene() { rpt_scene_desc d = scene_->describe(); check(rpt_upload_scene(ctx_, &d), ctx_); }

private:
    static void check(int rc, const rpt_ctx* ctx) { if (rc != RPT_OK) throw Error(rc, rpt_last_error(ctx)); }
    rpt_ctx* ctx_ = nullptr;
    Scene* scene_;
    uint64_t seed_;
};

}  // namespace rpt
