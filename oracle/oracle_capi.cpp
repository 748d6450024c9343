// oracle_capi.cpp — C entry points of the CPU oracle (ctypes-friendly).
// TEST INFRASTRUCTURE: see the header of rpt_oracle.hpp.  PARITY UNPINNED (ibid.).
#include "rpt_oracle.hpp"
#include "../include/rpt_test.h"          // the probe record layouts the oracle answers for (tests/test_gpu_probes.py)

#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace rpt_oracle;

namespace {

// renderer/src/analytical.rs:13-22 (light), :41-58 / :70-85 / :101-116 (primitives
// and their material writes), camera/pinhole.rs:14-25 (Pinhole::new), tracer.rs:16
// (eps), scene.rs:28-30 (depth) — written out independently of the product's
// rpt_scene_analytical(); tests compare the two byte for byte.
rpt_sphere g_spheres[2];
rpt_plane g_planes[1];
rpt_light g_lights[1];
rpt_material g_materials[3];

void build_analytical(rpt_scene_desc* out)
{
    std::memset(out, 0, sizeof(*out));
    std::memset(g_spheres, 0, sizeof(g_spheres));
    std::memset(g_planes, 0, sizeof(g_planes));
    std::memset(g_lights, 0, sizeof(g_lights));
    std::memset(g_materials, 0, sizeof(g_materials));

    // left sphere: analytical.rs:41, :56-58
    g_spheres[0].center[0] = -1.1f; g_spheres[0].radius = 1.0f; g_spheres[0].material = 0;
    g_materials[0].mask = RPT_MAT_RGB | RPT_MAT_ROUGHNESS | RPT_MAT_METALLIC;
    g_materials[0].rgb[0] = g_materials[0].rgb[1] = g_materials[0].rgb[2] = 1.0f;
    g_materials[0].roughness = 0.05f;
    g_materials[0].metallic = 1.0f;
    // right sphere: analytical.rs:70, :82-85
    g_spheres[1].center[0] = 1.1f; g_spheres[1].radius = 1.0f; g_spheres[1].material = 1;
    g_materials[1].mask = RPT_MAT_RGB | RPT_MAT_CLEARCOAT | RPT_MAT_CLEARCOAT_GLOSS | RPT_MAT_ROUGHNESS;
    g_materials[1].rgb[0] = 1.0f; g_materials[1].rgb[1] = 0.186f; g_materials[1].rgb[2] = 0.0f;
    g_materials[1].clearcoat = 1.0f;
    g_materials[1].clearcoat_gloss = 1.0f;
    g_materials[1].roughness = 0.1f;
    // plane y = -1: analytical.rs:194-198, :105-116
    g_planes[0].normal[1] = 1.0f; g_planes[0].point[1] = -1.0f; g_planes[0].min_denom = 0.0001f; g_planes[0].material = 2;
    g_materials[2].mask = RPT_MAT_ROUGHNESS;
    g_materials[2].roughness = 1.0f;
    g_materials[2].proc_kind = RPT_PROC_CHECKER_DIR;
    g_materials[2].proc_params[0] = 0.5f; g_materials[2].proc_params[1] = 100.0f;
    g_materials[2].proc_params[2] = 0.25f; g_materials[2].proc_params[3] = 0.1f;
    // light: analytical.rs:15-16, light.rs:13-28
    g_lights[0].type = RPT_LIGHT_SPHERICAL;
    g_lights[0].position[0] = 3.0f; g_lights[0].position[1] = 2.0f; g_lights[0].position[2] = 2.0f;
    g_lights[0].emission[0] = g_lights[0].emission[1] = g_lights[0].emission[2] = 3.0f;
    g_lights[0].radius = 1.0f;
    g_lights[0].area = 4.0f * PI_F * g_lights[0].radius * g_lights[0].radius;

    out->abi_version = RPT_ABI_VERSION;
    out->flags = 0;                                       // any_hit ignores max_dist: analytical.rs:130
    out->camera.origin[2] = 3.0f;                         // pinhole.rs:16-17
    out->camera.fov_deg = 80.0f;                          // pinhole.rs:23
    out->background.kind = RPT_BG_GRADIENT_Y;             // analytical.rs:28-32
    out->background.colour_a[0] = out->background.colour_a[1] = out->background.colour_a[2] = 1.0f;
    out->background.colour_b[0] = 0.5f; out->background.colour_b[1] = 0.7f; out->background.colour_b[2] = 1.0f;
    out->background.gamma = 2.2f;                         // scene.rs:33
    out->background.scale = 0.5f;
    out->eps = 0.005f;                                    // tracer.rs:16
    out->max_depth = 4;                                   // scene.rs:29
    out->n_spheres = 2; out->spheres = g_spheres;
    out->n_planes = 1; out->planes = g_planes;
    out->n_lights = 1; out->lights = g_lights;
    out->n_materials = 3; out->materials = g_materials;
}

Material material_from_array(const float* m)
{
    Material mat;
    mat.rgb = F3(m[0], m[1], m[2]); mat.emission = F3(m[3], m[4], m[5]);
    mat.anisotropic = m[6]; mat.metallic = m[7]; mat.roughness = m[8]; mat.subsurface = m[9];
    mat.specular_tint = m[10]; mat.sheen = m[11]; mat.sheen_tint = m[12]; mat.clearcoat = m[13];
    mat.clearcoat_gloss = m[14]; mat.spec_trans = m[15]; mat.ior = m[16];
    return mat;
}

}  // namespace

extern "C" {

int oracle_scene_analytical(rpt_scene_desc* out) { build_analytical(out); return 0; }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_undo_quirks(uint32_t mask) { g_undo_quirks = mask; }       // TESTS ONLY (rpt_oracle.hpp, g_undo_quirks)

const char* oracle_build_info(void)
{
#if defined(RPT_ORACLE_F64)
    return "oracle: f64";
#endif
#if defined(RPT_OPCOUNT)
    return "oracle: opcount";
#elif defined(RPT_ORACLE_LIBM)
    return "oracle: glibc libm";
#else
    return "oracle: strict math";
#endif
}

// Tracer::render for spp frames, rows [row_begin,row_end) (0,height = whole image).
int oracle_render(const rpt_scene_desc* desc, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done,
                  uint32_t spp, uint64_t seed, uint32_t row_begin, uint32_t row_end, int nthreads)
{
    if (!desc || !pixels || width == 0 || height == 0 || row_end > height || row_begin > row_end) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    Scene scene(*desc);
    Tracer tracer(scene);
    tracer.render(pixels, width, height, frames_done, spp, seed, row_begin, row_end);
    return 0;
}

// The same with RPT_RENDER_* flags (only RPT_RENDER_RUSSIAN_ROULETTE changes what the oracle computes).
int oracle_render_flags(const rpt_scene_desc* desc, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done,
                        uint32_t spp, uint64_t seed, uint32_t row_begin, uint32_t row_end, int nthreads, uint32_t render_flags)
{
    if (!desc || !pixels || width == 0 || height == 0 || row_end > height || row_begin > row_end) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    Scene scene(*desc);
    Tracer tracer(scene);
    tracer.russian_roulette = (render_flags & RPT_RENDER_RUSSIAN_ROULETTE) != 0;
    tracer.render(pixels, width, height, frames_done, spp, seed, row_begin, row_end);
    return 0;
}

// The same for a LIST of rows (each row one task, like the reference's scanline tasks, tracer.rs:29-32): what a bounded sample of a large
// frame spread evenly over it costs (bench.py, f64_reference).  Rows are independent of each other, so the pixels are oracle_render's.
int oracle_render_rows(const rpt_scene_desc* desc, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp,
                       uint64_t seed, const uint32_t* rows, uint32_t n_rows, int nthreads, uint32_t render_flags)
{
    if (!desc || !pixels || !rows || width == 0 || height == 0) return -1;
    for (uint32_t i = 0; i < n_rows; ++i) if (rows[i] >= height) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    Scene scene(*desc);
    Tracer tracer(scene);
    tracer.russian_roulette = (render_flags & RPT_RENDER_RUSSIAN_ROULETTE) != 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)n_rows; ++i) tracer.render(pixels, width, height, frames_done, spp, seed, rows[i], rows[i] + 1u);
    return 0;
}

// Radiance of single pixel-samples (no accumulation): out[3*k..] for k-th (col,row,frame).
int oracle_sample_pixels(const rpt_scene_desc* desc, const uint32_t* cols, const uint32_t* rows, const uint64_t* frames,
                         uint64_t n, uint32_t width, uint32_t height, uint64_t seed, float* out)
{
    Scene scene(*desc);
    Tracer tracer(scene);
    for (uint64_t k = 0; k < n; ++k) {
        F3 r = tracer.sample_pixel(cols[k], rows[k], width, height, frame_key(seed, frames[k]));
        out[3 * k + 0] = rawf(r.x); out[3 * k + 1] = rawf(r.y); out[3 * k + 2] = rawf(r.z);
    }
    return 0;
}

// The rays (closest_hit: max_dist = -1; any_hit: its max_dist) one pixel-sample queries, 7 floats each.
int oracle_sample_rays(const rpt_scene_desc* desc, uint32_t col, uint32_t row, uint64_t frame, uint32_t width, uint32_t height,
                       uint64_t seed, float* out, uint32_t max_rays)
{
    Scene scene(*desc);
    Tracer tracer(scene);
    std::vector<float> log;
    g_ray_log = &log;
    tracer.sample_pixel(col, row, width, height, frame_key(seed, frame));
    g_ray_log = nullptr;
    uint32_t n = (uint32_t)(log.size() / 7);
    if (n > max_rays) n = max_rays;
    std::memcpy(out, log.data(), (size_t)n * 7 * sizeof(float));
    return (int)n;
}

// Path events (g_event_log) of the samples [frame0, frame0 + spp) of the pixels of the rectangle [col0, col0 + cw) x
// [row0, row0 + ch), pixel-major then sample-major, '.'-terminated per sample.  Returns the number of bytes (or
// -needed when `cap` is too small).  tools/sched_sim.py replays them through candidate wave schedules.
int oracle_sample_events(const rpt_scene_desc* desc, uint32_t col0, uint32_t row0, uint32_t cw, uint32_t ch, uint64_t frame0, uint32_t spp,
                         uint32_t width, uint32_t height, uint64_t seed, uint8_t* out, uint64_t cap)
{
    Scene scene(*desc);
    Tracer tracer(scene);
    std::vector<uint8_t> log;
    g_event_log = &log;
    for (uint32_t r = row0; r < row0 + ch; ++r)
        for (uint32_t c = col0; c < col0 + cw; ++c)
            for (uint32_t s = 0; s < spp; ++s) tracer.sample_pixel(c, r, width, height, frame_key(seed, frame0 + s));
    g_event_log = nullptr;
    if (log.size() > cap) return -(int)log.size();
    std::memcpy(out, log.data(), log.size());
    return (int)log.size();
}

// Count floating-point operations over a render (RPT_OPCOUNT build; zeros otherwise).
// counts = {add, mul, div, sqrt, transcendental, compare}
int oracle_opcount(const rpt_scene_desc* desc, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t* counts)
{
    for (int i = 0; i < 6; ++i) counts[i] = 0;
#ifdef RPT_OPCOUNT
#ifdef _OPENMP
    omp_set_num_threads(1);
#endif
    g_ops = OpCounts();
    std::vector<float> px((size_t)width * height * 4, 0.0f);
    Scene scene(*desc);
    Tracer tracer(scene);
    tracer.render(px.data(), width, height, 0, spp, seed, 0, height);
    counts[0] = g_ops.add; counts[1] = g_ops.mul; counts[2] = g_ops.div;
    counts[3] = g_ops.sqrt; counts[4] = g_ops.transc; counts[5] = g_ops.cmp;
#else
    (void)desc; (void)width; (void)height; (void)spp; (void)seed;
#endif
    return 0;
}

// The same plus, in counts[6..11], the part of it spent in scene-sphere tests that missed (rpt_oracle.hpp, g_ops_missed).
int oracle_opcount_split(const rpt_scene_desc* desc, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t* counts)
{
    for (int i = 0; i < 12; ++i) counts[i] = 0;
#ifdef RPT_OPCOUNT
    g_ops_missed = OpCounts();
    oracle_opcount(desc, width, height, spp, seed, counts);
    counts[6] = g_ops_missed.add; counts[7] = g_ops_missed.mul; counts[8] = g_ops_missed.div;
    counts[9] = g_ops_missed.sqrt; counts[10] = g_ops_missed.transc; counts[11] = g_ops_missed.cmp;
#else
    (void)desc; (void)width; (void)height; (void)spp; (void)seed;
#endif
    return 0;
}

// ---- leaf probes (known-answer tests) ---------------------------------------
int oracle_sphere(const float* o, const float* d, const float* c, float radius, float* t)
{
    F tt(0.0f);
    bool hit = sphere(Ray(F3(o[0], o[1], o[2]), F3(d[0], d[1], d[2])), F3(c[0], c[1], c[2]), radius, tt);
    *t = rawf(tt);
    return hit ? 1 : 0;
}
int oracle_plane(const float* o, const float* d, const rpt_plane* p, float* t)
{
    F tt(0.0f);
    bool hit = plane(Ray(F3(o[0], o[1], o[2]), F3(d[0], d[1], d[2])), *p, tt);
    *t = rawf(tt);
    return hit ? 1 : 0;
}
float oracle_power_heuristic(float a, float b) { return rawf(Tracer::power_heuristic(a, b)); }
float oracle_schlick_fresnel(float u) { return rawf(Tracer::schlick_fresnel(u)); }
float oracle_dielectric_fresnel(float c, float eta) { return rawf(Tracer::dielectric_fresnel(c, eta)); }
float oracle_gtr1(float ndoth, float a) { return rawf(Tracer::gtr1(ndoth, a)); }
float oracle_smithg(float ndotv, float alphag) { return rawf(Tracer::smithg(ndotv, alphag)); }
float oracle_gtr2aniso(float ndoth, float hx, float hy, float ax, float ay) { return rawf(Tracer::gtr2aniso(ndoth, hx, hy, ax, ay)); }
float oracle_luminance(const float* c) { return rawf(Tracer::luminance(F3(c[0], c[1], c[2]))); }

// Material::new + field overrides + finalize; m = 17 user-set floats (rgb, emission,
// anisotropic, metallic, roughness, subsurface, specular_tint, sheen, sheen_tint,
// clearcoat, clearcoat_gloss, spec_trans, ior); out = {roughness, clearcoat_roughness, ax, ay}
void oracle_material_defaults(float* m)
{
    Material mat;
    float v[17] = {rawf(mat.rgb.x), rawf(mat.rgb.y), rawf(mat.rgb.z), rawf(mat.emission.x), rawf(mat.emission.y), rawf(mat.emission.z),
                   rawf(mat.anisotropic), rawf(mat.metallic), rawf(mat.roughness), rawf(mat.subsurface), rawf(mat.specular_tint),
                   rawf(mat.sheen), rawf(mat.sheen_tint), rawf(mat.clearcoat), rawf(mat.clearcoat_gloss), rawf(mat.spec_trans), rawf(mat.ior)};
    std::memcpy(m, v, sizeof(v));
}
void oracle_material_finalize(const float* m, float* out)
{
    Material mat = material_from_array(m);
    mat.finalize();
    out[0] = rawf(mat.roughness); out[1] = rawf(mat.clearcoat_roughness); out[2] = rawf(mat.ax); out[3] = rawf(mat.ay);
}

// Pinhole::gen_ray: cam = {origin[3], center[3], fov}; out = {origin[3], direction[3]}
void oracle_gen_ray(const float* cam, float px, float py, float offx, float offy, float width, float height, float* out)
{
    Pinhole p;
    p.origin = F3(cam[0], cam[1], cam[2]); p.center = F3(cam[3], cam[4], cam[5]); p.fov = cam[6];
    Ray r = p.gen_ray(px, py, offx, offy, width, height);
    out[0] = rawf(r.origin.x); out[1] = rawf(r.origin.y); out[2] = rawf(r.origin.z);
    out[3] = rawf(r.direction.x); out[4] = rawf(r.direction.y); out[5] = rawf(r.direction.z);
}

// disney_eval on a finalized material: returns f[3], pdf in out[0..3]
void oracle_disney_eval(const float* m, float eta, const float* v, const float* n, const float* l, float* out)
{
    rpt_scene_desc d; build_analytical(&d);
    Scene scene(d);
    Tracer tr(scene);
    State st;
    st.material = material_from_array(m);
    st.material.finalize();
    st.eta = eta;
    F pdf(0.0f);
    F3 f = tr.disney_eval(st, F3(v[0], v[1], v[2]), F3(n[0], n[1], n[2]), F3(l[0], l[1], l[2]), pdf);
    out[0] = rawf(f.x); out[1] = rawf(f.y); out[2] = rawf(f.z); out[3] = rawf(pdf);
}

// disney_sample with an explicit RNG position: out = {f[3], l[3], pdf, draws_used}
void oracle_disney_sample(const float* m, float eta, const float* v, const float* n, const float* l_stale,
                          uint32_t fkey, uint32_t pixel, uint32_t counter, float* out)
{
    rpt_scene_desc d; build_analytical(&d);
    Scene scene(d);
    Tracer tr(scene);
    State st;
    st.material = material_from_array(m);
    st.material.finalize();
    st.eta = eta;
    (void)counter;                                           // (the records' third RNG field is unused since the RNG is a stream)
    Rng rng(fkey, pixel);                                    // = (state, increment)
    const Rng rng0 = rng;
    F3 l(l_stale[0], l_stale[1], l_stale[2]);
    F pdf(0.0f);
    F3 f = tr.disney_sample(st, F3(v[0], v[1], v[2]), F3(n[0], n[1], n[2]), l, pdf, rng);
    out[0] = rawf(f.x); out[1] = rawf(f.y); out[2] = rawf(f.z);
    out[3] = rawf(l.x); out[4] = rawf(l.y); out[5] = rawf(l.z);
    out[6] = rawf(pdf); out[7] = (float)rng_draws_between(rng0, rng);
}

// sample_light with an explicit RNG position: light = rpt_light; out = {normal[3], emission[3], direction[3], dist, pdf, draws_used}
void oracle_sample_light(const rpt_light* light, const float* scatter_pos, uint32_t n_lights, uint32_t scene_flags,
                         uint32_t fkey, uint32_t pixel, uint32_t counter, float* out)
{
    rpt_scene_desc d; build_analytical(&d);
    std::vector<rpt_light> lights(n_lights ? n_lights : 1, *light);                // number_of_lights() scales the emission
    d.n_lights = n_lights; d.lights = lights.data();
    d.flags = scene_flags;
    Scene scene(d);
    Tracer tr(scene);
    (void)counter;
    Rng rng(fkey, pixel);                                    // = (state, increment)
    const Rng rng0 = rng;
    LightSampleRec ls;
    tr.sample_light(*light, F3(scatter_pos[0], scatter_pos[1], scatter_pos[2]), ls, rng);
    out[0] = rawf(ls.normal.x); out[1] = rawf(ls.normal.y); out[2] = rawf(ls.normal.z);
    out[3] = rawf(ls.emission.x); out[4] = rawf(ls.emission.y); out[5] = rawf(ls.emission.z);
    out[6] = rawf(ls.direction.x); out[7] = rawf(ls.direction.y); out[8] = rawf(ls.direction.z);
    out[9] = rawf(ls.dist); out[10] = rawf(ls.pdf); out[11] = (float)rng_draws_between(rng0, rng);
}

// The record-per-call layouts of include/rpt.h's rpt_probe_fn, evaluated by the oracle (one call for n records).
// cam = {origin[3], center[3], fov}; params = {width, height} (GEN_RAY only).
static inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
void oracle_probe_fn(uint32_t fn, const float* in, float* out, uint64_t n, const float* cam, const float* params)
{
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = in + i * RPT_PROBE_IN_STRIDE;
        float* o = out + i * RPT_PROBE_OUT_STRIDE;
        for (int k = 0; k < RPT_PROBE_OUT_STRIDE; ++k) o[k] = 0.0f;
        switch (fn) {
        case RPT_PROBE_FN_GEN_RAY: oracle_gen_ray(cam, r[0], r[1], r[2], r[3], params[0], params[1], o); break;
        case RPT_PROBE_FN_HIT_SPHERE: {
            float t = 0.0f;
            int h = oracle_sphere(r, r + 3, r + 6, r[9], &t);
            o[0] = h ? 1.0f : 0.0f; o[1] = h ? t : 0.0f;
            break;
        }
        case RPT_PROBE_FN_HIT_PLANE: {
            rpt_plane p;
            std::memset(&p, 0, sizeof(p));
            for (int k = 0; k < 3; ++k) { p.normal[k] = r[6 + k]; p.point[k] = r[9 + k]; }
            p.min_denom = r[12]; p.max_t = r[13];
            float t = 0.0f;
            int h = oracle_plane(r, r + 3, &p, &t);
            o[0] = h ? 1.0f : 0.0f; o[1] = h ? t : 0.0f;
            break;
        }
        case RPT_PROBE_FN_SAMPLE_LIGHT: {
            rpt_light L;
            std::memset(&L, 0, sizeof(L));
            L.type = fbits(r[0]);
            for (int k = 0; k < 3; ++k) { L.position[k] = r[1 + k]; L.emission[k] = r[4 + k]; L.u[k] = r[9 + k]; L.v[k] = r[12 + k]; }
            L.radius = r[7]; L.area = r[8];
            oracle_sample_light(&L, r + 15, (uint32_t)r[18], fbits(r[19]), fbits(r[20]), fbits(r[21]), fbits(r[22]), o);
            break;
        }
        case RPT_PROBE_FN_DISNEY_EVAL: oracle_disney_eval(r, r[17], r + 18, r + 21, r + 24, o); break;
        case RPT_PROBE_FN_DISNEY_SAMPLE: oracle_disney_sample(r, r[17], r + 18, r + 21, r + 24, fbits(r[27]), fbits(r[28]), fbits(r[29]), o); break;
        default: break;
        }
    }
}

// RNG stream: first n u32 draws of (seed, frame, pixel)
void oracle_rng_u32(uint64_t seed, uint64_t frame, uint32_t pixel, uint32_t n, uint32_t* out)
{
    Rng rng(frame_key(seed, frame), pixel);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_u32();
}
void oracle_rng_f32(uint64_t seed, uint64_t frame, uint32_t pixel, uint32_t n, float* out)
{
    Rng rng(frame_key(seed, frame), pixel);
    for (uint32_t i = 0; i < n; ++i) out[i] = rawf(rng.gen());
}

// element-wise math through the oracle's f_* layer (strict or glibc per build)
// fn: 0 sin, 1 cos, 2 log2, 3 pow(a,b), 4 a/b, 5 sqrt, 7 exp, 8 ln, 9 three quotients, 10 rpt_powf_log2x (strict header, any build), 100 tan
void oracle_math(uint32_t fn, const float* a, const float* b, float* out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = rawf(f_sin(a[i])); break;
        case 1: out[i] = rawf(f_cos(a[i])); break;
        case 2: out[i] = rawf(f_log2(a[i])); break;
        case 3: out[i] = rawf(f_powf(a[i], b[i])); break;
        case 4: out[i] = a[i] / b[i]; break;
        case 5: out[i] = std::sqrt(a[i]); break;
        case 7: out[i] = rawf(f_exp(a[i])); break;           // RPT_PROBE_EXP
        case 8: out[i] = rawf(f_ln(a[i])); break;            // RPT_PROBE_LOG
        case 9: {                                           // RPT_PROBE_DIV3: the same quotients, IEEE divides
            const float x = a[i], d = b[i];
            const float qx = (i & 4u) ? x / d : x / d, qy = (i & 4u) ? (0.5f * d) / d : (-d) / d, qz = (i & 4u) ? 0.0f / d : (0.75f * x) / d;
            out[i] = (i % 3u == 0u) ? qx : ((i % 3u == 1u) ? qy : qz);
            break;
        }
        case 10: {                                          // pow with the base's logarithm handed in (the library's material tables)
            const uint32_t ia = rpt_f2u(a[i]);
            const double la = (ia - 1u < 0x7f7fffffu) ? rpt_log2_core(a[i]) : 0.0;
            out[i] = rpt_powf_log2x(a[i], la, b[i]);
            break;
        }
        case 100: out[i] = rawf(f_tan(a[i])); break;         // host only (the camera's fov)
        default: out[i] = 0.0f;
        }
    }
}

// Media leaves (PROJECT-DEFINED, include/rpt.h): the phase function and its sampling.
float oracle_phase_hg(float cos_theta, float g) { return rawf(Tracer::phase_hg(cos_theta, g)); }
void oracle_sample_hg(const float* v, float g, float r1, float r2, float* out)
{
    F3 d = Tracer::sample_hg(F3(v[0], v[1], v[2]), g, r1, r2);
    out[0] = rawf(d.x); out[1] = rawf(d.y); out[2] = rawf(d.z);
}

// ColorBuffer::convert_to_u8, buffer.rs:55-64: (p.powf(0.4545) * 255.0) as u8 for
// r,g,b and (a * 255.0) as u8; Rust's `as u8` saturates and maps NaN to 0.
static inline uint8_t as_u8(float x)
{
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
void oracle_convert_to_u8(const float* pixels, uint8_t* frame, uint32_t width, uint32_t height)
{
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            size_t o = (size_t)x * 4 + (size_t)y * width * 4;
            frame[o + 0] = as_u8(rawf(f_powf(pixels[o + 0], 0.4545f) * F(255.0f)));
            frame[o + 1] = as_u8(rawf(f_powf(pixels[o + 1], 0.4545f) * F(255.0f)));
            frame[o + 2] = as_u8(rawf(f_powf(pixels[o + 2], 0.4545f) * F(255.0f)));
            frame[o + 3] = as_u8(pixels[o + 3] * 255.0f);
        }
}

// ColorBuffer::convert_to_u8_at, buffer.rs:67-89: blit the buffer (bw x bh) into a larger u8 frame
// (at.2 x at.3) at offset (at.0, at.1) — no gamma, strict `>` lower bounds, and the row shift that comes from
// y = height - j over reverse chunks.  `frame` must hold exactly width*height*4 bytes.
void oracle_convert_to_u8_at(const float* pixels, uint32_t bw, uint32_t bh, uint8_t* frame, uint32_t at0, uint32_t at1,
                             uint32_t width, uint32_t height)
{
    for (uint32_t j = 0; j < height; ++j) {                 // par_rchunks_exact_mut: j = 0 is the LAST row of `frame`
        uint8_t* line = frame + (size_t)(height - 1 - j) * width * 4;
        for (uint32_t ii = 0; ii < width; ++ii) {
            size_t i = (size_t)j * width + ii;
            size_t x = i % width;
            size_t y = height - (i / width);
            if (x > at0 && x < (size_t)at0 + bw) {
                if (y > at1 && y < (size_t)at1 + bh) {
                    size_t o = (x - at0) * 4 + (y - at1) * bw * 4;
                    for (int c = 0; c < 4; ++c) line[ii * 4 + c] = as_u8(pixels[o + c] * 255.0f);
                }
            }
        }
    }
}

// the denoiser of include/rpt.h (project-defined)
void oracle_denoise(const float* pixels, float* out, uint32_t width, uint32_t height, uint32_t iterations, float edge_k)
{
    denoise(pixels, out, width, height, iterations, edge_k);
}

}  // extern "C"
