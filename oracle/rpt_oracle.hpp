// rpt_oracle.hpp — CPU restatement of rust-pathtracer's per-pixel-sample path.
//
// TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, load or call anything under oracle/
// (plus the development scripts under tools/ that generate fixtures, time the CPU
// baseline or replay path events through schedule simulations; the package never
// imports tools/).  The shipped path (rust-pathtracer_amd/) never includes,
// links or loads anything from here, and has no CPU fallback.
//
// PARITY UNPINNED: the reference (/root/reference, crate v0.2.4) ships no tests,
// golden vectors or fixtures, cannot be compiled here (no Rust toolchain), and its
// RNG (rand::thread_rng, tracer.rs:44) is OS-seeded.  Nothing external pins these
// results.  What pins them is: this file follows the reference statement by
// statement (every function cites the lines it restates), hand-derived
// known-answer tests (tests/test_oracle_kat.py), and statistical self-checks
// (tests/test_oracle_stats.py).
//
// Two deliberate substitutions, both forced:
//   * RNG: a counter-based generator keyed by (seed, frame, pixel) replaces
//     thread_rng; draws are converted exactly like rand 0.8.5's Standard f32
//     ((u32 >> 8) * 2^-24) and consumed in the reference's order.
//   * libm: sin/cos/tan/powf/log2 come from include/rpt_strict_math.h (a
//     bit-reproducible libm stand-in shared with the device code) unless
//     RPT_ORACLE_LIBM is defined, in which case glibc's are used — the build the
//     reference itself would get on Linux; tests compare the two.
//
// Arithmetic is f32 with the reference's operation order; compile with
// -ffp-contract=off.  Define RPT_OPCOUNT to count floating-point operations.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/rpt.h"
#include "../include/rpt_strict_math.h"

namespace rpt_oracle {

// ---------------------------------------------------------------------------
// scalar type: plain float, or an op-counting wrapper (measured
// flops per sample replace the static estimate)
// ---------------------------------------------------------------------------
struct OpCounts {
    uint64_t add = 0, mul = 0, div = 0, sqrt = 0, transc = 0, cmp = 0;
    void operator+=(const OpCounts& o) { add += o.add; mul += o.mul; div += o.div; sqrt += o.sqrt; transc += o.transc; cmp += o.cmp; }
};

#if defined(RPT_ORACLE_F64)
// The whole path in f64 (BASELINE.json configs[0] says "f64"; the crate at this commit is f32, lib.rs:6, and its Readme's "default is
// f64" is stale: SURVEY.md fact 4).  Same statements, same draws (a draw is (u32 >> 8) * 2^-24: exact in both), same operation order,
// double arithmetic and glibc's double libm: what the f32 frames — the kernel's and the f32 oracle's — are a ROUNDED version of.  Used
// for error analysis only (bench.py `f64_reference`, tests/test_oracle_f64.py): the distance of an f32 frame from this one is the
// "stated float tolerance" of BASELINE.json's north_star, measured instead of assumed.  The running mean is accumulated in f64 too.
typedef double F;
typedef double Raw;
inline Raw raw(F a) { return a; }
inline void count_sqrt() {}
inline void count_transc() {}
struct MissedSphereTest { void missed() {} };
#elif defined(RPT_OPCOUNT)
typedef float Raw;
inline thread_local OpCounts g_ops;
// Operations spent in tests of SCENE spheres that missed (closest_hit / any_hit loops): what an ideal acceleration
// structure never executes.  bench.py prices a large scene's kernels against g_ops - g_ops_missed, not against the
// brute-force loop's count (10 000 tests per query).
inline thread_local OpCounts g_ops_missed;
struct MissedSphereTest {
    OpCounts before;
    MissedSphereTest() : before(g_ops) {}
    void missed()
    {
        g_ops_missed.add += g_ops.add - before.add; g_ops_missed.mul += g_ops.mul - before.mul; g_ops_missed.div += g_ops.div - before.div;
        g_ops_missed.sqrt += g_ops.sqrt - before.sqrt; g_ops_missed.transc += g_ops.transc - before.transc; g_ops_missed.cmp += g_ops.cmp - before.cmp;
    }
};
struct F {
    float v;
    F() : v(0.0f) {}
    F(float x) : v(x) {}
    F(double x) : v((float)x) {}
    F(int x) : v((float)x) {}
};
inline F operator+(F a, F b) { g_ops.add++; return F(a.v + b.v); }
inline F operator-(F a, F b) { g_ops.add++; return F(a.v - b.v); }
inline F operator*(F a, F b) { g_ops.mul++; return F(a.v * b.v); }
inline F operator/(F a, F b) { g_ops.div++; return F(a.v / b.v); }
inline F operator-(F a) { return F(-a.v); }
inline F& operator+=(F& a, F b) { a = a + b; return a; }
inline F& operator*=(F& a, F b) { a = a * b; return a; }
inline F& operator/=(F& a, F b) { a = a / b; return a; }
inline bool operator<(F a, F b) { g_ops.cmp++; return a.v < b.v; }
inline bool operator>(F a, F b) { g_ops.cmp++; return a.v > b.v; }
inline bool operator<=(F a, F b) { g_ops.cmp++; return a.v <= b.v; }
inline bool operator>=(F a, F b) { g_ops.cmp++; return a.v >= b.v; }
inline Raw raw(F a) { return a.v; }
inline void count_sqrt() { g_ops.sqrt++; }
inline void count_transc() { g_ops.transc++; }
#else
typedef float F;
typedef float Raw;
inline Raw raw(F a) { return a; }
inline void count_sqrt() {}
inline void count_transc() {}
struct MissedSphereTest { void missed() {} };
#endif

inline float rawf(F a) { return (float)raw(a); }              // for outputs that are f32 whatever F is (ray log, C ABI)

// crate constants: rust-pathtracer/src/lib.rs:8-10
static const float PI_F = 3.14159265358979323846f;          // std::f32::consts::PI
static const float INV_PI_F = 1.0f / 3.14159265358979323846f;
static const float TWO_PI_F = 3.14159265358979323846f * 2.0f;
static const float INV_4_PI_F = 0.0795774715459476679f;     // PROJECT-DEFINED (media, include/rpt.h): 1 / (4 pi) rounded to f32

// --- f32 intrinsics with Rust semantics -----------------------------------
inline F f_sqrt(F x) { count_sqrt(); return F(std::sqrt(raw(x))); }
inline F f_abs(F x) { return F(std::fabs(raw(x))); }
inline F f_floor(F x) { return F(std::floor(raw(x))); }
// f32::max / f32::min: a NaN operand yields the other operand.
inline F f_max(F self, F other)
{
    Raw a = raw(self), b = raw(other);
    if (a != a) return F(b);
    if (b != b) return F(a);
    return F(a > b ? a : b);
}
// f32::clamp: NaN stays NaN.
inline F f_clamp(F x, float lo, float hi)
{
    Raw a = raw(x);
    if (a < lo) return F(lo);
    if (a > hi) return F(hi);
    return F(a);
}
inline F f_min(F self, F other)
{
    Raw a = raw(self), b = raw(other);
    if (a != a) return F(b);
    if (b != b) return F(a);
    return F(a < b ? a : b);
}
// Rust's `%` on f32 is C fmodf (exact).
inline F f_rem(F a, float b) { return F(std::fmod(raw(a), b)); }

#if defined(RPT_ORACLE_F64)
inline F f_sin(F x) { return ::sin(x); }
inline F f_cos(F x) { return ::cos(x); }
inline F f_tan(F x) { return ::tan(x); }
inline F f_powf(F x, F y) { return ::pow(x, y); }
inline F f_log2(F x) { return ::log2(x); }
inline F f_exp(F x) { return ::exp(x); }
inline F f_ln(F x) { return ::log(x); }
#elif defined(RPT_ORACLE_LIBM)
inline F f_sin(F x) { count_transc(); return F(::sinf(raw(x))); }
inline F f_cos(F x) { count_transc(); return F(::cosf(raw(x))); }
inline F f_tan(F x) { count_transc(); return F(::tanf(raw(x))); }
inline F f_powf(F x, F y) { count_transc(); return F(::powf(raw(x), raw(y))); }
inline F f_log2(F x) { count_transc(); return F(::log2f(raw(x))); }
inline F f_exp(F x) { count_transc(); return F(::expf(raw(x))); }
inline F f_ln(F x) { count_transc(); return F(::logf(raw(x))); }
#else
inline F f_sin(F x) { count_transc(); return F(rpt_sinf(raw(x))); }
inline F f_cos(F x) { count_transc(); return F(rpt_cosf(raw(x))); }
inline F f_tan(F x) { count_transc(); return F(rpt_tanf(raw(x))); }
inline F f_powf(F x, F y) { count_transc(); return F(rpt_powf(raw(x), raw(y))); }
inline F f_log2(F x) { count_transc(); return F(rpt_log2f(raw(x))); }
inline F f_exp(F x) { count_transc(); return F(rpt_expf(raw(x))); }
inline F f_ln(F x) { count_transc(); return F(rpt_logf(raw(x))); }
#endif

// ---------------------------------------------------------------------------
// F3 — rust-pathtracer/src/fx.rs:209-351 (type, normalize, length, dot, cross,
// mult_f) and :437-515 (operators)
// ---------------------------------------------------------------------------
struct F3 {
    F x, y, z;
    F3() : x(0.0f), y(0.0f), z(0.0f) {}
    F3(F x_, F y_, F z_) : x(x_), y(y_), z(z_) {}
    static F3 zeros() { return F3(0.0f, 0.0f, 0.0f); }                 // fx.rs:225
    static F3 new_x(F v) { return F3(v, v, v); }                       // fx.rs:233
    F length() const { return f_sqrt(x * x + y * y + z * z); }         // fx.rs:331-333
    F3 normalize() const { F l = length(); return F3(x / l, y / l, z / l); }   // fx.rs:307-313
    F dot(const F3& o) const { return x * o.x + y * o.y + z * o.z; }   // fx.rs:335-337
    F3 cross(const F3& o) const                                        // fx.rs:339-344
    {
        return F3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
    }
    F3 mult_f(F f) const { return F3(x * f, y * f, z * f); }           // fx.rs:346-351
};
inline F3 operator+(F3 a, F3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }   // fx.rs:437-443
inline F3 operator-(F3 a, F3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }   // fx.rs:453-459
inline F3 operator*(F3 a, F3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }   // fx.rs:469-475
inline F3 operator*(F a, F3 b) { return F3(a * b.x, a * b.y, a * b.z); }          // fx.rs:477-483 (f32 * F3)
inline F3 operator/(F3 a, F3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }   // fx.rs:485-491
inline F3 operator-(F3 a) { return F3(-a.x, -a.y, -a.z); }                        // fx.rs:509-515
inline F3& operator+=(F3& a, F3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }   // fx.rs:445-451
inline F3& operator/=(F3& a, F3 b) { a.x /= b.x; a.y /= b.y; a.z /= b.z; return a; }   // fx.rs:493-499

// free functions: rust-pathtracer/src/math.rs:4-60
inline F3 cross(const F3& a, const F3& b) { return a.cross(b); }
inline F dot(const F3& a, const F3& b) { return a.dot(b); }
inline F3 normalize(const F3& a) { return a.normalize(); }
inline F length(const F3& a) { return a.length(); }
inline F3 mix(const F3& a, const F3& b, F v)                                     // math.rs:34-40
{
    return F3((F(1.0f) - v) * a.x + b.x * v, (F(1.0f) - v) * a.y + b.y * v, (F(1.0f) - v) * a.z + b.z * v);
}
inline F3 pow3(const F3& a, const F3& e)                                         // math.rs:53-60
{
    return F3(f_powf(a.x, e.x), f_powf(a.y, e.y), f_powf(a.z, e.z));
}
inline F mix_ptf(F a, F b, F v) { return (F(1.0f) - v) * a + b * v; }            // tracer.rs:229-231, material.rs:121-123

// ---------------------------------------------------------------------------
// Ray — rust-pathtracer/src/ray.rs:6-33.  inv_direction / sign_* (ray.rs:24-27)
// are never read by the tracer and are not restated.
// ---------------------------------------------------------------------------
struct Ray {
    F3 origin, direction;
    Ray() {}
    Ray(F3 o, F3 d) : origin(o), direction(d) {}                                  // ray.rs:18-29
    F3 at(F dist) const { return origin + dist * direction; }                     // ray.rs:31-33
};

// ---------------------------------------------------------------------------
// Material — rust-pathtracer/src/material.rs:48-131 (fields the tracer reads)
// ---------------------------------------------------------------------------
struct Medium {                                                                   // material.rs:16-34
    uint32_t medium_type;                                                         // MediumType, material.rs:8-13: RPT_MEDIUM_*
    F density;
    F3 color;
    F anisotropy;
    Medium() : medium_type(RPT_MEDIUM_NONE), density(0.0f), color(0.0f, 0.0f, 0.0f), anisotropy(0.0f) {}   // material.rs:25-32
};

struct Material {
    F3 rgb, emission;
    F anisotropic, metallic, roughness, subsurface, specular_tint, sheen, sheen_tint;
    F clearcoat, clearcoat_gloss, clearcoat_roughness, spec_trans, ior, ax, ay;
    Medium medium;                                                                // material.rs:75,107; read only under RPT_SCENE_MEDIA

    Material()                                                                    // material.rs:82-114
        : rgb(1.5f, 1.5f, 1.5f), emission(0.0f, 0.0f, 0.0f), anisotropic(0.0f), metallic(0.0f),
          roughness(0.5f), subsurface(0.0f), specular_tint(0.0f), sheen(0.0f), sheen_tint(0.0f),
          clearcoat(0.0f), clearcoat_gloss(0.0f), clearcoat_roughness(0.0f), spec_trans(0.0f),
          ior(1.45f), ax(0.0f), ay(0.0f)
    {
    }

    void finalize()                                                               // material.rs:117-131
    {
        roughness = f_max(roughness, 0.01f);
        clearcoat_roughness = mix_ptf(0.1f, 0.001f, clearcoat_gloss);
        medium.anisotropy = f_clamp(medium.anisotropy, -0.9f, 0.9f);              // material.rs:126
        F aspect = f_sqrt(F(1.0f) - anisotropic * F(0.9f));
        ax = f_max(roughness / aspect, 0.001f);
        ay = f_max(roughness * aspect, 0.001f);
    }
};

// globals.rs:6-62
struct State {
    uint16_t depth;
    F eta, hit_dist;
    F3 fhp, normal, ffnormal;
    bool is_emitter;
    Material material;
    Medium medium;                                                                // globals.rs:19,37; the medium the path is in (media, include/rpt.h)

    State() : depth(4), eta(0.0f), hit_dist(-1.0f), is_emitter(false) {}          // globals.rs:23-39

    void finalize(const Ray& ray)                                                 // globals.rs:50-62
    {
        fhp = ray.at(hit_dist);
        if (dot(normal, ray.direction) <= F(0.0f)) ffnormal = normal;
        else ffnormal = -normal;
        material.finalize();
        eta = (dot(ray.direction, normal) < F(0.0f)) ? (F(1.0f) / material.ior) : material.ior;
    }
};

struct ScatterSampleRec { F3 l, f; F pdf; ScatterSampleRec() : pdf(0.0f) {} };    // globals.rs:89-104
struct LightSampleRec {                                                           // globals.rs:109-130
    F3 normal, emission, direction;
    F dist, pdf;
    LightSampleRec() : dist(0.0f), pdf(0.0f) {}
};

// ---------------------------------------------------------------------------
// RNG — replaces rand::thread_rng (tracer.rs:44).  The hash that builds the keys is the PCG hash of Jarzynski & Olano
// ("Hash Functions for GPU Rendering", JCGT 2020); the draws come from one PCG stream per path (below);
// u32 -> f32 exactly as rand 0.8.5's Standard distribution.
// ---------------------------------------------------------------------------
inline uint32_t pcg_hash(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// (seed, frame) -> two independently folded words: 64 bits of key per frame
struct FrameKey {
    uint32_t k0, k1;
};
inline FrameKey frame_key(uint64_t seed, uint64_t frame)
{
    FrameKey fk;
    uint32_t k = pcg_hash((uint32_t)(seed >> 32));
    k = pcg_hash(k ^ (uint32_t)seed);
    k = pcg_hash(k ^ (uint32_t)(frame >> 32));
    fk.k0 = pcg_hash(k ^ (uint32_t)frame);
    uint32_t j = pcg_hash((uint32_t)(seed >> 32) ^ 0x85EBCA6Bu);
    j = pcg_hash(j ^ (uint32_t)seed);
    j = pcg_hash(j ^ (uint32_t)(frame >> 32));
    fk.k1 = pcg_hash(j ^ (uint32_t)frame);
    return fk;
}
// One PCG stream per path: PCG-RXS-M-XS-32 (O'Neill, "PCG: A Family of Simple Fast Space-Efficient Statistically Good Algorithms
// for Random Number Generation", 2014) — a 32-bit LCG whose odd increment selects the stream, the RXS-M-XS output permutation —
// started from (state, increment) hashed from (seed, frame, pixel).  (Until round 3: pcg_hash(key + counter) with a 32-bit key —
// paths with nearby keys read overlapping windows of one sequence.)
struct Rng {
    uint32_t state, inc;
    Rng(FrameKey fk, uint32_t pixel_index)
    {
        const uint32_t a = pcg_hash(pixel_index), b = pcg_hash(a);
        state = pcg_hash(a ^ fk.k0);
        inc = pcg_hash(b ^ fk.k1) | 1u;
    }
    Rng(uint32_t state_, uint32_t inc_) : state(state_), inc(inc_ | 1u) {}          // tests: a stream at an explicit position
    uint32_t next_u32()
    {
        state = state * 747796405u + inc;
        uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        return (word >> 22u) ^ word;
    }
    F gen() { return F((float)(next_u32() >> 8) * (1.0f / 16777216.0f)); }
};
// tests: how many draws lie between two positions of one stream (at most `limit`)
inline uint32_t rng_draws_between(Rng from, const Rng& to, uint32_t limit = 64u)
{
    uint32_t n = 0u;
    while (from.state != to.state && n < limit) { (void)from.next_u32(); ++n; }
    return n;
}

// ---------------------------------------------------------------------------
// Scene — data-driven restatement of trait Scene (scene.rs:5-90) as implemented
// by AnalyticalScene (renderer/src/analytical.rs:11-205)
// ---------------------------------------------------------------------------

// analytical.rs:166-190 and, identically, scene.rs:39-63
inline bool sphere(const Ray& ray, F3 center, F radius, F& t_out)
{
    F3 l = center - ray.origin;
    F tca = l.dot(ray.direction);
    F d2 = l.dot(l) - tca * tca;
    F radius2 = radius * radius;
    if (d2 > radius2) return false;
    F thc = f_sqrt(radius2 - d2);
    F t0 = tca - thc;
    F t1 = tca + thc;
    if (t0 > t1) { F tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < F(0.0f)) {
        t0 = t1;
        if (t0 < F(0.0f)) return false;
    }
    t_out = t0;
    return true;
}

// analytical.rs:193-204 with the plane's normal / point / threshold as data
inline bool plane(const Ray& ray, const rpt_plane& p, F& t_out)
{
    F3 normal(p.normal[0], p.normal[1], p.normal[2]);
    F denom = dot(normal, ray.direction);
    if (f_abs(denom) > F(p.min_denom)) {
        F t = dot(F3(p.point[0], p.point[1], p.point[2]) - ray.origin, normal) / denom;
        if (t >= F(0.0f) && (!(p.max_t > 0.0f) || t <= F(p.max_t))) { t_out = t; return true; }   // max_t: project extension, 0 in the reference
    }
    return false;
}

struct Pinhole {                                                                   // camera/pinhole.rs:6-60
    F3 origin, center;
    F fov;
    Ray gen_ray(F px, F py, F offx, F offy, F width, F height) const               // pinhole.rs:38-60
    {
        F ratio = width / height;
        F pixel_size_x = F(1.0f) / width, pixel_size_y = F(1.0f) / height;
        // f32::to_radians = self * (PI / 180.0)
        F half_width = f_tan((fov * F(PI_F / 180.0f)) * F(0.5f));
        F half_height = half_width / ratio;
        F3 up_vector(0.0f, 1.0f, 0.0f);
        F3 w = (origin - center).normalize();
        F3 u = up_vector.cross(w);
        F3 v = w.cross(u);
        F3 lower_left = origin - u.mult_f(half_width) - v.mult_f(half_height) - w;
        F3 horizontal = u.mult_f(half_width * F(2.0f));
        F3 vertical = v.mult_f(half_height * F(2.0f));
        F3 rd = lower_left - origin;
        rd += horizontal.mult_f(pixel_size_x * offx + px);
        rd += vertical.mult_f(pixel_size_y * offy + py);
        return Ray(origin, rd.normalize());
    }
};

// Debug aid for tests: when set, every closest_hit / any_hit query appends {o, d, max_dist} (max_dist = -1
// for closest_hit) so a test can replay the exact rays of a pixel-sample through the device probes.
inline thread_local std::vector<float>* g_ray_log = nullptr;
// TESTS ONLY (tests/test_reference_screenshot.py): render with a quirk of the reference UNDONE, to show that the comparison with the
// reference's screenshot tells the two apart.  Bit 0: GTR1 with ln(a^2), the textbook form, instead of the reference's log2 (Q5).
// (Q3, any_hit honouring max_dist, is the scene flag RPT_SCENE_ANYHIT_USES_MAX_DIST.)  Always 0 otherwise.
inline uint32_t g_undo_quirks = 0u;
// Debug aid for tools/sched_sim.py: when set, sample_pixel appends one byte per path event: per bounce
// 'M' miss (path over) | 'E' emitter (over) | 'H' surface hit ('k' follows when its material has clearcoat != 0), then 'n' light sample not facing / 's' shadowed /
// 'v' unshadowed (disney_eval runs) / 'z' no lights, then the sampled lobe 'D' 'C' 'S', then 'x' if pdf <= 0 (over);
// '.' closes the sample.
inline thread_local std::vector<uint8_t>* g_event_log = nullptr;
inline void log_event(char c) { if (g_event_log) g_event_log->push_back((uint8_t)c); }

struct Scene {
    rpt_scene_desc d;
    std::vector<rpt_sphere> spheres;
    std::vector<rpt_plane> planes;
    std::vector<rpt_light> lights;
    std::vector<rpt_material> materials;
    std::vector<rpt_sdf_prim> sdf_prims;
    Pinhole pinhole;

    explicit Scene(const rpt_scene_desc& desc) : d(desc)
    {
        spheres.assign(desc.spheres, desc.spheres + desc.n_spheres);
        planes.assign(desc.planes, desc.planes + desc.n_planes);
        lights.assign(desc.lights, desc.lights + desc.n_lights);
        materials.assign(desc.materials, desc.materials + desc.n_materials);
        if (desc.sdf.n_prims) sdf_prims.assign(desc.sdf.prims, desc.sdf.prims + desc.sdf.n_prims);
        pinhole.origin = F3(desc.camera.origin[0], desc.camera.origin[1], desc.camera.origin[2]);
        pinhole.center = F3(desc.camera.center[0], desc.camera.center[1], desc.camera.center[2]);
        pinhole.fov = desc.camera.fov_deg;
    }

    size_t number_of_lights() const { return lights.size(); }                      // analytical.rs:152-154
    uint16_t recursion_depth() const { return (uint16_t)d.max_depth; }             // scene.rs:28-30

    // scene.rs:32-34 with the exponent as data (2.2 in the reference)
    F3 to_linear(F3 c) const
    {
        F g(d.background.gamma);
        return F3(f_powf(c.x, g), f_powf(c.y, g), f_powf(c.z, g));
    }

    // analytical.rs:28-32
    F3 background(const Ray& ray) const
    {
        const rpt_background& b = d.background;
        F3 ca(b.colour_a[0], b.colour_a[1], b.colour_a[2]);
        if (b.kind == RPT_BG_CONSTANT) return ca * F3::new_x(b.scale);
        F3 cb(b.colour_b[0], b.colour_b[1], b.colour_b[2]);
        F t = F(0.5f) * (ray.direction.y + F(1.0f));
        return to_linear((F(1.0f) - t) * ca + t * cb) * F3::new_x(b.scale);
    }

    // The material writes of analytical.rs:56-58 / 82-85 / 115-116 as a patch.
    void apply_material(const rpt_material& m, const Ray& ray, Material& out) const
    {
        if (m.mask & RPT_MAT_RGB) out.rgb = F3(m.rgb[0], m.rgb[1], m.rgb[2]);
        if (m.mask & RPT_MAT_EMISSION) out.emission = F3(m.emission[0], m.emission[1], m.emission[2]);
        if (m.mask & RPT_MAT_ANISOTROPIC) out.anisotropic = m.anisotropic;
        if (m.mask & RPT_MAT_METALLIC) out.metallic = m.metallic;
        if (m.mask & RPT_MAT_ROUGHNESS) out.roughness = m.roughness;
        if (m.mask & RPT_MAT_SUBSURFACE) out.subsurface = m.subsurface;
        if (m.mask & RPT_MAT_SPECULAR_TINT) out.specular_tint = m.specular_tint;
        if (m.mask & RPT_MAT_SHEEN) out.sheen = m.sheen;
        if (m.mask & RPT_MAT_SHEEN_TINT) out.sheen_tint = m.sheen_tint;
        if (m.mask & RPT_MAT_CLEARCOAT) out.clearcoat = m.clearcoat;
        if (m.mask & RPT_MAT_CLEARCOAT_GLOSS) out.clearcoat_gloss = m.clearcoat_gloss;
        if (m.mask & RPT_MAT_SPEC_TRANS) out.spec_trans = m.spec_trans;
        if (m.mask & RPT_MAT_IOR) out.ior = m.ior;
        if (m.mask & RPT_MAT_MEDIUM) {                                             // Material.medium as one more field (material.rs:75)
            out.medium.medium_type = m.medium_type;
            out.medium.density = m.medium_density;
            out.medium.color = F3(m.medium_color[0], m.medium_color[1], m.medium_color[2]);
            out.medium.anisotropy = m.medium_anisotropy;
        }
        if (m.proc_kind == RPT_PROC_CHECKER_DIR) {                                 // analytical.rs:107-115
            F s(m.proc_params[0]), o(m.proc_params[1]);
            F x = ray.direction.x / ray.direction.y * s + o;
            F y = ray.direction.z / ray.direction.y * s + o;
            F x1 = f_rem(f_floor(x), 2.0f);
            F y1 = f_rem(f_floor(y), 2.0f);
            F c = (f_rem(x1 + y1, 2.0f) < F(1.0f)) ? F(m.proc_params[2]) : F(m.proc_params[3]);
            out.rgb = F3(c, c, c);
        }
    }

    // ---- procedural SDF object (project-defined: include/rpt.h, rpt_sdf; no reference counterpart) ----
    F sdf_prim(const rpt_sdf_prim& pr, const F3& p) const
    {
        F3 q = p - F3(pr.center[0], pr.center[1], pr.center[2]);
        if (pr.kind == RPT_SDF_TORUS_Y) {
            F qx = f_sqrt(q.x * q.x + q.z * q.z) - F(pr.params[0]);
            return f_sqrt(qx * qx + q.y * q.y) - F(pr.params[1]);
        }
        return q.length() - F(pr.params[0]);
    }
    F sdf_eval(const F3& p) const
    {
        F k(d.sdf.smooth_k);
        F inv_k = F(1.0f) / k;                       // the spec multiplies by this f32 reciprocal (include/rpt.h)
        F dd = sdf_prim(sdf_prims[0], p);
        for (size_t i = 1; i < sdf_prims.size(); ++i) {
            F b = sdf_prim(sdf_prims[i], p);
            F h = f_max(k - f_abs(dd - b), 0.0f) * inv_k;
            F m = (dd < b) ? dd : b;
            dd = m - h * h * k * F(0.25f);
        }
        return dd;
    }
    bool sdf_march(const Ray& ray, F& t_out) const
    {
        F t(0.0f);
        for (uint32_t step = 0; step < d.sdf.max_steps; ++step) {
            F dist = sdf_eval(ray.at(t));
            if (dist < F(d.sdf.hit_eps) * t) { t_out = t; return true; }
            t = t + dist;
            if (t > F(d.sdf.max_t)) break;
        }
        return false;
    }
    F3 sdf_normal(const F3& p) const
    {
        F e(d.sdf.normal_eps);
        F3 k0(1.0f, -1.0f, -1.0f), k1(-1.0f, -1.0f, 1.0f), k2(-1.0f, 1.0f, -1.0f), k3(1.0f, 1.0f, 1.0f);
        F3 n = sdf_eval(p + e * k0) * k0 + sdf_eval(p + e * k1) * k1 + sdf_eval(p + e * k2) * k2 + sdf_eval(p + e * k3) * k3;
        return n.normalize();
    }

    // scene.rs:36-86 (default method Scene::sample_lights)
    bool sample_lights(const Ray& ray, State& state, LightSampleRec& light_sample) const
    {
        bool hit = false;
        F dist = state.hit_dist;                                                   // scene.rs:66 (stale across bounces)
        for (const rpt_light& light : lights) {
            if (light.type == RPT_LIGHT_SPHERICAL) {
                F3 pos(light.position[0], light.position[1], light.position[2]);
                F dd;
                if (sphere(ray, pos, light.radius, dd)) {
                    if (dd < dist) {
                        dist = dd;
                        F3 hit_point = ray.at(dd);
                        F cos_theta = dot(-ray.direction, normalize(hit_point - pos));
                        light_sample.pdf = (dist * dist) / (F(light.area) * cos_theta * F(0.5f));
                        light_sample.emission = F3(light.emission[0], light.emission[1], light.emission[2]);
                        state.is_emitter = true;
                        state.hit_dist = dd;
                        hit = true;
                    }
                }
            } else if (light.type == RPT_LIGHT_RECTANGULAR && (d.flags & RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES)) {
                // PROJECT-DEFINED (include/rpt.h, rpt_light): the reference intersects spherical lights only (scene.rs:68).
                // The parallelogram position + a*u + b*v, seen from the side its normal points to.
                F3 pos(light.position[0], light.position[1], light.position[2]);
                F3 u(light.u[0], light.u[1], light.u[2]), v(light.v[0], light.v[1], light.v[2]);
                F3 n = normalize(cross(u, v));
                if (dot(n, ray.direction) > F(0.0f)) continue;                     // back side: invisible
                F plane_w = dot(n, pos);
                F3 uu = u.mult_f(F(1.0f) / dot(u, u));
                F3 vv = v.mult_f(F(1.0f) / dot(v, v));
                F dt = dot(ray.direction, n);
                F dd = (plane_w - dot(n, ray.origin)) / dt;
                if (dd >= F(0.0f)) {                                               // (NaN and -inf fail this test)
                    F3 vi = ray.at(dd) - pos;
                    F a1 = dot(uu, vi);
                    if (a1 >= F(0.0f) && a1 <= F(1.0f)) {
                        F a2 = dot(vv, vi);
                        if (a2 >= F(0.0f) && a2 <= F(1.0f)) {
                            if (dd < dist) {
                                dist = dd;
                                F cos_theta = dot(-ray.direction, n);
                                light_sample.pdf = (dist * dist) / (F(light.area) * cos_theta);
                                light_sample.emission = F3(light.emission[0], light.emission[1], light.emission[2]);
                                state.is_emitter = true;
                                state.hit_dist = dd;
                                hit = true;
                            }
                        }
                    }
                }
            }
        }
        return hit;
    }

    // analytical.rs:36-127.  The first primitive is accepted whenever it is hit
    // (analytical.rs:43 has no `d < dist` test); the others only when nearer.
    bool closest_hit(const Ray& ray, State& state, LightSampleRec& light_sample) const
    {
        if (g_ray_log) for (float v : {rawf(ray.origin.x), rawf(ray.origin.y), rawf(ray.origin.z), rawf(ray.direction.x), rawf(ray.direction.y), rawf(ray.direction.z), -1.0f}) g_ray_log->push_back(v);
        F dist(3.40282347e+38f);                                                   // F::MAX, analytical.rs:38
        bool hit = false;
        bool first = true;
        for (const rpt_sphere& s : spheres) {
            F3 center(s.center[0], s.center[1], s.center[2]);
            F dd;
            MissedSphereTest mt;
            if (!sphere(ray, center, s.radius, dd)) mt.missed();
            else {
                if (first || dd < dist) {
                    F3 hp = ray.at(dd);
                    state.hit_dist = dd;
                    state.normal = normalize(hp - center);
                    apply_material(materials[s.material], ray, state.material);
                    hit = true;
                    dist = dd;
                }
            }
            first = false;
        }
        for (const rpt_plane& p : planes) {
            F dd;
            if (plane(ray, p, dd)) {
                if (first || dd < dist) {
                    state.hit_dist = dd;
                    state.normal = F3(p.normal[0], p.normal[1], p.normal[2]);
                    apply_material(materials[p.material], ray, state.material);
                    hit = true;
                    dist = dd;                                                     // (analytical.rs:118 omits this for the last primitive; no later test reads it)
                }
            }
            first = false;
        }
        if (!sdf_prims.empty()) {                                                  // the SDF object, tested last
            F dd;
            if (sdf_march(ray, dd)) {
                if (first || dd < dist) {
                    state.hit_dist = dd;
                    state.normal = sdf_normal(ray.at(dd));
                    apply_material(materials[d.sdf.material], ray, state.material);
                    hit = true;
                    dist = dd;
                }
            }
            first = false;
        }
        if (sample_lights(ray, state, light_sample)) hit = true;                   // analytical.rs:122-124
        return hit;
    }

    // analytical.rs:130-145; max_dist is ignored there (flag off)
    bool any_hit(const Ray& ray, F max_dist) const
    {
        if (g_ray_log) for (float v : {rawf(ray.origin.x), rawf(ray.origin.y), rawf(ray.origin.z), rawf(ray.direction.x), rawf(ray.direction.y), rawf(ray.direction.z), rawf(max_dist)}) g_ray_log->push_back(v);
        bool use_max = (d.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
        for (const rpt_sphere& s : spheres) {
            F dd;
            MissedSphereTest mt;
            if (!sphere(ray, F3(s.center[0], s.center[1], s.center[2]), s.radius, dd)) mt.missed();
            else if (!use_max || dd < max_dist) return true;
        }
        for (const rpt_plane& p : planes) {
            F dd;
            if (plane(ray, p, dd))
                if (!use_max || dd < max_dist) return true;
        }
        if (!sdf_prims.empty()) {
            F dd;
            if (sdf_march(ray, dd))
                if (!use_max || dd < max_dist) return true;
        }
        return false;
    }
};

// ---------------------------------------------------------------------------
// Tracer — rust-pathtracer/src/tracer.rs
// ---------------------------------------------------------------------------
struct Tracer {
    F eps;
    const Scene& scene;
    bool russian_roulette = false;                                                 // project extension, see sample_pixel
    explicit Tracer(const Scene& s) : eps(s.d.eps), scene(s) {}                    // tracer.rs:13-19

    static F power_heuristic(F a, F b) { F t = a * a; return t / (b * b + t); }    // tracer.rs:223-226

    static F gtr1(F ndoth, F a)                                                    // tracer.rs:233-240
    {
        if (a >= F(1.0f)) return F(INV_PI_F);
        F a2 = a * a;
        F t = F(1.0f) + (a2 - F(1.0f)) * ndoth * ndoth;
        if (g_undo_quirks & 1u) return (a2 - F(1.0f)) / (F(PI_F) * f_ln(a2) * t);     // (tests only: the textbook form, NOT the reference's)
        return (a2 - F(1.0f)) / (F(PI_F) * f_log2(a2) * t);                           // quirk Q5: log2, tracer.rs:239
    }

    static F3 sample_gtr1(F rgh, F r1, F /*r2*/)                                   // tracer.rs:242-254
    {
        F a = f_max(0.001f, rgh);
        F a2 = a * a;
        F phi = r1 * F(TWO_PI_F);
        F cos_theta = f_sqrt((F(1.0f) - f_powf(a2, F(1.0f) - r1)) / (F(1.0f) - a2));
        F sin_theta = f_clamp(f_sqrt(F(1.0f) - (cos_theta * cos_theta)), 0.0f, 1.0f);
        F sin_phi = f_sin(phi);
        F cos_phi = f_cos(phi);
        return F3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
    }

    static F3 sample_ggxvndf(const F3& v, F ax, F ay, F r1, F r2)                  // tracer.rs:256-274
    {
        F3 vh = normalize(F3(ax * v.x, ay * v.y, v.z));
        F lensq = vh.x * vh.x + vh.y * vh.y;
        F3 t_1 = (lensq > F(0.0f)) ? F3(-vh.y, vh.x, 0.0f).mult_f(F(1.0f) / f_sqrt(lensq)) : F3(1.0f, 0.0f, 0.0f);
        F3 t_2 = cross(vh, t_1);
        F r = f_sqrt(r1);
        F phi = F(2.0f * PI_F) * r2;
        F t1 = r * f_cos(phi);
        F t2 = r * f_sin(phi);
        F s = F(0.5f) * (F(1.0f) + vh.z);
        t2 = (F(1.0f) - s) * f_sqrt(F(1.0f) - t1 * t1) + s * t2;
        F3 nh = t1 * t_1 + t2 * t_2 + f_sqrt(f_max(0.0f, F(1.0f) - t1 * t1 - t2 * t2)) * vh;
        return normalize(F3(ax * nh.x, ay * nh.y, f_max(0.0f, nh.z)));
    }

    static F smithg(F ndotv, F alphag)                                             // tracer.rs:276-280
    {
        F a = alphag * alphag;
        F b = ndotv * ndotv;
        return (F(2.0f) * ndotv) / (ndotv + f_sqrt(a + b - a * b));
    }

    static F luminance(const F3& c)                                                // tracer.rs:284-286
    {
        return F(0.212671f) * c.x + F(0.715160f) * c.y + F(0.072169f) * c.z;
    }

    static F schlick_fresnel(F u)                                                  // tracer.rs:288-292
    {
        F m = f_clamp(F(1.0f) - u, 0.0f, 1.0f);
        F m2 = m * m;
        return m2 * m2 * m;
    }

    static F gtr2aniso(F ndoth, F hdotx, F hdoty, F ax, F ay)                      // tracer.rs:294-299
    {
        F a = hdotx / ax;
        F b = hdoty / ay;
        F c = a * a + b * b + ndoth * ndoth;
        return F(1.0f) / (F(PI_F) * ax * ay * c * c);
    }

    static F smithganiso(F ndotv, F vdotx, F vdoty, F ax, F ay)                    // tracer.rs:301-306
    {
        F a = vdotx * ax;
        F b = vdoty * ay;
        F c = ndotv;
        return (F(2.0f) * ndotv) / (ndotv + f_sqrt(a * a + b * b + c * c));
    }

    static F dielectric_fresnel(F cos_theta_i, F eta)                              // tracer.rs:308-322
    {
        F sin_theta_tsq = eta * eta * (F(1.0f) - cos_theta_i * cos_theta_i);
        if (sin_theta_tsq > F(1.0f)) return F(1.0f);
        F cos_theta_t = f_sqrt(f_max(F(1.0f) - sin_theta_tsq, 0.0f));
        F rs = (eta * cos_theta_t - cos_theta_i) / (eta * cos_theta_t + cos_theta_i);
        F rp = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
        return F(0.5f) * (rs * rs + rp * rp);
    }

    static F3 cosine_sample_hemisphere(F r1, F r2)                                 // tracer.rs:324-333
    {
        F3 dir;
        F r = f_sqrt(r1);
        F phi = F(TWO_PI_F) * r2;
        dir.x = r * f_cos(phi);
        dir.y = r * f_sin(phi);
        dir.z = f_sqrt(f_max(0.0f, F(1.0f) - dir.x * dir.x - dir.y * dir.y));
        return dir;
    }

    static void get_spec_color(const Material& material, F eta, F3& spec_col, F3& sheen_col)   // tracer.rs:335-341
    {
        F lum = luminance(material.rgb);
        F3 ctint = (lum > F(0.0f)) ? (material.rgb / F3::new_x(lum)) : F3(1.0f, 1.0f, 1.0f);
        F f0 = (F(1.0f) - eta) / (F(1.0f) + eta);
        spec_col = mix(f0 * f0 * mix(F3(1.0f, 1.0f, 1.0f), ctint, material.specular_tint), material.rgb, material.metallic);
        sheen_col = mix(F3(1.0f, 1.0f, 1.0f), ctint, material.sheen_tint);
    }

    static F3 eval_diffuse(const Material& material, const F3& c_sheen, const F3& v, const F3& l, const F3& h, F& pdf)   // tracer.rs:343-366
    {
        pdf = 0.0f;
        if (l.z <= F(0.0f)) return F3::zeros();
        F fl = schlick_fresnel(l.z);
        F fv = schlick_fresnel(v.z);
        F fh = schlick_fresnel(dot(l, h));
        F fd90 = F(0.5f) + F(2.0f) * dot(l, h) * dot(l, h) * material.roughness;
        F fd = mix_ptf(1.0f, fd90, fl) * mix_ptf(1.0f, fd90, fv);
        F fss90 = dot(l, h) * dot(l, h) * material.roughness;
        F fss = mix_ptf(1.0f, fss90, fl) * mix_ptf(1.0f, fss90, fv);
        F ss = F(1.25f) * (fss * (F(1.0f) / (l.z + v.z) - F(0.5f)) + F(0.5f));
        F3 fsheen = fh * material.sheen * c_sheen;
        pdf = l.z * F(INV_PI_F);
        return (F(1.0f) - material.metallic) * (F(1.0f) - material.spec_trans) *
               (F(INV_PI_F) * mix_ptf(fd, ss, material.subsurface) * material.rgb + fsheen);
    }

    static F disney_fresnel(const Material& material, F eta, F ldot_h, F vdot_h)   // tracer.rs:435-439
    {
        F metallic_fresnel = schlick_fresnel(ldot_h);
        F dielectric = dielectric_fresnel(f_abs(vdot_h), eta);
        return mix_ptf(dielectric, metallic_fresnel, material.metallic);
    }

    static F3 eval_spec_reflection(const Material& material, F eta, const F3& spec_col, const F3& v, const F3& l, const F3& h, F& pdf)   // tracer.rs:368-382
    {
        pdf = 0.0f;
        if (l.z <= F(0.0f)) return F3::zeros();
        F fm = disney_fresnel(material, eta, dot(l, h), dot(v, h));
        F3 f = mix(spec_col, F3(1.0f, 1.0f, 1.0f), fm);
        F d = gtr2aniso(h.z, h.x, h.y, material.ax, material.ay);
        F g1 = smithganiso(f_abs(v.z), v.x, v.y, material.ax, material.ay);
        F g2 = g1 * smithganiso(f_abs(l.z), l.x, l.y, material.ax, material.ay);
        pdf = g1 * d / (F(4.0f) * v.z);
        return d * g2 * f / F3::new_x(F(4.0f) * l.z * v.z);
    }

    static F3 eval_spec_refraction(const Material& material, F eta, const F3& v, const F3& l, const F3& h, F& pdf)   // tracer.rs:384-402
    {
        pdf = 0.0f;
        if (l.z >= F(0.0f)) return F3::zeros();
        F f = dielectric_fresnel(f_abs(dot(v, h)), eta);
        F d = gtr2aniso(h.z, h.x, h.y, material.ax, material.ay);
        F g1 = smithganiso(f_abs(v.z), v.x, v.y, material.ax, material.ay);
        F g2 = g1 * smithganiso(f_abs(l.z), l.x, l.y, material.ax, material.ay);
        F denom = dot(l, h) + dot(v, h) * eta;
        denom *= denom;
        F eta2 = eta * eta;
        F jacobian = f_abs(dot(l, h)) / denom;
        pdf = g1 * f_max(0.0f, dot(v, h)) * d * jacobian / v.z;
        return (F(1.0f) - material.metallic) * material.spec_trans * (F(1.0f) - f) * d * g2 * f_abs(dot(v, h)) * jacobian * eta2 /
               f_abs(l.z * v.z) * pow3(material.rgb, F3(0.5f, 0.5f, 0.5f));
    }

    static F3 eval_clearcoat(const Material& material, const F3& v, const F3& l, const F3& h, F& pdf)   // tracer.rs:404-419
    {
        pdf = 0.0f;
        if (l.z <= F(0.0f)) return F3::zeros();
        F fh = dielectric_fresnel(dot(v, h), F(1.0f / 1.5f));
        F f = mix_ptf(0.04f, 1.0f, fh);
        F d = gtr1(h.z, material.clearcoat_roughness);
        F g = smithg(l.z, 0.25f) * smithg(v.z, 0.25f);
        F jacobian = F(1.0f) / (F(4.0f) * dot(v, h));
        pdf = d * h.z * jacobian;
        return material.clearcoat * f * d * g / (F(4.0f) * l.z * v.z) * F3(0.25f, 0.25f, 0.25f);
    }

    static void get_lobe_probabilities(const Material& material, const F3& spec_col, F approx_fresnel,
                                       F& diffuse_wt, F& spec_reflect_wt, F& spec_refract_wt, F& clearcoat_wt)   // tracer.rs:421-433
    {
        diffuse_wt = luminance(material.rgb) * (F(1.0f) - material.metallic) * (F(1.0f) - material.spec_trans);
        spec_reflect_wt = luminance(mix(spec_col, F3(1.0f, 1.0f, 1.0f), approx_fresnel));
        spec_refract_wt = (F(1.0f) - approx_fresnel) * (F(1.0f) - material.metallic) * material.spec_trans * luminance(material.rgb);
        clearcoat_wt = F(0.25f) * material.clearcoat * (F(1.0f) - material.metallic);
        F total_wt = diffuse_wt + spec_reflect_wt + spec_refract_wt + clearcoat_wt;
        diffuse_wt /= total_wt;
        spec_reflect_wt /= total_wt;
        spec_refract_wt /= total_wt;
        clearcoat_wt /= total_wt;
    }

    // nested helpers of disney_sample / disney_eval / sample_light
    static void onb(const F3& n, F3& t, F3& b)                                     // tracer.rs:184-189, 449-454, 559-564
    {
        F3 up = (f_abs(n.z) < F(0.999f)) ? F3(0.0f, 0.0f, 1.0f) : F3(1.0f, 0.0f, 0.0f);
        t = normalize(cross(up, n));
        b = cross(n, t);
    }
    static F3 to_local(const F3& x, const F3& y, const F3& z, const F3& v) { return F3(dot(v, x), dot(v, y), dot(v, z)); }   // tracer.rs:456-458
    static F3 to_world(const F3& x, const F3& y, const F3& z, const F3& v) { return v.x * x + v.y * y + v.z * z; }             // tracer.rs:460-462
    static F3 reflect(F3 i, F3 n) { return i - F3(2.0f, 2.0f, 2.0f) * n * F3::new_x(dot(n, i)); }                              // tracer.rs:464-466
    static F3 refract(F3 i, F3 n, F eta)                                                                                      // tracer.rs:468-475
    {
        F k = F(1.0f) - eta * eta * (F(1.0f) - dot(n, i) * dot(n, i));
        if (k < F(0.0f)) return F3::zeros();
        return eta * i - (eta * dot(n, i) + f_sqrt(k)) * n;
    }

    // tracer.rs:441-553.  `l` is in/out: on entry it still holds the previous
    // bounce's world-space direction (zeros on the first bounce), and the specular
    // branch reads it before overwriting it (tracer.rs:531).
    F3 disney_sample(const State& state, F3 v, const F3& n, F3& l, F& pdf, Rng& rng) const
    {
        pdf = 0.0f;
        F3 f;
        F r1 = rng.gen();
        F r2 = rng.gen();

        F3 t, b;
        onb(n, t, b);
        v = to_local(t, b, n, v);

        F3 spec_col, sheen_col;
        get_spec_color(state.material, state.eta, spec_col, sheen_col);

        F diffuse_wt, spec_reflect_wt, spec_refract_wt, clearcoat_wt;
        F approx_fresnel = disney_fresnel(state.material, state.eta, v.z, v.z);
        get_lobe_probabilities(state.material, spec_col, approx_fresnel, diffuse_wt, spec_reflect_wt, spec_refract_wt, clearcoat_wt);

        F cdf[4];
        cdf[0] = diffuse_wt;
        cdf[1] = cdf[0] + clearcoat_wt;
        cdf[2] = cdf[1] + spec_reflect_wt;
        cdf[3] = cdf[2] + spec_refract_wt;

        if (r1 < cdf[0]) {                                                         // diffuse reflection lobe
            log_event('D');
            r1 /= cdf[0];
            l = cosine_sample_hemisphere(r1, r2);
            F3 h = normalize(l + v);
            f = eval_diffuse(state.material, sheen_col, v, l, h, pdf);
            pdf *= diffuse_wt;
        } else if (r1 < cdf[1]) {                                                  // clearcoat lobe
            log_event('C');
            r1 = (r1 - cdf[0]) / (cdf[1] - cdf[0]);
            F3 h = sample_gtr1(state.material.clearcoat_roughness, r1, r2);
            if (h.z < F(0.0f)) h = -h;
            l = normalize(reflect(-v, h));
            f = eval_clearcoat(state.material, v, l, h, pdf);
            pdf *= clearcoat_wt;
        } else {                                                                   // specular reflection / refraction lobes
            log_event('S');
            r1 = (r1 - cdf[1]) / (F(1.0f) - cdf[1]);
            F3 h = sample_ggxvndf(v, state.material.ax, state.material.ay, r1, r2);
            if (h.z < F(0.0f)) h = -h;
            F fresnel = disney_fresnel(state.material, state.eta, dot(l, h), dot(v, h));   // stale l: tracer.rs:531
            F ff = F(1.0f) - ((F(1.0f) - fresnel) * state.material.spec_trans * (F(1.0f) - state.material.metallic));
            F rand = rng.gen();
            if (rand < ff) {
                l = normalize(reflect(-v, h));
                f = eval_spec_reflection(state.material, state.eta, spec_col, v, l, h, pdf);
                pdf *= ff;
            } else {
                l = normalize(refract(-v, h, state.eta));
                f = eval_spec_refraction(state.material, state.eta, v, l, h, pdf);
                pdf *= F(1.0f) - ff;
            }
            pdf *= spec_reflect_wt + spec_refract_wt;
        }

        l = to_world(t, b, n, l);
        return f_abs(dot(n, l)) * f;
    }

    // tracer.rs:555-626
    F3 disney_eval(const State& state, F3 v_in, const F3& n, const F3& l_in, F& bsdf_pdf) const
    {
        bsdf_pdf = 0.0f;
        F3 f = F3::zeros();
        F3 t, b;
        onb(n, t, b);
        F3 v = to_local(t, b, n, v_in);
        F3 l = to_local(t, b, n, l_in);

        F3 h;
        if (l.z > F(0.0f)) h = normalize(l + v);
        else h = normalize(l + state.eta * v);
        if (h.z < F(0.0f)) h = -h;

        F3 spec_col, sheen_col;
        get_spec_color(state.material, state.eta, spec_col, sheen_col);

        F diffuse_wt, spec_reflect_wt, spec_refract_wt, clearcoat_wt;
        F fresnel = disney_fresnel(state.material, state.eta, dot(l, h), dot(v, h));
        get_lobe_probabilities(state.material, spec_col, fresnel, diffuse_wt, spec_reflect_wt, spec_refract_wt, clearcoat_wt);

        F pdf(0.0f);
        if (diffuse_wt > F(0.0f) && l.z > F(0.0f)) {
            f += eval_diffuse(state.material, sheen_col, v, l, h, pdf);
            bsdf_pdf += pdf * diffuse_wt;
        }
        if (spec_reflect_wt > F(0.0f) && l.z > F(0.0f) && v.z > F(0.0f)) {
            f += eval_spec_reflection(state.material, state.eta, spec_col, v, l, h, pdf);
            bsdf_pdf += pdf * spec_reflect_wt;
        }
        if (spec_refract_wt > F(0.0f) && l.z < F(0.0f)) {
            f += eval_spec_refraction(state.material, state.eta, v, l, h, pdf);
            bsdf_pdf += pdf * spec_refract_wt;
        }
        if (clearcoat_wt > F(0.0f) && l.z > F(0.0f) && v.z > F(0.0f)) {
            f += eval_clearcoat(state.material, v, l, h, pdf);
            bsdf_pdf += pdf * clearcoat_wt;
        }
        return f_abs(l.z) * f;
    }

    // tracer.rs:173-220 (only LightType::Spherical does anything)
    void sample_light(const rpt_light& light, const F3& scatter_pos, LightSampleRec& light_sample, Rng& rng) const
    {
        if (light.type != RPT_LIGHT_SPHERICAL) {
            // tracer.rs:217: `_ => {}` — the reference does nothing for the other two declared types.
            if (!(scene.d.flags & RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES)) return;
            // PROJECT-DEFINED from here (include/rpt.h, rpt_light)
            F3 position(light.position[0], light.position[1], light.position[2]);
            F3 emission(light.emission[0], light.emission[1], light.emission[2]);
            if (light.type == RPT_LIGHT_RECTANGULAR) {
                F r1 = rng.gen();
                F r2 = rng.gen();
                F3 u(light.u[0], light.u[1], light.u[2]), v(light.v[0], light.v[1], light.v[2]);
                F3 light_surface_pos = position + r1 * u + r2 * v;
                light_sample.direction = light_surface_pos - scatter_pos;
                light_sample.dist = length(light_sample.direction);
                F dist_sq = light_sample.dist * light_sample.dist;
                light_sample.direction /= F3::new_x(light_sample.dist);
                light_sample.normal = normalize(cross(u, v));
                light_sample.emission = F((float)scene.number_of_lights()) * emission;
                light_sample.pdf = dist_sq / (F(light.area) * f_abs(dot(light_sample.normal, light_sample.direction)));
            } else {                                                               // RPT_LIGHT_DISTANT: no draws
                light_sample.direction = normalize(position);
                light_sample.normal = normalize(scatter_pos - position);
                light_sample.emission = F((float)scene.number_of_lights()) * emission;
                light_sample.dist = F(INFINITY);
                light_sample.pdf = F(1.0f);
            }
            return;
        }
        F r1 = rng.gen();
        F r2 = rng.gen();
        F3 light_position(light.position[0], light.position[1], light.position[2]);
        F3 sphere_center_to_surface = scatter_pos - light_position;
        F dist_to_sphere_center = length(sphere_center_to_surface);
        // uniform_sample_hemisphere, tracer.rs:178-182
        F r = f_sqrt(f_max(0.0f, F(1.0f) - r1 * r1));
        F phi = F(TWO_PI_F) * r2;
        F3 sampled_dir(r * f_cos(phi), r * f_sin(phi), r1);
        sphere_center_to_surface /= F3::new_x(dist_to_sphere_center);
        F3 t, b;
        onb(sphere_center_to_surface, t, b);
        sampled_dir = sampled_dir.x * t + sampled_dir.y * b + sampled_dir.z * sphere_center_to_surface;
        F3 light_surface_pos = light_position + F(light.radius) * sampled_dir;
        light_sample.direction = light_surface_pos - scatter_pos;
        light_sample.dist = length(light_sample.direction);
        F dist_sq = light_sample.dist * light_sample.dist;
        light_sample.direction /= F3::new_x(light_sample.dist);
        light_sample.normal = normalize(light_surface_pos - light_position);
        light_sample.emission = F((float)scene.number_of_lights()) * F3(light.emission[0], light.emission[1], light.emission[2]);
        light_sample.pdf = dist_sq / (F(light.area) * F(0.5f) * f_abs(dot(light_sample.normal, light_sample.direction)));
    }

    // ---- participating media: PROJECT-DEFINED (include/rpt.h, "participating media"); no reference counterpart ----
    static F phase_hg(F cos_theta, F g)                                            // Henyey-Greenstein, cos against the direction back along the ray
    {
        F denom = F(1.0f) + g * g + F(2.0f) * g * cos_theta;
        return F(INV_4_PI_F) * (F(1.0f) - g * g) / (denom * f_sqrt(denom));
    }
    static F3 sample_hg(const F3& v, F g, F r1, F r2)
    {
        F cos_theta;
        if (f_abs(g) < F(0.001f)) cos_theta = F(1.0f) - F(2.0f) * r2;
        else {
            F sqr_term = (F(1.0f) - g * g) / (F(1.0f) + g - F(2.0f) * g * r2);
            cos_theta = -(F(1.0f) + g * g - sqr_term * sqr_term) / (F(2.0f) * g);
        }
        F phi = r1 * F(TWO_PI_F);
        F sin_theta = f_clamp(f_sqrt(F(1.0f) - (cos_theta * cos_theta)), 0.0f, 1.0f);
        F sin_phi = f_sin(phi);
        F cos_phi = f_cos(phi);
        F3 t, b;
        onb(v, t, b);
        return (sin_theta * cos_phi) * t + (sin_theta * sin_phi) * b + cos_theta * v;
    }
    // what is left of a light's radiance after `dist` inside the medium
    static F3 medium_transmittance(const Medium& md, F dist)
    {
        if (md.medium_type == RPT_MEDIUM_ABSORB)
            return F3(f_exp(-(((F(1.0f) - md.color.x) * dist) * md.density)), f_exp(-(((F(1.0f) - md.color.y) * dist) * md.density)),
                      f_exp(-(((F(1.0f) - md.color.z) * dist) * md.density)));
        if (md.medium_type == RPT_MEDIUM_SCATTER) return F3::new_x(f_exp(-(dist * md.density)));
        return F3(1.0f, 1.0f, 1.0f);
    }

    // tracer.rs:126-170.  The two extra arguments are the media extension (both off = the reference): `phase` — the
    // estimate is taken at a medium scatter point (state.fhp, no offset) with the phase function of that medium in place
    // of the BSDF; `inside` — the medium the path is in, which attenuates the light on its way to the point.
    F3 direct_light(const Ray& ray, const State& state, Rng& rng, const Medium* phase = nullptr, const Medium* inside = nullptr) const
    {
        F3 ld = F3::zeros();
        F3 scatter_pos = phase ? state.fhp : state.fhp + eps * state.ffnormal;
        ScatterSampleRec scatter_sample;
        size_t number_lights = scene.number_of_lights();
        if (number_lights > 0) {
            F random = rng.gen();
            random = random * F((float)number_lights);
            size_t index = (size_t)raw(random);                                    // `as usize`
            if (index >= number_lights) index = number_lights - 1;                 // the reference would panic here; unreachable for n < 2^24
            const rpt_light& light = scene.lights[index];
            LightSampleRec light_sample;
            sample_light(light, scatter_pos, light_sample, rng);
            F3 li = light_sample.emission;
            const bool facing = dot(light_sample.direction, light_sample.normal) < F(0.0f);
            if (!facing) log_event('n');
            if (facing) {
                Ray shadow_ray(scatter_pos, light_sample.direction);
                bool in_shadow = scene.any_hit(shadow_ray, light_sample.dist - eps);
                log_event(in_shadow ? 's' : 'v');
                if (!in_shadow) {
                    if (inside && raw(light_sample.dist) <= 3.40282347e+38f) li = li * medium_transmittance(*inside, light_sample.dist);
                    if (phase) {
                        F p = phase_hg(dot(-ray.direction, light_sample.direction), phase->anisotropy);
                        scatter_sample.f = F3::new_x(p);
                        scatter_sample.pdf = p;
                    } else {
                        scatter_sample.f = disney_eval(state, -ray.direction, state.ffnormal, light_sample.direction, scatter_sample.pdf);
                    }
                    F mis_weight(1.0f);
                    if (F(light.area) > F(0.0f)) mis_weight = power_heuristic(light_sample.pdf, scatter_sample.pdf);
                    if (scatter_sample.pdf > F(0.0f)) ld += mis_weight * li * (scatter_sample.f / F3::new_x(light_sample.pdf));
                }
            }
        } else {
            log_event('z');
        }
        return ld;
    }

    // One pixel-sample: the closure body of tracer.rs:33-117 up to `color`.
    //   col,row_mem: pixel position in the top-down buffer; returns the radiance.
    F3 sample_pixel(uint32_t col, uint32_t row_mem, uint32_t width_u, uint32_t height_u, FrameKey fkey) const
    {
        F width = F((float)width_u);
        F height = F((float)height_u);
        // tracer.rs:29-40: par_rchunks hands out rows from the END of the buffer, so
        // j = 0 is the last row in memory; i = j*width + col.
        uint32_t j = height_u - 1u - row_mem;
        F x = F((float)col);                                                       // (i % width) as F
        F y = height - F((float)j);                                                // height - (i / width) as F
        F xx = x / width;
        F yy = y / height;

        Rng rng(fkey, row_mem * width_u + col);
        F cam_off_x = rng.gen();                                                   // tracer.rs:45
        F cam_off_y = rng.gen();
        Ray ray = scene.pinhole.gen_ray(xx, F(1.0f) - yy, cam_off_x, cam_off_y, width, height);   // tracer.rs:46-47

        F3 radiance(0.0f, 0.0f, 0.0f);
        F3 throughput(1.0f, 1.0f, 1.0f);
        State state;
        LightSampleRec light_sample;
        ScatterSampleRec scatter_sample;
        state.depth = scene.recursion_depth();                                     // tracer.rs:57

        // PROJECT-DEFINED, off by default (include/rpt.h, RPT_RENDER_RUSSIAN_ROULETTE): the reference's loop has no
        // roulette (tracer.rs:61-103).  One more draw, after all other draws of the bounce.  True: the path ends.
        auto roulette = [&](uint16_t bounce) {
            if (russian_roulette && (uint32_t)bounce + 1u >= 2u && (uint32_t)bounce + 1u < (uint32_t)state.depth) {
                F q = f_max(f_max(throughput.x, throughput.y), throughput.z);
                q = f_clamp(q, 0.05f, 1.0f);
                F r = rng.gen();
                if (r >= q) { log_event('r'); return true; }
                throughput = throughput / F3::new_x(q);
            }
            return false;
        };
        // PROJECT-DEFINED, off by default (include/rpt.h, RPT_SCENE_MEDIA): the reference never reads a Medium.
        const bool media = (scene.d.flags & RPT_SCENE_MEDIA) != 0;
        bool in_medium = false;

        for (uint16_t bounce = 0; bounce < state.depth; ++bounce) {                // tracer.rs:61
            state.material = Material();
            if (media) state.is_emitter = false;                                   // (media: a path can go on after an emitter was seen)
            bool hit = scene.closest_hit(ray, state, light_sample);
            if (!hit) {
                log_event('M');
                radiance += scene.background(ray) * throughput;
                break;
            }
            if (media && in_medium) {                                              // the medium acts on the segment [0, hit_dist] first
                const Medium md = state.medium;
                F seg = state.hit_dist;
                if (md.medium_type == RPT_MEDIUM_ABSORB) {
                    throughput = throughput * medium_transmittance(md, seg);
                } else if (md.medium_type == RPT_MEDIUM_EMISSIVE) {
                    radiance += md.color.mult_f(seg).mult_f(md.density) * throughput;
                } else if (md.medium_type == RPT_MEDIUM_SCATTER) {
                    F r = rng.gen();
                    F d = f_min(-f_ln(r) / md.density, seg);
                    if (d < seg) {                                                 // a scatter event before the segment's end
                        log_event('V');
                        throughput = throughput * md.color;
                        ray.origin = ray.at(d);
                        state.fhp = ray.origin;
                        radiance += direct_light(ray, state, rng, &md, &md) * throughput;
                        F r1 = rng.gen();
                        F r2 = rng.gen();
                        F3 dir = sample_hg(-ray.direction, md.anisotropy, r1, r2);
                        scatter_sample.pdf = phase_hg(dot(-ray.direction, dir), md.anisotropy);
                        scatter_sample.l = dir;
                        ray.direction = dir;
                        if (roulette(bounce)) break;
                        continue;
                    }
                }
            }
            log_event(state.is_emitter ? 'E' : 'H');
            if (!state.is_emitter && raw(state.material.clearcoat) != 0.0f) log_event('k');   // surface with a clearcoat lobe
            state.finalize(ray);
            radiance += state.material.emission * throughput;
            if (state.is_emitter) {
                F mis_weight(1.0f);
                if (state.depth > 0) mis_weight = power_heuristic(scatter_sample.pdf, light_sample.pdf);
                radiance += mis_weight * light_sample.emission * throughput;
                break;
            }
            radiance += direct_light(ray, state, rng, nullptr, (media && in_medium) ? &state.medium : nullptr) * throughput;
            scatter_sample.f = disney_sample(state, -ray.direction, state.ffnormal, scatter_sample.l, scatter_sample.pdf, rng);
            if (scatter_sample.pdf > F(0.0f)) throughput = throughput * (scatter_sample.f / F3::new_x(scatter_sample.pdf));
            else { log_event('x'); break; }
            ray.direction = scatter_sample.l;
            ray.origin = state.fhp + eps * ray.direction;
            if (media && state.material.medium.medium_type != RPT_MEDIUM_NONE) {   // crossing (or staying on one side of) a medium's boundary
                in_medium = dot(ray.direction, state.normal) < F(0.0f);
                if (in_medium) state.medium = state.material.medium;
            }
            if (roulette(bounce)) break;
        }
        log_event('.');
        return radiance;
    }

    // tracer.rs:22-123 for `spp` consecutive frames on a top-down RGBA f32 buffer.
    // Rows [row_begin, row_end) only (the whole image by default) so callers can
    // time a bounded sample of a large frame.
    void render(float* pixels, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp, uint64_t seed,
                uint32_t row_begin, uint32_t row_end) const
    {
#ifdef RPT_ORACLE_F64
        // (the f64 instantiation keeps the call's running mean in f64 as well: `pixels` is read once and written once)
        std::vector<Raw> mean_store((size_t)(row_end - row_begin) * width * 4);
        Raw* const mean = mean_store.data() - (size_t)row_begin * width * 4;
        for (size_t i = (size_t)row_begin * width * 4; i < (size_t)row_end * width * 4; ++i) mean[i] = pixels[i];
#else
        float* const mean = pixels;
#endif
        for (uint32_t s = 0; s < spp; ++s) {
            uint64_t frames = frames_done + s;
            FrameKey fkey = frame_key(seed, frames);
            F v = F(1.0f) / F((float)(frames + 1));                                // tracer.rs:115
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t jj = 0; jj < (int64_t)(row_end - row_begin); ++jj) {      // one task per scanline, tracer.rs:24,29-32
                uint32_t row_mem = row_end - 1u - (uint32_t)jj;                    // bottom-up like par_rchunks
                for (uint32_t col = 0; col < width; ++col) {
                    F3 radiance = sample_pixel(col, row_mem, width, height, fkey);
                    Raw* pixel = mean + ((size_t)row_mem * width + col) * 4;
                    F color[4] = {radiance.x, radiance.y, radiance.z, F(1.0f)};    // tracer.rs:59,105
                    for (int c = 0; c < 4; ++c)                                    // mix_color, tracer.rs:108-113
                        pixel[c] = raw((F(1.0f) - v) * F(pixel[c]) + color[c] * v);
                }
            }
        }
#ifdef RPT_ORACLE_F64
        for (size_t i = (size_t)row_begin * width * 4; i < (size_t)row_end * width * 4; ++i) pixels[i] = (float)mean[i];
#endif
    }
};

// ---------------------------------------------------------------------------
// Denoiser — PROJECT-DEFINED (include/rpt.h, "denoiser"): "Implement a denoiser" is a Todo of the reference (Readme.md:14);
// nothing to restate.  This IS the specification the device kernels are compared with, bit for bit.
// ---------------------------------------------------------------------------
inline void denoise(const float* in, float* out, uint32_t width, uint32_t height, uint32_t iterations, float edge_k)
{
    const size_t n = (size_t)width * height;
    std::vector<float> a(n * 3), b(n * 3);
    for (size_t p = 0; p < n; ++p)
        for (int c = 0; c < 3; ++c) a[p * 3 + c] = in[p * 4 + c] / (1.0f + in[p * 4 + c]);
    static const float H[3] = {0.25f, 0.5f, 0.25f};
    float k = edge_k;
    for (uint32_t it = 0; it < iterations; ++it) {
        const int64_t s = (int64_t)1 << it;
        for (int64_t y = 0; y < (int64_t)height; ++y)
            for (int64_t x = 0; x < (int64_t)width; ++x) {
                const float* cp = &a[((size_t)y * width + x) * 3];
                float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int64_t qx = x + s * dx, qy = y + s * dy;
                        if (qx < 0 || qy < 0 || qx >= (int64_t)width || qy >= (int64_t)height) continue;
                        const float* cq = &a[((size_t)qy * width + qx) * 3];
                        const float d0 = cp[0] - cq[0], d1 = cp[1] - cq[1], d2c = cp[2] - cq[2];
                        // (fused multiply-adds, written out: part of the specification since round 4 — a third fewer operations on a
                        //  pass that was instruction-bound, and nothing here restates the reference)
                        const float d2 = __builtin_fmaf(d0, d0, __builtin_fmaf(d1, d1, d2c * d2c));
                        if (!(d2 == d2)) continue;
                        const float t = __builtin_fmaf(-d2, k, 1.0f);
                        const float g = t > 0.0f ? t : 0.0f;
                        const float wt = (H[dy + 1] * H[dx + 1]) * (g * g);
                        for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf(cq[c], wt, acc[c]);
                        wsum = wsum + wt;
                    }
                float* o = &b[((size_t)y * width + x) * 3];
                for (int c = 0; c < 3; ++c) o[c] = wsum > 0.0f ? acc[c] / wsum : cp[c];
            }
        a.swap(b);
        k = k * 4.0f;
    }
    for (size_t p = 0; p < n; ++p) {
        const float* i4 = in + p * 4;
        const bool finite = std::isfinite(i4[0]) && std::isfinite(i4[1]) && std::isfinite(i4[2]);
        for (int c = 0; c < 3; ++c) out[p * 4 + c] = finite ? a[p * 3 + c] / (1.0f - a[p * 3 + c]) : i4[c];
        out[p * 4 + 3] = i4[3];
    }
}

}  // namespace rpt_oracle
