// dev_integrator.h — scene queries and the per-sample path of
// rust-pathtracer/src/tracer.rs:33-117 for small analytical scenes (SceneSmall).
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_INTEGRATOR_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_INTEGRATOR_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_INTEGRATOR_H_PLAIN
#else
#define RPT_DEV_INTEGRATOR_H_NORMAL
#endif

#include "dev_bsdf.h"
#include "dev_media.h"
#include "dev_scene.h"

namespace RPT_NS {
using namespace rptscene;

// Defined in dev_scene_large.h.  Declared here because the templates below call them and nothing in their argument types (a scene of
// namespace rptscene, an index) leads argument-dependent lookup to this namespace.
RPT_DEV DevLight light_at(const SceneLarge& sc, uint32_t index);
RPT_DEV DevMedium medium_at(const SceneLarge& sc, uint32_t index);

struct RayD {
    v3 o, d;
};

// A wave-uniform scene constant (an SGPR) passed through an empty asm: what is computed from it — radius * radius, the
// f64 image of the background's exponent — is then recomputed where it is used (one instruction) instead of being hoisted
// out of the path loop into a VGPR that lives, and spills, for the whole kernel.
// (readfirstlane: a no-op where the value already is in an SGPR; where register pressure made the compiler keep the constant
// in a VGPR it is what lets the constraint be met at all.)
RPT_DEV uint32_t uniform_here(uint32_t x)
{
    x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    asm volatile("" : "+s"(x));
    return x;
}
RPT_DEV float uniform_here(float x) { return rpt_u2f(uniform_here(rpt_f2u(x))); }

// (Round 4 tried fetching a whole table record — sphere, plane, light, material patch — in as few wide scalar loads as it has
// power-of-two pieces, all requested up front, instead of the field-by-field loads the compiler emits, each behind its own wait:
// geometry and light records +0.3 % / -0.5 % on configs[1] / [3], material patches -2.2 % / -1.4 % (a patch of 29 dwords of which
// the mask selects three or four; four more spilled SGPRs).  Not taken; the SDF object's primitive records, read every march
// step and used whole, are the exception: sdf_prim_at.  profiles/r4/experiments/wide_record_loads.txt)
// analytical.rs:166-190 == scene.rs:39-63
RPT_DEV bool hit_sphere(const RayD& ray, v3 center, float radius, float& t)
{
    v3 l = center - ray.o;
    float tca = dot3(l, ray.d);
    float d2 = dot3(l, l) - tca * tca;
    float radius2 = radius * radius;
    if (d2 > radius2) return false;
    float thc = fsqrt(radius2 - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0.0f) {
        t0 = t1;
        if (t0 < 0.0f) return false;
    }
    t = t0;
    return true;
}

// analytical.rs:193-204 (normal / point / threshold from the table)
RPT_DEV bool hit_plane(const RayD& ray, const DevPlane& p, float& t)
{
    v3 n = mk3(p.nx, p.ny, p.nz);
    float denom = dot3(n, ray.d);
    if (__builtin_fabsf(denom) > p.min_denom) {
        float tt = fdiv(dot3(mk3(p.px, p.py, p.pz) - ray.o, n), denom);
        if (tt >= 0.0f && (!(p.max_t > 0.0f) || tt <= p.max_t)) { t = tt; return true; }
    }
    return false;
}

// Overlay one material patch on `m` for the lanes where `on` holds: the field
// writes of analytical.rs:56-58 / 82-85 / 115-116.  The patch and its mask are
// wave-uniform (SGPRs); only the select is per lane.
RPT_DEV void apply_patch_fields(Mat& m, const DevMaterial& p, bool on)
{
    if (p.mask & RPT_MAT_RGB) { m.rgb.x = on ? p.rgb[0] : m.rgb.x; m.rgb.y = on ? p.rgb[1] : m.rgb.y; m.rgb.z = on ? p.rgb[2] : m.rgb.z; }
    if (p.mask & RPT_MAT_EMISSION) { m.emission.x = on ? p.emission[0] : m.emission.x; m.emission.y = on ? p.emission[1] : m.emission.y; m.emission.z = on ? p.emission[2] : m.emission.z; }
    if (p.mask & RPT_MAT_ANISOTROPIC) m.anisotropic = on ? p.anisotropic : m.anisotropic;
    if (p.mask & RPT_MAT_METALLIC) m.metallic = on ? p.metallic : m.metallic;
    if (p.mask & RPT_MAT_ROUGHNESS) m.roughness = on ? p.roughness : m.roughness;
    if (p.mask & RPT_MAT_SUBSURFACE) m.subsurface = on ? p.subsurface : m.subsurface;
    if (p.mask & RPT_MAT_SPECULAR_TINT) m.specular_tint = on ? p.specular_tint : m.specular_tint;
    if (p.mask & RPT_MAT_SHEEN) m.sheen = on ? p.sheen : m.sheen;
    if (p.mask & RPT_MAT_SHEEN_TINT) m.sheen_tint = on ? p.sheen_tint : m.sheen_tint;
    if (p.mask & RPT_MAT_CLEARCOAT) m.clearcoat = on ? p.clearcoat : m.clearcoat;
    if (p.mask & RPT_MAT_CLEARCOAT_GLOSS) m.clearcoat_gloss = on ? p.clearcoat_gloss : m.clearcoat_gloss;
    if (p.mask & RPT_MAT_SPEC_TRANS) m.spec_trans = on ? p.spec_trans : m.spec_trans;
    if (p.mask & RPT_MAT_IOR) m.ior = on ? p.ior : m.ior;
}
// The checker of analytical.rs:107-115 along a ray's direction: true on the squares of the SECOND colour (proc_params[3]).
RPT_DEV bool checker_second(float scale, float offset, v3 dir)
{
    float x = fdiv(dir.x, dir.y) * scale + offset;
    float y = fdiv(dir.z, dir.y) * scale + offset;
    float x1 = rem2(__builtin_floorf(x));
    float y1 = rem2(__builtin_floorf(y));
    return !(rem2(x1 + y1) < 1.0f);
}
RPT_DEV bool checker_second(const DevMaterial& p, v3 dir) { return checker_second(p.proc_params[0], p.proc_params[1], dir); }
RPT_DEV void apply_patch(Mat& m, const DevMaterial& p, bool on, v3 dir)
{
    apply_patch_fields(m, p, on);
    if (p.proc_kind == RPT_PROC_CHECKER_DIR) {                      // analytical.rs:107-115
        if (on) {
            float c = checker_second(p, dir) ? p.proc_params[3] : p.proc_params[2];
            m.rgb = mk3(c, c, c);
        }
    }
}
// The same patch for a row of a material table (MaterialTable, below): which of the checker's two colours is the row's, not the ray's.
RPT_DEV void apply_patch_row(Mat& m, const DevMaterial& p, bool on, bool second)
{
    apply_patch_fields(m, p, on);
    if (p.proc_kind == RPT_PROC_CHECKER_DIR) {
        if (on) {
            float c = second ? p.proc_params[3] : p.proc_params[2];
            m.rgb = mk3(c, c, c);
        }
    }
}

// ---- procedural SDF object (include/rpt.h rpt_sdf; restated in oracle/rpt_oracle.hpp) ----
RPT_DEV float sdf_prim(const DevSdfPrim& pr, v3 p)
{
    v3 q = p - mk3(pr.cx, pr.cy, pr.cz);
    if (pr.kind == RPT_SDF_TORUS_Y) {
        float qx = fsqrt(q.x * q.x + q.z * q.z) - pr.p0;
        return fsqrt(qx * qx + q.y * q.y) - pr.p1;
    }
    return len3(q) - pr.p0;
}

// A primitive's record as the two scalar loads it is laid out for (dev_scene.h, DevSdfPrim): {cx, cy, cz, p0} and {p1, kind}, both
// requested before anything is computed.  (Left to itself the compiler fetches the fields one by one, each right before its use and
// each behind its own wait: it is saving scalar registers.)
// One primitive's distance from its record's values.
RPT_DEV float sdf_prim_value(float cx, float cy, float cz, float p0, float p1, uint32_t kind, v3 p)
{
    const v3 q = p - mk3(cx, cy, cz);
    if (kind == RPT_SDF_TORUS_Y) {
        const float qx = fsqrt(q.x * q.x + q.z * q.z) - p0;
        return fsqrt(qx * qx + q.y * q.y) - p1;
    }
    return len3(q) - p0;
}
RPT_DEV float sdf_prim_at(const DevSdf& sd, uint32_t i, v3 p)
{
    typedef float rpt_f4 __attribute__((ext_vector_type(4)));
    typedef float rpt_f2 __attribute__((ext_vector_type(2)));
    const RPT_CONST_AS char* rec = (const RPT_CONST_AS char*)&sd.prims[i];
    const rpt_f4 a = *(const RPT_CONST_AS rpt_f4*)rec;
    const rpt_f2 b = *(const RPT_CONST_AS rpt_f2*)(rec + 16);
    return sdf_prim_value(a.x, a.y, a.z, a.w, b.x, rpt_f2u(b.y), p);
}
// the smooth union's step: the object so far (dd) and one more primitive (b)
RPT_DEV float sdf_smooth_union(float dd, float b, float k, float inv_k)
{
    float h = rmax(k - __builtin_fabsf(dd - b), 0.0f) * inv_k;
    float m = (dd < b) ? dd : b;
    return m - h * h * k * 0.25f;
}

RPT_DEV float sdf_eval(const DevSdf& sd, v3 p)
{
    const float k = sd.smooth_k;
    float dd = sdf_prim_at(sd, 0u, p);
    for (uint32_t i = 1; i < sd.n_prims; ++i) dd = sdf_smooth_union(dd, sdf_prim_at(sd, i, p), k, sd.inv_smooth_k);
    return dd;
}

// The object's records held in scalar registers across a loop of evaluations (a march phase of k_sdf.hip's kernels that know the number
// of primitives): sdf_eval above fetches every record again at every evaluation, three scalar loads each behind its own wait per march
// step for three primitives.  Same values, same operations.
template <uint32_t N>
struct SdfRegs {
    float k, inv_k;
    float cx[N], cy[N], cz[N], p0[N], p1[N];
    uint32_t kind[N];
};
template <uint32_t N>
RPT_DEV void sdf_regs_load(const DevSdf& sd, SdfRegs<N>& r)
{
    typedef float rpt_f4 __attribute__((ext_vector_type(4)));
    typedef float rpt_f2 __attribute__((ext_vector_type(2)));
    r.k = uniform_here(sd.smooth_k); r.inv_k = uniform_here(sd.inv_smooth_k);
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) {
        const RPT_CONST_AS char* rec = (const RPT_CONST_AS char*)&sd.prims[i];
        const rpt_f4 a = *(const RPT_CONST_AS rpt_f4*)rec;
        const rpt_f2 b = *(const RPT_CONST_AS rpt_f2*)(rec + 16);
        r.cx[i] = uniform_here(a.x); r.cy[i] = uniform_here(a.y); r.cz[i] = uniform_here(a.z); r.p0[i] = uniform_here(a.w);
        r.p1[i] = uniform_here(b.x); r.kind[i] = uniform_here(rpt_f2u(b.y));
    }
}
template <uint32_t N>
RPT_DEV float sdf_eval(const SdfRegs<N>& r, v3 p)
{
    float dd = sdf_prim_value(r.cx[0], r.cy[0], r.cz[0], r.p0[0], r.p1[0], r.kind[0], p);
#pragma unroll
    for (uint32_t i = 1; i < N; ++i) dd = sdf_smooth_union(dd, sdf_prim_value(r.cx[i], r.cy[i], r.cz[i], r.p0[i], r.p1[i], r.kind[i], p), r.k, r.inv_k);
    return dd;
}

// Sphere marching.  Lanes leave the loop at different step counts; the loop runs until
// the wave's last lane is done (no cross-lane compaction inside a bounce: the parked-lane
// vote of the kernel works at bounce granularity).
// `t_useful`: a hit at or beyond this t would be rejected by the caller (it is not nearer than the
// primitive already found, or not within the shadow ray's max_dist), and t only grows along the
// march, so the march may stop there — the reference-order march would go on and find a hit the caller
// then discards.  Pass +inf when every hit counts.
RPT_DEV bool sdf_march(const DevSdf& sd, const RayD& ray, float t_useful, float& t_out)
{
    float t = 0.0f;
    for (uint32_t step = 0; step < sd.max_steps; ++step) {
        float dist = sdf_eval(sd, ray.o + t * ray.d);
        if (dist < sd.hit_eps * t) { t_out = t; return true; }
        t = t + dist;
        if (t > sd.max_t) break;
        if (t > t_useful) break;
    }
    return false;
}

RPT_DEV v3 sdf_normal(const DevSdf& sd, v3 p)
{
    const float e = sd.normal_eps;
    const v3 k0 = mk3(1.0f, -1.0f, -1.0f), k1 = mk3(-1.0f, -1.0f, 1.0f), k2 = mk3(-1.0f, 1.0f, -1.0f), k3 = mk3(1.0f, 1.0f, 1.0f);
    v3 n = sdf_eval(sd, p + e * k0) * k0 + sdf_eval(sd, p + e * k1) * k1 + sdf_eval(sd, p + e * k2) * k2 + sdf_eval(sd, p + e * k3) * k3;
    return norm3(n);
}

// Path state that survives from one bounce to the next (tracer.rs:51-57):
// hit_dist is deliberately NOT reset per bounce (scene.rs:66 reads the stale value).
// The reference also keeps light_sample and state.is_emitter alive across bounces
// (tracer.rs:53-54), but they are only ever read in the bounce that wrote them (an
// emitter hit ends the path, tracer.rs:77-87), so here they are per-bounce locals.
struct PathState {
    float hit_dist;            // State.hit_dist, starts at -1 (globals.rs:28)
    float scatter_pdf;         // ScatterSampleRec.pdf of the previous bounce (MIS, tracer.rs:81)
};


// Result of a sphere march done elsewhere (the resumable-march kernel runs the march as its own
// scheduling state and hands the outcome to the unchanged closest_hit / any_hit arithmetic).
struct SdfMarchResult {
    bool hit;
    float t;
};

// The spheres-then-planes part of closest_hit: nearest accepted distance and who won.
struct AnalyticHit {
    float dist;
    bool hit;
    uint32_t accepted;                                              // bit i: primitive i's material writes happened
    v3 c;                                                           // centre of the winning sphere
    v3 pn;                                                          // normal of the winning plane
    bool win_plane;
};

RPT_DEV void analytic_closest(const SceneSmall& sc, const RayD& ray, AnalyticHit& a)
{
    a.dist = 3.40282347e+38f;                                       // F::MAX
    a.hit = false;
    a.accepted = 0;
    a.c = mk3(0.0f, 0.0f, 0.0f);
    a.pn = mk3(0.0f, 0.0f, 0.0f);
    a.win_plane = false;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const DevSphere& s = sc.spheres[i];
        float t;
        bool h = hit_sphere(ray, mk3(s.cx, s.cy, s.cz), uniform_here(s.radius), t);
        bool acc = h && (i == 0 || t < a.dist);                     // analytical.rs:43 has no distance test for the first one
        if (acc) {
            a.dist = t;
            a.c = mk3(s.cx, s.cy, s.cz);
            a.win_plane = false;
            a.hit = true;
            a.accepted |= 1u << i;
        }
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevPlane& p = sc.planes[k];
        float t;
        bool h = hit_plane(ray, p, t);
        bool acc = h && ((sc.n_spheres == 0 && k == 0) || t < a.dist);
        if (acc) {
            a.dist = t;
            a.pn = mk3(p.nx, p.ny, p.nz);
            a.win_plane = true;
            a.hit = true;
            a.accepted |= 1u << (kMaxSpheres + k);
        }
    }
}

// How far the SDF march of closest_hit needs to look: a hit at or beyond the analytic winner loses.
RPT_DEV float sdf_primary_t_useful(const SceneSmall& sc, const AnalyticHit& a)
{
    const bool first = (sc.n_spheres == 0 && sc.n_planes == 0);
    return first ? __builtin_inff() : a.dist;
}

// What closest_hit's geometry pass leaves behind for the passes that only a surface hit needs (normal,
// material): one dword.  Small scenes: the mask of accepted primitives (bit i: sphere i, bit kMaxSpheres + k:
// plane k, bit kMaxSpheres + kMaxPlanes: the SDF object); the winner is the last accepted one, since the
// reference tests spheres, then planes, then (project extension) the SDF object, each against the running
// distance.  Large scenes: see dev_scene_large.h.
struct GeomHit {
    uint32_t code;
};

// What Scene::sample_lights reports when the ray reaches a light first (scene.rs:71-80).
struct EmitterHit {
    bool is_emitter;
    float light_pdf;
    v3 light_emission;
};

// One light of Scene::sample_lights (scene.rs:65-85): spherical lights as in the reference; rectangular ones
// (project-defined, include/rpt.h rpt_light) when the scene opts in.  `ldist` is the loop's running distance.
RPT_DEV bool light_intersect(const DevLight& L, uint32_t scene_flags, const RayD& ray, PathState& ps, EmitterHit& e, float& ldist)
{
    v3 pos = mk3(L.px, L.py, L.pz);
    if (L.type == RPT_LIGHT_SPHERICAL) {
        float t;
        if (hit_sphere(ray, pos, L.radius, t)) {
            if (t < ldist) {
                ldist = t;
                v3 hit_point = ray.o + t * ray.d;
                float cos_theta = dot3(-ray.d, norm3(hit_point - pos));
                e.light_pdf = fdiv(ldist * ldist, L.area * cos_theta * 0.5f);
                e.light_emission = mk3(L.ex, L.ey, L.ez);
                e.is_emitter = true;
                ps.hit_dist = t;
                return true;
            }
        }
    } else if (L.type == RPT_LIGHT_RECTANGULAR && (scene_flags & RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES)) {
        const v3 u = mk3(L.ux, L.uy, L.uz), v = mk3(L.vx, L.vy, L.vz);
        const v3 n = norm3(cross3(u, v));
        if (dot3(n, ray.d) > 0.0f) return false;                    // back side: invisible
        const float plane_w = dot3(n, pos);
        const v3 uu = scale3(u, fdiv(1.0f, dot3(u, u)));
        const v3 vv = scale3(v, fdiv(1.0f, dot3(v, v)));
        const float dt = dot3(ray.d, n);
        const float t = fdiv(plane_w - dot3(n, ray.o), dt);
        if (t >= 0.0f) {
            const v3 vi = (ray.o + t * ray.d) - pos;
            const float a1 = dot3(uu, vi);
            if (a1 >= 0.0f && a1 <= 1.0f) {
                const float a2 = dot3(vv, vi);
                if (a2 >= 0.0f && a2 <= 1.0f) {
                    if (t < ldist) {
                        ldist = t;
                        const float cos_theta = dot3(-ray.d, n);
                        e.light_pdf = fdiv(ldist * ldist, L.area * cos_theta);
                        e.light_emission = mk3(L.ex, L.ey, L.ez);
                        e.is_emitter = true;
                        ps.hit_dist = t;
                        return true;
                    }
                }
            }
        }
    }
    return false;
}

// Scene::sample_lights, scene.rs:36-86, over the kernarg light table
RPT_DEV bool sample_lights_small(const SceneSmall& sc, const RayD& ray, PathState& ps, EmitterHit& e, bool hit)
{
    float ldist = ps.hit_dist;
    for (uint32_t i = 0; i < sc.n_lights; ++i) {
        const DevLight& L = sc.lights[i];
        hit = light_intersect(L, sc.flags, ray, ps, e, ldist) || hit;
    }
    return hit;
}

// Geometry pass of AnalyticalScene::closest_hit (analytical.rs:36-127) + Scene::sample_lights
// (scene.rs:36-86) over the tables: who was hit and how far; no normal, no material.
// The spheres-then-planes result when it was computed earlier (the march kernel needs it before the march, to know
// how far the march has to look): nearest accepted distance and the accepted mask.
struct AnalyticPre {
    float dist;
    uint32_t accepted;
};

template <bool SDF>
RPT_DEV bool closest_geom_small(const SceneSmall& sc, const DevSdf* sdf, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e,
                                const SdfMarchResult* pre = nullptr, const AnalyticPre* apre = nullptr)
{
    float dist;
    bool hit;
    uint32_t accepted;
    if (apre) {
        dist = apre->dist; accepted = apre->accepted; hit = accepted != 0u;
    } else {
        AnalyticHit a;
        analytic_closest(sc, ray, a);
        dist = a.dist; hit = a.hit; accepted = a.accepted;
    }
    if (SDF) {                                                      // the SDF object, tested last
        float t;
        const bool first = (sc.n_spheres == 0 && sc.n_planes == 0);
        bool h;
        if (pre) { h = pre->hit; t = pre->t; }
        else h = sdf_march(*sdf, ray, first ? __builtin_inff() : dist, t);   // sdf_primary_t_useful
        bool acc = h && (first || t < dist);
        if (acc) {
            dist = t;
            hit = true;
            accepted |= 1u << (kMaxSpheres + kMaxPlanes);
        }
    }
    if (hit) ps.hit_dist = dist;                                    // analytical.rs:48,79,104
    g.code = accepted;
    return sample_lights_small(sc, ray, ps, e, hit);
}

// material = Material::new() then the accepted primitives' writes, in order (analytical.rs:56-58, 82-85, 115-116)
template <bool SDF>
RPT_DEV void material_small(const SceneSmall& sc, const DevSdf* sdf, const RayD& ray, uint32_t accepted, Mat& mat)
{
    mat_defaults(mat);
    for (uint32_t i = 0; i < sc.n_spheres; ++i)
        apply_patch(mat, sc.materials[sc.spheres[i].material], (accepted >> i) & 1u, ray.d);
    for (uint32_t k = 0; k < sc.n_planes; ++k)
        apply_patch(mat, sc.materials[sc.planes[k].material], (accepted >> (kMaxSpheres + k)) & 1u, ray.d);
    if (SDF)
        apply_patch(mat, sc.materials[sdf->material], (accepted >> (kMaxSpheres + kMaxPlanes)) & 1u, ray.d);
}

// Only Material.emission of the same layering (what an emitter exit needs, tracer.rs:74).
RPT_DEV void apply_patch_emission(v3& em, const DevMaterial& p, bool on)
{
    if (p.mask & RPT_MAT_EMISSION) { em.x = on ? p.emission[0] : em.x; em.y = on ? p.emission[1] : em.y; em.z = on ? p.emission[2] : em.z; }
}
template <bool SDF>
RPT_DEV v3 emission_small(const SceneSmall& sc, const DevSdf* sdf, uint32_t accepted)
{
    v3 em = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < sc.n_spheres; ++i)
        apply_patch_emission(em, sc.materials[sc.spheres[i].material], (accepted >> i) & 1u);
    for (uint32_t k = 0; k < sc.n_planes; ++k)
        apply_patch_emission(em, sc.materials[sc.planes[k].material], (accepted >> (kMaxSpheres + k)) & 1u);
    if (SDF)
        apply_patch_emission(em, sc.materials[sdf->material], (accepted >> (kMaxSpheres + kMaxPlanes)) & 1u);
    return em;
}

// Surface pass: normal (only the final one: the reference also computes the normals of
// accepted-then-superseded primitives, which nothing reads) of a surface hit at `dist`.
template <bool SDF>
RPT_DEV v3 normal_small(const SceneSmall& sc, const DevSdf* sdf, const RayD& ray, float dist, uint32_t accepted)
{
    v3 c = mk3(0.0f, 0.0f, 0.0f), pn = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const DevSphere& s = sc.spheres[i];
        const bool acc = (accepted >> i) & 1u;
        c.x = acc ? s.cx : c.x; c.y = acc ? s.cy : c.y; c.z = acc ? s.cz : c.z;
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevPlane& p = sc.planes[k];
        const bool acc = (accepted >> (kMaxSpheres + k)) & 1u;
        pn.x = acc ? p.nx : pn.x; pn.y = acc ? p.ny : pn.y; pn.z = acc ? p.nz : pn.z;
    }
    const bool win_sdf = SDF && ((accepted >> (kMaxSpheres + kMaxPlanes)) & 1u);
    const bool win_plane = ((accepted >> kMaxSpheres) & ((1u << kMaxPlanes) - 1u)) != 0u;
    v3 hp = ray.o + dist * ray.d;                                   // ray.at(d)
    v3 sn;
    if (SDF && win_sdf) sn = sdf_normal(*sdf, hp);
    else sn = norm3(hp - c);
    const bool use_pn = win_plane && !win_sdf;
    return mk3(use_pn ? pn.x : sn.x, use_pn ? pn.y : sn.y, use_pn ? pn.z : sn.z);   // (per component: a struct select goes through scratch)
}

// closest_hit in two passes: geometry (TRACE) and surface (normal, material: SHADE)
RPT_DEV bool closest_geom(const SceneSmall& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) { return closest_geom_small<false>(sc, nullptr, ray, ps, g, e); }
RPT_DEV bool closest_geom(const SceneSmallSdf& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) { return closest_geom_small<true>(sc, &sc.sdf, ray, ps, g, e); }
RPT_DEV v3 hit_emission(const SceneSmall& sc, const GeomHit& g) { return emission_small<false>(sc, nullptr, g.code); }
RPT_DEV v3 hit_emission(const SceneSmallSdf& sc, const GeomHit& g) { return emission_small<true>(sc, &sc.sdf, g.code); }
RPT_DEV v3 hit_normal(const SceneSmall& sc, const RayD& ray, float dist, const GeomHit& g) { return normal_small<false>(sc, nullptr, ray, dist, g.code); }
RPT_DEV v3 hit_normal(const SceneSmallSdf& sc, const RayD& ray, float dist, const GeomHit& g) { return normal_small<true>(sc, &sc.sdf, ray, dist, g.code); }
RPT_DEV void hit_material(const SceneSmall& sc, const RayD& ray, const GeomHit& g, Mat& mat) { material_small<false>(sc, nullptr, ray, g.code, mat); }
RPT_DEV void hit_material(const SceneSmallSdf& sc, const RayD& ray, const GeomHit& g, Mat& mat) { material_small<true>(sc, &sc.sdf, ray, g.code, mat); }

// Media (dev_media.h): which material's Medium the layered material of a hit carries — the last accepted primitive whose
// patch writes the medium (RPT_MAT_MEDIUM), kNoMediumIdx when none does (Material::new()'s Medium is MediumType::None).
template <bool SDF>
RPT_DEV uint32_t medium_index_small(const SceneSmall& sc, const DevSdf* sdf, uint32_t accepted)
{
    uint32_t idx = kNoMediumIdx;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const uint32_t mi = sc.spheres[i].material;
        if (sc.materials[mi].mask & RPT_MAT_MEDIUM) idx = ((accepted >> i) & 1u) ? mi : idx;
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const uint32_t mi = sc.planes[k].material;
        if (sc.materials[mi].mask & RPT_MAT_MEDIUM) idx = ((accepted >> (kMaxSpheres + k)) & 1u) ? mi : idx;
    }
    if (SDF) {
        const uint32_t mi = sdf->material;
        if (sc.materials[mi].mask & RPT_MAT_MEDIUM) idx = ((accepted >> (kMaxSpheres + kMaxPlanes)) & 1u) ? mi : idx;
    }
    return idx;
}
RPT_DEV uint32_t hit_medium_index(const SceneSmall& sc, const GeomHit& g) { return medium_index_small<false>(sc, nullptr, g.code); }
RPT_DEV uint32_t hit_medium_index(const SceneSmallSdf& sc, const GeomHit& g) { return medium_index_small<true>(sc, &sc.sdf, g.code); }

// The Medium of material `index` (a per-lane index: a select chain over the wave-uniform table), finalized.
RPT_DEV DevMedium medium_at(const SceneSmall& sc, uint32_t index)
{
    uint32_t type = RPT_MEDIUM_NONE;
    float density = 0.0f, cx = 0.0f, cy = 0.0f, cz = 0.0f, aniso = 0.0f;
    for (uint32_t i = 0; i < sc.n_materials; ++i) {
        const DevMaterial& m = sc.materials[i];
        const bool pick = (index == i);
        type = pick ? m.medium_type : type;
        density = pick ? m.medium_density : density;
        cx = pick ? m.medium_color[0] : cx; cy = pick ? m.medium_color[1] : cy; cz = pick ? m.medium_color[2] : cz;
        aniso = pick ? m.medium_anisotropy : aniso;
    }
    return DevMedium{type, density, mk3(cx, cy, cz), clampf(aniso, -0.9f, 0.9f)};             // material.rs:126
}

// AnalyticalScene::any_hit (analytical.rs:130-145); it ignores max_dist unless the
// scene opts in.
RPT_DEV bool any_hit_analytic(const SceneSmall& sc, const RayD& ray, float max_dist)
{
    bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    bool occluded = false;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const DevSphere& s = sc.spheres[i];
        float t;
        bool h = hit_sphere(ray, mk3(s.cx, s.cy, s.cz), uniform_here(s.radius), t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        float t;
        bool h = hit_plane(ray, sc.planes[k], t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}

// How far the SDF march of any_hit needs to look.
RPT_DEV float sdf_shadow_t_useful(const SceneSmall& sc, float max_dist)
{
    return (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) ? max_dist : __builtin_inff();
}

template <bool SDF>
RPT_DEV bool any_hit_small(const SceneSmall& sc, const DevSdf* sdf, const RayD& ray, float max_dist, const SdfMarchResult* pre = nullptr)
{
    bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    bool occluded = any_hit_analytic(sc, ray, max_dist);
    if (SDF) {
        float t;
        bool h;
        if (pre) { h = pre->hit; t = pre->t; }
        else h = sdf_march(*sdf, ray, sdf_shadow_t_useful(sc, max_dist), t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}
RPT_DEV bool any_hit(const SceneSmall& sc, const RayD& ray, float max_dist) { return any_hit_small<false>(sc, nullptr, ray, max_dist); }
RPT_DEV bool any_hit(const SceneSmallSdf& sc, const RayD& ray, float max_dist) { return any_hit_small<true>(sc, &sc.sdf, ray, max_dist); }

// analytical.rs:28-32 + scene.rs:32-34
template <class S>
RPT_DEV v3 background(const S& sc, const RayD& ray)
{
    const DevBackground& b = sc.bg;
    v3 ca = mk3(b.ax, b.ay, b.az);
    if (b.kind == RPT_BG_CONSTANT) return ca * splat3(b.scale);
    v3 cb = mk3(b.bx, b.by, b.bz);
    float t = 0.5f * (ray.d.y + 1.0f);
    v3 c = (1.0f - t) * ca + t * cb;
    const float gamma = uniform_here(b.gamma);
    v3 lin = mk3(rpt_powf(c.x, gamma), rpt_powf(c.y, gamma), rpt_powf(c.z, gamma));
    return lin * splat3(b.scale);
}

struct LightSample {
    v3 normal, emission, direction;
    float dist, pdf;
};

// tracer.rs:173-220 (LightType::Spherical; the other types are no-ops there)
template <class S>
RPT_DEV void sample_light(const S& sc, const DevLight& L, v3 scatter_pos, LightSample& ls, Rng& rng)
{
    ls.normal = mk3(0.0f, 0.0f, 0.0f); ls.emission = mk3(0.0f, 0.0f, 0.0f); ls.direction = mk3(0.0f, 0.0f, 0.0f);
    ls.dist = 0.0f; ls.pdf = 0.0f;                                  // LightSampleRec::new, globals.rs:119-129
    if (L.type != RPT_LIGHT_SPHERICAL) {
        // tracer.rs:217: the reference does nothing for the other declared types; project-defined when the scene opts in
        if (!(sc.flags & RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES)) return;
        const v3 position = mk3(L.px, L.py, L.pz);
        if (L.type == RPT_LIGHT_RECTANGULAR) {
            float r1 = rng.gen();
            float r2 = rng.gen();
            const v3 u = mk3(L.ux, L.uy, L.uz), v = mk3(L.vx, L.vy, L.vz);
            v3 surface_pos = position + r1 * u + r2 * v;
            ls.direction = surface_pos - scatter_pos;
            ls.dist = len3(ls.direction);
            float dist_sq = ls.dist * ls.dist;
            ls.direction = divs3(ls.direction, ls.dist);
            ls.normal = norm3(cross3(u, v));
            ls.emission = sc.n_lights_f * mk3(L.ex, L.ey, L.ez);
            ls.pdf = fdiv(dist_sq, L.area * __builtin_fabsf(dot3(ls.normal, ls.direction)));
        } else {                                                    // RPT_LIGHT_DISTANT: no draws
            ls.direction = norm3(position);
            ls.normal = norm3(scatter_pos - position);
            ls.emission = sc.n_lights_f * mk3(L.ex, L.ey, L.ez);
            ls.dist = __builtin_inff();
            ls.pdf = 1.0f;
        }
        return;
    }
    float r1 = rng.gen();
    float r2 = rng.gen();
    v3 lpos = mk3(L.px, L.py, L.pz);
    v3 c2s = scatter_pos - lpos;
    float dist_to_center = len3(c2s);
    float r = fsqrt(rmax(0.0f, 1.0f - r1 * r1));          // uniform_sample_hemisphere
    float phi = kTwoPi * r2;
    float sn, cs;
    rpt_sincosf(phi, &sn, &cs);
    v3 sampled = mk3(r * cs, r * sn, r1);
    c2s = divs3(c2s, dist_to_center);
    v3 t, b;
    onb(c2s, t, b);
    sampled = sampled.x * t + sampled.y * b + sampled.z * c2s;
    v3 surface_pos = lpos + L.radius * sampled;
    ls.direction = surface_pos - scatter_pos;
    ls.dist = len3(ls.direction);
    float dist_sq = ls.dist * ls.dist;
    ls.direction = divs3(ls.direction, ls.dist);
    ls.normal = norm3(surface_pos - lpos);
    ls.emission = sc.n_lights_f * mk3(L.ex, L.ey, L.ez);
    ls.pdf = fdiv(dist_sq, L.area * 0.5f * __builtin_fabsf(dot3(ls.normal, ls.direction)));
}

// Scene::light_at (analytical.rs:148-150) for a per-lane index: a select chain over the
// (wave-uniform, SGPR-resident) table of a small scene.
RPT_DEV DevLight light_at(const SceneSmall& sc, uint32_t index)
{
    DevLight L = sc.lights[0];
    const uint32_t n_lights = uniform_here(sc.n_lights);
    for (uint32_t i = 1; i < n_lights; ++i) {
        const DevLight& Li = sc.lights[i];
        bool pick = (index == i);
        L.type = pick ? Li.type : L.type;
        L.px = pick ? Li.px : L.px; L.py = pick ? Li.py : L.py; L.pz = pick ? Li.pz : L.pz;
        L.ex = pick ? Li.ex : L.ex; L.ey = pick ? Li.ey : L.ey; L.ez = pick ? Li.ez : L.ez;
        L.radius = pick ? Li.radius : L.radius; L.area = pick ? Li.area : L.area;
        L.ux = pick ? Li.ux : L.ux; L.uy = pick ? Li.uy : L.uy; L.uz = pick ? Li.uz : L.uz;
        L.vx = pick ? Li.vx : L.vx; L.vy = pick ? Li.vy : L.vy; L.vz = pick ? Li.vz : L.vz;
    }
    return L;
}

// How the integrator asks the scene its two questions.  The default asks directly; the resumable-march
// kernel substitutes a query that carries the outcome of a march it ran as a separate scheduling state.
// `any` may also leave the question open (`pending`: the wavefront form walks the shadow ray in another kernel and is
// handed what the radiance gains if the answer turns out to be "free", through park()); the direct forms never do.
struct DirectQuery {
    template <class S> RPT_DEV bool any(const S& sc, const RayD& ray, float max_dist, v3, bool& pending) const { pending = false; return any_hit(sc, ray, max_dist); }
    template <class S> RPT_DEV bool geom(const S& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) const { return closest_geom(sc, ray, ps, g, e); }
    RPT_DEV void park(v3) const {}
};

// Where SHADE takes the material of a hit from.  The reference builds it at every hit: Material::new(), the accepted primitives'
// writes, State::finalize, and then — in disney_sample and again in disney_eval — the specular and sheen colours (material_small,
// mat_finalize, get_spec_color).  All of it is a function of WHICH primitives were accepted, of which colour of a checker the ray
// looks at, and of the side the ray comes from (eta): a handful of cases when the scene has a handful of primitives.
struct MaterialPerHit {
    static constexpr bool kTable = false;
    typedef Mat MatType;
};
// A kernel for scenes of at most kMatTableBits primitives (k_small.hip, k_compact.hip, k_sdf.hip) computes every case once per workgroup, with the same
// functions, into rows of LDS (material_table_row) and SHADE keeps its row's address: what the BSDF code needs of the material it
// reads there, where it needs it (dev_bsdf.h, MatRow) — for what was ~45 selects behind uniform branches on the patches' masks, a
// square root and nine divides per hit, and more inside disney_eval / disney_sample.  The checker itself stays per ray.
// Row index, nb = ns + np + SDF: accepted spheres | accepted planes << ns | SDF object accepted << (ns + np) | the checker's second
// colour << nb | (normal . ray < 0) << (nb + 1).  The host launches such a kernel only when nb <= kMatTableBits and at most ONE
// primitive's material is procedural (one bit for "second colour").
constexpr uint32_t kMatRowFloat4s = 8u;          // 128 B
constexpr uint32_t kMatTableBits = 3u;
constexpr uint32_t kMatTableRows = 1u << (kMatTableBits + 2u);       // 4 KB
// ... and of at most kMatTableBitsWide primitives in the megakernel that takes the table's shape as data (k_small.hip, render_small_regen_table_kernel:
// it has the LDS for 64 rows; the compacting and the SDF kernels do not)
constexpr uint32_t kMatTableBitsWide = 4u;
constexpr uint32_t kMatTableRowsWide = 1u << (kMatTableBitsWide + 2u);       // 8 KB
template <bool SDF>
struct MaterialTable {
    static constexpr bool kTable = true;
    typedef MatRow MatType;
    const float4* rows;
    uint32_t ns, np;                           // the scene's n_spheres and n_planes (literals in the kernels that know them)
    // the one procedural material there may be: its primitive's bit in the accepted mask (0: none) and the checker's scale and offset —
    // found once per workgroup (material_table_procedural), not behind two dependent scalar loads per primitive at every hit
    uint32_t proc_bit;
    float proc_scale, proc_offset;
    template <class S>
    RPT_DEV void fetch(const S& sc, const RayD& ray, uint32_t accepted, bool ndd_negative, MatRow& m, float& eta, v3& emission) const
    {
        (void)sc;
        bool second = false;
        if (proc_bit != 0u) { if (accepted & proc_bit) second = checker_second(proc_scale, proc_offset, ray.d); }
        uint32_t row = (accepted & ((1u << ns) - 1u)) | (((accepted >> kMaxSpheres) & ((1u << np) - 1u)) << ns);
        uint32_t nb = ns + np;
        if constexpr (SDF) {
            row |= ((accepted >> (kMaxSpheres + kMaxPlanes)) & 1u) << nb;
            nb += 1u;
        }
        row |= ((second ? 1u : 0u) << nb) | ((ndd_negative ? 1u : 0u) << (nb + 1u));
        const float4* r = rows + row * kMatRowFloat4s;
        const float4 em = r[1];
        emission = mk3(em.x, em.y, em.z);
        eta = r[4].w;
        m.more = r + kMatRowMore;
    }
};
// The table's procedural primitive: wave-uniform, looked up once per workgroup.
template <bool SDF, class S, class T>
RPT_DEV void material_table_procedural(const S& sc, uint32_t ns, uint32_t np, T& t)
{
    t.proc_bit = 0u; t.proc_scale = 0.0f; t.proc_offset = 0.0f;
    for (uint32_t i = 0; i < ns; ++i) {
        const DevMaterial& pm = sc.materials[sc.spheres[i].material];
        if (pm.proc_kind == RPT_PROC_CHECKER_DIR) { t.proc_bit = 1u << i; t.proc_scale = pm.proc_params[0]; t.proc_offset = pm.proc_params[1]; }
    }
    for (uint32_t k = 0; k < np; ++k) {
        const DevMaterial& pm = sc.materials[sc.planes[k].material];
        if (pm.proc_kind == RPT_PROC_CHECKER_DIR) { t.proc_bit = 1u << (kMaxSpheres + k); t.proc_scale = pm.proc_params[0]; t.proc_offset = pm.proc_params[1]; }
    }
    if constexpr (SDF) {
        const DevMaterial& pm = sc.materials[sc.sdf.material];
        if (pm.proc_kind == RPT_PROC_CHECKER_DIR) { t.proc_bit = 1u << (kMaxSpheres + kMaxPlanes); t.proc_scale = pm.proc_params[0]; t.proc_offset = pm.proc_params[1]; }
    }
}
// One row of that table (the lane that owns it calls this; any pass's functions: the row holds their results): the material of the
// accepted set `set` (spheres | planes << ns | SDF object << (ns + np)) on the squares of the checker's first or `second` colour, seen
// from the side `ndd_negative` says, into rows[row].
template <bool SDF, class S>
RPT_DEV void material_table_row_of(const S& sc, uint32_t ns, uint32_t np, uint32_t set, bool second, bool ndd_negative, uint32_t row, float4* rows)
{
    Mat m;
    mat_defaults(m);
    for (uint32_t i = 0; i < ns; ++i) apply_patch_row(m, sc.materials[sc.spheres[i].material], (set >> i) & 1u, second);
    for (uint32_t k = 0; k < np; ++k) apply_patch_row(m, sc.materials[sc.planes[k].material], (set >> (ns + k)) & 1u, second);
    if constexpr (SDF) apply_patch_row(m, sc.materials[sc.sdf.material], (set >> (ns + np)) & 1u, second);
    mat_finalize(m);
    const float eta = ndd_negative ? fdiv(1.0f, m.ior) : m.ior;
    v3 spec_col, sheen_col;
    get_spec_color(m, eta, spec_col, sheen_col);
    MatRowValues x;
    mat_row_derive(m, x);
    const uint64_t l2 = rpt_d2u(x.cc_log2_a2);
    float4* r = rows + row * kMatRowFloat4s;
    r[0] = make_float4(m.rgb.x, m.rgb.y, m.rgb.z, m.metallic);
    r[1] = make_float4(m.emission.x, m.emission.y, m.emission.z, m.roughness);
    r[2] = make_float4(spec_col.x, spec_col.y, spec_col.z, m.subsurface);
    r[3] = make_float4(sheen_col.x, sheen_col.y, sheen_col.z, m.sheen);
    r[4] = make_float4(m.clearcoat, m.clearcoat_roughness, m.spec_trans, eta);
    r[5] = make_float4(m.ax, m.ay, rpt_u2f((uint32_t)l2), rpt_u2f((uint32_t)(l2 >> 32)));
    r[kMatRowMore + 0] = make_float4(x.lum, x.w_diffuse, x.w_clearcoat, x.one_m_metallic);
    r[kMatRowMore + 1] = make_float4(x.dm, x.gtr1_a2m1, x.gtr1_k, x.cc_a2);
    static_assert(kMatRowMore + 2 == (int)kMatRowFloat4s, "row layout");
}
template <bool SDF, class S>
RPT_DEV void material_table_row(const S& sc, uint32_t ns, uint32_t np, uint32_t row, float4* rows)
{
    const uint32_t nb = ns + np + (SDF ? 1u : 0u);
    material_table_row_of<SDF>(sc, ns, np, row & ((1u << nb) - 1u), (row >> nb) & 1u, (row >> (nb + 1u)) & 1u, row, rows);
}

// The table by CLASS of accepted set (launch.h, MatClassMap: scenes of five to twelve primitives): row = class | the checker's second
// colour << 4 | (normal . ray < 0) << 5, the class looked up in a byte table in LDS.  Otherwise MaterialTable<false>.
constexpr uint32_t kMatClassBits = 4u;
struct MaterialTableMapped {
    static constexpr bool kTable = true;
    typedef MatRow MatType;
    const float4* rows;
    const uint8_t* cls;                        // [4096] in LDS
    uint32_t ns, np;
    uint32_t proc_bit;
    float proc_scale, proc_offset;
    template <class S>
    RPT_DEV void fetch(const S& sc, const RayD& ray, uint32_t accepted, bool ndd_negative, MatRow& m, float& eta, v3& emission) const
    {
        (void)sc;
        bool second = false;
        if (proc_bit != 0u) { if (accepted & proc_bit) second = checker_second(proc_scale, proc_offset, ray.d); }
        const uint32_t set = (accepted & ((1u << ns) - 1u)) | (((accepted >> kMaxSpheres) & ((1u << np) - 1u)) << ns);
        const uint32_t row = (uint32_t)cls[set] | ((second ? 1u : 0u) << kMatClassBits) | ((ndd_negative ? 1u : 0u) << (kMatClassBits + 1u));
        const float4* r = rows + row * kMatRowFloat4s;
        const float4 em = r[1];
        emission = mk3(em.x, em.y, em.z);
        eta = r[4].w;
        m.more = r + kMatRowMore;
    }
};

// Head of direct_light (tracer.rs:130-145): pick a light, sample it.  Returns the facing test of tracer.rs:147.
// OFFSET false: the estimate is taken at a point inside a medium (media, dev_media.h): scatter_pos is `fhp` itself.
template <bool OFFSET = true, class S>
RPT_DEV bool nee_sample(const S& sc, v3 fhp, v3 ffnormal, Rng& rng, v3& scatter_pos, float& light_area, LightSample& ls)
{
    if (OFFSET) scatter_pos = fhp + sc.eps * ffnormal;
    else scatter_pos = fhp;
    float random = rng.gen();
    random = random * sc.n_lights_f;
    uint32_t index = (uint32_t)random;                              // `as usize`
    const uint32_t n_lights = uniform_here(sc.n_lights);
    index = (index >= n_lights) ? n_lights - 1u : index;            // the reference would panic; unreachable for n < 2^24

    const DevLight L = light_at(sc, index);                         // Scene::light_at for a per-lane index
    light_area = L.area;
    sample_light(sc, L, scatter_pos, ls, rng);
    return dot3(ls.direction, ls.normal) < 0.0f;
}

// direct_light (tracer.rs:126-170) in two halves: the GEOMETRIC half — pick a light, sample it, shadow ray
// (tracer.rs:130-152) — needs only the hit point and the facing normal; the RADIOMETRIC half — disney_eval, MIS weight,
// contribution (tracer.rs:155-164) — needs the material.  (Running the first half BEFORE the material is built keeps the
// shadow query out of the shading block's register peak; measured: no spills left in a 4-wave large-scene kernel, but 1.2 %
// slower on BASELINE configs[1], so the reference's order is kept.)
struct NeeQuery {
    bool lit;                  // a light was sampled, it faces the point, and the shadow ray is free (or its answer is pending)
    bool pending;              // the shadow query was left open (DirectQuery)
    float light_area;
    LightSample ls;
};

template <bool OFFSET = true, class S, class Q>
RPT_DEV NeeQuery nee_query(const S& sc, const Q& q, v3 fhp, v3 ffnormal, Rng& rng, v3 throughput)
{
    NeeQuery n;
    n.lit = false;
    n.pending = false;
    n.light_area = 0.0f;
    n.ls.normal = mk3(0.0f, 0.0f, 0.0f); n.ls.emission = mk3(0.0f, 0.0f, 0.0f); n.ls.direction = mk3(0.0f, 0.0f, 0.0f);
    n.ls.dist = 0.0f; n.ls.pdf = 0.0f;
    if (sc.n_lights == 0) return n;
    v3 scatter_pos;
    bool facing;
    { RPT_PROF(PB_NEE_SAMPLE); facing = nee_sample<OFFSET>(sc, fhp, ffnormal, rng, scatter_pos, n.light_area, n.ls); }
    if (facing) {
        RayD shadow{scatter_pos, n.ls.direction};
        bool in_shadow;
        { RPT_PROF(PB_ANYHIT); in_shadow = q.any(sc, shadow, n.ls.dist - sc.eps, throughput, n.pending); }
        n.lit = !in_shadow;
    }
    return n;
}

template <class MT>
RPT_DEV v3 nee_eval(const NeeQuery& n, const MT& mat, float eta, const ShadeFrame& fr, v3 ffnormal)
{
    v3 ld = mk3(0.0f, 0.0f, 0.0f);
    if (n.lit) {
        RPT_PROF(PB_EVAL);
        v3 li = n.ls.emission;
        float bsdf_pdf;
        v3 f = disney_eval(mat, eta, fr, ffnormal, n.ls.direction, bsdf_pdf);
        float mis_weight = 1.0f;
        if (n.light_area > 0.0f) mis_weight = power_heuristic(n.ls.pdf, bsdf_pdf);
        if (bsdf_pdf > 0.0f) ld = ld + (mis_weight * li) * divs3(f, n.ls.pdf);
    }
    return ld;
}

// Camera ray for pixel-relative coordinates (px, py) = coord of tracer.rs:46 and
// jitter (offx, offy): the per-sample tail of Pinhole::gen_ray (pinhole.rs:56-59).
RPT_DEV RayD camera_ray(const DevCamera& cam, float px, float py, float offx, float offy)
{
    v3 rd = mk3(cam.rdx, cam.rdy, cam.rdz);
    rd = rd + scale3(mk3(cam.hx, cam.hy, cam.hz), cam.psx * offx + px);
    rd = rd + scale3(mk3(cam.vx, cam.vy, cam.vz), cam.psy * offy + py);
    return RayD{mk3(cam.ox, cam.oy, cam.oz), norm3(rd)};
}

// ---------------------------------------------------------------------------
// The path as a resumable state machine.
//
// tracer.rs:44-103 is "for each sample { for each bounce { ... break ... } }".  Run
// literally on a 64-wide wave, lanes whose path ended early idle until the longest
// path of the wave ends (measured: 40 % VALU lane utilisation on the stock scene).
// Here each lane owns the whole sample loop of its pixel: path_begin() starts a
// sample, path_bounce() advances it by ONE bounce and says whether the path ended,
// so the kernel can immediately regenerate a new camera path in that lane.  Each
// sample's arithmetic and its position in the pixel's running mean are unchanged,
// so the image is bit-identical to the nested-loop form.
// ---------------------------------------------------------------------------
struct PathRegs {
    RayD ray;
    v3 radiance, throughput;
    PathState ps;
    uint32_t bounce;           // ScatterSampleRec.l needs no register: it is zeros before the first
                               // bounce and equals ray.d afterwards (tracer.rs:100)
    Rng rng;
    uint32_t medium;           // media kernels only (dev_media.h): 0, or 1 + the material whose Medium the path is in [| kMediumScatterNow]
};

// tracer.rs:44-57.  HASHED: `pixel` and `pixel_b` are the two hashes of the pixel index (dev_math.h, Rng::init; the state-machine
// kernels keep them in LDS: left to itself the compiler hoists the hashes out of the sample loop into registers that then spill).
template <bool HASHED = false, class S>
RPT_DEV void path_begin(const S& sc, PathRegs& p, float px, float py, FrameKey fkey, uint32_t pixel, uint32_t pixel_b = 0u)
{
    if (HASHED) p.rng.init_hashed(fkey, pixel, pixel_b);
    else p.rng.init(fkey, pixel);
    float offx = p.rng.gen();
    float offy = p.rng.gen();
    p.ray = camera_ray(sc.cam, px, py, offx, offy);
    p.radiance = mk3(0.0f, 0.0f, 0.0f);
    p.throughput = mk3(1.0f, 1.0f, 1.0f);
    p.ps.hit_dist = -1.0f;
    p.ps.scatter_pdf = 0.0f;
    p.bounce = 0;
    p.medium = 0u;
}

// One iteration of the loop at tracer.rs:61-103 in two halves, split where the work stops being needed by every ray:
// TRACE (path_trace_geom): the geometry pass of closest_hit, the miss and emitter exits (tracer.rs:64-87).  What a surface hit
// parks is one dword (GeomHit) next to the path's own registers.  SHADE (path_shade_full): normal, material
// layering, State::finalize, then next-event estimation and BSDF sampling as in path_shade.  The normal, the
// material writes and finalize cost about as much as the three sphere tests; in TRACE they ran for the 62 % of its
// lanes that hit a surface (42 % of the wave), in SHADE they run with the shading block's 84 %.  Per lane the
// operations and their order are unchanged, so images stay bit-identical.
// 0: the ray left the scene (the background is still to be added), 1: it ended on an emitter (radiance updated), 2: the bounce
// goes on in SHADE — a surface was hit, or (media kernels) the path scatters inside the medium it is in.
template <class S, class Q>
RPT_DEV uint32_t path_trace_geom_split(const S& sc, const Q& q, PathRegs& p, GeomHit& g)
{
    EmitterHit e;
    e.is_emitter = false;
    e.light_pdf = 0.0f;
    e.light_emission = mk3(0.0f, 0.0f, 0.0f);
    bool hit;
    { RPT_PROF(PB_CLOSEST); hit = q.geom(sc, p.ray, p.ps, g, e); }
    if (!hit) return 0u;
    if constexpr (S::kMedia) {
        // the medium acts on the segment [0, hit_dist] before what lies at its end is looked at (include/rpt.h, media, step 2)
        if (p.medium != 0u) {
            const DevMedium md = medium_at(sc, p.medium - 1u);
            const float seg = p.ps.hit_dist;
            if (md.type == RPT_MEDIUM_ABSORB) {
                p.throughput = p.throughput * medium_transmittance(md, seg);
            } else if (md.type == RPT_MEDIUM_EMISSIVE) {
                p.radiance = p.radiance + scale3(scale3(md.color, seg), md.density) * p.throughput;
            } else if (md.type == RPT_MEDIUM_SCATTER) {
                const float r = p.rng.gen();
                const float d = rmin(fdiv(-rpt_logf(r), md.density), seg);
                if (d < seg) {                                      // a scatter event before the segment's end: SHADE does the rest
                    p.throughput = p.throughput * md.color;
                    p.ray.o = p.ray.o + d * p.ray.d;
                    p.medium |= kMediumScatterNow;
                    return 2u;
                }
            }
        }
    }
    if (e.is_emitter) {
        RPT_PROF(PB_FINALIZE);
        p.radiance = p.radiance + hit_emission(sc, g) * p.throughput;                      // tracer.rs:74
        // state.depth > 0 always holds (tracer.rs:57,80): the MIS weight is always applied
        float mis_weight = power_heuristic(p.ps.scatter_pdf, e.light_pdf);
        p.radiance = p.radiance + (mis_weight * e.light_emission) * p.throughput;
        return 1u;
    }
    return 2u;
}

template <class S, class Q>
RPT_DEV bool path_trace_geom(const S& sc, const Q& q, PathRegs& p, GeomHit& g)
{
    const uint32_t what = path_trace_geom_split(sc, q, p, g);
    if (what == 0u) {
        RPT_PROF(PB_BACKGROUND);
        p.radiance = p.radiance + background(sc, p.ray) * p.throughput;
    }
    return what == 2u;
}

// What every bounce ends with once the next ray is set: the depth test of tracer.rs:61 and the project's Russian roulette
// (include/rpt.h RPT_RENDER_RUSSIAN_ROULETTE).  True: the path is over.
template <class S>
RPT_DEV bool path_next_bounce(const S& sc, PathRegs& p)
{
    p.bounce += 1;
    if (p.bounce >= sc.max_depth) return true;
    if ((sc.flags & kSceneFlagRussianRoulette) && p.bounce >= 2u) {
        const v3 thr = p.throughput;
        float q = rmax(rmax(thr.x, thr.y), thr.z);
        q = clampf(q, 0.05f, 1.0f);
        const float r = p.rng.gen();
        if (r >= q) return true;
        p.throughput = divs3(thr, q);
    }
    return false;
}

// Media kernels: the bounce is a scatter event inside the medium (p.ray.o is the scatter point — `cold` when the caller
// lent p.ray.o to a shadow march): next-event estimation with the phase function, then a Henyey-Greenstein direction.
template <class S, class Q>
RPT_DEV bool path_shade_medium(const S& sc, const Q& q, PathRegs& p, const volatile float4* cold)
{
    p.medium &= ~kMediumScatterNow;
    const DevMedium md = medium_at(sc, p.medium - 1u);
    const v3 fhp = cold ? mk3(cold->x, cold->y, cold->z) : p.ray.o;
    const v3 wo = -p.ray.d;
    {
        NeeQuery nq = nee_query<false>(sc, q, fhp, mk3(0.0f, 0.0f, 0.0f), p.rng, p.throughput);
        v3 ld = mk3(0.0f, 0.0f, 0.0f);
        if (nq.lit) {
            v3 li = nq.ls.emission;
            if (nq.ls.dist <= 3.40282347e+38f) li = li * medium_transmittance(md, nq.ls.dist);
            const float ph = phase_hg(dot3(wo, nq.ls.direction), md.anisotropy);
            float mis_weight = 1.0f;
            if (nq.light_area > 0.0f) mis_weight = power_heuristic(nq.ls.pdf, ph);
            if (ph > 0.0f) ld = ld + (mis_weight * li) * divs3(mk3(ph, ph, ph), nq.ls.pdf);
        }
        const v3 gain = ld * p.throughput;
        if (nq.pending) q.park(gain);
        else p.radiance = p.radiance + gain;
    }
    const float r1 = p.rng.gen();
    const float r2 = p.rng.gen();
    const v3 dir = sample_hg(wo, md.anisotropy, r1, r2);
    p.ps.scatter_pdf = phase_hg(dot3(wo, dir), md.anisotropy);
    p.ray.o = fhp;
    p.ray.d = dir;
    return path_next_bounce(sc, p);
}

// Returns true when the path is over (pdf <= 0 or depth exhausted).
// `n_pre`: the normal when the caller already has it (the march kernel needs it before the shadow march); `cold`: the
// hit point parked in LDS when p.ray.o no longer holds the path's origin (the march kernel lends it to the shadow
// march).  Both null: everything is rebuilt from the unchanged ray and p.ps.hit_dist.
template <class S, class Q, class M = MaterialPerHit>
RPT_DEV bool path_shade_full(const S& sc, const Q& q, PathRegs& p, const GeomHit& g, const v3* n_pre = nullptr, const volatile float4* cold = nullptr,
                             const M& materials = M{})
{
    if constexpr (S::kMedia) {
        if (p.medium & kMediumScatterNow) return path_shade_medium(sc, q, p, cold);
    }
    const v3 normal = n_pre ? *n_pre : hit_normal(sc, p.ray, p.ps.hit_dist, g);
    const float ndd = dot3(normal, p.ray.d);
    const bool front = (ndd <= 0.0f);
    const v3 ffnormal = mk3(front ? normal.x : -normal.x, front ? normal.y : -normal.y, front ? normal.z : -normal.z);
    typename M::MatType mat;
    float eta;
    ShadeFrame fr;
    if constexpr (M::kTable) {
        RPT_PROF(PB_FINALIZE);
        v3 emission;
        materials.fetch(sc, p.ray, g.code, ndd < 0.0f, mat, eta, emission);
        fr.spec_col = fr.sheen_col = mk3(0.0f, 0.0f, 0.0f);          // (not read: mat_spec_col)
        p.radiance = p.radiance + emission * p.throughput;
    } else {
        RPT_PROF(PB_FINALIZE);
        hit_material(sc, p.ray, g, mat);
        mat_finalize(mat);
        eta = (ndd < 0.0f) ? fdiv(1.0f, mat.ior) : mat.ior;
        p.radiance = p.radiance + mat.emission * p.throughput;
    }
    const v3 fhp = cold ? mk3(cold->x, cold->y, cold->z) : (p.ray.o + p.ps.hit_dist * p.ray.d);
    if constexpr (M::kTable) {
        RPT_PROF(PB_FRAME);
        onb(ffnormal, fr.t, fr.b);
        fr.v = to_local(fr.t, fr.b, ffnormal, -p.ray.d);
    } else {
        RPT_PROF(PB_FRAME);
        fr = make_frame(mat, eta, -p.ray.d, ffnormal);
    }
    {
        NeeQuery nq = nee_query(sc, q, fhp, ffnormal, p.rng, p.throughput);
        if constexpr (S::kMedia) {
            // inside a medium the light arrives attenuated (include/rpt.h, media, step 3)
            if (p.medium != 0u && nq.lit && nq.ls.dist <= 3.40282347e+38f)
                nq.ls.emission = nq.ls.emission * medium_transmittance(medium_at(sc, p.medium - 1u), nq.ls.dist);
        }
        const v3 gain = nee_eval(nq, mat, eta, fr, ffnormal) * p.throughput;
        if (nq.pending) q.park(gain);
        else p.radiance = p.radiance + gain;
    }
    float pdf;
    v3 scatter_l = (p.bounce > 0) ? p.ray.d : mk3(0.0f, 0.0f, 0.0f);   // the stale `l` of tracer.rs:531
    v3 f;
    { RPT_PROF(PB_SAMPLE_HEAD); f = disney_sample(mat, eta, fr, ffnormal, scatter_l, pdf, p.rng); }
    p.ps.scatter_pdf = pdf;
    if (!(pdf > 0.0f)) return true;
    RPT_PROF(PB_SAMPLE_TAIL);
    p.throughput = p.throughput * divs3(f, pdf);
    // the hit point again (not kept across the BSDF code): from LDS, or from the old ray before the direction changes
    const v3 fhp2 = cold ? mk3(cold->x, cold->y, cold->z) : (p.ray.o + p.ps.hit_dist * p.ray.d);
    p.ray.o = fhp2 + sc.eps * scatter_l;
    p.ray.d = scatter_l;
    if constexpr (S::kMedia) {
        // crossing, or staying on one side of, the boundary of a medium (include/rpt.h, media, step 4)
        const uint32_t mi = hit_medium_index(sc, g);
        if (mi != kNoMediumIdx) {
            if (medium_at(sc, mi).type != RPT_MEDIUM_NONE) p.medium = (dot3(scatter_l, normal) < 0.0f) ? (mi + 1u) : 0u;
        }
    }
    return path_next_bounce(sc, p);
}


// One whole iteration of tracer.rs:61-103; true when the path is over.
template <class S>
RPT_DEV bool path_bounce(const S& sc, PathRegs& p)
{
    GeomHit g;
    if (!path_trace_geom(sc, DirectQuery{}, p, g)) return true;
    return path_shade_full(sc, DirectQuery{}, p, g);
}

// One pixel-sample start to end: the nested-loop kernel, and (namespace rptplain) what sample_guard recomputes a sample with.
template <bool HASHED = false, class S>
RPT_DEV v3 trace_sample(const S& sc, float px, float py, FrameKey fkey, uint32_t pixel_index, uint32_t pixel_b = 0u)
{
    PathRegs p;
    path_begin<HASHED>(sc, p, px, py, fkey, pixel_index, pixel_b);
    if (sc.max_depth == 0) return p.radiance;
    while (!path_bounce(sc, p)) {}
    return p.radiance;
}

}  // namespace RPT_NS
#endif  // this pass
