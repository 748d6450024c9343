// dev_scene_large.h — scene queries for analytical scenes too large for the kernarg
// tables of dev_scene.h (BASELINE.json configs[4]: thousands of spheres, tens of lights).
//
// The tables live in HBM.  Loops over primitives use a wave-uniform index and read through
// constant-address-space pointers, so hipcc emits s_load_dwordx4 and the sphere lands in
// SGPRs (scalar cache / L2 resident: 16 B per sphere, the whole table is re-read by every
// wave for every ray — brute force; the acceleration structure is the next step,
// DESIGN.md §8).  Data that depends on the lane (the winning sphere's centre, its material,
// the sampled light) is gathered with ordinary vector loads.
//
// Material layering (dev_integrator.h, apply_patch) needs one bit per primitive and does not
// scale; large scenes therefore require every SPHERE material to be a full patch
// (mask == RPT_MAT_ALL, no procedural part), which rpt_upload_scene checks.  With full
// patches the ordered-acceptance rule of analytical.rs:36-120 reduces to "the last accepted
// sphere's material", i.e. the nearest sphere's (first index on ties), so the result is still
// exactly what the ordered loop gives.  Plane materials stay arbitrary patches.
#pragma once

#include "dev_integrator.h"

namespace rptdev {

#define RPT_CONST_AS __attribute__((address_space(4)))

struct SceneLarge {
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
};

// Wave-uniform table reads: plain dwords through the constant address space, which the
// compiler merges into s_load_dwordx4/x8.
typedef const RPT_CONST_AS float* cfloat_p;
typedef const RPT_CONST_AS uint32_t* cuint_p;

RPT_DEV float4 sphere_uniform(const SceneLarge& sc, uint32_t i)     // i wave-uniform -> scalar load
{
    cfloat_p p = (cfloat_p)sc.spheres + 4u * i;
    return make_float4(p[0], p[1], p[2], p[3]);
}

RPT_DEV DevLight light_uniform(const SceneLarge& sc, uint32_t i)
{
    static_assert(sizeof(DevLight) == 9 * 4, "DevLight is 9 dwords");
    cfloat_p p = (cfloat_p)sc.lights + 9u * i;
    DevLight L;
    L.type = ((cuint_p)p)[0];
    L.px = p[1]; L.py = p[2]; L.pz = p[3];
    L.ex = p[4]; L.ey = p[5]; L.ez = p[6];
    L.radius = p[7]; L.area = p[8];
    return L;
}

RPT_DEV DevMaterial material_uniform(const SceneLarge& sc, uint32_t i)
{
    static_assert(sizeof(DevMaterial) == 23 * 4, "DevMaterial is 23 dwords");
    cfloat_p p = (cfloat_p)sc.materials + 23u * i;
    DevMaterial m;
    m.mask = ((cuint_p)p)[0]; m.proc_kind = ((cuint_p)p)[1];
    for (int c = 0; c < 3; ++c) { m.rgb[c] = p[2 + c]; m.emission[c] = p[5 + c]; }
    m.anisotropic = p[8]; m.metallic = p[9]; m.roughness = p[10]; m.subsurface = p[11]; m.specular_tint = p[12];
    m.sheen = p[13]; m.sheen_tint = p[14]; m.clearcoat = p[15]; m.clearcoat_gloss = p[16]; m.spec_trans = p[17]; m.ior = p[18];
    for (int c = 0; c < 4; ++c) m.proc_params[c] = p[19 + c];
    return m;
}

RPT_DEV DevLight light_at(const SceneLarge& sc, uint32_t index)     // per-lane index -> gather
{
    return sc.lights[index];
}

// AnalyticalScene::closest_hit + Scene::sample_lights, as in dev_integrator.h, for N spheres.
RPT_DEV bool closest_hit(const SceneLarge& sc, const RayD& ray, PathState& ps, HitInfo& hi)
{
    float dist = 3.40282347e+38f;
    bool hit = false;
    uint32_t best = 0xFFFFFFFFu;                                    // nearest sphere so far
    uint32_t accepted_planes = 0;
    v3 pn = mk3(0.0f, 0.0f, 0.0f);
    bool win_plane = false;

    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const float4 s = sphere_uniform(sc, i);
        float t;
        bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
        bool acc = h && (i == 0 || t < dist);                       // analytical.rs:43 / :74
        if (acc) {
            dist = t;
            best = i;
            hit = true;
        }
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevPlane& p = sc.planes[k];
        float t;
        bool h = hit_plane(ray, p, t);
        bool acc = h && ((sc.n_spheres == 0 && k == 0) || t < dist);
        if (acc) {
            dist = t;
            pn = mk3(p.nx, p.ny, p.nz);
            win_plane = true;
            hit = true;
            accepted_planes |= 1u << k;
        }
    }

    mat_defaults(hi.mat);
    if (best != 0xFFFFFFFFu) {                                      // the nearest sphere's full patch
        const DevMaterial m = sc.materials[sc.sphere_material[best]];
        hi.mat.rgb = mk3(m.rgb[0], m.rgb[1], m.rgb[2]);
        hi.mat.emission = mk3(m.emission[0], m.emission[1], m.emission[2]);
        hi.mat.anisotropic = m.anisotropic; hi.mat.metallic = m.metallic; hi.mat.roughness = m.roughness;
        hi.mat.subsurface = m.subsurface; hi.mat.specular_tint = m.specular_tint; hi.mat.sheen = m.sheen;
        hi.mat.sheen_tint = m.sheen_tint; hi.mat.clearcoat = m.clearcoat; hi.mat.clearcoat_gloss = m.clearcoat_gloss;
        hi.mat.spec_trans = m.spec_trans; hi.mat.ior = m.ior;
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevMaterial pm = material_uniform(sc, sc.planes[k].material);   // wave-uniform patch
        apply_patch(hi.mat, pm, (accepted_planes >> k) & 1u, ray.d);
    }

    if (hit) {
        ps.hit_dist = dist;
        v3 c = mk3(0.0f, 0.0f, 0.0f);
        if (!win_plane) {
            const float4 s = sc.spheres[best];                      // per-lane gather
            c = mk3(s.x, s.y, s.z);
        }
        v3 hp = ray.o + dist * ray.d;
        v3 sn = norm3(hp - c);
        hi.normal.x = win_plane ? pn.x : sn.x;
        hi.normal.y = win_plane ? pn.y : sn.y;
        hi.normal.z = win_plane ? pn.z : sn.z;
    }

    // Scene::sample_lights, scene.rs:65-85
    float ldist = ps.hit_dist;
    for (uint32_t i = 0; i < sc.n_lights; ++i) {
        const DevLight L = light_uniform(sc, i);
        if (L.type != RPT_LIGHT_SPHERICAL) continue;
        v3 pos = mk3(L.px, L.py, L.pz);
        float t;
        if (hit_sphere(ray, pos, L.radius, t)) {
            if (t < ldist) {
                ldist = t;
                v3 hit_point = ray.o + t * ray.d;
                float cos_theta = dot3(-ray.d, norm3(hit_point - pos));
                hi.light_pdf = (ldist * ldist) / (L.area * cos_theta * 0.5f);
                hi.light_emission = mk3(L.ex, L.ey, L.ez);
                hi.is_emitter = true;
                ps.hit_dist = t;
                hit = true;
            }
        }
    }
    return hit;
}

RPT_DEV bool any_hit(const SceneLarge& sc, const RayD& ray, float max_dist)
{
    bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    bool occluded = false;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const float4 s = sphere_uniform(sc, i);
        float t;
        bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        float t;
        bool h = hit_plane(ray, sc.planes[k], t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}

}  // namespace rptdev
